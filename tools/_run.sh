python tools/bench_wgridder_dirty.py | cut -c1-200; timeout 900 python -m pytest tests/test_gpu_wgridder.py -q -x 2>&1 | tail -3
