"""
bench.py's multi-rank path on the one GPU of the test box: two ranks launched exactly as the driver launches N > 1
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), both on device 0 (AFHIP_BENCH_DEVICE) with
the gloo backend (RCCL refuses two ranks on one device; on an 8-GPU node the same code runs with backend nccl = RCCL
over xGMI).  What it pins: row shards per rank, the chi^2 all-reduce, max-over-ranks timing, one JSON line from rank 0
with whole-job totals (BASELINE configs[3] is this with 8 ranks of 1e6 rows).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_share_the_rows_and_reduce_chi2():
    env = dict(os.environ, AFHIP_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
           "--warmup", "1", "--rows", "30000", "--sources", "200", "--no-cpu-baseline", "--check-rows", "64"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    out = p.stdout.decode("utf-8", "replace")
    assert p.returncode == 0, (out[-2000:], p.stderr.decode("utf-8", "replace")[-4000:])
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]                     # rank 0 only
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["steps"] == 2
    assert r["config"]["rows_per_gpu"] == 30000 and r["config"]["rows_total"] == 60000
    assert "gloo" in r["config"]["sharding"]
    assert r["value"] > 0 and abs(r["value"] - 60000 * 64 / (r["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * r["value"]
    assert r["fp64_max_abs_err"] < 1e-8
    assert r["roofline"]["kernel_ms"] > 0
