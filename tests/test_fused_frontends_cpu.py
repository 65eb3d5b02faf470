"""
CPU side of the fused predict's row-chunk front-ends (``-m "not gpu"``): the chunk arithmetic of
``chunked.fused_predict_vis`` and ``sharding.fused_predict_shard`` (world-size-2 gloo included) and, where an
interpreter with dask exists, the graph ``rime.dask.fused_predict_vis`` builds -- with the CPU oracle chain standing in
for the one device call per block (tests/fused_cases.py::oracle_fused_predict_vis; the product path has no such
fallback: these tests patch it in).  Expected values: G14, the REFERENCE's dask graph
(tests/golden/make_golden_fused_dask.py; africanus/rime/examples/predict.py:404-525).  The oracle chain itself is pinned
to G14 here too: it is the reference's arithmetic, so the only difference is the order of the source-chunk sum.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden
from fused_cases import CASES, CHUNKINGS, case_arrays, oracle_fused_predict_vis, scale_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g14():
    return load_golden("g14_fused_dask.npz")


@pytest.fixture()
def oracle_kernel(monkeypatch):
    import oracle
    from codex_africanus_amd import chunked
    from codex_africanus_amd.rime import fused
    monkeypatch.setattr(fused, "fused_predict_vis", oracle_fused_predict_vis)
    monkeypatch.setattr(chunked, "_predict_vis", oracle.predict_vis)        # base_vis + DIE stage of a block


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_chain_equals_the_reference_graph(g14, name):
    out = oracle_fused_predict_vis(**case_arrays(g14, name, "one"))
    ref = g14["vis_%s_one" % name]
    assert np.abs(out - ref).max() <= 1e-13 * scale_of(g14, name)


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("ck", list(CHUNKINGS))
def test_chunked_front_end_pairs_row_and_time_chunks(g14, oracle_kernel, name, ck):
    from codex_africanus_amd import chunked
    s, r, t, c = CHUNKINGS[ck]
    out = chunked.fused_predict_vis(chunks={"source": s, "row": r, "time": t, "chan": c}, **case_arrays(g14, name, ck))
    assert np.abs(out - g14["vis_%s_%s" % (name, ck)]).max() <= 1e-13 * scale_of(g14, name)


def test_chunk_errors(g14):
    from codex_africanus_amd import chunked
    a = case_arrays(g14, "beam_die", "one")
    with pytest.raises(ValueError, match="does not equal number of time chunks"):
        chunked.fused_predict_vis(chunks={"row": (20, 20, 20), "time": (3, 3)}, **a)
    with pytest.raises(ValueError, match="Subdivision of antenna dimension"):
        chunked.fused_predict_vis(chunks={"ant": (2, 3)}, **a)
    with pytest.raises(ValueError, match="Both die1_jones and die2_jones"):
        chunked.fused_predict_vis(**dict(a, die2_jones=None))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _shard_worker(rank, world, port, name, q):
    """one rank of the row-sharded fused predict: oracle chain as the kernel, numpy chi^2, gloo all-reduce"""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from codex_africanus_amd import sharding
    from codex_africanus_amd.rime import fused
    import fused_cases
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "g14_fused_dask.npz"))
        a = fused_cases.case_arrays(g, name, "one")
        a["time_index"] = a["time_index"] + 3
        fused.fused_predict_vis = lambda *x, **k: torch.from_numpy(fused_cases.oracle_fused_predict_vis(
            *[None if v is None else np.asarray(v) for v in x], **{kk: (np.asarray(v) if hasattr(v, "shape") else v) for kk, v in k.items()}))
        sharding.chi2 = lambda model, data, weight=None: (torch.abs(data - model) ** 2).sum(dim=(0, 2, 3))
        data = torch.from_numpy(g["vis_%s_one" % name] + 0.01)
        vis, c2, (lo, hi) = sharding.fused_predict_shard(rank, world, data=data, **a)
        q.put((rank, lo, hi, vis.numpy(), c2.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["beam_feed", "nobeam_gauss_die"])
def test_fused_predict_shard_gloo_world2(g14, name):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, name, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, lo0, hi0, v0, c0), (_, lo1, hi1, v1, c1) = res
    ref = g14["vis_%s_one" % name]
    assert lo0 == 0 and hi0 == lo1 and hi1 == ref.shape[0] and hi0 % 10 == 0 and 0 < hi0 < hi1
    assert np.abs(np.concatenate([v0, v1]) - ref).max() <= 1e-13 * scale_of(g14, name)
    assert np.array_equal(c0, c1)                                     # both ranks hold the reduced vector
    want = (np.abs(ref + 0.01 - np.concatenate([v0, v1])) ** 2).sum(axis=(0, 2, 3))
    assert np.allclose(c0, want, rtol=1e-12, atol=0)


DASK_SCRIPT = r"""
import os, sys
import numpy as np
import dask, dask.array as da
from codex_africanus_amd.rime import fused
from codex_africanus_amd.rime import dask as rdask
import fused_cases
from fused_cases import CASES, CHUNKINGS, NANT, case_arrays, scale_of
fused.fused_predict_vis = fused_cases.oracle_fused_predict_vis     # CPU stand-in for the device call of a block
fused.cached_plan = lambda *a, **k: None
from codex_africanus_amd.rime import predict as _p
import oracle
rdask._np_predict_vis = oracle.predict_vis
g14 = np.load(os.path.join(sys.argv[1], "tests", "golden", "g14_fused_dask.npz"))
n = 0
for name in CASES:
    for ck, (s, r, t, c) in CHUNKINGS.items():
        a = case_arrays(g14, name, ck)
        ch = {"time_index": (r,), "antenna1": (r,), "antenna2": (r,), "lm": (s, 2), "uvw": (r, 3), "frequency": (c,),
              "brightness": (s, c, 2, 2), "stokes": (s, 4), "spi": (s, 2, 4), "ref_freq": (s,), "gauss_shape": (s, 3),
              "beam": a["beam"].shape if "beam" in a else None, "beam_lm_extents": (2, 2), "beam_freq_map": (4,),
              "parallactic_angles": (t, NANT), "point_errors": (t, NANT, c, 2), "antenna_scaling": (NANT, c, 2),
              "feed_rotation": (t, NANT, 2, 2), "die1_jones": (t, NANT, c, 2, 2), "die2_jones": (t, NANT, c, 2, 2),
              "base_vis": (r, c, 2, 2)}
        d = {k: da.from_array(v, chunks=ch[k]) for k, v in a.items()}
        for streams in (None, True):
            vis = rdask.fused_predict_vis(streams=streams, **d)
            assert vis.chunks[0] == r and vis.chunks[1] == c
            out = vis.compute(scheduler="threads", num_workers=4)
            assert np.abs(out - g14["vis_%s_%s" % (name, ck)]).max() <= 1e-13 * scale_of(g14, name), (name, ck)
            n += 1
print("FUSED_DASK_GRAPH_OK", n)
"""


def test_dask_graph_with_real_dask_and_the_oracle_as_block_kernel(tmp_path):
    """rime.dask.fused_predict_vis builds the graph; real dask (conda interpreter of the image) runs it; the block's
    device call is replaced by the oracle chain.  Skipped only where no interpreter has dask."""
    exe = None
    for cand in (sys.executable, "/opt/conda/bin/python3.9", "/opt/conda/bin/python3"):
        if os.path.exists(cand) and subprocess.call([cand, "-c", "import dask.array, numpy"], stdout=subprocess.DEVNULL,
                                                    stderr=subprocess.DEVNULL) == 0:
            exe = cand
            break
    if exe is None:
        pytest.skip("no interpreter with dask on this box")
    script = tmp_path / "fused_dask_graph.py"
    script.write_text(DASK_SCRIPT)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    p = subprocess.run([exe, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode("utf-8", "replace")
    assert p.returncode == 0 and "FUSED_DASK_GRAPH_OK 54" in out, out[-4000:]
