"""beam_cube_dde at tools/bench_api_kernels.py's shape (100 sources x 100 timesteps x 64 antennas x 64 chan, 2 x 2 complex128)
and two smaller layouts: the block-indexed kernel against the flat-index kernel (AFHIP_BEAM_BLOCK=0), interleaved, results
compared bit for bit."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd import rime
dev = torch.device("cuda:0")
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
rng = np.random.default_rng(0)
for (nsrc, ntime, nant, nchan, corr) in ((100, 100, 64, 64, (2, 2)), (1000, 8, 64, 16, (2,)), (37, 11, 7, 50, (1,)), (20, 5, 9, 33, (3,))):
    lw, mh, nud = 65, 65, 33
    beam = rng.standard_normal((lw, mh, nud) + corr) + 1j * rng.standard_normal((lw, mh, nud) + corr)
    ext = np.array([[-1.0, 1.0], [-1.0, 1.0]]); fmap = np.linspace(0.8e9, 1.8e9, nud)
    lm = (rng.random((nsrc, 2)) - 0.5) * 1.2
    pa = rng.random((ntime, nant)) * 6.28
    pe = (rng.random((ntime, nant, nchan, 2)) - 0.5) * 0.01
    asc = 1.0 + (rng.random((nant, nchan, 2)) - 0.5) * 0.02
    freq = np.linspace(0.75e9, 1.9e9, nchan)
    args = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (beam, ext, fmap, lm, pa, pe, asc, freq)]
    res = {}
    line = []
    for rnd in range(2):
        for blk in ("0", "1"):
            os.environ["AFHIP_BEAM_BLOCK"] = blk
            res[blk] = rime.beam_cube_dde(*args)
            dt = timeit(lambda: rime.beam_cube_dde(*args))
            njones = nsrc * ntime * nant * nchan
            line.append("block=%s %.3f ms %.1f GJones/s" % (blk, dt * 1e3, njones / dt / 1e9))
    same = bool(torch.equal(res["0"].view(torch.float64), res["1"].view(torch.float64)))
    print((nsrc, ntime, nant, nchan, corr), " | ".join(line), "bit-equal", same)
