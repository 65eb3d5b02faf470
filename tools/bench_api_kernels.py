#!/usr/bin/env python
"""Achieved HBM bandwidth of the API-compatible (materialised-input) kernels on one MI355X:
predict_vis (coh only / dde+coh+die), phase_delay, beam_cube_dde, chi2.  Device-resident torch
tensors, HIP-event timing on torch's stream, algorithmic bytes = inputs read once + output written once."""
import json, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd import rime, sharding

dev = torch.device("cuda:0")


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def rc(*shape):
    return torch.randn(*shape, dtype=torch.complex128, device=dev)


out = {}
# predict_vis, coh only: (src=16, row=262144, chan=64, 2, 2) c128 = 17.2 GB read, 1.07 GB written
# (AF_BENCH_ROWS overrides the row count: 262144 rows make every source slab exactly 2^30 bytes)
s, r, c, t, a = 16, int(os.environ.get("AF_BENCH_ROWS", 262144)), 64, 130, 64
nbl = a * (a - 1) // 2
ti = torch.arange(r, device=dev, dtype=torch.int32) // nbl
a1 = torch.randint(0, a, (r,), device=dev, dtype=torch.int32)
a2 = torch.randint(0, a, (r,), device=dev, dtype=torch.int32)
coh = rc(s, r, c, 2, 2)
dt = timeit(lambda: rime.predict_vis(ti, a1, a2, None, coh, None, None, None, None))
b = coh.numel() * 16 + r * c * 64
out["predict_vis coh-only c128 (16 src x %d rows x 64 chan)" % r] = dict(ms=dt * 1e3, GBs=b / dt / 1e9, bytes=b)
ntime = int(ti.max().item()) + 1
dde = rc(s, ntime, a, c, 2, 2)
die = rc(ntime, a, c, 2, 2)
bv = rc(r, c, 2, 2)
dt = timeit(lambda: rime.predict_vis(ti, a1, a2, dde, coh, dde, die, bv, die))
b2 = b + dde.numel() * 16 + die.numel() * 16 + bv.numel() * 16
out["predict_vis dde+coh+die+bvis c128 (same shape, 64 ant)"] = dict(ms=dt * 1e3, GBs=b2 / dt / 1e9, bytes=b2)
del coh, dde, bv
coh64 = torch.randn(s, r, c, 2, 2, dtype=torch.complex64, device=dev)
dt = timeit(lambda: rime.predict_vis(ti, a1, a2, None, coh64, None, None, None, None))
b = coh64.numel() * 8 + r * c * 32
out["predict_vis coh-only c64"] = dict(ms=dt * 1e3, GBs=b / dt / 1e9, bytes=b)
del coh64
if os.environ.get("AF_BENCH_ONLY_PREDICT"):
    print(json.dumps(out, indent=1)); sys.exit(0)
# phase_delay: (100 src, 100k rows, 64 chan) c128 = 10.2 GB written
lm = (torch.rand(100, 2, dtype=torch.float64, device=dev) - 0.5) * 0.1
uvw = (torch.rand(100000, 3, dtype=torch.float64, device=dev) - 0.5) * 8000
fr = torch.linspace(0.856e9, 1.712e9, 64, dtype=torch.float64, device=dev)
dt = timeit(lambda: rime.phase_delay(lm, uvw, fr))
b = 100 * 100000 * 64 * 16
out["phase_delay f64 (100 src x 100k rows x 64 chan)"] = dict(ms=dt * 1e3, GBs=b / dt / 1e9, bytes=b)
# beam_cube_dde: (100 src, 100 time, 64 ant, 64 chan, 2, 2) = 2.6 GB written
g = torch.linspace(-1, 1, 257, dtype=torch.float64, device=dev)
beam = (torch.exp(-(g[:, None] ** 2 + g[None, :] ** 2) / 0.5)[:, :, None, None, None]
        * torch.ones(1, 1, 33, 2, 2, dtype=torch.complex128, device=dev)).contiguous()
ext = torch.tensor([[-0.06, 0.06], [-0.06, 0.06]], dtype=torch.float64, device=dev)
fmap = torch.linspace(0.856e9, 1.712e9, 33, dtype=torch.float64, device=dev)
pa = torch.rand(100, 64, dtype=torch.float64, device=dev) * np.pi / 6
pe = 1e-3 * torch.randn(100, 64, 64, 2, dtype=torch.float64, device=dev)
asc = 1 + 1e-3 * torch.randn(64, 64, 2, dtype=torch.float64, device=dev)
dt = timeit(lambda: rime.beam_cube_dde(beam, ext, fmap, lm, pa, pe, asc, fr), reps=3)
b = 100 * 100 * 64 * 64 * 64
out["beam_cube_dde c128 (100 src x 100 t x 64 ant x 64 chan)"] = dict(ms=dt * 1e3, GBs=b / dt / 1e9, bytes=b,
                                                                      MJones_per_s=100 * 100 * 64 * 64 / dt / 1e6)
# chi2: 2 x (1e6 x 64 x 4) c128 = 8.2 GB read
m, d = rc(1000000, 64, 4), rc(1000000, 64, 4)
dt = timeit(lambda: sharding.chi2(m, d))
b = 2 * m.numel() * 16
out["chi2 c128 (1e6 rows x 64 chan x 4 corr)"] = dict(ms=dt * 1e3, GBs=b / dt / 1e9, bytes=b)
# vis_to_im at the C2 shape: 1e6 rows x 64 chan x 4 corr visibilities -> 1000 sources
from codex_africanus_amd import dft
del m
lm1k = (torch.rand(1000, 2, dtype=torch.float64, device=dev) - 0.5) * 0.07
uvw1m = (torch.rand(1000000, 3, dtype=torch.float64, device=dev) - 0.5) * 8000
fl = torch.zeros(1000000, 64, 4, dtype=torch.bool, device=dev)
dt = timeit(lambda: dft.vis_to_im(d, uvw1m, lm1k, fr, fl), reps=2)
out["vis_to_im f64 (1e6 rows x 64 chan x 1000 src x 4 corr)"] = dict(
    ms=dt * 1e3, Mvis_per_s=1e6 * 64 / dt / 1e6, TFLOPs_algorithmic=1e6 * 64 * 1000 * 20 / dt / 1e12)
# calibration consumers, FULL 2x2 gains, 2 directions: 1e6 rows x 64 chan
del d, fl
from codex_africanus_amd.calibration.utils import corrupt_vis, residual_vis, correct_vis
nrow_c, nant_c, nbl_c = 1000000, 64, 2016
ntime_c = -(-nrow_c // nbl_c)
tbi = torch.arange(ntime_c, device=dev, dtype=torch.int64) * nbl_c
tbc = torch.full((ntime_c,), nbl_c, device=dev, dtype=torch.int64)
tbc[-1] = nrow_c - (ntime_c - 1) * nbl_c
a1c = torch.randint(0, nant_c, (nrow_c,), device=dev, dtype=torch.int64)
a2c = torch.randint(0, nant_c, (nrow_c,), device=dev, dtype=torch.int64)
# Measurement-Set order (antenna1 < antenna2 enumerated per timestep): consecutive rows share antenna1
_p, _q = np.triu_indices(nant_c, 1)
a1m = torch.from_numpy(np.tile(_p, ntime_c)[:nrow_c].astype(np.int64)).to(dev)
a2m = torch.from_numpy(np.tile(_q, ntime_c)[:nrow_c].astype(np.int64)).to(dev)
jn = rc(ntime_c, nant_c, 64, 2, 2, 2)
md = rc(nrow_c, 64, 2, 2, 2)
vs = rc(nrow_c, 64, 2, 2)
flg = torch.rand(nrow_c, 64, 2, 2, device=dev) < 0.01
j1 = jn[:, :, :, :1].contiguous()
for tag, A1, A2 in ((" (random antenna pairs)", a1c, a2c), (", Measurement-Set row order", a1m, a2m)):
    dt = timeit(lambda: corrupt_vis(tbi, tbc, A1, A2, jn, md), reps=3)
    b = md.numel() * 16 + nrow_c * 64 * 64
    key = "" if "random" in tag else tag
    out["corrupt_vis FULL (1e6 rows x 64 chan x 2 dir)" + key] = dict(ms=dt * 1e3, GBs=b / dt / 1e9, bytes=b)
    dt = timeit(lambda: residual_vis(tbi, tbc, A1, A2, jn, vs, flg, md), reps=3)
    b = md.numel() * 16 + 2 * nrow_c * 64 * 64 + flg.numel()
    out["residual_vis FULL (1e6 rows x 64 chan x 2 dir)" + key] = dict(ms=dt * 1e3, GBs=b / dt / 1e9, bytes=b)
    dt = timeit(lambda: correct_vis(tbi, tbc, A1, A2, j1, vs, flg), reps=3)
    b = 2 * nrow_c * 64 * 64 + flg.numel()
    out["correct_vis FULL (1e6 rows x 64 chan)" + key] = dict(ms=dt * 1e3, GBs=b / dt / 1e9, bytes=b)
# Stokes <-> correlation conversion at the C2 visibility shape
del jn, md, vs, flg, j1
from codex_africanus_amd.model.coherency import convert
vis4 = rc(1000000, 64, 4)
for label, isch, osch in (("corr->Stokes", ["XX", "XY", "YX", "YY"], ["I", "Q", "U", "V"]),
                          ("Stokes->corr", ["I", "Q", "U", "V"], [["XX", "XY"], ["YX", "YY"]])):
    dt = timeit(lambda: convert(vis4, isch, osch), reps=5)
    b = 2 * vis4.numel() * 16
    out["convert %s c128 (1e6 rows x 64 chan x 4)" % label] = dict(ms=dt * 1e3, GBs=b / dt / 1e9, bytes=b)
st_real = torch.randn(1000000, 64, 4, dtype=torch.float64, device=dev)
dt = timeit(lambda: convert(st_real, ["I", "Q", "U", "V"], ["XX", "XY", "YX", "YY"]), reps=5)
b = st_real.numel() * 8 + st_real.numel() * 16
out["convert Stokes->corr f64 -> c128 (1e6 rows x 64 chan x 4)"] = dict(ms=dt * 1e3, GBs=b / dt / 1e9, bytes=b)
print(json.dumps(out, indent=1))
