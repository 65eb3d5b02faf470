#!/usr/bin/env python
"""predict_vis with DDE + DIE gathers at tools/bench_api_kernels.py's shape (16 sources x 262144 rows x 64 chan, 64
antennas, 2x2 c128): the (row block, chan tile) kernel with LDS-staged Jones against the lane-per-cell kernel, and
the tile shapes of the measurement hook.  One process per variant is not needed: the switches are read per call."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd import rime

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


s, r, c, a = 16, int(os.environ.get("AF_BENCH_ROWS", 262144)), 64, 64
nbl = a * (a - 1) // 2
ti = torch.arange(r, device=dev, dtype=torch.int32) // nbl
a1 = torch.randint(0, a, (r,), device=dev, dtype=torch.int32)
a2 = torch.randint(0, a, (r,), device=dev, dtype=torch.int32)
ntime = int(ti.max().item()) + 1
coh, dde, die, bv = rc(s, r, c, 2, 2), rc(s, ntime, a, c, 2, 2), rc(ntime, a, c, 2, 2), rc(r, c, 2, 2)
b_coh = coh.numel() * 16 + r * c * 64
b_all = b_coh + dde.numel() * 16 + die.numel() * 16 + bv.numel() * 16
out = {}
variants = [("lane per cell", dict(AFHIP_PREDICT_TILE="0")),
            ("streamed form (coherencies through LDS, 2 sub-blocks share a Jones copy)", dict(AFHIP_PREDICT_TILE="1", AFHIP_PREDICT_STREAM="1"))]
for ct, tb, cpt in ((4, 512, 1), (4, 512, 2), (8, 1024, 1)):
    for ts in (1, 2):
        variants.append(("tile CT=%d TB=%d cells/thread=%d stage-of-%d-timestep(s)" % (ct, tb, cpt, ts),
                         dict(AFHIP_PREDICT_TILE="1", AFHIP_PREDICT_STREAM="0", AFHIP_PREDICT_TILE_CT=str(ct), AFHIP_PREDICT_TILE_TB=str(tb),
                              AFHIP_PREDICT_TILE_TS=str(ts), AFHIP_PREDICT_TILE_CPT=str(cpt))))
for name, env in list(variants[2:]):
    variants.append((name + ", rows first", dict(env, AFHIP_PREDICT_ROWS_FIRST="1")))
for g in (4, 8, 32, 64):
    variants.append(("tile CT=4 TB=512 cells/thread=1 stage-of-1-timestep(s), rows first, %d row blocks per XCD turn" % g,
                     dict(variants[2][1], AFHIP_PREDICT_ROWS_FIRST="1", AFHIP_PREDICT_GROUP=str(g))))
ref = None
for name, env in variants:
    os.environ["AFHIP_PREDICT_ROWS_FIRST"] = "0"
    os.environ["AFHIP_PREDICT_GROUP"] = "0"
    os.environ.update(env)
    v = rime.predict_vis(ti, a1, a2, dde, coh, dde, die, bv, die)
    if ref is None:
        ref = v
    same = bool(torch.equal(v, ref))
    dt1 = timeit(lambda: rime.predict_vis(ti, a1, a2, dde, coh, dde, None, None, None))
    dt2 = timeit(lambda: rime.predict_vis(ti, a1, a2, dde, coh, dde, die, bv, die))
    out[name] = dict(dde_coh_ms=dt1 * 1e3, dde_coh_TBs=b_coh / dt1 / 1e12, all_ms=dt2 * 1e3, all_TBs=b_all / dt2 / 1e12,
                     bit_equal_to_lane_kernel=same)
dtc = timeit(lambda: rime.predict_vis(ti, a1, a2, None, coh, None, None, None, None))
out["coh only (no gathers)"] = dict(ms=dtc * 1e3, TBs=b_coh / dtc / 1e12)
print(json.dumps(out, indent=1))
