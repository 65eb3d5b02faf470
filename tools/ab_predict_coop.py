#!/usr/bin/env python
"""predict_vis without DDE terms, lane = cell with cooperative IO (round 4) against round 3's lane kernel
(AFHIP_PREDICT_COOP=0), same process, interleaved: the coherency stream (16 src x 262144 rows x 64 chan, 2 x 2) in
complex128 / complex64, alone and with DIE terms + base_vis."""
import os, sys, json
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


s, r, c, a = 16, 262144, 64, 64
nbl = a * (a - 1) // 2
ti = torch.arange(r, device=dev, dtype=torch.int32) // nbl
a1 = torch.randint(0, a, (r,), device=dev, dtype=torch.int32)
a2 = torch.randint(0, a, (r,), device=dev, dtype=torch.int32)
ntime = int(ti.max().item()) + 1
out = {}
for dt_ in (torch.complex128, torch.complex64):
    rc = lambda *shape: torch.randn(*shape, dtype=dt_, device=dev)
    coh, die, bv = rc(s, r, c, 2, 2), rc(ntime, a, c, 2, 2), rc(r, c, 2, 2)
    esz = coh.element_size()
    for name, args, b in (("coh only", (None, coh, None, None, None, None), (coh.numel() + bv.numel()) * esz),
                          ("coh + die + bvis", (None, coh, None, die, bv, die), (coh.numel() + 2 * bv.numel()) * esz)):
        for rnd in range(2):
            for env in ("1", "0"):
                os.environ["AFHIP_PREDICT_COOP"] = env
                t = timeit(lambda: rime.predict_vis(ti, a1, a2, *args))
                key = "%s %s %s" % (str(dt_).split(".")[-1], name, "coop" if env == "1" else "lane")
                out.setdefault(key, []).append(round(b / t / 1e12, 3))
    del coh, die, bv
os.environ.pop("AFHIP_PREDICT_COOP", None)
print(json.dumps(out, indent=1))
