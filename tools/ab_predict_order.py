"""predict_vis (row block, chan tile) kernel: block orders and row blocks per XCD turn, interleaved, four rounds (same-box A/B)."""
import json, os, sys
import torch
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
rc = lambda *shape: torch.randn(*shape, dtype=torch.complex128, device=dev)
s, r, c, a = 16, 262144, 64, 64
nbl = a * (a - 1) // 2
ti = torch.arange(r, device=dev, dtype=torch.int32) // nbl
a1 = torch.randint(0, a, (r,), device=dev, dtype=torch.int32)
a2 = torch.randint(0, a, (r,), device=dev, dtype=torch.int32)
ntime = int(ti.max().item()) + 1
coh, dde = rc(s, r, c, 2, 2), rc(s, ntime, a, c, 2, 2)
b = coh.numel() * 16 + r * c * 64
V = [("chan tiles first", dict(AFHIP_PREDICT_ROWS_FIRST="0", AFHIP_PREDICT_GROUP="0")),
     ("rows first g16", dict(AFHIP_PREDICT_ROWS_FIRST="1", AFHIP_PREDICT_GROUP="16")),
     ("rows first g8", dict(AFHIP_PREDICT_ROWS_FIRST="1", AFHIP_PREDICT_GROUP="8")),
     ("rows first g4", dict(AFHIP_PREDICT_ROWS_FIRST="1", AFHIP_PREDICT_GROUP="4")),
     ("rows first g2", dict(AFHIP_PREDICT_ROWS_FIRST="1", AFHIP_PREDICT_GROUP="2"))]
for rnd in range(4):
    line = []
    for name, env in V:
        os.environ.update(env)
        dt = timeit(lambda: rime.predict_vis(ti, a1, a2, dde, coh, dde, None, None, None))
        line.append("%s %.2f" % (name, b / dt / 1e12))
    dt = timeit(lambda: rime.predict_vis(ti, a1, a2, None, coh, None, None, None, None))
    line.append("coh only %.2f" % (b / dt / 1e12))
    print(" | ".join(line))
