#!/usr/bin/env python
"""predict_vis (row block, chan tile) kernel at tools/bench_predict_tile.py's shape with the two block orders, a few
calls each: for a FETCH_SIZE pass of rocprofv3 (the order is read per call: AFHIP_PREDICT_ROWS_FIRST)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd import rime
dev = torch.device("cuda:0")
s, r, c, a = 16, 262144, 64, 64
nbl = a * (a - 1) // 2
ti = torch.arange(r, device=dev, dtype=torch.int32) // nbl
a1 = torch.randint(0, a, (r,), device=dev, dtype=torch.int32)
a2 = torch.randint(0, a, (r,), device=dev, dtype=torch.int32)
ntime = int(ti.max().item()) + 1
rc = lambda *shape: torch.randn(*shape, dtype=torch.complex128, device=dev)
coh, dde = rc(s, r, c, 2, 2), rc(s, ntime, a, c, 2, 2)
for order in sys.argv[1:] or ["0", "1"]:
    os.environ["AFHIP_PREDICT_ROWS_FIRST"] = order
    for _ in range(3):
        rime.predict_vis(ti, a1, a2, dde, coh, dde, None, None, None)
    torch.cuda.synchronize()
print("algorithmic bytes per call: %.3f GB" % ((coh.numel() * 16 + r * c * 64) / 1e9))
