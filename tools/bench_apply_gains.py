"""apply_gains (africanus/rime/predict.py:623-647: predict_vis with DIE terms and base_vis only) at C2's counts: 1e6 rows x 64
chan x 2x2 complex128, 64 antennas, 32 timesteps; bytes = visibilities read + written (the gains are 8 MB)."""
import os, sys
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
r, c, a = int(os.environ.get("AF_BENCH_ROWS", 1000000)), 64, 64
nbl = a * (a - 1) // 2
ti = torch.arange(r, device=dev, dtype=torch.int32) // nbl
a1 = torch.randint(0, a, (r,), device=dev, dtype=torch.int32)
a2 = torch.randint(0, a, (r,), device=dev, dtype=torch.int32)
ntime = int(ti.max().item()) + 1
rc = lambda *shape: torch.randn(*shape, dtype=torch.complex128, device=dev)
vis, die = rc(r, c, 2, 2), rc(ntime, a, c, 2, 2)
b = 2 * vis.numel() * 16
# default: lane = cell with cooperative IO (round 4); AFHIP_APPLY_COOP=0: round 3's LDS-staged tile kernel
for env in ({}, {"AFHIP_APPLY_COOP": "0"}):
    os.environ.update(env)
    ref = rime.apply_gains(ti, a1, a2, die, vis, die)
    for _ in range(3):
        dt = timeit(lambda: rime.apply_gains(ti, a1, a2, die, vis, die))
        print(env, "apply_gains %.3f ms  %.2f TB/s" % (dt * 1e3, b / dt / 1e12))
dt = timeit(lambda: vis.clone())
print("device copy of the visibilities %.3f ms  %.2f TB/s" % (dt * 1e3, b / dt / 1e12))
