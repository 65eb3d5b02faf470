"""How the rate of the coherency-only predict_vis (16 read streams + 1 write stream) depends on WHERE the output lies
relative to the coherencies and on the stride between the per-source arrays: one big device buffer, explicit pointers
through the C ABI (af_predict_vis_c128), HIP-event timing."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
s, r, c = 16, 262144, 64
cell = 64
big = torch.empty(20 * 2**30 // 8, dtype=torch.float64, device=dev)
big.normal_()
base = big.data_ptr()
base = (base + 2**21 - 1) // 2**21 * 2**21          # 2 MB aligned
ws = torch.empty(256, dtype=torch.uint8, device=dev)
ti = torch.zeros(r, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run(coh_ptr, out_ptr, nsrc, rows):
    _lib.call("af_predict_vis_c128", ti.data_ptr(), ti.data_ptr(), ti.data_ptr(), 4, rows, None, coh_ptr, None, None, None, None,
              nsrc, 0, 0, c, 4, _lib.AF_JONES_2X2, out_ptr, ws.data_ptr(), 256, st)


def rate(coh_ptr, out_ptr, nsrc, rows, reps=5):
    run(coh_ptr, out_ptr, nsrc, rows); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(coh_ptr, out_ptr, nsrc, rows)
    e1.record(); torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / reps * 1e-3
    return (nsrc + 1) * rows * c * cell / dt / 1e12


coh_bytes = s * r * c * cell
for flat in ("1", "0"):
    os.environ["AFHIP_PREDICT_FLAT"] = flat
    print("flat kernel" if flat == "1" else "lane-per-cell kernel")
    for delta in (0, 4096, 16384, 65536, 262144, 2**20, 2**21, 2**21 + 4096, 2**22 + 65536, 2**24, 2**24 + 2**19, 2**26 + 2**20 + 8192):
        print("  source stride 2^30, out at coh_end + %9d B: %.2f TB/s" % (delta, rate(base, base + coh_bytes + delta, s, r)))
    for rows in (262144 - 64, 262144 + 16, 262144 + 64, 262144 + 1024, 250000, 1000000 // 4):
        cb = s * rows * c * cell
        print("  rows %7d (stride 2^30 %+d B), out at coh_end + 0: %.2f TB/s ; + 2 MB + 4 KB: %.2f TB/s" % (
            rows, rows * c * cell - 2**30, rate(base, base + cb, s, rows), rate(base, base + cb + 2**21 + 4096, s, rows)))
