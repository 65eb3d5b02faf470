import ctypes, sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
nrow, nchan = 1000000, 64
ncorr = int(sys.argv[1]) if len(sys.argv) > 1 else 4      # 1: the gridders' visibilities
m = torch.randn(nrow, nchan, ncorr, dtype=torch.complex128, device=dev)
d = torch.randn(nrow, nchan, ncorr, dtype=torch.complex128, device=dev)
out = torch.zeros(nchan, dtype=torch.float64, device=dev)
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def run():
    _lib.call("af_chi2_c128", P(m), P(d), None, nrow, nchan, ncorr, P(out), st)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
ref = ((d - m).abs() ** 2).sum(dim=(0, 2))
print(json.dumps(dict(ncorr=ncorr, ms=ms, TBs=2 * m.numel() * 16 / ms / 1e9, relerr=float(((out - ref).abs() / ref).max()))))
