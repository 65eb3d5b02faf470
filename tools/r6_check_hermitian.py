#!/usr/bin/env python3
"""One-off evidence for DESIGN 3.4 "Dispatch by symmetry": the GEMM form FORCED onto brightness matrices that are not
Hermitian (fused._hermitian patched to True), against the oracle chain -- how many rows differ, and which."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["AFHIP_GEMM_MIN_FILL"] = "0"
import numpy as np
import test_gpu_fused as FU
import test_gpu_fused_gemm as FG
from codex_africanus_amd.rime import fused
nant = 19
d = FG._decomposable(FU._problem(31, 1200, 4, 11, nant), nant, seed=3)
rng = np.random.default_rng(7)
d["X"] = d["X"] + 0.3 * (rng.standard_normal(d["X"].shape) + 1j * rng.standard_normal(d["X"].shape))
ref = FU._oracle_chain(d, True)
good = FG._call(d)
fused._hermitian = lambda x: True
forced = FG._call(d)
bad_rows = (np.abs(forced - ref).reshape(ref.shape[0], -1).max(1) > 1e-9 * FU._scale(d))
print("dispatcher's route: max |err| %.2e of scale; GEMM form forced: %d of %d rows differ, max |err| %.2e of scale" % (
    np.abs(good - ref).max() / FU._scale(d), int(bad_rows.sum()), ref.shape[0], np.abs(forced - ref).max() / FU._scale(d)))
