#!/usr/bin/env python
"""Own row transforms against the hipFFT route on DENSE random images (every pixel non-zero): max |difference| relative to
the largest visibility, fp64 planes; a few sizes and seeds."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd.gridding.wgridder import model
C = 2.99792458e8
for nx, ny in ((512, 512), (1024, 2048), (2048, 1024), (4096, 512)):
    for seed in range(2):
        rng = np.random.default_rng(seed)
        cell = np.deg2rad(1.0) / max(nx, ny)
        freq = np.linspace(1.0e9, 1.3e9, 4)
        nrow = 20000
        uvw = (rng.random((nrow, 3)) - 0.5) / (cell * freq[-1] / C) * np.array([0.9, 0.9, 0.02])
        image = rng.standard_normal((1, nx, ny))
        args = (uvw, freq, image, np.array([0]), np.array([4]), cell)
        os.environ.pop("AFHIP_WGRID_FFT1", None); os.environ.pop("AFHIP_WGRID_FFT2", None)
        own = model(*args, epsilon=1e-7)
        os.environ["AFHIP_WGRID_FFT1"] = os.environ["AFHIP_WGRID_FFT2"] = "0"
        lib = model(*args, epsilon=1e-7)
        print(nx, ny, seed, "max rel diff %.3e" % (np.abs(own - lib).max() / np.abs(lib).max()), "equal" if np.array_equal(own, lib) else "")
