"""Convolutional degridding (BASELINE configs[4]): the Perley degridder and the wgridder-shaped `model`."""
import ctypes
import os
import time

import numpy as np

from .common import (FP32_PEAK_TFLOPS, FP64_PEAK_TFLOPS, L2_PEAK_GBS, NUMBA_CALIBRATION, parallel_rows as _parallel_rows,
                     sized_cpu_sample, threads_available as _threads)


class Degrid(object):
    """BASELINE configs[4]: convolutional degridding (africanus/gridding/perleypolyhedron/degridder.py:79-175) of a
    4096^2 complex grid onto 1e6 rows x 64 chan with a 7x7-tap kernel (oversampling 63, packed gather policy),
    XX / YY from Stokes I; uniformly random uv inside 0.45 of the grid (no track locality at all)."""
    W, OS, CELL = 7, 63, 2.0

    def __init__(self, args, rank, dev, lib, _lib, t):
        import torch
        from codex_africanus_amd.gridding.perleypolyhedron import kernels
        self.args, self._lib = args, _lib
        nrow, nchan, npix = args.rows, args.chans, args.npix
        freq = np.linspace(0.856e9, 1.712e9, nchan)
        self.wl = 299792458.0 / freq
        rng = np.random.default_rng(1000 + args.seed + rank)
        umax = 0.45 / np.deg2rad(self.CELL / 3600.0) * self.wl.min()
        uvw = np.zeros((nrow, 3))
        uvw[:, :2] = rng.uniform(-1, 1, (nrow, 2)) * umax
        uvw[:, 2] = rng.uniform(-400, 400, nrow)
        self.uvw = uvw
        g = torch.Generator(device="cpu").manual_seed(args.seed)
        grid = torch.randn(1, npix, npix, 2, dtype=torch.float64, generator=g)
        self.d_grid = torch.view_as_complex(grid).to(dev)
        self.kernel = kernels.pack_kernel(kernels.kbsinc(self.W, oversample=self.OS), self.W, self.OS)
        self.chanmap = np.zeros(nchan, dtype=np.int64)
        self.coef = np.array([1, 1], dtype=np.complex128)      # XXYY_FROM_I
        self.ncorr = 2
        self.dv = dict(uvw=t(uvw), wl=t(self.wl), cm=t(self.chanmap), k=t(self.kernel), cf=t(self.coef))
        self.ws_bytes = int(lib.af_degridder_workspace_bytes(nrow))
        self.d_ws = torch.empty(max(self.ws_bytes, 256), dtype=torch.uint8, device=dev)
        self.centre = np.zeros(2)
        self.label = ("convolutional degridding %d^2 grid, 7x7 taps, oversampling 63, 2 corr from Stokes I "
                      "(BASELINE configs[4])" % npix)

    def predict(self, d_vis, stream, P):
        a, v = self.args, self.dv
        self._lib.call("af_degridder_c128", P(v["uvw"]), P(self.d_grid), P(v["wl"]), P(v["cm"]), self.CELL,
                       self.centre.ctypes.data, self.centre.ctypes.data, P(v["k"]), self.W, self.OS, 0, P(v["cf"]),
                       2, 1, a.rows, a.chans, a.npix, P(d_vis), P(self.d_ws), self.ws_bytes, stream)

    def _oracle(self, rows, grid_host):
        import oracle
        return oracle.degridder(self.uvw[rows], grid_host, self.wl, self.chanmap, self.CELL, (0.0, 0.0), (0.0, 0.0),
                                self.kernel, self.W, self.OS, "None", "None", "XXYY_FROM_I",
                                "conv_1d_axisymmetric_packed_gather")

    def reference_rows(self, rows):
        return self._oracle(rows, self.d_grid.cpu().numpy()), rows

    def roofline(self, kernel_s):
        a = self.args
        nvis = float(a.rows) * a.chans
        # algorithmic HBM bytes: the visibilities written (ncorr x 16 B each), uvw, and the grid read ONCE (it
        # is re-read ~12x through L2 / Infinity Cache by the 49-tap gathers: "gather" below)
        alg_bytes = nvis * self.ncorr * 16 + a.rows * 24 + float(a.npix) ** 2 * 16
        taps = nvis * self.W * self.W
        return dict(kernel="degrid_coop_kernel<7>", bound="hbm", alg_bytes=alg_bytes, alg_flops=taps * 8.0,
                    channels_in_kernel=a.chans,
                    gather={"achieved": taps * 16 / kernel_s / 1e9, "peak": L2_PEAK_GBS, "unit": "GB/s",
                            "frac": taps * 16 / kernel_s / 1e9 / L2_PEAK_GBS,
                            "note": "16-byte grid cells gathered per tap (49 per visibility), served by L2 / "
                                    "Infinity Cache: the resource that actually bounds the kernel"},
                    note="HBM view: 32 B written per visibility + the grid once; the kernel is bound by the "
                         "gather path (784 B of grid cells per visibility through L2), see 'gather'")

    def cpu_baseline(self, min_seconds):
        threads = _threads()
        gh = self.d_grid.cpu().numpy()
        rows = np.arange(self.args.rows)
        s = sized_cpu_sample(lambda n: self._oracle(rows[:n], gh),
                             lambda n: _parallel_rows(lambda lo, hi: self._oracle(rows[lo:hi], gh), n, threads),
                             self.args.rows, threads, min_seconds)
        one = self.args.chans / s["per_row_s"] / 1e6
        return {
            "value": s["rows"] * self.args.chans / s["seconds"] / 1e6, "unit": "Mvis/s", "cores": threads, "kind": "port",
            "sample": "oracle degridder (C restatement of africanus/gridding/perleypolyhedron/degridder.py:15-175, "
                      "packed gather policy), %d rows x %d chan on %d threads in %.2f s after a warm-up call; linear in "
                      "rows; single-thread probe %d rows in %.2f s = %.3f Mvis/s"
                      % (s["rows"], self.args.chans, threads, s["seconds"], s["probe_rows"], s["probe_s"], one),
            "single_thread_value": one, "probe_rows": s["probe_rows"], "probe_seconds": s["probe_s"],
            "sample_rows": s["rows"], "sample_seconds": s["seconds"],
        }


class Wgrid(object):
    """BASELINE configs[4] as named -- wgridder-style degridding of a 4096^2 model IMAGE onto 1e6 rows x 64 chan at
    epsilon 1e-5 (7 x 7 x 7 taps, w-stacking): africanus/gridding/wgridder/im2vis.py:14-99 (arithmetic in the un-vendored
    ducc0: the accuracy contract of gridding/wgridder/tests/test_wgridder.py:18-113 is what is checked).  Uniformly
    random uv inside 0.45 of the grid, |w| <= 400 m; a sparse image (3000 non-zero pixels) so that the direct transform
    of a row sample is affordable for the checker and the CPU baseline."""
    EPS, CELL = 1e-5, 2.0

    def __init__(self, args, rank, dev, lib, _lib, t):
        import torch
        from codex_africanus_amd.gridding.wgridder.im2vis import kernel_parameters, kernel_correction, _quadrature
        self.args, self._lib = args, _lib
        nrow, nchan, npix = args.rows, args.chans, args.npix
        self.freq = np.linspace(0.856e9, 1.712e9, nchan)
        self.cell = cell = np.deg2rad(self.CELL / 3600.0)
        rng = np.random.default_rng(2000 + args.seed + rank)
        umax = 0.45 / cell * (299792458.0 / self.freq.max())
        uvw = np.zeros((nrow, 3))
        uvw[:, :2] = rng.uniform(-1, 1, (nrow, 2)) * umax
        uvw[:, 2] = rng.uniform(-400, 400, nrow)
        self.uvw = uvw
        image = np.zeros((npix, npix))
        nz = np.random.default_rng(args.seed).integers(0, npix, (3000, 2))
        image[nz[:, 0], nz[:, 1]] = np.random.default_rng(args.seed + 1).lognormal(0, 1, 3000)
        self.image = image
        self.W, self.beta = kernel_parameters(self.EPS)
        nu = int(lib.af_wgrid_padded(npix))
        self.nu = nu
        corr = kernel_correction(npix, nu, self.W, self.beta)
        qt, qw = _quadrature()
        emax = 2 * (npix / 2.0 * cell) ** 2
        self.max_nm1 = emax / (np.sqrt(1.0 - emax) + 1.0)
        fl = self.freq / 299792458.0
        w = uvw[:, 2]
        cands = (w.min() * fl.min(), w.min() * fl.max(), w.max() * fl.min(), w.max() * fl.max())
        self.wl = (float(min(cands)), float(max(cands)))
        self.nplanes = int(lib.af_wgrid_planes(self.wl[0], self.wl[1], float(self.max_nm1), self.W, 1))
        self.dv = dict(uvw=t(uvw), freq=t(self.freq), image=t(image), cu=t(corr), qt=t(qt), qw=t(qw))
        self.ws_bytes = int(lib.af_wgrid_workspace_bytes(npix, npix, self.nplanes, nrow, nchan, self.nplanes, self.W))
        self.d_ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.ncorr = 1
        self.label = ("wgridder-style degridding of a %d^2 image, epsilon %g: %d taps per axis, %d w-planes of %d^2 "
                      "(BASELINE configs[4] as named)" % (npix, self.EPS, self.W, self.nplanes, nu))

    def predict(self, d_vis, stream, P):
        a, v = self.args, self.dv
        self._lib.call("af_wgrid_im2vis_f64", P(v["uvw"]), P(v["freq"]), a.rows, a.chans, 0, a.chans, P(v["image"]),
                       a.npix, a.npix, self.cell, self.cell, P(v["cu"]), P(v["cu"]), P(v["qt"]), P(v["qw"]), self.W,
                       self.beta, self.wl[0], self.wl[1], float(self.max_nm1), 1, None, None, P(d_vis), P(self.d_ws),
                       self.ws_bytes, stream)

    def _direct(self, rows, omp):
        import oracle
        npix, cell = self.args.npix, self.cell
        ix, iy = np.nonzero(self.image)
        x, y = (ix - npix / 2) * cell, (iy - npix / 2) * cell
        n = np.sqrt(1 - x * x - y * y)
        src = np.broadcast_to((self.image[ix, iy] / n)[:, None, None], (ix.size, self.freq.size, 1)).copy()
        return oracle.im_to_vis(src, self.uvw[rows] * np.array([1, 1, -1.0]), np.stack([x, y], 1), self.freq, omp=omp)

    def reference_rows(self, rows):
        return self._direct(rows, True), rows

    def roofline(self, kernel_s):
        a = self.args
        nvis = float(a.rows) * a.chans
        # dominant kernel of the call = the visibility pass wg_degrid_tiles<W> (the other ~30 ms are hipFFT row
        # transforms, transposes and the device sort).  Algorithmic HBM bytes of that launch: every cell of every
        # w-plane read once + 16 B written per visibility + the sorted index (4 B) and uvw.
        alg_bytes = float(self.nplanes) * self.nu * self.nu * 16 + nvis * 16 + nvis * 4 + a.rows * 24
        taps = nvis * self.W ** 3
        return dict(kernel="wg_degrid_tiles<%d>" % self.W, bound="hbm", alg_bytes=alg_bytes, alg_flops=taps * 4.0,
                    channels_in_kernel=a.chans,
                    note="the visibility pass of the call (sorted (tile, plane) chunks, tiles staged through LDS); the "
                         "step also runs %d pruned plane transforms (hipFFT rows + transposes) and the device sort; "
                         "fp64_max_abs_err here is against the direct transform, whose contract is an l2 error <= "
                         "epsilon" % self.nplanes)

    def cpu_baseline(self, min_seconds):
        threads = _threads()

        def parallel(n):
            t0 = time.perf_counter()
            self._direct(np.arange(n), True)
            return time.perf_counter() - t0

        s = sized_cpu_sample(lambda n: self._direct(np.arange(n), False), parallel, self.args.rows, threads, min_seconds)
        one = self.args.chans / s["per_row_s"] / 1e6
        return {
            "value": s["rows"] * self.args.chans / s["seconds"] / 1e6, "unit": "Mvis/s", "cores": threads, "kind": "port",
            "sample": "the reference's CPU path for this entry is ducc0.wgridder.dirty2ms (absent here: not vendored, "
                      "not installed); timed instead: the direct transform the accuracy contract is stated against "
                      "(oracle im_to_vis over the image's 3000 non-zero pixels, OpenMP over rows), %d rows x %d chan "
                      "in %.2f s on %d threads; its cost grows with the number of non-zero pixels, the wgridder's does not"
                      % (s["rows"], self.args.chans, s["seconds"], threads),
            "single_thread_value": one, "probe_rows": s["probe_rows"], "probe_seconds": s["probe_s"],
            "sample_rows": s["rows"], "sample_seconds": s["seconds"],
        }


class WgridF32Planes(Wgrid):
    """The same call with the w-planes in float32 (af_wgrid_plane_precision(AF_WGRID_PLANES_F32)): what a float32
    image gets -- the reference's single-precision call, africanus/gridding/wgridder/im2vis.py:41-47 -- and what a
    float64 caller may opt into at epsilon >= 1e-5 (gridding.wgridder.plane_precision("single")); same checker, same
    accuracy contract.  Not the default for float64 images: adjointness with `dirty` then holds to ~1e-7, and the
    reference's double-precision test pins 1e-12."""

    def __init__(self, *a):
        Wgrid.__init__(self, *a)
        self.label += "; float32 w-planes"

    def predict(self, d_vis, stream, P):
        lib = self._lib.load()
        prev = lib.af_wgrid_plane_precision(1)           # per thread
        try:
            Wgrid.predict(self, d_vis, stream, P)
        finally:
            lib.af_wgrid_plane_precision(prev)

    def roofline(self, kernel_s):
        r = Wgrid.roofline(self, kernel_s)
        nvis = float(self.args.rows) * self.args.chans
        r["kernel"] = "wg_degrid_tiles<%d, float2>" % self.W
        r["alg_bytes"] = float(self.nplanes) * self.nu * self.nu * 8 + nvis * 16 + nvis * 4 + self.args.rows * 24
        r["note"] += "; float32 planes: 8 bytes per cell"
        return r

