"""im_to_vis workloads: the headline (BASELINE configs[1]), complex brightness, Gaussian sources, float32."""
import ctypes
import os
import time

import numpy as np

from .common import (FP32_PEAK_TFLOPS, FP64_PEAK_TFLOPS, L2_PEAK_GBS, NUMBA_CALIBRATION, parallel_rows as _parallel_rows,
                     sized_cpu_sample, threads_available as _threads)


class Dft(object):
    """im_to_vis (africanus/dft/kernels.py:14-69): real image (headline) or complex brightness."""

    def __init__(self, args, rank, dev, lib, _lib, t):
        from codex_africanus_amd.testing import synthetic_inputs, real_image
        self.args, self._lib, self.lib = args, _lib, lib
        self.cplx = args.workload == "dft_complex"
        # chi^2 in the transform's epilogue pays where two waves share a SIMD (real images, 32-channel tiles: +0.6 ms in the
        # kernel for a 1.5 ms pass); with complex pixels (64-channel tiles, one wave per SIMD) it costs what the pass costs
        self.chi2_in_epilogue = not self.cplx
        nrow, nchan, nsrc = args.rows, args.chans, args.sources
        self.ncorr = 4
        d = synthetic_inputs(seed=args.seed, nrow=16, nchan=nchan, nsrc=nsrc, nant=64)
        rng = np.random.default_rng(1000 + args.seed + rank)
        uvw = np.empty((nrow, 3))
        uvw[:, 0] = rng.uniform(-4000, 4000, nrow)
        uvw[:, 1] = rng.uniform(-4000, 4000, nrow)
        uvw[:, 2] = rng.uniform(-400, 400, nrow)
        if self.cplx:   # linear-feed coherency matrices [I+Q, U+iV, U-iV, I-Q], flat spectrum
            image = np.ascontiguousarray(np.broadcast_to(d["brightness"][:, None, :], (nsrc, nchan, 4)))
        else:
            image = real_image(d)
        self.image, self.uvw, self.lm, self.freq = image, uvw, d["lm"], d["frequency"]
        self.d_image, self.d_uvw, self.d_lm, self.d_freq = t(image), t(uvw), t(self.lm), t(self.freq)
        self.ws_bytes = int(lib.af_im_to_vis_workspace_bytes(nsrc, nchan, 4, int(self.cplx)))
        import torch
        self.d_ws = torch.empty(max(self.ws_bytes, 256), dtype=torch.uint8, device=dev)
        self.mode = {"auto": _lib.AF_DFT_AUTO, "exact": _lib.AF_DFT_EXACT,
                     "recurrence": _lib.AF_DFT_RECURRENCE}[args.mode]
        self.label = ("im_to_vis DFT predict, complex brightness (the fused predict without DDEs)" if self.cplx
                      else "im_to_vis DFT predict (BASELINE configs[1])")

    def predict(self, d_vis, stream, P):
        a = self.args
        self._lib.call("af_im_to_vis_f64", P(self.d_image), int(self.cplx), P(self.d_uvw), P(self.d_lm),
                       P(self.d_freq), a.sources, a.rows, a.chans, 4, self._lib.CONVENTION["fourier"], self.mode,
                       P(d_vis), P(self.d_ws), self.ws_bytes, stream)

    def predict_chi2(self, d_vis, d_data, d_chi2, stream, P):
        """the step's transform AND its chi^2 in one call: summed in the transform's epilogue (af_im_to_vis_chi2_f64)"""
        a = self.args
        self._lib.call("af_im_to_vis_chi2_f64", P(self.d_image), int(self.cplx), P(self.d_uvw), P(self.d_lm),
                       P(self.d_freq), a.sources, a.rows, a.chans, 4, self._lib.CONVENTION["fourier"], self.mode,
                       P(d_vis), P(d_data), None, P(d_chi2), P(self.d_ws), self.ws_bytes, stream)

    def reference_rows(self, rows):
        import oracle
        return oracle.im_to_vis(self.image, self.uvw[rows], self.lm, self.freq, omp=True), rows

    def end_to_end(self):
        """The drop-in call as a reference user makes it: numpy in -> numpy out, `dft.im_to_vis(image, uvw, lm, frequency)`
        (upload, transform, 64 B per visibility back over PCIe into a fresh array).  SURVEY 8(d): reported separately,
        never `value`.  Best of two calls after one warm call."""
        from codex_africanus_amd import dft
        a = self.args
        best = None
        for k in range(3):
            t0 = time.perf_counter()
            vis = dft.im_to_vis(self.image, self.uvw, self.lm, self.freq)
            dt = time.perf_counter() - t0
            if k:
                best = dt if best is None else min(best, dt)
        assert vis.shape == (a.rows, a.chans, 4)
        del vis
        return {"ms": best * 1e3, "value": a.rows * a.chans / best / 1e6, "unit": "Mvis/s",
                "call": "codex_africanus_amd.dft.im_to_vis(numpy arrays) -> numpy array: H2D + kernels + D2H of %.2f GB"
                        % (a.rows * a.chans * 64 / 1e9)}

    def roofline(self, kernel_s):
        a = self.args
        nrow, nchan, nsrc, ncorr = a.rows, a.chans, a.sources, 4
        # Dominant kernel = the one the library's measurement hook brackets.  4-correlation images on a
        # one-spacing band run dft_mfma_kernel<CT>: every CT-channel tile in ONE launch (C2: all 64 channels); CT = 32
        # for real images (two waves per SIMD), 64 for complex ones (af_im_to_vis_mfma.hip main_ct).
        mfma = a.mode != "exact" and nchan >= 14
        px = 16 if self.cplx else 8
        if mfma:
            ct = 64 if self.cplx else 32
            ntile = nchan // ct + (1 if nchan % ct > ct // 2 else 0)
            dom_chans = min(nchan, ntile * ct) if ntile else nchan
            if not ntile:
                ct = 16 if nchan <= 16 else 32
            name = "dft_mfma_kernel<%d,%s>" % (ct, str(self.cplx).lower())
            nstep = -(-nsrc // 4)
            # algorithmic HBM bytes of that launch: 64 B written per vis + uvw 24 B/row + its
            # records ((CT x {1 real | 3 complex: Re, Im, -Im} + 1 header) x 16 doubles per tile and 4-source step)
            alg_bytes = nrow * dom_chans * ncorr * 16 + nrow * 24 + max(ntile, 1) * nstep * (ct * (3 if self.cplx else 1) + 1) * 16 * 8
            if getattr(self, "fused_chi2", False):
                # chi^2 in the epilogue: the launch also reads the observed data (64 B per vis).  Its PMC traffic is
                # higher by another 64 B per vis: the epilogue reads the visibilities it has just stored back
                # (keeping them in registers costs the second wave per SIMD) -- on a bus used at < 10 %
                alg_bytes += nrow * dom_chans * ncorr * 16
        else:
            dom_chans = nchan
            name = "dft_exact_kernel"
            alg_bytes = nrow * nchan * ncorr * 16 + nrow * 24 + nsrc * nchan * ncorr * px
        # algorithmic flops per (row, chan, src): one phasor step by the three-term recurrence (2 FMA: re, im)
        # + ncorr MACs: complex x real pixel = 2 FMA, complex x complex = 4 FMA.  FMA = 2 flop.
        fma = 2 + ncorr * (4 if self.cplx else 2)
        alg_flops = float(nrow) * dom_chans * nsrc * fma * 2
        return dict(kernel=name, bound="mfma", alg_flops=alg_flops, alg_bytes=float(alg_bytes),
                    channels_in_kernel=dom_chans,
                    note="fp64-pipe bound (MFMA f64 and VALU f64 share one 78.6 TFLOP/s pipe on gfx950), not "
                         "HBM-bound: nsrc phasors per 64-byte visibility; %d flop per (row, chan, src)" % (2 * fma)
                         + ("; the kernel time includes the step's chi^2 epilogue (reads the observed data, +0.6 ms at the "
                            "default shape: the transform alone is 3 % higher in frac)" if getattr(self, "fused_chi2", False) and mfma else ""))

    def cpu_baseline(self, min_seconds):
        import oracle
        nchan, nsrc = self.freq.shape[0], self.lm.shape[0]
        threads = _threads()

        def single(n):
            oracle.im_to_vis(self.image, self.uvw[:n], self.lm, self.freq, omp=False)

        def parallel(n):
            t0 = time.perf_counter()
            oracle.im_to_vis(self.image, self.uvw[:n], self.lm, self.freq, omp=True)
            return time.perf_counter() - t0

        s = sized_cpu_sample(single, parallel, self.uvw.shape[0], threads, min_seconds)
        one = nchan / s["per_row_s"] / 1e6
        return {
            "value": s["rows"] * nchan / s["seconds"] / 1e6, "unit": "Mvis/s", "cores": threads, "kind": "port",
            "sample": "oracle im_to_vis (C restatement of africanus/dft/kernels.py:33-67, OpenMP over rows), "
                      "%d rows x %d chan x %d src x 4 corr fp64, %s image, in %.2f s on %d threads after a warm-up call; "
                      "linear in rows; single-thread probe %d rows in %.2f s = %.4f Mvis/s"
                      % (s["rows"], nchan, nsrc, "complex" if self.cplx else "real", s["seconds"], threads,
                         s["probe_rows"], s["probe_s"], one),
            "single_thread_value": one, "probe_rows": s["probe_rows"], "probe_seconds": s["probe_s"],
            "sample_rows": s["rows"], "sample_seconds": s["seconds"],
            "numba_calibration": NUMBA_CALIBRATION,
        }


class GaussDft(object):
    """Gaussian and point sources without DDEs (af_gauss_predict_c128): the reference chain phase_delay x gaussian_shape x
    brightness summed over sources (africanus/rime/examples/predict.py:107-134, model/shape/gaussian_shape.py:21-62),
    BASELINE configs[1]'s counts, three sources in four extended."""

    def __init__(self, args, rank, dev, lib, _lib, t):
        import torch
        from codex_africanus_amd.testing import synthetic_inputs
        self.args, self._lib = args, _lib
        nrow, nchan, nsrc = args.rows, args.chans, args.sources
        self.ncorr = 4
        d = synthetic_inputs(seed=args.seed, nrow=16, nchan=nchan, nsrc=nsrc, nant=64)
        rng = np.random.default_rng(1000 + args.seed + rank)
        uvw = np.empty((nrow, 3))
        uvw[:, 0] = rng.uniform(-4000, 4000, nrow)
        uvw[:, 1] = rng.uniform(-4000, 4000, nrow)
        uvw[:, 2] = rng.uniform(-400, 400, nrow)
        shapes = np.stack([rng.uniform(0, 3e-4, nsrc), rng.uniform(0, 2e-4, nsrc), rng.uniform(0, np.pi, nsrc)], axis=1)
        shapes[::4] = 0.0                 # point sources in between
        self.X = np.ascontiguousarray(np.broadcast_to(d["brightness"][:, None, :], (nsrc, nchan, 4)))
        self.uvw, self.lm, self.freq, self.shapes = uvw, d["lm"], d["frequency"], shapes
        self.dv = [t(a) for a in (self.lm, self.uvw, self.freq, self.X, self.shapes)]
        self.ws_bytes = int(lib.af_gauss_predict_workspace_bytes(nsrc, nchan))
        self.d_ws = torch.empty(max(self.ws_bytes, 256), dtype=torch.uint8, device=dev)
        self.label = "predict of Gaussian + point sources without DDEs (phase_delay x gaussian_shape x brightness, fused)"

    def predict(self, d_vis, stream, P):
        a, v = self.args, self.dv
        self._lib.call("af_gauss_predict_c128", P(v[0]), P(v[1]), P(v[2]), P(v[3]), P(v[4]), a.sources, a.rows, a.chans,
                       self._lib.CONVENTION["fourier"], P(d_vis), P(self.d_ws), self.ws_bytes, stream)

    # (af_gauss_predict_chi2_c128 -- chi^2 in the kernel's epilogue -- is not used for the step: with 64-channel tiles, one
    # wave per SIMD, the epilogue costs what the separate pass costs: 48.99 + 0.08 against 47.65 + 1.40 ms)

    def _chain(self, uvw):
        import oracle
        ks = oracle.phase_delay(self.lm, uvw, self.freq) * oracle.gaussian_shape(uvw, self.freq, self.shapes)
        return np.einsum("srf,sfc->rfc", ks, self.X)

    def reference_rows(self, rows):
        return self._chain(self.uvw[rows]), rows

    def roofline(self, kernel_s):
        a = self.args
        nrow, nchan, nsrc = a.rows, a.chans, a.sources
        mfma = nchan >= 14
        # per (row, chan, src): the phasor step (2 FMA), the envelope (e, r and the two products: 4 multiplies) and
        # four complex x complex MACs (16 FMA)
        alg_flops = float(nrow) * nchan * nsrc * (2 * 2 + 4 + 16 * 2)
        alg_bytes = float(nrow) * nchan * 64 + nrow * 24.0 + nsrc * nchan * 64.0
        return dict(kernel="dft_mfma_kernel<64,true,false,true>" if mfma else "gauss_dft_kernel", bound="mfma",
                    alg_flops=alg_flops, alg_bytes=alg_bytes, channels_in_kernel=nchan,
                    note="fp64-pipe bound: 40 flop per (row, chan, src) = phasor recurrence + envelope recurrence + 4 complex MACs")

    def cpu_baseline(self, min_seconds):
        threads = _threads()

        def single(n):
            self._chain(self.uvw[:n])

        def parallel(n):
            return _parallel_rows(lambda lo, hi: self._chain(self.uvw[lo:hi]), n, threads)

        s = sized_cpu_sample(single, parallel, min(self.uvw.shape[0], 4096), threads, min_seconds)
        nchan = self.freq.shape[0]
        return {"value": s["rows"] * nchan / s["seconds"] / 1e6, "unit": "Mvis/s", "cores": threads, "kind": "port",
                "sample": "oracle phase_delay (C) x gaussian_shape x einsum over sources (numpy), %d rows x %d chan x %d src in "
                          "%.2f s on %d threads" % (s["rows"], nchan, self.lm.shape[0], s["seconds"], threads),
                "single_thread_value": nchan / s["per_row_s"] / 1e6}


class DftF32(object):
    """im_to_vis with every input float32 -> complex64 (af_im_to_vis_f32: fp64 phases, float32 phasors and sums): the
    single-precision call of africanus/dft/kernels.py:26-31, at BASELINE configs[1]'s counts.  Not the headline (that is
    fp64); its step has no chi^2 (the chi^2 entry is complex128).  Errors are against the float64 transform of the
    same float32 inputs."""
    vis_dtype, chi2 = "complex64", False

    def __init__(self, args, rank, dev, lib, _lib, t):
        import torch
        from codex_africanus_amd.testing import synthetic_inputs, real_image
        self.args, self._lib = args, _lib
        nrow, nchan, nsrc = args.rows, args.chans, args.sources
        self.ncorr = 4
        d = synthetic_inputs(seed=args.seed, nrow=16, nchan=nchan, nsrc=nsrc, nant=64)
        rng = np.random.default_rng(1000 + args.seed + rank)
        uvw = np.empty((nrow, 3), np.float32)
        uvw[:, 0] = rng.uniform(-4000, 4000, nrow)
        uvw[:, 1] = rng.uniform(-4000, 4000, nrow)
        uvw[:, 2] = rng.uniform(-400, 400, nrow)
        self.image, self.uvw = real_image(d).astype(np.float32), uvw
        self.lm, self.freq = d["lm"].astype(np.float32), d["frequency"].astype(np.float32)
        self.dv = [t(a) for a in (self.image, self.uvw, self.lm, self.freq)]
        self.ws_bytes = int(lib.af_im_to_vis_f32_workspace_bytes(nsrc, nchan, 4, 0))
        self.d_ws = torch.empty(max(self.ws_bytes, 256), dtype=torch.uint8, device=dev)
        self.label = "im_to_vis DFT predict in single precision (float32 in, complex64 out; BASELINE configs[1]'s counts)"

    def predict(self, d_vis, stream, P):
        a, v = self.args, self.dv
        self._lib.call("af_im_to_vis_f32", P(v[0]), 0, P(v[1]), P(v[2]), P(v[3]), a.sources, a.rows, a.chans, 4,
                       self._lib.CONVENTION["fourier"], self._lib.AF_DFT_AUTO, P(d_vis), P(self.d_ws), self.ws_bytes, stream)

    def reference_rows(self, rows):
        import oracle
        f = lambda x: x.astype(np.float64)
        return oracle.im_to_vis(f(self.image), f(self.uvw[rows]), f(self.lm), f(self.freq), omp=True), rows

    def roofline(self, kernel_s):
        a = self.args
        units = float(a.rows) * a.chans * a.sources
        # per (row, chan, src), kilometre baselines (the float32 "chain" form): one complex rotation of the recurrence
        # (2 mul + 2 fma = 6 flop, two packed instructions) on the fp32 VALU + 8 fp32 MACs (16 flop) on the matrix pipe
        # (v_mfma_f32_4x4x1_16b: 8 issue cycles per 256 MACs = the fp32 vector rate): 22 flop per unit against the fp32
        # peak (the band correction of a rounded float32 axis, 2 more fma, is overhead, not algorithm)
        return dict(kernel="dft_f32_kernel<16,4,false,true,true>", bound="mfma", alg_flops=units * 22.0,
                    peak_tflops=FP32_PEAK_TFLOPS,
                    alg_bytes=float(a.rows) * a.chans * 32 + a.rows * 12.0 + a.sources * a.chans * 16.0,
                    channels_in_kernel=a.chans,
                    note="single precision: fp64 phase -> float32 anchor and step phasors, packed float32 rotation recurrence "
                         "(VALU) + fp32 MACs (matrix pipe, 4x4x1 blocks, pixels broadcast by CBSZ/ABID); 22 flop per "
                         "(row, chan, src) against the fp32 peak (157.3 TFLOP/s)")

    def cpu_baseline(self, min_seconds):
        import oracle
        threads = _threads()
        f = lambda x: x.astype(np.float64)
        img, lm, fr = f(self.image), f(self.lm), f(self.freq)

        def parallel(n):
            t0 = time.perf_counter()
            oracle.im_to_vis(img, f(self.uvw[:n]), lm, fr, omp=True, dtype=np.complex64)
            return time.perf_counter() - t0

        s = sized_cpu_sample(lambda n: oracle.im_to_vis(img, f(self.uvw[:n]), lm, fr, omp=False, dtype=np.complex64), parallel,
                             self.uvw.shape[0], threads, min_seconds)
        one = a_chans = self.args.chans / s["per_row_s"] / 1e6
        return {"value": s["rows"] * self.args.chans / s["seconds"] / 1e6, "unit": "Mvis/s", "cores": threads, "kind": "port",
                "sample": "oracle im_to_vis with complex64 accumulation (the reference's dtype=complex64 loop, fp64 phases), "
                          "%d rows in %.2f s on %d threads" % (s["rows"], s["seconds"], threads),
                "single_thread_value": one}

