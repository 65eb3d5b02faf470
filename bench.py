#!/usr/bin/env python
"""
bench.py -- benchmarks of the RIME visibility-predict hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input resident in HBM:
    vis = predict(...)                               # the workload's transform (below)
    chi2[nu] = sum |data - vis|^2                    # per-channel chi^2 of the shard
    (N > 1) all-reduce of chi2                        # the only cross-GPU exchange of the path
PER GPU (rows shard across GPUs, weak scaling: BASELINE configs[3] is 8e6 rows on 8 GPUs).
Metric: Mvis/s = rows x chans / second / 1e6 (whole job).

Workloads (--workload), one per BASELINE config a single GPU can run:
    dft          im_to_vis, real 4-correlation image: BASELINE configs[1], the HEADLINE and the default
                 (1e6 rows x 64 chan x 1000 point sources x 4 corr, fp64)
    dft_complex  the same transform with complex brightness matrices (= the fused predict without DDEs)
    dft_f32      the same transform for single-precision callers (float32 in, complex64 out; af_im_to_vis_f32)
    fused_dde    fused predict with per-antenna beam-cube DDEs, 64 antennas: BASELINE configs[2]
    fused_dde_ant  the same with antenna-decomposable uvw (uvw_pq = uvw_p - uvw_q, as in a Measurement Set): one complex
                 GEMM per (timestep, channel) on the fp64 matrix cores (= fused_dde --uvw antennas)
    degrid       convolutional degridding of a 4096^2 grid, 1e6 rows x 64 chan, 7x7 taps: BASELINE configs[4]
    wgrid        wgridder-style degridding of a 4096^2 image at epsilon 1e-5: BASELINE configs[4] as named
The default run (N = 1, headline shape) also times every other workload for a few steps and reports them under
"workloads" in the same JSON line (--extras none switches that off; --extras a,b picks some).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload W] [--rows R --chans C --sources S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: rows shard, nothing else moves.  Two executors:
  --executor ranks    (default) one process per GPU, torch.distributed, chi^2 all-reduced by RCCL over xGMI.  Under
                      a launcher (WORLD_SIZE set) this process is one rank.  WITHOUT a launcher `--gpus N` starts
                      its N rank processes itself -- before this process makes any GPU call -- relays rank 0's
                      JSON line and exits non-zero if any rank failed: it never falls through to one rank.
  --executor threads  ONE process, N worker threads, row block k on device k % N through
                      codex_africanus_amd.placement.block(k) -- the reference's "dask chunks on a thread pool"
                      (africanus/rime/dask_predict.py:311-369, africanus/dft/dask.py:37-51) mapped to the GPUs of
                      one node; the chi^2 partials are peer-copied to the first device and summed there.
Fewer visible devices than N is an error unless AFHIP_BENCH_DEVICE=d puts every rank / worker on device d
(then the ranks use gloo: RCCL refuses two ranks on one device).  "n_gpus" is the number of ranks / workers that
actually reported.

Rank 0 prints ONE JSON line.  Besides the driver's contract it carries
  "roofline"     for the workload's dominant kernel: algorithmic flops (or bytes) per launch / its average
                 duration measured with HIP events on the stream it is launched on (the library's measurement
                 hook af_profile_events brackets exactly that kernel), against the peak that bounds it
                 ("Rooflines"); "traffic" = HBM bytes per launch from the PMC passes of the same
                 command committed under profiles/ (null when no such summary exists);
  "cpu_baseline" the CPU oracle (C restatement of the reference's numba loops) timed on this box's host
                 cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # 256 CU x 4 SIMD x 16 FMA lanes/clk x 2 flop x 2.4 GHz, vector or matrix
FP32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32 vector = fp32-input MFMA peak (64 flop/clk/SIMD)
L2_PEAK_GBS = 34500.0          # MI355X_MICROARCH.md: aggregate L2 bandwidth (degridder's gather view)
PMC_ROUNDS = ("r04", "r03", "r02")    # profiles/<round>_<workload>_pmc_summary.json, newest first
# the REAL reference (numba 0.54) measured in the build container: im_to_vis 10k x 16 x 100 x 4
# on one core 0.263 Mvis/s = 38 ns per (row, chan, src); linear in sources -> 0.026 Mvis/s/core at 1000 sources
NUMBA_CALIBRATION = {"value": 0.026, "unit": "Mvis/s per core at 1000 sources",
                     "source": "SURVEY.md section 6: africanus.dft.im_to_vis under numba on one Xeon core of the build "
                               "container, 38 ns per (row, chan, src); not measurable on the GPU box (no numba there)"}
EXTRA_WORKLOADS = ("dft_complex", "dft_f32", "gauss", "fused_dde", "fused_dde_ant", "degrid", "wgrid", "wgrid_f32planes")
DEFAULT_SHAPE = dict(rows=1000000, chans=64, sources=1000, mode="auto", pa="random", npix=4096)


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--rows", type=int, default=DEFAULT_SHAPE["rows"], help="rows PER GPU")
    p.add_argument("--chans", type=int, default=DEFAULT_SHAPE["chans"])
    p.add_argument("--sources", type=int, default=DEFAULT_SHAPE["sources"])
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--mode", default="auto", choices=["auto", "exact", "recurrence"])
    p.add_argument("--workload", default="dft", choices=["dft", "dft_complex", "dft_f32", "gauss", "fused_dde", "fused_dde_ant", "degrid", "wgrid",
                                                         "wgrid_f32planes"])
    p.add_argument("--uvw", default="random", choices=["random", "antennas"],
                   help="fused_dde: uvw drawn per row (BASELINE's recipe: not antenna-decomposable, lane-per-row kernel) or "
                        "differences of per-(time, antenna) coordinates as in a Measurement Set (the GEMM form on the "
                        "matrix cores); workload fused_dde_ant = fused_dde --uvw antennas")
    p.add_argument("--extras", default="auto",
                   help="other workloads timed for a few steps into \"workloads\" of the same JSON line: auto (all of "
                        "them when N = 1, the workload is the headline and the shape is the default), all, none, or a "
                        "comma list of " + ",".join(EXTRA_WORKLOADS))
    p.add_argument("--extra-steps", type=int, default=5)
    p.add_argument("--executor", default="ranks", choices=["ranks", "threads"],
                   help="N > 1: one process per GPU (torch.distributed) or one process with N worker threads")
    p.add_argument("--pa", default="random", choices=["random", "common"],
                   help="fused_dde: parallactic angles iid U(0, pi/6) per (time, antenna) (SURVEY 8(d), the "
                        "reference's own test recipe) or one angle per timestep + 1e-3 rad antenna jitter "
                        "(a real array: coherent beam gathers)")
    p.add_argument("--npix", type=int, default=DEFAULT_SHAPE["npix"], help="degrid / wgrid: grid size")
    p.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"],
                   help="torch.distributed backend; nccl = RCCL over xGMI.  auto = nccl, or gloo when "
                        "AFHIP_BENCH_DEVICE puts the ranks on one device (the N > 1 code path on a one-GPU box)")
    p.add_argument("--force-dist", action="store_true",
                   help="N = 1: still create the process group (world size 1) and all-reduce the chi^2 vector every step -- "
                        "on a one-GPU box this loads librccl, builds a communicator and runs the collective on the device")
    p.add_argument("--launch-timeout", type=int, default=3600, help="self-launched ranks: seconds before they are killed")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=2.0,
                   help="minimum wall time of the all-cores CPU baseline sample (the single-thread probe runs "
                        ">= a quarter of it)")
    p.add_argument("--check-rows", type=int, default=256, help="rows checked against the oracle")
    return p.parse_args(argv)


def _threads():
    import oracle
    return oracle.num_threads(omp=True)


def _parallel_rows(fn, nrows, threads):
    """Run fn(lo, hi) on `threads` host threads over equal row blocks (the oracle's C loops release the
    GIL): the reference's own parallelism is exactly this, dask row chunks on a thread pool."""
    from concurrent.futures import ThreadPoolExecutor
    edges = np.linspace(0, nrows, threads + 1).astype(np.int64)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda k: fn(int(edges[k]), int(edges[k + 1])), range(threads)))
    return time.perf_counter() - t0


def sized_cpu_sample(single, parallel, max_rows, threads, min_parallel_s):
    """Sizes and times a CPU-baseline sample: `single(n)` runs n rows on one thread,
    `parallel(n)` runs n rows on all `threads` and returns its wall time.
      1. one discarded warm-up call (library paged in, OpenMP pool started, inputs touched);
      2. single-thread probe grown until it runs >= max(0.5 s, min_parallel_s / 4);
      3. all-threads sample grown until it runs >= min_parallel_s.
    Returns dict(per_row_s, probe_rows, probe_s, rows, seconds)."""
    min_probe_s = max(0.5, min_parallel_s / 4.0) if min_parallel_s >= 1.0 else min_parallel_s / 2.0
    single(min(16, max_rows))
    n, dt = min(16, max_rows), 0.0
    for _ in range(8):
        t0 = time.perf_counter()
        single(n)
        dt = time.perf_counter() - t0
        if dt >= min_probe_s or n >= max_rows:
            break
        n = int(min(max_rows, max(2 * n, 1.25 * n * min_probe_s / max(dt, 1e-5))))
    per_row, probe_rows, probe_s = dt / n, n, dt
    q = max(threads, 1)
    rows = int(min(max_rows, max(q * 8, 0.1 * min_parallel_s * q / per_row)))
    rows = max(q, rows - rows % q)
    parallel(min(rows, q * 2))                       # warm the worker threads
    sec = 0.0
    for _ in range(6):
        sec = parallel(rows)
        if sec >= min_parallel_s or rows >= max_rows - max_rows % q:
            break
        rows = int(min(max_rows, max(2 * rows, 1.25 * rows * min_parallel_s / max(sec, 1e-5))))
        rows = max(q, rows - rows % q)
    return dict(per_row_s=per_row, probe_rows=probe_rows, probe_s=probe_s, rows=rows, seconds=sec)


# ------------------------------------------------------------------------------------------ workloads
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


class FusedDde(object):
    """BASELINE configs[2]: 64 antennas, 2016 baselines per timestep, beam cube 257 x 257 x 33
    x 2 x 2 complex128, parallactic angles U(0, pi/6), pointing errors 1e-3 N(0,1), antenna scaling 1 +- 1e-3;
    brightness = flat-spectrum coherency matrices of the synthetic sky.  The reference chain it replaces:
    phase_delay -> einsum -> beam_cube_dde -> predict_vis (africanus/rime/examples/predict.py:404-525)."""
    NANT, LW, MH, NUD = 64, 257, 257, 33

    def __init__(self, args, rank, dev, lib, _lib, t):
        import torch
        from codex_africanus_amd.testing import synthetic_inputs
        self.args, self._lib = args, _lib
        nrow, nchan, nsrc, nant = args.rows, args.chans, args.sources, self.NANT
        d = synthetic_inputs(seed=args.seed, nrow=16, nchan=nchan, nsrc=nsrc, nant=64)
        rng = np.random.default_rng(1000 + args.seed + rank)
        uvw = np.empty((nrow, 3))
        uvw[:, 0] = rng.uniform(-4000, 4000, nrow)
        uvw[:, 1] = rng.uniform(-4000, 4000, nrow)
        uvw[:, 2] = rng.uniform(-400, 400, nrow)
        lm, freq = d["lm"], d["frequency"]
        a1, a2 = np.triu_indices(nant, 1)
        nbl = a1.shape[0]
        ntime = -(-nrow // nbl)
        ant1 = np.tile(a1, ntime)[:nrow].astype(np.int32)
        ant2 = np.tile(a2, ntime)[:nrow].astype(np.int32)
        time_index = np.repeat(np.arange(ntime, dtype=np.int64), nbl)[:nrow]
        self.antennas = args.workload == "fused_dde_ant" or getattr(args, "uvw", "random") == "antennas"
        if self.antennas:
            # a Measurement Set's uvw: per-(time, antenna) coordinates, baselines are their differences (same extent as
            # the per-row recipe: |u|, |v| <= 4000 m, |w| <= 400 m)
            xyz = rng.uniform(-1, 1, (ntime, nant, 3)) * np.array([2000.0, 2000.0, 200.0])
            uvw = xyz[time_index, ant1] - xyz[time_index, ant2]
        g = np.linspace(-1, 1, self.LW)
        ll, mm = np.meshgrid(g, g, indexing="ij")
        pattern = np.exp(-(ll**2 + mm**2) / 0.5) * np.exp(1j * (0.3 * ll + 0.2 * mm))
        gains = (1 + 0.02 * np.arange(self.NUD))[:, None] * np.array([1.0, 0.05j, -0.04j, 0.95])[None, :]
        beam = (pattern[:, :, None, None] * gains[None, None]).reshape(self.LW, self.MH, self.NUD, 2, 2)
        extents = np.array([[-0.06, 0.06], [-0.06, 0.06]])
        beam_freq_map = np.linspace(freq[0], freq[-1], self.NUD)
        pa = rng.uniform(0, np.pi / 6, (ntime, nant))
        if args.pa == "common":
            pa = np.linspace(0, np.pi / 6, ntime)[:, None] + 1e-3 * rng.standard_normal((ntime, nant))
        pe = 1e-3 * rng.standard_normal((ntime, nant, nchan, 2))
        asc = 1.0 + 1e-3 * rng.standard_normal((nant, nchan, 2))
        X = np.ascontiguousarray(np.broadcast_to(d["brightness"][:, None, :], (nsrc, nchan, 4))).reshape(nsrc, nchan, 2, 2)
        # the plan (host side, once per row layout): 2 x 2 blocks of baselines that share their antennas' Jones terms
        # (AFHIP_FUSED_GROUPS=0: plain row ranges, for A/B runs)
        n_items, n_groups = ctypes.c_int64(0), ctypes.c_int64(0)
        tip = time_index.ctypes.data_as(ctypes.c_void_p)
        pa1, pa2 = ant1.ctypes.data_as(ctypes.c_void_p), ant2.ctypes.data_as(ctypes.c_void_p)
        if os.environ.get("AFHIP_FUSED_GROUPS", "1") != "0":
            _lib.call("af_fused_plan_groups", tip, pa1, pa2, nrow, nant, None, 0, ctypes.byref(n_items), None, 0,
                      ctypes.byref(n_groups))
            items = np.zeros((n_items.value, 4), dtype=np.int32)
            groups = np.zeros((n_groups.value, 8), dtype=np.int32)
            _lib.call("af_fused_plan_groups", tip, pa1, pa2, nrow, nant, items.ctypes.data_as(ctypes.c_void_p),
                      n_items.value, ctypes.byref(n_items), groups.ctypes.data_as(ctypes.c_void_p), n_groups.value,
                      ctypes.byref(n_groups))
        else:
            groups = None
            _lib.call("af_fused_plan_rows", tip, nrow, None, 0, ctypes.byref(n_items))
            items = np.zeros((n_items.value, 4), dtype=np.int32)
            _lib.call("af_fused_plan_rows", tip, nrow, items.ctypes.data_as(ctypes.c_void_p), n_items.value,
                      ctypes.byref(n_items))
        self.n_items, self.ntime, self.nbl = n_items.value, ntime, nbl
        if self.antennas:
            nap = 8 * ((nant + 7) // 8)
            au, rm = np.zeros((ntime, nant, 3)), np.zeros((ntime, nap, nap), np.int32)
            res, ok = ctypes.c_double(), ctypes.c_int()
            HP = lambda x: x.ctypes.data_as(ctypes.c_void_p)
            _lib.call("af_fused_plan_antennas", tip, pa1, pa2, HP(uvw), nrow, nant, 1e-10, ntime, HP(au), HP(rm),
                      ctypes.byref(res), ctypes.byref(ok))
            if not ok.value:
                raise SystemExit("fused_dde_ant: the synthetic uvw did not decompose (residual %g m)" % res.value)
            self.plan_residual = res.value
            self.d_au, self.d_rm = t(au), t(rm)
        self.dv = dict(items=t(items), groups=None if groups is None else t(groups), a1=t(ant1), a2=t(ant2), X=t(X), beam=t(beam), ext=t(extents),
                       fmap=t(beam_freq_map), pa=t(pa), pe=t(pe), asc=t(asc), lm=t(lm), uvw=t(uvw), freq=t(freq))
        self.ws_bytes = int(lib.af_fused_predict_workspace_bytes(nsrc, nchan, self.LW, self.MH, self.NUD))
        self.d_ws = torch.empty(max(self.ws_bytes, 256), dtype=torch.uint8, device=dev)
        self.h = dict(time_index=time_index, ant1=ant1, ant2=ant2, X=X, beam=beam, extents=extents,
                      beam_freq_map=beam_freq_map, pa=pa, pe=pe, asc=asc, lm=lm, uvw=uvw, freq=freq)
        self.ncorr = 4
        self.label = ("fused predict with per-antenna beam-cube DDEs, 64 antennas (BASELINE configs[2]), "
                      "parallactic angles %s" % args.pa)
        if self.antennas:
            self.label += "; antenna-decomposable uvw (Measurement-Set geometry): GEMM form on the fp64 matrix cores"

    def predict(self, d_vis, stream, P):
        a, v = self.args, self.dv
        if self.antennas:
            self._lib.call("af_fused_predict_antennas_c128", P(self.d_au), P(self.d_rm), self.ntime, a.rows, P(v["lm"]),
                           P(v["freq"]), P(v["X"]), a.sources, a.chans, P(v["beam"]), self.LW, self.MH, self.NUD, P(v["ext"]),
                           P(v["fmap"]), P(v["pa"]), self.ntime, self.NANT, P(v["pe"]), P(v["asc"]), None,
                           self._lib.CONVENTION["fourier"], P(d_vis), P(self.d_ws), self.ws_bytes, stream)
            return
        self._lib.call("af_fused_predict_c128", P(v["items"]), self.n_items, P(v["a1"]), P(v["a2"]),
                       None if v["groups"] is None else P(v["groups"]), a.rows,
                       P(v["lm"]), P(v["uvw"]), P(v["freq"]), P(v["X"]), a.sources, a.chans, P(v["beam"]), self.LW,
                       self.MH, self.NUD, P(v["ext"]), P(v["fmap"]), P(v["pa"]), self.ntime, self.NANT, P(v["pe"]),
                       P(v["asc"]), None, None, self._lib.CONVENTION["fourier"], P(d_vis), P(self.d_ws),
                       self.ws_bytes, stream)

    def front_end_check(self, d_vis, rank, world, dev):
        """The row-shard front-end a multi-GPU job goes through -- sharding.fused_predict_shard with this rank's rows
        and timesteps (bounds given) -- on the arrays of the benchmark: its visibilities must equal the direct C-ABI
        call's (d_vis) in every bit.  One extra predict before the timed region."""
        import torch
        from codex_africanus_amd import sharding
        v, a = self.dv, self.args
        ti = torch.from_numpy(self.h["time_index"]).to(dev)
        vis, _, bounds = sharding.fused_predict_shard(
            rank, world, ti, v["a1"], v["a2"], v["lm"], v["uvw"], v["freq"], v["X"], v["beam"], v["ext"], v["fmap"],
            v["pa"], v["pe"], v["asc"], bounds=(rank * a.rows, (rank + 1) * a.rows))
        same = bool(torch.equal(vis.reshape(d_vis.shape), d_vis))
        if not same:
            raise SystemExit("rank %d: sharding.fused_predict_shard differs from the C-ABI call" % rank)
        return "sharding.fused_predict_shard(rank %d of %d, rows %s) == the direct C-ABI call (%s): bit-equal" % (
            rank, world, bounds, "af_fused_predict_antennas_c128" if self.antennas else "af_fused_predict_c128")

    def _chain(self, rows, dde=None, tinv=None):
        """The reference chain on `rows` (only their timesteps' Jones terms are built)."""
        import oracle
        h = self.h
        if dde is None:
            tsel, tinv = np.unique(h["time_index"][rows], return_inverse=True)
            dde = oracle.beam_cube_dde(h["beam"], h["extents"], h["beam_freq_map"], h["lm"], h["pa"][tsel],
                                       h["pe"][tsel], h["asc"], h["freq"])
        phase = oracle.phase_delay(h["lm"], h["uvw"][rows], h["freq"])
        coh = np.einsum("srf,sfij->srfij", phase, h["X"])
        return oracle.predict_vis(tinv, h["ant1"][rows], h["ant2"][rows], dde, coh, dde, None, None, None)

    def reference_rows(self, rows):
        # the oracle's beam terms cost ~2.4 s per timestep on one host thread (1000 sources x 64 antennas x 64 channels):
        # check 10 rows of each of THREE timesteps (first, middle, last) instead of rows spread over all of them
        a = self.args
        picks = []
        for t in sorted({0, self.ntime // 2, self.ntime - 1}):
            lo, hi = t * self.nbl, min((t + 1) * self.nbl, a.rows)
            if hi > lo:
                picks.append(np.linspace(lo, hi - 1, min(10, hi - lo)).astype(np.int64))
        rows = np.unique(np.concatenate(picks))
        return self._chain(rows).reshape(len(rows), a.chans, 4), rows

    def roofline(self, kernel_s):
        a = self.args
        nrow, nchan, nsrc = a.rows, a.chans, a.sources
        # 64 B written per vis + uvw and indices 36 B/row + the beam cube (with |.|: 24 B per complex)
        # + parangles / pointing errors / scaling + brightness; ~150 flop per (row, chan, src): phasor 8 +
        # E X E^H 112 + 4 complex MACs 32 (SURVEY's count, kept so that rounds compare)
        alg_bytes = (nrow * nchan * 64 + nrow * 36 + self.LW * self.MH * self.NUD * 4 * 24
                     + self.ntime * self.NANT * (8 + nchan * 16) + self.NANT * nchan * 16 + nsrc * nchan * 64)
        # what the kernel actually issues (counted in the ISA of the unrolled, grouped, wave-specialised instantiation:
        # tools/count_fused_isa.sh): 63 fp64 VALU instructions per (row, chan, src) in the accumulating waves + 344
        # per 512 Jones terms in the sampling waves (one term per 31.5 units at 64 antennas); an fp64 instruction
        # occupies its SIMD for 4 cycles, so the pipe's capacity is 256 CU x 4 SIMD x 16 lanes x clock lane-instructions/s
        units = float(nrow) * nchan * nsrc
        terms = float(nsrc) * self.ntime * self.NANT * nchan
        if self.antennas:
            # the GEMM form: 8 complex MACs = 64 flop per (row, chan, source) of needed output; executed: 36 of the 64
            # 16 x 16 tiles of M per (timestep, channel, source), 2 MFMA 16x16x4 (2048 flop each) per tile
            mfma_flops = 36.0 * 2 * 2048 * nsrc * self.ntime * nchan
            executed = {"mfma_flop_per_unit": mfma_flops / units, "mfma_tflops": mfma_flops / kernel_s / 1e12,
                        "mfma_pipe_occupancy_at_2.4GHz": mfma_flops / kernel_s / 1e12 / FP64_PEAK_TFLOPS,
                        "note": "matrix-core flops actually issued (upper block triangle incl. the diagonal blocks' lower "
                                "halves and baselines a short last timestep lacks) against the 78.6 TFLOP/s fp64 pipe"}
            return dict(kernel="fused_gemm3_kernel", bound="mfma", alg_flops=units * 64.0, alg_bytes=float(alg_bytes),
                        channels_in_kernel=nchan, executed=executed,
                        note="antenna-decomposable uvw: V(t, nu) = G H^H, M = N = 128, K = 2 nsrc per (timestep, channel) on "
                             "v_mfma_f64_16x16x4; 64 flop per (row, chan, src) (8 complex MACs) against the fp64 pipe")
        fp64_lane_instr = 63.0 * units + (344.0 * 64 / 512) * terms
        cap = 256 * 4 * 16 * 2.4e9
        executed = {"fp64_instructions_per_unit": fp64_lane_instr / units, "flop_equivalent_per_unit": 2 * fp64_lane_instr / units,
                    "fp64_pipe_occupancy_at_2.4GHz": fp64_lane_instr / kernel_s / cap,
                    "note": "fraction of the fp64 pipe's issue slots (4 cycles per wave instruction) the kernel fills at the "
                            "nominal 2.4 GHz; the chip holds ~2.03 GHz under this all-VALU fp64 mix, i.e. x 1.18 at the "
                            "clock it runs at"}
        return dict(kernel="fused_predict_kernel", bound="mfma", alg_flops=float(nrow) * nchan * nsrc * 150.0,
                    alg_bytes=float(alg_bytes), channels_in_kernel=nchan, executed=executed,
                    note="fp64 VALU bound (same 78.6 TFLOP/s fp64 pipe as the matrix path): 2x2 complex Jones "
                         "algebra per (row, chan, src), 150 flop (SURVEY 8(d))")

    def cpu_baseline(self, min_seconds):
        """One timestep of the workload through the oracle chain: beam_cube_dde for the timestep's 64 antennas
        (single thread, as the reference's numba kernel), then phase_delay -> einsum -> predict_vis on a row
        sample spread over the host threads (dask row chunks in the reference); the row part is scaled to the
        timestep's 2016 rows, so the Jones terms are amortised as in the full job."""
        import oracle
        h, a = self.h, self.args
        threads = min(_threads(), 64)
        rows_t = np.arange(min(self.nbl, a.rows))
        oracle.beam_cube_dde(h["beam"], h["extents"], h["beam_freq_map"], h["lm"][:8], h["pa"][:1], h["pe"][:1],
                             h["asc"], h["freq"])                                        # warm-up, discarded
        t0 = time.perf_counter()
        dde = oracle.beam_cube_dde(h["beam"], h["extents"], h["beam_freq_map"], h["lm"], h["pa"][:1], h["pe"][:1],
                                   h["asc"], h["freq"])
        t_beam = time.perf_counter() - t0
        tinv = np.zeros(len(rows_t), dtype=np.int64)
        self._chain(rows_t[:2], dde, tinv[:2])                                           # warm-up, discarded
        n1 = min(16, len(rows_t))
        t0 = time.perf_counter()
        self._chain(rows_t[:n1], dde, tinv[:n1])
        per_row = (time.perf_counter() - t0) / n1
        per_thread = int(max(2, min(32, 0.3 * min_seconds / per_row)))                   # coh: 4 MB per row
        n = min(len(rows_t), per_thread * threads)
        dt = _parallel_rows(lambda lo, hi: self._chain(rows_t[lo:hi], dde, tinv[lo:hi]) if hi > lo else None, n, threads)
        t_step = t_beam + dt * len(rows_t) / n
        return {
            "value": len(rows_t) * a.chans / t_step / 1e6, "unit": "Mvis/s", "cores": threads, "kind": "port",
            "sample": "oracle chain beam_cube_dde -> phase_delay -> einsum -> predict_vis (C restatements of "
                      "africanus/rime/fast_beam_cubes.py:57-240, phase.py:20-63, predict.py:193-252) for ONE "
                      "timestep (%d rows x %d chan x %d src, 64 antennas): beam terms %.2f s on 1 thread + %d rows on "
                      "%d threads in %.2f s scaled to the timestep's rows" % (len(rows_t), a.chans, a.sources,
                                                                             t_beam, n, threads, dt),
            "single_thread_value": a.chans / (per_row + t_beam / len(rows_t)) / 1e6,
            "probe_rows": n1, "sample_rows": n, "sample_seconds": dt,
        }


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


WORKLOADS = {"dft": Dft, "dft_complex": Dft, "dft_f32": DftF32, "gauss": GaussDft, "fused_dde": FusedDde, "fused_dde_ant": FusedDde, "degrid": Degrid,
             "wgrid": Wgrid,
             "wgrid_f32planes": WgridF32Planes}
METRIC = "Mvis/s (rows x chans) for predict_vis at 1e6 rows/64 ch/1000 src; fp64 max-abs err"


def pmc_traffic(workload, is_default_shape):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs of THIS command; KB units; FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950 streaming reads -- an upper bound where reads are narrower).  A constant of the committed
    profile, not a measurement of this run (counters cannot be read from inside the process): "traffic_source" says
    which file."""
    if not is_default_shape:
        return None, None
    names = ["%s_%s_pmc_summary.json" % (r, workload) for r in PMC_ROUNDS]
    names += ["r01_pmc_summary.json"] if workload == "dft" else []
    names += ["r01_fused_pmc_summary.json"] if workload == "fused_dde" else []
    for name in names:
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            c = json.load(open(path))
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, "profiles/" + name
    return None, None


def is_default_shape(args):
    return all(getattr(args, k) == v for k, v in DEFAULT_SHAPE.items())


def roofline_entry(wl, args, workload, kernel_s):
    r = wl.roofline(kernel_s)
    traffic, traffic_src = pmc_traffic(workload, is_default_shape(args))
    hbm_ach = r["alg_bytes"] / kernel_s / 1e9
    fp_ach = r["alg_flops"] / kernel_s / 1e12
    hbm = {"achieved": hbm_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_ach / HBM_PEAK_GBS,
           "algorithmic_bytes": r["alg_bytes"]}
    peak = r.get("peak_tflops", FP64_PEAK_TFLOPS)      # the pipe the workload computes on (fp64 unless it says fp32)
    fp64 = {"achieved": fp_ach, "peak": peak, "unit": "TFLOP/s", "frac": fp_ach / peak,
            "algorithmic_flops": r["alg_flops"]}
    top = fp64 if r["bound"] == "mfma" else hbm
    roof = {"kernel": r["kernel"], "bound": r["bound"], "achieved": top["achieved"], "peak": top["peak"],
            "unit": top["unit"], "frac": top["frac"], "traffic": traffic, "traffic_source": traffic_src,
            "kernel_ms": kernel_s * 1e3, "channels_in_kernel": r["channels_in_kernel"], "note": r["note"],
            "hbm": hbm, "fp64": fp64}
    for k in ("gather", "executed"):
        if k in r:
            roof[k] = r[k]
    return roof


class Events(object):
    """HIP events of the library's measurement hook (af_profile_events brackets the workload's dominant kernel on
    the stream it is launched on); one pair per timed step."""

    def __init__(self, _lib, steps):
        self._lib, self.evs = _lib, []
        for _ in range(steps):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            _lib.call("af_event_create", ctypes.byref(a))
            _lib.call("af_event_create", ctypes.byref(b))
            self.evs.append((a, b))

    def arm(self, k):
        self._lib.call("af_profile_events", self.evs[k][0], self.evs[k][1])

    def disarm(self):
        self._lib.call("af_profile_events", None, None)

    def collect(self):
        """Mean kernel seconds; destroys the events (call after the device is idle)."""
        out = []
        for a, b in self.evs:
            ms = ctypes.c_float(0)
            self._lib.call("af_event_elapsed_ms", a, b, ctypes.byref(ms))
            out.append(ms.value)
            self._lib.call("af_event_destroy", a)
            self._lib.call("af_event_destroy", b)
        self.evs = []
        return float(np.mean(out)) / 1e3 if out else float("nan")


def check_rows(wl, d_vis, nrow, n, dev):
    """max |HIP - oracle| over a row sample of the benchmarked output (checker only)."""
    import torch
    rows = np.linspace(0, nrow - 1, min(n, nrow)).astype(np.int64)
    ref, rows = wl.reference_rows(rows)
    got = d_vis[torch.from_numpy(rows).to(dev)].cpu().numpy()
    return float(np.abs(got - ref.reshape(got.shape)).max())


def measure(args, workload, steps, warmup, rank, world, dev, dist, cpu_seconds, collective=None):
    """Times `steps` steps of one workload on this rank's device (all ranks call it together).  Returns the result
    dict on rank 0, None elsewhere.  `collective`: all-reduce the chi^2 vector (default: when world > 1)."""
    import torch
    from codex_africanus_amd import _lib, sharding
    lib = _lib.load()
    collective = world > 1 if collective is None else collective
    wargs = argparse.Namespace(**vars(args))
    wargs.workload = workload
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    wl = WORKLOADS[workload](wargs, rank, dev, lib, _lib, t)
    nrow, nchan, nsrc, ncorr = args.rows, args.chans, args.sources, wl.ncorr
    have_chi2 = getattr(wl, "chi2", True)
    d_vis = torch.empty((nrow, nchan, ncorr), dtype=getattr(torch, getattr(wl, "vis_dtype", "complex128")), device=dev)
    d_chi2 = torch.zeros(nchan, dtype=torch.float64, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    # "observed" data for the chi^2: the model itself plus a fixed perturbation (one extra predict)
    wl.predict(d_vis, stream, P)
    if have_chi2:
        d_data = d_vis.clone()
        d_data += 0.01

    fused_chi2 = (have_chi2 and hasattr(wl, "predict_chi2") and getattr(wl, "chi2_in_epilogue", True)
                  and os.environ.get("AFHIP_BENCH_FUSED_CHI2", "1") != "0")
    wl.fused_chi2 = fused_chi2          # the dominant kernel then also reads the data: counted in its algorithmic bytes

    def step():
        if fused_chi2:
            wl.predict_chi2(d_vis, d_data, d_chi2, stream, P)
        else:
            wl.predict(d_vis, stream, P)
        if have_chi2:
            if not fused_chi2:
                _lib.call("af_chi2_c128", P(d_vis), P(d_data), None, nrow, nchan, ncorr, P(d_chi2), stream)
            if collective:
                sharding.allreduce_chi2(d_chi2)       # RCCL over xGMI (gloo in the one-device tests)

    # (AFHIP_FUSED_STAGE runs one stage of the fused kernels for profiling: their output is meaningless)
    staged = os.environ.get("AFHIP_FUSED_STAGE", "0") != "0"
    front_end = wl.front_end_check(d_vis, rank, world, dev) if hasattr(wl, "front_end_check") and not staged else None
    for _ in range(warmup):
        step()
    ev = Events(_lib, steps)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        ev.arm(k)
        step()
    ev.disarm()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    reported = 1
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        one = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        reported = int(round(float(one.item())))
    kernel_s = ev.collect()
    if have_chi2:        # the step's chi^2 against a separate pass over the final visibilities (checker)
        ref_chi2 = torch.zeros_like(d_chi2)
        _lib.call("af_chi2_c128", P(d_vis), P(d_data), None, nrow, nchan, ncorr, P(ref_chi2), stream)
        if collective:
            dist.all_reduce(ref_chi2, op=dist.ReduceOp.SUM)
        if not torch.allclose(d_chi2, ref_chi2, rtol=1e-10, atol=0):
            raise SystemExit("rank %d: the step's chi^2 differs from a separate pass over its visibilities" % rank)
    if rank != 0:
        return None
    max_err = check_rows(wl, d_vis, nrow, args.check_rows, dev) if args.check_rows > 0 else None
    res = {
        "label": wl.label + ("; chi^2 summed in the transform's epilogue (the entry's _chi2 form)" if fused_chi2 else ""),
        "has_chi2": have_chi2, "ranks_reported": reported, "elapsed": elapsed, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "value": reported * nrow * nchan / (elapsed / steps) / 1e6,
        "corrs": ncorr, "fp64_max_abs_err": max_err, "roofline": roofline_entry(wl, wargs, workload, kernel_s),
    }
    if front_end is not None:
        res["front_end"] = front_end
    if hasattr(wl, "end_to_end") and world == 1:
        res["end_to_end"] = wl.end_to_end()
    if cpu_seconds > 0 and world == 1:
        res["cpu_baseline"] = wl.cpu_baseline(cpu_seconds)
    return res


def headline_json(args, res, world_desc, backend_desc):
    nrow, nchan, nsrc = args.rows, args.chans, args.sources
    n = res["ranks_reported"]
    out = {
        "metric": METRIC, "value": res["value"], "unit": "Mvis/s",
        "n_gpus": n, "steps": res["steps"], "warmup": res["warmup"], "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": res["label"] + (" + per-channel chi^2" if res.get("has_chi2", True) else " (no chi^2 in the step)")
                        + ("" if n == 1 else " + " + backend_desc),
            "rows_per_gpu": nrow, "chans": nchan, "sources": nsrc, "corrs": res["corrs"],
            "rows_total": n * nrow, "phasor_mode": args.mode,
            "sharding": "rows over %d GPU(s), no data-path collective; chi2 (nchan,) all-reduce (%s)"
                        % (n, "none" if n == 1 else backend_desc),
            "executor": world_desc,
        },
        "fp64_max_abs_err": res["fp64_max_abs_err"],
        "roofline": res["roofline"],
    }
    if "cpu_baseline" in res:
        out["cpu_baseline"] = res["cpu_baseline"]
    if "front_end" in res:
        out["config"]["front_end"] = res["front_end"]
    return out


def compact_summary(out, res, extras):
    """Per-workload numbers where the driver's record keeps them (its `parsed` copy keeps the scalar
    entries of "config", "roofline" and "cpu_baseline" and a 2 000-character tail of the line; the long "workloads"
    block falls outside both).  Three copies of the same few numbers: scalar keys `<workload>_<field>` inside
    "roofline", the same table as lists under roofline["others"] ([ms_per_step, kernel_ms, frac, max_abs_err,
    Mvis/s]), and -- as the LAST key of the line, i.e. inside the tail -- "summary"."""
    roof, table = out["roofline"], {}
    for name, e in extras.items():
        if "error" in e:
            table[name] = None
            roof["%s_error" % name] = e["error"][:120]
            continue
        r = e["roofline"]
        table[name] = [round(e["ms_per_step"], 4), round(e["kernel_ms"], 4), round(r["frac"], 4), e["fp64_max_abs_err"],
                       round(e["value"], 2)]
        roof["%s_ms_per_step" % name] = e["ms_per_step"]
        roof["%s_kernel_ms" % name] = e["kernel_ms"]
        roof["%s_frac" % name] = r["frac"]
        roof["%s_bound" % name] = r["bound"]
        roof["%s_max_abs_err" % name] = e["fp64_max_abs_err"]
        for k, v in e.get("variants", {}).items():
            roof["%s_%s" % (name, k)] = v
    if table:
        roof["others"] = table
        roof["others_columns"] = "ms_per_step, kernel_ms, roofline frac, max_abs_err, Mvis/s"
    e2e = res.get("end_to_end")
    if e2e:
        out["end_to_end"] = e2e
        out["config"]["end_to_end_ms"] = e2e["ms"]
        out["config"]["end_to_end_mvis_s"] = e2e["value"]
        roof["end_to_end_ms"] = e2e["ms"]
        roof["end_to_end_mvis_s"] = e2e["value"]
    if table or e2e:
        out["summary"] = {"headline": [round(out["ms_per_step"], 4), round(roof["kernel_ms"], 4), round(roof["frac"], 4),
                                       out["fp64_max_abs_err"], round(out["value"], 2)],
                          "columns": "ms_per_step, kernel_ms, roofline frac, max_abs_err, Mvis/s",
                          "end_to_end_ms": None if not e2e else round(e2e["ms"], 3),
                          "end_to_end_mvis_s": None if not e2e else round(e2e["value"], 2)}
        out["summary"].update(table)


def extras_requested(args, world):
    e = args.extras
    if e == "none":
        return ()
    if e == "auto":
        return EXTRA_WORKLOADS if (world == 1 and args.workload == "dft" and is_default_shape(args)) else ()
    names = EXTRA_WORKLOADS if e == "all" else tuple(x for x in e.split(",") if x)
    bad = [x for x in names if x not in EXTRA_WORKLOADS]
    if bad:
        raise SystemExit("--extras: unknown workload(s) %s (choose from %s)" % (bad, ",".join(EXTRA_WORKLOADS)))
    return tuple(x for x in names if x != args.workload)


# ------------------------------------------------------------------------------------------ executor: ranks
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_environments(n, port, base_env):
    """The environment of each of the n rank processes `--gpus n` starts when no launcher did (pure arithmetic,
    pinned by tests/test_bench_launcher.py): what `torch.distributed.run --nnodes=1 --nproc-per-node n
    --master-addr 127.0.0.1` would export."""
    envs = []
    for r in range(n):
        e = dict(base_env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), GROUP_RANK="0",
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                 AFHIP_BENCH_SELF_LAUNCHED="1")
        envs.append(e)
    return envs


def visible_devices():
    """Device count without initialising the GPU in this process (torch.cuda.device_count() does not, on this
    image) -- the self-launching parent must stay GPU-free."""
    import torch
    return int(torch.cuda.device_count())


def require_devices(n, what):
    have = visible_devices()
    if have == 0:
        raise SystemExit("bench.py needs a ROCm device (the HIP path has no CPU fallback)")
    shared = os.environ.get("AFHIP_BENCH_DEVICE")
    if shared is not None:
        if not (0 <= int(shared) < have):
            raise SystemExit("AFHIP_BENCH_DEVICE=%s but %d device(s) are visible" % (shared, have))
        return have
    if have < n:
        raise SystemExit("--gpus %d (%s) but only %d device(s) are visible; refusing to report a %d-GPU number "
                         "(set AFHIP_BENCH_DEVICE=d to put every rank on device d for a functional test)"
                         % (n, what, have, have))
    return have


def supervise(procs, logs, timeout):
    """Waits for rank processes started with Popen (rank 0's stdout a pipe, `logs[r]` the file rank r > 0 writes to or
    None).  The first rank that exits non-zero ends the job: the others -- blocked in the rendezvous or the all-reduce
    it never joins -- are terminated and reaped.  Returns (exit code, rank 0's stdout, message)."""
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    def rank_log(r):
        if logs[r] is None:
            return ""
        logs[r].seek(0)
        return logs[r].read().decode("utf-8", "replace")[-2000:]

    deadline = time.time() + timeout
    failed, message = None, ""
    while any(p.poll() is None for p in procs):
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad:
            failed = bad[0]
            stop_all()
            break
        if time.time() > deadline:
            stop_all()
            reader.join(timeout=10)
            return 1, b"".join(chunks), "the ranks did not finish within %d s; stopped" % timeout
        time.sleep(0.05)
    reader.join(timeout=10)
    codes = [p.returncode for p in procs]
    if failed is None:
        bad = [(r, c) for r, c in enumerate(codes) if c]
        failed = bad[0] if bad else None
    if failed is not None:
        message = "rank %d exited with code %d; exit codes of all ranks %s\n%s" % (failed[0], failed[1], codes, rank_log(failed[0]))
        return (failed[1] if failed[1] > 0 else 1), b"".join(chunks), message
    return 0, b"".join(chunks), ""


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher: start the N ranks (children of this GPU-free process), relay
    rank 0's JSON line, fail if any rank fails."""
    import tempfile
    require_devices(args.gpus, "self-launched ranks")
    envs = rank_environments(args.gpus, free_port(), os.environ)
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs, logs = [], []
    for r, e in enumerate(envs):
        # rank 0's stdout carries the JSON line; the other ranks' output is kept (a rank that dies says why)
        log = None if r == 0 else tempfile.TemporaryFile()
        logs.append(log)
        procs.append(subprocess.Popen(cmd, env=e, cwd=ROOT, stdout=subprocess.PIPE if r == 0 else log,
                                      stderr=None if r == 0 else subprocess.STDOUT))
    code, out0, message = supervise(procs, logs, args.launch_timeout)
    text = out0.decode("utf-8", "replace")
    if code:
        sys.stderr.write("bench.py: %s\n%s\n" % (message, text[-2000:]))
        return code
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        sys.stderr.write("bench.py: rank 0 printed %d JSON line(s)\n%s\n" % (len(lines), text[-2000:]))
        return 1
    sys.stdout.write(lines[0] + "\n")
    sys.stdout.flush()
    return 0


def run_ranks(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    import torch
    import torch.distributed as dist
    have = require_devices(1 if "AFHIP_BENCH_DEVICE" in os.environ else local_rank + 1, "rank %d" % rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the HIP path has no CPU fallback)")
    dev_index = int(os.environ.get("AFHIP_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = args.backend
    if backend == "auto":
        backend = "gloo" if "AFHIP_BENCH_DEVICE" in os.environ else "nccl"
    grouped = world > 1 or args.force_dist
    if grouped:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:                       # --force-dist without a launcher: this process is the whole job
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
        # build the communicator now (RCCL sets its rings up lazily, at the first collective): the timed
        # region must not pay for it even when --warmup is 0
        warm = torch.zeros(1, dtype=torch.float64, device=dev)
        dist.all_reduce(warm)
        torch.cuda.synchronize(dev)
    cpu_s = 0.0 if args.no_cpu_baseline else args.cpu_seconds
    res = measure(args, args.workload, args.steps, args.warmup, rank, world, dev, dist, cpu_s, collective=grouped)
    if rank == 0:
        launcher = ("self-launched" if os.environ.get("AFHIP_BENCH_SELF_LAUNCHED") else "external launcher") if world > 1 else "single process"
        desc = "ranks: one process per GPU (%s), %d of %d device(s) visible in use%s" % (
            launcher, 1 if "AFHIP_BENCH_DEVICE" in os.environ else world, have,
            ", all ranks on device %s" % os.environ["AFHIP_BENCH_DEVICE"] if "AFHIP_BENCH_DEVICE" in os.environ and world > 1 else "")
        out = headline_json(args, res, desc, "RCCL all-reduce over xGMI" if backend == "nccl" else "gloo all-reduce")
        if grouped and world == 1:
            out["config"]["collective"] = ("world-size-1 process group (--force-dist), backend %s: chi2 all-reduced by "
                                           "sharding.allreduce_chi2 every step" % backend)
            if backend == "nccl":
                out["config"]["rccl_loaded"] = any("librccl" in ln for ln in open("/proc/self/maps"))
        extras = {}
        for name in extras_requested(args, world):
            # the previous workload's buffers go back to the driver before the next one allocates (a free that lands
            # inside the timed steps shows as one slow step in five); two warm-up steps
            import gc
            gc.collect()
            torch.cuda.synchronize(dev)
            torch.cuda.empty_cache()
            try:
                r = measure(args, name, max(1, min(args.extra_steps, args.steps)), 2, 0, 1, dev, dist, min(cpu_s, 1.0))
            except Exception as exc:       # an extra must never cost the headline its line
                extras[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
                continue
            roof = r["roofline"]
            extras[name] = {
                "label": r["label"] + ("" if r.get("has_chi2", True) else " (no chi^2 in the step)"), "steps": r["steps"],
                "ms_per_step": r["ms_per_step"], "value": r["value"],
                "unit": "Mvis/s", "kernel_ms": roof["kernel_ms"], "fp64_max_abs_err": r["fp64_max_abs_err"],
                "roofline": {k: roof[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                  "traffic_source")},
            }
            for k in ("gather", "executed"):
                if k in roof:
                    extras[name]["roofline"][k] = roof[k]
            if "cpu_baseline" in r:
                extras[name]["cpu_baseline"] = {k: r["cpu_baseline"][k] for k in
                                                ("value", "unit", "cores", "kind", "sample", "single_thread_value")
                                                if k in r["cpu_baseline"]}
        if extras:
            out["workloads"] = extras
        compact_summary(out, res, extras)
        print(json.dumps(out))
        sys.stdout.flush()
    if grouped:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------- executor: threads
def run_threads(args):
    """One process, N worker threads, N devices: row block k -> device k % N through placement.block(k) (the dask
    shape of africanus/rime/dask_predict.py:311-369).  Each worker's inputs are resident on its device; a step
    submits one task per row block to the thread pool, every task enqueues transform + chi^2 on its worker's own
    stream and peer-copies its chi^2 partial to the first device, where the partials are summed (stream-ordered by
    events: no host synchronisation inside a step)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from codex_africanus_amd import _lib, placement
    n = args.gpus
    if not getattr(WORKLOADS[args.workload], "chi2", True):
        raise SystemExit("--executor threads reduces the chi^2 across devices: workload %s has none" % args.workload)
    have = require_devices(n, "worker threads")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the HIP path has no CPU fallback)")
    lib = _lib.load()
    shared = os.environ.get("AFHIP_BENCH_DEVICE")
    if shared is not None:
        devs = (int(shared),) * n
    else:
        devs = placement.parse_device_list(os.environ.get("AFHIP_DEVICES"), have)[:n]
        if len(devs) < n:
            raise SystemExit("--gpus %d but AFHIP_DEVICES names %d device(s)" % (n, len(devs)))
    placement.set_devices(devs)
    placement.set_policy("block")
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    nrow, nchan = args.rows, args.chans

    class Worker(object):
        pass

    workers = []
    for k in range(n):
        w = Worker()
        w.k, w.index = k, devs[k]
        w.dev = torch.device("cuda", w.index)
        with torch.cuda.device(w.dev):
            t = lambda a, d=w.dev: torch.from_numpy(np.ascontiguousarray(a)).to(d)
            w.wl = WORKLOADS[args.workload](args, k, w.dev, lib, _lib, t)
            w.stream = torch.cuda.Stream(device=w.dev)
            w.sp = ctypes.c_void_p(w.stream.cuda_stream)
            w.d_vis = torch.empty((nrow, nchan, w.wl.ncorr), dtype=torch.complex128, device=w.dev)
            w.d_chi2 = torch.zeros(nchan, dtype=torch.float64, device=w.dev)
            w.wl.predict(w.d_vis, w.sp, P)
            w.stream.synchronize()
            w.d_data = w.d_vis.clone()
            w.d_data += 0.01
            w.done = torch.cuda.Event()
            torch.cuda.synchronize(w.dev)
        workers.append(w)
    ncorr = workers[0].wl.ncorr
    fused_chi2 = (hasattr(workers[0].wl, "predict_chi2") and getattr(workers[0].wl, "chi2_in_epilogue", True)
                  and os.environ.get("AFHIP_BENCH_FUSED_CHI2", "1") != "0")
    for w in workers:
        w.wl.fused_chi2 = fused_chi2
    dev0 = workers[0].dev
    staging = torch.zeros((n, nchan), dtype=torch.float64, device=dev0)
    total = torch.zeros(nchan, dtype=torch.float64, device=dev0)
    reduce_stream = torch.cuda.Stream(device=dev0)
    with torch.cuda.device(dev0):
        reduced = torch.cuda.Event()
        reduced.record(reduce_stream)
    history = []
    evs = []
    for w in workers:                    # a HIP event belongs to the device that is current when it is created
        with torch.cuda.device(w.dev):
            evs.append(Events(_lib, args.steps))
    placed = [None] * n

    def task(k, step_no):
        w = workers[k]
        with placement.block(k):                      # row block k -> devs[k % n]; af_set_device on this thread
            placed[k] = placement.activate()[0]
        if step_no is not None:
            evs[k].arm(step_no)
        if fused_chi2:
            w.wl.predict_chi2(w.d_vis, w.d_data, w.d_chi2, w.sp, P)
        else:
            w.wl.predict(w.d_vis, w.sp, P)
            _lib.call("af_chi2_c128", P(w.d_vis), P(w.d_data), None, nrow, nchan, ncorr, P(w.d_chi2), w.sp)
        if step_no is not None:
            evs[k].disarm()
        with torch.cuda.stream(w.stream):
            w.stream.wait_event(reduced)                       # the previous step's sum has read staging[k]
            staging[k].copy_(w.d_chi2, non_blocking=True)      # xGMI peer copy (nchan doubles)
            w.done.record(w.stream)
        return k

    pool = ThreadPoolExecutor(n)

    def step(step_no):
        done = list(pool.map(lambda k: task(k, step_no), range(n)))
        for w in workers:
            reduce_stream.wait_event(w.done)
        with torch.cuda.stream(reduce_stream):
            torch.sum(staging, dim=0, out=total)
            history.append(total.clone())                      # every step's reduced vector is checked below
            reduced.record(reduce_stream)
        return len(done)

    def sync_all():
        for w in workers:
            w.stream.synchronize()
        reduce_stream.synchronize()

    for _ in range(args.warmup):
        step(None)
    sync_all()
    if list(placed) != list(devs) and args.warmup:
        raise SystemExit("placement put the row blocks on %s, expected %s" % (placed, list(devs)))
    t0 = time.perf_counter()
    reported = n
    for s in range(args.steps):
        reported = min(reported, step(s))
    sync_all()
    elapsed = time.perf_counter() - t0
    pool.shutdown()
    kernel_s = []
    for w, e in zip(workers, evs):
        with torch.cuda.device(w.dev):
            kernel_s.append(e.collect())
    chi2_sum = total.cpu().numpy()
    chi2_check = sum(w.d_chi2.cpu().numpy() for w in workers)
    if not np.allclose(chi2_sum, chi2_check, rtol=1e-12, atol=0):
        raise SystemExit("chi^2 reduced across devices differs from the sum of the partials")
    for k, h in enumerate(history):          # identical inputs every step: every step's reduction must be the same vector
        if not np.allclose(h.cpu().numpy(), chi2_sum, rtol=1e-12, atol=0):     # (to the order of the chi^2 kernel's atomics)
            raise SystemExit("step %d reduced a different chi^2 vector than the last step (staging overwritten early?)" % k)
    w0 = workers[0]
    with torch.cuda.device(dev0):
        max_err = check_rows(w0.wl, w0.d_vis, nrow, args.check_rows, dev0) if args.check_rows > 0 else None
    res = {
        "label": w0.wl.label, "ranks_reported": reported, "elapsed": elapsed, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "value": reported * nrow * nchan / (elapsed / args.steps) / 1e6,
        "corrs": ncorr, "fp64_max_abs_err": max_err,
        "roofline": roofline_entry(w0.wl, args, args.workload, float(np.mean(kernel_s))),
    }
    desc = "threads: one process, %d worker threads, row block k on device %s[k %% %d] (placement.block)" % (n, list(devs), n)
    out = headline_json(args, res, desc, "peer copies of the partials to device %d, summed there" % devs[0])
    out["per_device_kernel_ms"] = [1e3 * x for x in kernel_s]
    out["config"]["devices"] = list(devs)
    out["config"]["physical_devices"] = len(set(devs))
    print(json.dumps(out))
    sys.stdout.flush()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.executor == "threads":
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--executor threads is one process; do not start it under a multi-rank launcher")
        return run_threads(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))        # this process makes no GPU call, before or after
    return run_ranks(args)


if __name__ == "__main__":
    main()
