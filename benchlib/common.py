"""Shared pieces of the benchmark: hardware peaks, the CPU-baseline sampler, the roofline entry, HIP-event timing of the
dominant kernel (af_profile_events), the oracle row check."""
import ctypes
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # 256 CU x 4 SIMD x 16 FMA lanes/clk x 2 flop x 2.4 GHz, vector or matrix
FP32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32 vector = fp32-input MFMA peak (64 flop/clk/SIMD)
L2_PEAK_GBS = 34500.0          # MI355X_MICROARCH.md: aggregate L2 bandwidth (degridder's gather view)
PMC_ROUNDS = ("r05", "r04", "r03", "r02")    # profiles/<round>_<workload>_pmc_summary.json, newest first
# the REAL reference (numba 0.54) measured in the build container: im_to_vis 10k x 16 x 100 x 4
# on one core 0.263 Mvis/s = 38 ns per (row, chan, src); linear in sources -> 0.026 Mvis/s/core at 1000 sources
NUMBA_CALIBRATION = {"value": 0.026, "unit": "Mvis/s per core at 1000 sources",
                     "source": "SURVEY.md section 6: africanus.dft.im_to_vis under numba on one Xeon core of the build "
                               "container, 38 ns per (row, chan, src); not measurable on the GPU box (no numba there)"}
EXTRA_WORKLOADS = ("dft_complex", "dft_f32", "gauss", "fused_dde", "fused_dde_ant", "fused_dde_ant128", "fused_dde_ant_c64", "fused_dde_c64", "degrid", "wgrid",
                   "wgrid_f32planes")
DEFAULT_SHAPE = dict(rows=1000000, chans=64, sources=1000, mode="auto", pa="random", npix=4096, antennas=64)



def threads_available():
    import oracle
    return oracle.num_threads(omp=True)


def parallel_rows(fn, nrows, threads):
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

