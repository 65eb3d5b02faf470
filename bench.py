#!/usr/bin/env python
"""
bench.py -- headline benchmark of the RIME visibility-predict hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input resident in HBM:
    vis = im_to_vis(image, uvw, lm, frequency)       # direct-transform predict, BASELINE configs[1]
    chi2[nu] = sum |data - vis|^2                    # per-channel chi^2 of the shard
    (N > 1) RCCL all-reduce of chi2 over xGMI         # the only cross-GPU exchange of the path
at the shape BASELINE.json's metric is quoted on: 1e6 rows x 64 chan x 1000 point sources x
4 corr, fp64, PER GPU (rows shard across GPUs, weak scaling: BASELINE configs[3] is 8e6 rows
on 8 GPUs).  Metric: Mvis/s = rows x chans / second / 1e6 (whole job).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R --chans C --sources S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Besides the driver's contract it carries
  "roofline"     for the dominant kernel (dft_mfma_kernel<64>): algorithmic fp64 flops per launch /
                 its average duration measured with HIP events on its own stream, against the
                 fp64 peak (MFMA and VALU f64 share one 78.6 TFLOP/s pipe) -- plus, under "hbm",
                 the algorithmic HBM bytes against the 8 TB/s peak ("Rooflines");
  "cpu_baseline" the CPU oracle (C restatement of the numba loop, OpenMP over rows) timed on
                 this box's host cores on a bounded row sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # 256 CU x 4 SIMD x 16 FMA lanes/clk x 2 flop x 2.4 GHz, vector or matrix


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--rows", type=int, default=1000000, help="rows PER GPU")
    p.add_argument("--chans", type=int, default=64)
    p.add_argument("--sources", type=int, default=1000)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--mode", default="auto", choices=["auto", "exact", "recurrence"])
    p.add_argument("--workload", default="dft", choices=["dft", "fused_dde"],
                   help="dft: im_to_vis (BASELINE configs[1], the headline); fused_dde: fused predict with "
                        "per-antenna beam-cube DDEs, 64 antennas (BASELINE configs[2])")
    p.add_argument("--pa", default="random", choices=["random", "common"],
                   help="fused_dde: parallactic angles iid U(0, pi/6) per (time, antenna) (SURVEY 8(d), the "
                        "reference's own test recipe) or one angle per timestep + 1e-3 rad antenna jitter "
                        "(a real array: coherent beam gathers)")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="torch.distributed backend; nccl = RCCL over xGMI (default).  gloo exists to "
                        "exercise the N>1 code path on a one-GPU box (with AFHIP_BENCH_DEVICE=0)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU work of the baseline sample")
    p.add_argument("--check-rows", type=int, default=256, help="rows checked against the oracle")
    return p.parse_args()


def cpu_baseline(image, uvw, lm, freq, target_core_seconds):
    """Time the CPU oracle (kind 'port') on a bounded row sample; OpenMP over rows."""
    import oracle
    nchan, nsrc = freq.shape[0], lm.shape[0]
    threads = oracle.num_threads(omp=True)
    # calibrate on a few rows, single thread
    t0 = time.perf_counter()
    oracle.im_to_vis(image, uvw[:16], lm, freq, omp=False)
    per_row = (time.perf_counter() - t0) / 16
    rows = int(max(threads * 8, min(uvw.shape[0], target_core_seconds / per_row)))
    rows -= rows % threads
    t0 = time.perf_counter()
    oracle.im_to_vis(image, uvw[:rows], lm, freq, omp=True)
    dt = time.perf_counter() - t0
    return {
        "value": rows * nchan / dt / 1e6, "unit": "Mvis/s", "cores": threads, "kind": "port",
        "sample": "oracle im_to_vis (C restatement of africanus/dft/kernels.py:33-67, OpenMP over rows), "
                  "%d rows x %d chan x %d src x 4 corr fp64 in %.2f s; linear in rows; "
                  "single-thread rate %.4f Mvis/s" % (rows, nchan, nsrc, dt, nchan / per_row / 1e6),
        "single_thread_value": nchan / per_row / 1e6,
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the HIP path has no CPU fallback)")
    dev_index = int(os.environ.get("AFHIP_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
        # build the communicator now (RCCL sets its rings up lazily, at the first collective): the timed
        # region must not pay for it even when --warmup is 0
        warm = torch.zeros(1, dtype=torch.float64, device=dev)
        dist.all_reduce(warm)
        torch.cuda.synchronize(dev)

    from codex_africanus_amd import _lib
    from codex_africanus_amd.testing import synthetic_inputs, real_image
    lib = _lib.load()

    nrow, nchan, nsrc, ncorr = args.rows, args.chans, args.sources, 4
    # every rank draws the same sky and its own uvw shard (seed + rank): rows are independent
    d = synthetic_inputs(seed=args.seed, nrow=16, nchan=nchan, nsrc=nsrc, nant=64)
    rng = np.random.default_rng(1000 + args.seed + rank)
    uvw = np.empty((nrow, 3))
    uvw[:, 0] = rng.uniform(-4000, 4000, nrow)
    uvw[:, 1] = rng.uniform(-4000, 4000, nrow)
    uvw[:, 2] = rng.uniform(-400, 400, nrow)
    image = real_image(d)
    lm, freq = d["lm"], d["frequency"]

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_image, d_uvw, d_lm, d_freq = t(image), t(uvw), t(lm), t(freq)
    d_vis = torch.empty((nrow, nchan, ncorr), dtype=torch.complex128, device=dev)
    d_chi2 = torch.zeros(nchan, dtype=torch.float64, device=dev)
    ws_bytes = int(lib.af_im_to_vis_workspace_bytes(nsrc, nchan, ncorr, 0))
    d_ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
    mode = {"auto": _lib.AF_DFT_AUTO, "exact": _lib.AF_DFT_EXACT, "recurrence": _lib.AF_DFT_RECURRENCE}[args.mode]
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    P = lambda x: ctypes.c_void_p(x.data_ptr())

    if args.workload == "dft":
        def predict():
            _lib.call("af_im_to_vis_f64", P(d_image), 0, P(d_uvw), P(d_lm), P(d_freq), nsrc, nrow, nchan, ncorr,
                      _lib.CONVENTION["fourier"], mode, P(d_vis), P(d_ws), ws_bytes, stream)
    else:
        # BASELINE configs[2]: 64 antennas, 2016 baselines per timestep, beam cube
        # 257 x 257 x 33 x 2 x 2 complex128, parallactic angles U(0, pi/6), pointing errors 1e-3 N(0,1),
        # antenna scaling 1 +- 1e-3; brightness = flat-spectrum coherency matrices of the synthetic sky
        nant = 64
        a1, a2 = np.triu_indices(nant, 1)
        nbl = a1.shape[0]
        ntime = -(-nrow // nbl)
        ant1 = np.tile(a1, ntime)[:nrow].astype(np.int32)
        ant2 = np.tile(a2, ntime)[:nrow].astype(np.int32)
        time_index = np.repeat(np.arange(ntime, dtype=np.int64), nbl)[:nrow]
        g = np.linspace(-1, 1, 257)
        ll, mm = np.meshgrid(g, g, indexing="ij")
        pattern = np.exp(-(ll**2 + mm**2) / 0.5) * np.exp(1j * (0.3 * ll + 0.2 * mm))
        gains = (1 + 0.02 * np.arange(33))[:, None] * np.array([1.0, 0.05j, -0.04j, 0.95])[None, :]
        beam = (pattern[:, :, None, None] * gains[None, None]).reshape(257, 257, 33, 2, 2)
        extents = np.array([[-0.06, 0.06], [-0.06, 0.06]])
        beam_freq_map = np.linspace(freq[0], freq[-1], 33)
        pa = rng.uniform(0, np.pi / 6, (ntime, nant))
        if args.pa == "common":
            pa = np.linspace(0, np.pi / 6, ntime)[:, None] + 1e-3 * rng.standard_normal((ntime, nant))
        pe = 1e-3 * rng.standard_normal((ntime, nant, nchan, 2))
        asc = 1.0 + 1e-3 * rng.standard_normal((nant, nchan, 2))
        X = np.broadcast_to(d["brightness"][:, None, :], (nsrc, nchan, 4)).reshape(nsrc, nchan, 2, 2)
        n_items = ctypes.c_int64(0)
        tip = time_index.ctypes.data_as(ctypes.c_void_p)
        _lib.call("af_fused_plan_rows", tip, nrow, None, 0, ctypes.byref(n_items))
        items = np.zeros((n_items.value, 4), dtype=np.int32)
        _lib.call("af_fused_plan_rows", tip, nrow, items.ctypes.data_as(ctypes.c_void_p), n_items.value,
                  ctypes.byref(n_items))
        fd = dict(items=t(items), a1=t(ant1), a2=t(ant2), X=t(X), beam=t(beam), ext=t(extents),
                  fmap=t(beam_freq_map), pa=t(pa), pe=t(pe), asc=t(asc))
        fws_bytes = int(lib.af_fused_predict_workspace_bytes(nsrc, nchan, 257, 257, 33))
        d_fws = torch.empty(max(fws_bytes, 256), dtype=torch.uint8, device=dev)
        fused_host = dict(time_index=time_index, ant1=ant1, ant2=ant2, X=X, beam=beam, extents=extents,
                          beam_freq_map=beam_freq_map, pa=pa, pe=pe, asc=asc)

        def predict():
            _lib.call("af_fused_predict_c128", P(fd["items"]), n_items.value, P(fd["a1"]), P(fd["a2"]), nrow,
                      P(d_lm), P(d_uvw), P(d_freq), P(fd["X"]), nsrc, nchan, P(fd["beam"]), 257, 257, 33,
                      P(fd["ext"]), P(fd["fmap"]), P(fd["pa"]), ntime, nant, P(fd["pe"]), P(fd["asc"]), None, None,
                      _lib.CONVENTION["fourier"], P(d_vis), P(d_fws), fws_bytes, stream)

    # "observed" data for the chi^2: the model itself plus a fixed perturbation (one extra predict)
    predict()
    d_data = d_vis.clone()
    d_data += 0.01

    def step():
        predict()
        _lib.call("af_chi2_c128", P(d_vis), P(d_data), None, nrow, nchan, ncorr, P(d_chi2), stream)
        if world > 1:
            dist.all_reduce(d_chi2, op=dist.ReduceOp.SUM)

    for _ in range(args.warmup):
        step()

    # HIP events around the dominant kernel of every timed step, on its own stream
    evs = []
    for _ in range(args.steps):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.call("af_event_create", ctypes.byref(a))
        _lib.call("af_event_create", ctypes.byref(b))
        evs.append((a, b))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        _lib.call("af_profile_events", evs[k][0], evs[k][1])
        step()
    _lib.call("af_profile_events", None, None)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    kernel_ms = []
    for a, b in evs:
        ms = ctypes.c_float(0)
        _lib.call("af_event_elapsed_ms", a, b, ctypes.byref(ms))
        kernel_ms.append(ms.value)
        _lib.call("af_event_destroy", a)
        _lib.call("af_event_destroy", b)
    kernel_s = float(np.mean(kernel_ms)) / 1e3 if kernel_ms else float("nan")

    # parity of the benchmarked output against the CPU oracle on a row sample (checker only)
    max_err = None
    if rank == 0 and args.check_rows > 0:
        import oracle
        rows = np.linspace(0, nrow - 1, min(args.check_rows, nrow)).astype(np.int64)
        got = d_vis[torch.from_numpy(rows).to(dev)].cpu().numpy()
        if args.workload == "dft":
            ref = oracle.im_to_vis(image, uvw[rows], lm, freq, omp=True)
        else:
            # the reference chain on the sampled rows (only their timesteps' Jones terms are built)
            h = fused_host
            rows = rows[:32]
            got = got[:32].reshape(32, nchan, 2, 2)
            tsel, tinv = np.unique(h["time_index"][rows], return_inverse=True)
            dde = oracle.beam_cube_dde(h["beam"], h["extents"], h["beam_freq_map"], lm, h["pa"][tsel],
                                       h["pe"][tsel], h["asc"], freq)
            phase = oracle.phase_delay(lm, uvw[rows], freq)
            coh = np.einsum("srf,sfij->srfij", phase, h["X"])
            ref = oracle.predict_vis(tinv, h["ant1"][rows], h["ant2"][rows], dde, coh, dde, None, None, None)
        max_err = float(np.abs(got - ref).max())

    if rank == 0:
        total_vis = world * nrow * nchan
        ms_per_step = elapsed / args.steps * 1e3
        # Dominant kernel = the one the library's measurement hook brackets.  Real 4-correlation images
        # on a one-spacing band run dft_mfma_kernel<64>: every 64-channel tile in ONE launch (C2: all
        # 64 channels).  --mode exact runs dft_exact_kernel over all channels.
        mfma = args.mode != "exact" and ncorr == 4 and nchan >= 14
        if mfma:
            ntile = nchan // 64 + (1 if nchan % 64 > 32 else 0)
            dom_chans = min(nchan, ntile * 64) if ntile else nchan
            kernel_name = "dft_mfma_kernel<64>" if ntile else "dft_mfma_kernel<%d>" % (16 if nchan <= 16 else 32)
            nstep = -(-nsrc // 4)
            # algorithmic HBM bytes of that launch: 64 B written per vis + uvw 24 B/row +
            # its records ((64 + 1) x 16 doubles per tile and 4-source step); it reads nothing else from HBM
            alg_bytes = nrow * dom_chans * ncorr * 16 + nrow * 24 + max(ntile, 1) * nstep * 65 * 16 * 8
        else:
            ct = 13
            ntile = -(-nchan // ct)
            dom_chans = nchan
            kernel_name = "dft_exact_kernel<13,4,false>"
            alg_bytes = nrow * dom_chans * ncorr * 16 + nrow * 24 + ntile * nsrc * 512
        # algorithmic flops: per (row, chan, src) one complex phasor step (recurrence, 2 FMA) +
        # ncorr complex-by-real MACs (2 FMA each) = 10 FMA = 20 flop; the MACs are fp64 MFMA
        # (v_mfma_f64_4x4x4_4b), the recurrence fp64 VALU -- one shared fp64 pipe on this chip
        alg_flops = float(nrow) * dom_chans * nsrc * (2 + 2 * ncorr) * 2
        workload = "im_to_vis DFT predict (BASELINE configs[1])"
        if args.workload == "fused_dde":
            # + indices 12 B/row, beam cube + |beam| + parangles/pointing/scaling; ~150 flop
            # per (row, chan, src): phasor 8 + E X E^H 112 + 4 complex MACs 32
            alg_bytes = (nrow * nchan * 64 + nrow * 36 + 257 * 257 * 33 * 4 * 24 + ntime * nant * (8 + nchan * 16)
                         + nant * nchan * 16 + nsrc * nchan * 64)
            alg_flops = float(nrow) * nchan * nsrc * 150.0
            kernel_name = "fused_predict_kernel"
            workload = ("fused predict with per-antenna beam-cube DDEs, 64 antennas (BASELINE configs[2]), "
                        "parallactic angles %s" % args.pa)
        achieved = alg_bytes / kernel_s / 1e9
        # HBM bytes per launch from the PMC passes of THIS command committed under profiles/
        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs; KB units; FETCH_SIZE doubled
        # as MI355X_MICROARCH.md prescribes for gfx950 streaming reads -- an upper bound here).
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pmc) and (nrow, nchan, nsrc, args.mode, args.workload) == (1000000, 64, 1000, "auto", "dft"):
            c = json.load(open(pmc))
            traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            traffic_src = "profiles/r01_pmc_summary.json"
        out = {
            "metric": "Mvis/s (rows x chans) for predict_vis at 1e6 rows/64 ch/1000 src; fp64 max-abs err",
            "value": total_vis / (elapsed / args.steps) / 1e6,
            "unit": "Mvis/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": workload + " + per-channel chi^2" + (
                    "" if world == 1 else " + RCCL all-reduce" if args.backend == "nccl" else " + gloo all-reduce"),
                "rows_per_gpu": nrow, "chans": nchan, "sources": nsrc, "corrs": ncorr,
                "rows_total": world * nrow, "phasor_mode": args.mode,
                "sharding": "rows over %d GPU(s), no data-path collective; chi2 (nchan,) all-reduce (%s)"
                            % (world, "none" if world == 1 else args.backend),
            },
            "fp64_max_abs_err": max_err,
            "roofline": {
                "kernel": kernel_name,
                # both workloads are bound by the fp64 pipe ("mfma": the dense fp64 matrix peak equals the vector
                # peak on this chip): im_to_vis issues fp64 MFMA + VALU, fused_dde fp64 VALU (Jones algebra);
                # the HBM view of the same launch is in "hbm"
                "bound": "mfma",
                "achieved": alg_flops / kernel_s / 1e12,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": alg_flops / kernel_s / 1e12 / FP64_PEAK_TFLOPS,
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel_ms": kernel_s * 1e3, "algorithmic_flops": alg_flops,
                "channels_in_kernel": dom_chans if args.workload == "dft" else nchan,
                "note": "fp64-pipe bound (MFMA f64 and VALU f64 share it; 78.6 TFLOP/s spec for either), not "
                        "HBM-bound: nsrc=1000 phasors per 64-byte visibility",
                "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "algorithmic_bytes": alg_bytes},
                "fp64": {"achieved": alg_flops / kernel_s / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": alg_flops / kernel_s / 1e12 / FP64_PEAK_TFLOPS},
            },
        }
        if not args.no_cpu_baseline and world == 1 and args.workload == "dft":
            out["cpu_baseline"] = cpu_baseline(image, uvw, lm, freq, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
