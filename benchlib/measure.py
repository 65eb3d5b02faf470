"""One workload timed on one rank (`measure`), the driver's JSON line (`headline_json`) and the compact per-workload
summary the driver's record keeps (`compact_summary`)."""
import argparse
import ctypes
import os
import time

import numpy as np

from .common import EXTRA_WORKLOADS, Events, check_rows, is_default_shape, roofline_entry
from .workloads import METRIC, WORKLOADS


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
    for _ in range(int(os.environ.get("AFHIP_BENCH_FRONT_END_REPEATS", "1")) - 1):      # stress runs (tools/): the same check again
        if front_end is not None:
            wl.front_end_check(d_vis, rank, world, dev)
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
    kernel_s = ev.collect()
    per_rank = None
    if world > 1:
        # every rank's own wall time and dominant-kernel time (straggler diagnosis: the job's time is the MAX), then the
        # max over ranks and the number of ranks that reported
        mine = torch.zeros((world, 2), dtype=torch.float64, device=dev)
        mine[rank, 0], mine[rank, 1] = elapsed, kernel_s
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        per_rank = mine.cpu().numpy()
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        one = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        reported = int(round(float(one.item())))
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
        # the arithmetic type the path computes in: single-precision workloads (complex64 out) say so
        "dtype": "f32" if getattr(wl, "vis_dtype", "complex128") == "complex64" else "f64",
        "label": wl.label + ("; chi^2 summed in the transform's epilogue (the entry's _chi2 form)" if fused_chi2 else ""),
        "has_chi2": have_chi2, "ranks_reported": reported, "elapsed": elapsed, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "value": reported * nrow * nchan / (elapsed / steps) / 1e6,
        "corrs": ncorr, "fp64_max_abs_err": max_err, "roofline": roofline_entry(wl, wargs, workload, kernel_s),
    }
    if front_end is not None:
        res["front_end"] = front_end
    if per_rank is not None:
        k_ms, e_ms = 1e3 * per_rank[:, 1], 1e3 * per_rank[:, 0] / steps
        res["per_rank"] = {"kernel_ms": [round(float(x), 4) for x in k_ms], "kernel_ms_min": float(k_ms.min()),
                           "kernel_ms_max": float(k_ms.max()), "ms_per_step": [round(float(x), 4) for x in e_ms],
                           "ms_per_step_min": float(e_ms.min()), "ms_per_step_max": float(e_ms.max())}
    if hasattr(wl, "end_to_end") and world == 1 and not getattr(args, "no_end_to_end", False):
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
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": res.get("dtype", "f64"), "data": "synthetic",
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
    if "per_rank" in res:
        out["per_rank"] = res["per_rank"]
        for k in ("kernel_ms_min", "kernel_ms_max", "ms_per_step_min", "ms_per_step_max"):
            out["config"]["rank_" + k] = res["per_rank"][k]      # scalars: the driver's record keeps them
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

