"""The two multi-GPU executors of bench.py: one process per GPU (torch.distributed; RCCL over xGMI) and one process with
N worker threads (placement.block: the reference's dask chunks on a thread pool)."""
import ctypes
import json
import os
import sys
import time

import numpy as np

from .common import Events, check_rows, roofline_entry
from .launcher import free_port, require_devices
from .measure import compact_summary, extras_requested, headline_json, measure
from .workloads import WORKLOADS


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

