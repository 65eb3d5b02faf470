#!/usr/bin/env python
"""Condense gpurun_out/prof_d2h (tools/profile_d2h_overlap.sh) into profiles/<round>_d2h_overlap_*: the bench lines and
a timeline summary of the LAST chunked pass of the traced run -- per chunk the transform's interval and its download's
interval, and how much of the copy time ran while a kernel of the library was executing."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_d2h")
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"


def find(suffix):
    hits = sorted(glob.glob(os.path.join(SRC, "trace", "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def union_len(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def overlap(a, b):
    """total time covered by both interval sets (each made disjoint first)"""
    def flat(iv):
        out = []
        for s, e in sorted(iv):
            if out and s <= out[-1][1]:
                out[-1][1] = max(out[-1][1], e)
            else:
                out.append([s, e])
        return out
    a, b = flat(a), flat(b)
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


out = {}
for name in ("bench.json", "host_path.json"):
    path = os.path.join(SRC, name)
    if os.path.exists(path):
        lines = [x for x in open(path).read().splitlines() if x.startswith("{")]
        if lines:
            out[name.replace(".json", "")] = json.loads(lines[-1])
kt, mt = find("kernel_trace.csv"), find("memory_copy_trace.csv")
if kt and mt:
    kernels = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(kt))]
    copies = []
    for r in csv.DictReader(open(mt)):
        direction = r.get("Direction", r.get("Kind", ""))
        copies.append((direction, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    d2h = [(s, e) for d, s, e in copies if "DEVICE_TO_HOST" in d.upper() or "D2H" in d.upper()]
    # under rocprofv3 the runtime performs the downloads as blit kernels (__amd_rocclr_copyBuffer in the KERNEL trace; the
    # memory-copy trace then only holds the small uploads); outside the profiler they are SDMA transfers
    blit = [(s, e) for n, s, e in kernels if "copyBuffer" in n]
    out["copy_engine_in_trace"] = "SDMA (memory-copy trace)" if d2h else "blit kernels (__amd_rocclr_copyBuffer in the kernel trace)"
    d2h = d2h or blit
    kernels = [k for k in kernels if "copyBuffer" not in k[0]]
    big = [(s, e) for s, e in d2h if e - s > 1e6]                  # the chunk downloads: 256 MB each, milliseconds
    nch = out.get("bench", {}).get("chunks", 16)
    last = sorted(big)[-nch:]                                       # the last chunked pass
    t0, t1 = last[0][0], last[-1][1]
    main = [(s, e) for n, s, e in kernels if "dft_mfma_kernel" in n and s >= t0 - 5e7 and e <= t1]
    main = sorted(main)[-nch * 2:]
    main = [iv for iv in main if iv[1] > t0 - 3e7]
    allk = [(s, e) for n, s, e in kernels if s >= min(m[0] for m in main) and e <= t1]
    span0 = min(m[0] for m in main)
    out["timeline_under_rocprofv3"] = {
        "chunks": nch,
        "pass_ms": (t1 - span0) / 1e6,
        "copy_busy_ms": union_len(last) / 1e6,
        "kernel_busy_ms": union_len(allk) / 1e6,
        "dft_mfma_kernel_busy_ms": union_len(main) / 1e6,
        "copy_and_kernel_concurrent_ms": overlap(last, allk) / 1e6,
        "fraction_of_kernel_time_under_a_copy": overlap(last, allk) / max(union_len(allk), 1),
        "per_chunk": [{"download_ms": [round((s - span0) / 1e6, 3), round((e - span0) / 1e6, 3)]} for s, e in last],
        "dft_mfma_kernel_intervals_ms": [[round((s - span0) / 1e6, 3), round((e - span0) / 1e6, 3)] for s, e in main],
        "note": "times relative to the start of the pass's first transform kernel.  UNDER THE PROFILER the downloads are blit "
                "kernels and every dispatch waits for the previous one (pass_ms here against chunked_ms of the un-profiled "
                "bench line): the trace cannot show the overlap; the un-profiled pass is timed by HIP events inside the "
                "process, bench.hip_event_timeline",
    }
    for src, dst in ((kt, "kernel_trace"), (mt, "memory_copy_trace")):
        rows = list(csv.DictReader(open(src)))
        keep = [r for r in rows if int(r["Start_Timestamp"]) >= span0 - 1e6 and int(r["End_Timestamp"]) <= t1 + 1e6]
        with open(os.path.join(ROOT, "profiles", "%s_d2h_overlap_%s.csv" % (tag, dst)), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(keep)
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_d2h_overlap_summary.json" % tag), "w"), indent=1)
b = dict(out.get("bench", {}))
tl = b.pop("hip_event_timeline", None)
print(json.dumps(b, indent=1))
if tl:
    print(json.dumps({k: v for k, v in tl.items() if k != "per_chunk"}, indent=1), tl["per_chunk"][:3])
if "timeline_under_rocprofv3" in out:
    print(json.dumps({k: v for k, v in out["timeline_under_rocprofv3"].items() if k not in ("per_chunk", "dft_mfma_kernel_intervals_ms")}, indent=1))
