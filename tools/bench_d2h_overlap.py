#!/usr/bin/env python
"""
Does the device -> host copy of a result overlap the transform that produces the next rows?  
BASELINE configs[1] (1e6 rows x 64 chan x 1000 src x 4 corr: 4.1 GB of visibilities) through the C ABI:
  mono     one af_im_to_vis_f64 call, then one 4.1 GB af_memcpy_d2h into page-locked memory
  chunked  NCHUNK row chunks: the transform of chunk k on stream A, its download on stream B behind an event
           (af_stream_wait_event), so that download k runs while chunk k + 1 is transformed
Prints wall times (best of --repeats); under rocprofv3 --kernel-trace --memory-copy-trace the same run gives the
timeline tools/summarize_d2h_overlap.py condenses.   python tools/bench_d2h_overlap.py [--mode both] [--chunks 16]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd import _lib
from codex_africanus_amd.testing import synthetic_inputs, real_image

p = argparse.ArgumentParser()
p.add_argument("--mode", default="both", choices=["mono", "chunked", "both"])
p.add_argument("--chunks", type=int, default=16)
p.add_argument("--rows", type=int, default=1000000)
p.add_argument("--repeats", type=int, default=3)
args = p.parse_args()

lib = _lib.load()
nrow, nchan, nsrc = args.rows, 64, 1000
d = synthetic_inputs(seed=0, nrow=16, nchan=nchan, nsrc=nsrc, nant=64)
rng = np.random.default_rng(1)
uvw = np.empty((nrow, 3))
uvw[:, 0] = rng.uniform(-4000, 4000, nrow); uvw[:, 1] = rng.uniform(-4000, 4000, nrow); uvw[:, 2] = rng.uniform(-400, 400, nrow)
image = real_image(d)
vp = ctypes.c_void_p


def dev(a):
    ptr = vp()
    _lib.call("af_malloc", ctypes.byref(ptr), a.nbytes)
    _lib.call("af_memcpy_h2d", ptr, a.ctypes.data_as(vp), a.nbytes, None)
    return ptr


d_img, d_uvw, d_lm, d_fr = dev(image), dev(uvw), dev(d["lm"]), dev(d["frequency"])
out_bytes = nrow * nchan * 4 * 16
d_out, h_out = vp(), vp()
_lib.call("af_malloc", ctypes.byref(d_out), out_bytes)
_lib.call("af_malloc_host", ctypes.byref(h_out), out_bytes)            # page-locked
ws_bytes = int(lib.af_im_to_vis_workspace_bytes(nsrc, nchan, 4, 0))
d_ws = vp()
_lib.call("af_malloc", ctypes.byref(d_ws), max(ws_bytes, 256))
sa, sb = vp(), vp()
_lib.call("af_stream_create", ctypes.byref(sa))
_lib.call("af_stream_create", ctypes.byref(sb))
_lib.call("af_device_synchronize")


def transform(r0, r1, stream):
    _lib.call("af_im_to_vis_f64", d_img, 0, vp(d_uvw.value + r0 * 24), d_lm, d_fr, nsrc, r1 - r0, nchan, 4,
              _lib.CONVENTION["fourier"], _lib.AF_DFT_AUTO, vp(d_out.value + r0 * nchan * 64), d_ws, ws_bytes, stream)


def mono():
    t0 = time.perf_counter()
    transform(0, nrow, sa)
    _lib.call("af_memcpy_d2h", h_out, d_out, out_bytes, sa)
    _lib.call("af_stream_synchronize", sa)
    return time.perf_counter() - t0


edges = [nrow * k // args.chunks for k in range(args.chunks + 1)]
events = []
for _ in range(args.chunks):
    e = vp()
    _lib.call("af_event_create", ctypes.byref(e))
    events.append(e)


def chunked():
    t0 = time.perf_counter()
    for k in range(args.chunks):
        r0, r1 = edges[k], edges[k + 1]
        transform(r0, r1, sa)
        _lib.call("af_event_record", events[k], sa)
        _lib.call("af_stream_wait_event", sb, events[k])
        _lib.call("af_memcpy_d2h", vp(h_out.value + r0 * nchan * 64), vp(d_out.value + r0 * nchan * 64), (r1 - r0) * nchan * 64, sb)
    _lib.call("af_stream_synchronize", sb)
    return time.perf_counter() - t0


def kernel_only():
    t0 = time.perf_counter()
    transform(0, nrow, sa)
    _lib.call("af_stream_synchronize", sa)
    return time.perf_counter() - t0


def copy_only():
    t0 = time.perf_counter()
    _lib.call("af_memcpy_d2h", h_out, d_out, out_bytes, sb)
    _lib.call("af_stream_synchronize", sb)
    return time.perf_counter() - t0


def chunked_timeline():
    """the same pass with HIP events around every chunk's transform (stream A) and download (stream B): a timeline taken
    inside the process, without a profiler (rocprofv3 turns the downloads into blit kernels and serialises them with
    the transforms: profiles/r05_d2h_overlap_summary.json "timeline_under_rocprofv3")"""
    def ev():
        e = vp()
        _lib.call("af_event_create", ctypes.byref(e))
        return e
    base = ev()
    marks = [[ev() for _ in range(4)] for _ in range(args.chunks)]
    _lib.call("af_device_synchronize")
    _lib.call("af_event_record", base, sa)
    for k in range(args.chunks):
        r0, r1 = edges[k], edges[k + 1]
        _lib.call("af_event_record", marks[k][0], sa)
        transform(r0, r1, sa)
        _lib.call("af_event_record", marks[k][1], sa)
        _lib.call("af_stream_wait_event", sb, marks[k][1])
        _lib.call("af_event_record", marks[k][2], sb)
        _lib.call("af_memcpy_d2h", vp(h_out.value + r0 * nchan * 64), vp(d_out.value + r0 * nchan * 64), (r1 - r0) * nchan * 64, sb)
        _lib.call("af_event_record", marks[k][3], sb)
    _lib.call("af_stream_synchronize", sb)
    _lib.call("af_stream_synchronize", sa)
    out = []
    for k in range(args.chunks):
        t = []
        for e in marks[k]:
            ms = ctypes.c_float(0)
            _lib.call("af_event_elapsed_ms", base, e, ctypes.byref(ms))
            t.append(round(ms.value, 3))
        out.append({"transform_ms": t[:2], "download_ms": t[2:]})
    # time during which a transform and a download were both in progress
    both = 0.0
    for a in out:
        for b in out:
            both += max(0.0, min(a["transform_ms"][1], b["download_ms"][1]) - max(a["transform_ms"][0], b["download_ms"][0]))
    return {"per_chunk": out, "pass_ms": out[-1]["download_ms"][1],
            "transform_busy_ms": round(sum(a["transform_ms"][1] - a["transform_ms"][0] for a in out), 3),
            "download_busy_ms": round(sum(a["download_ms"][1] - a["download_ms"][0] for a in out), 3),
            "transform_and_download_concurrent_ms": round(both, 3)}


res = {"rows": nrow, "result_GB": out_bytes / 1e9, "chunks": args.chunks}
kernel_only(); copy_only()                                                # warm: first launch, first touch of the pinned pages
res["kernel_only_ms"] = 1e3 * min(kernel_only() for _ in range(args.repeats))
res["copy_only_ms"] = 1e3 * min(copy_only() for _ in range(args.repeats))
if args.mode in ("mono", "both"):
    res["mono_ms"] = 1e3 * min(mono() for _ in range(args.repeats))
if args.mode in ("chunked", "both"):
    res["chunked_ms"] = 1e3 * min(chunked() for _ in range(args.repeats))
    host = np.frombuffer((ctypes.c_char * out_bytes).from_address(h_out.value), dtype=np.complex128).reshape(nrow, nchan, 4)
    res["checksum_abs"] = float(np.abs(host[:: max(1, nrow // 997)]).sum())
    res["hip_event_timeline"] = chunked_timeline()
for k in ("mono_ms", "chunked_ms"):
    if k in res:
        res[k.replace("_ms", "_Mvis_s")] = nrow * nchan / res[k] / 1e3
print(json.dumps(res))
