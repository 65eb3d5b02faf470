"""
Device-buffer plumbing between the reference's array-in / array-out signatures
and the device-pointer C ABI.

Two modes, chosen per call:
  * host mode   -- every array argument is a numpy array (or array-like): inputs are
                   uploaded with af_malloc / af_memcpy_h2d, the result comes back as a
                   fresh numpy array (the reference's contract);
  * device mode -- at least one argument is a torch tensor on a ROCm device: tensors are
                   used in place (zero copy), numpy arguments are uploaded, and the result
                   is a torch tensor on that device, enqueued on torch's current stream.
PyTorch is only plumbing here (device memory and streams); no torch op does arithmetic
of the path.
"""
import ctypes
import os

import numpy as np

from . import _lib, placement


# measurement hook (tools/bench_small_calls.py): AFHIP_POOL=0 restores round 1's host path -- hipMalloc / hipFree
# per array per call and the NULL stream -- for before / after comparisons
_USE_POOL = os.environ.get("AFHIP_POOL", "1") != "0"


def _is_torch(x):
    return type(x).__module__.split(".")[0] == "torch" and hasattr(x, "data_ptr")


_TORCH_DTYPES = None


def _torch_dtype(np_dtype):
    global _TORCH_DTYPES
    import torch
    if _TORCH_DTYPES is None:
        _TORCH_DTYPES = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
                         np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128,
                         np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64,
                         np.dtype(np.uint8): torch.uint8, np.dtype(np.bool_): torch.bool}
    return _TORCH_DTYPES[np.dtype(np_dtype)]


def np_dtype_of(x):
    """numpy dtype of a numpy array / torch tensor / array-like."""
    if _is_torch(x):
        import torch
        return np.dtype({torch.float32: np.float32, torch.float64: np.float64,
                         torch.complex64: np.complex64, torch.complex128: np.complex128,
                         torch.int32: np.int32, torch.int64: np.int64, torch.int16: np.int16,
                         torch.int8: np.int8, torch.uint8: np.uint8, torch.bool: np.bool_}[x.dtype])
    return np.asarray(x).dtype if not hasattr(x, "dtype") else np.dtype(x.dtype)


# Debugging aid (AFHIP_POISON=1, or set by a test): every scratch block and every result buffer of a call starts as 0xFF
# bytes (NaN in every floating-point format) instead of whatever the allocator hands out -- a kernel that reads a cell
# before writing it, or leaves a result cell unwritten, then shows in the result instead of depending on what ran before
# (tests/test_gpu_uninitialised.py requires bit-identical results with and without it).
POISON = os.environ.get("AFHIP_POISON", "0") not in ("", "0")
POISON_BYTE = int(os.environ.get("AFHIP_POISON_BYTE", "255"))   # other patterns: 1 (small integers, denormal doubles), 64 (doubles ~ 32)


class _OwnedBuffer(object):
    """Device memory of one host-mode call, drawn from libafhip's per-device scratch pool (af_pool_malloc): in
    steady state no hipMalloc / hipFree happens on this path.  Returned to the pool only after the call's stream
    has been synchronised (Call.__exit__), so a block is idle when another thread picks it up."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        _lib.call("af_pool_malloc" if _USE_POOL else "af_malloc", ctypes.byref(p), max(self.nbytes, 1))
        self.ptr = p.value

    def free(self):
        if self.ptr:
            _lib.call("af_pool_free" if _USE_POOL else "af_free", self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _PinnedPool(object):
    """Page-locked host buffers for results returned as numpy arrays (host mode).

    A device -> host copy into pageable memory runs at 10-15 GB/s (the runtime stages it through its own pinned
    bounce buffers); into page-locked memory it runs at PCIe rate.  The result array is a view of a pinned
    buffer from libafhip's host pool (af_pool_malloc_host: size-bucketed, least-recently-freed eviction, capped by
    AFHIP_PINNED_LIMIT); when the array (and every view of it) is garbage collected the buffer goes back to the
    pool, so a loop that drops its previous result re-uses the same pages, ragged chunk sizes included.
    """
    MIN_BYTES = 1 << 20          # small results: plain numpy arrays, not worth pinning

    def take(self, nbytes):
        p = ctypes.c_void_p()
        lib = _lib.load()
        if lib.af_pool_malloc_host(ctypes.byref(p), ctypes.c_size_t(nbytes)) != 0 or not p.value:
            return None            # no page-locked memory to be had: the caller falls back to pageable
        return p.value

    @staticmethod
    def give(ptr):
        try:
            _lib.load().af_pool_free_host(ctypes.c_void_p(ptr))
        except Exception:
            pass

    def array(self, shape, dtype):
        """numpy array backed by a pinned buffer (or None)."""
        import weakref
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        if nbytes < self.MIN_BYTES:
            return None
        ptr = self.take(nbytes)
        if ptr is None:
            return None
        raw = (ctypes.c_char * nbytes).from_address(ptr)
        weakref.finalize(raw, self.give, ptr)   # raw is the base of every view handed out
        return np.frombuffer(raw, dtype=dtype).reshape(shape)


_pinned = _PinnedPool()


class _DeferredStatus(object):
    """Status words of device-mode calls that could not be read when the call returned (nothing synchronises there).

    A kernel that met an out-of-range index writes a flag into its workspace (and NaN into the rows concerned); the
    wrapper copies that word into a slot of ONE page-locked int32 buffer on the call's stream and records an event.
    Whenever the package next has a reason to look -- the start of any later call on any thread, a host-mode result,
    ``codex_africanus_amd.check_status()`` (which waits) -- the completed slots are read and a set flag raises
    ``ValueError`` with the message the call registered.  No host synchronisation is added to the device path."""
    SLOTS = 256

    def __init__(self):
        import threading
        self._lock = threading.Lock()
        self._buf = None
        self._free = []
        self._pending = []          # (slot, event, message)
        self._carry = []            # messages of flagged calls drained to make room, not yet reported

    def _ensure(self):
        if self._buf is None:
            import torch
            self._buf = torch.zeros(self.SLOTS, dtype=torch.int32).pin_memory()
            self._free = list(range(self.SLOTS))

    def watch(self, ws_tensor, offset_bytes, message):
        import torch
        with self._lock:
            self._ensure()
            if not self._free:
                # every slot is in flight (more than SLOTS device-mode calls enqueued before the GPU retired one):
                # wait for them, and KEEP what they report -- poll() raises it first
                self._carry.extend(self._drain_locked(wait=True))
            slot = self._free.pop()
        word = ws_tensor[offset_bytes:offset_bytes + 4].view(torch.int32)
        self._buf[slot:slot + 1].copy_(word, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(ws_tensor.device))
        with self._lock:
            self._pending.append((slot, ev, message))

    def _drain_locked(self, wait):
        bad, keep = [], []
        for slot, ev, message in self._pending:
            if wait:
                ev.synchronize()
            if wait or ev.query():
                flags = int(self._buf[slot])
                self._free.append(slot)
                if flags:
                    bad.append(message(flags))
            else:
                keep.append((slot, ev, message))
        self._pending = keep
        return bad

    def poll(self, wait=False):
        if not self._pending and not self._carry:
            return
        with self._lock:
            bad = self._carry + self._drain_locked(wait)
            self._carry = []
        if bad:
            raise ValueError("; ".join(bad) + " (reported by an earlier device-mode call; its rows are NaN)")


_deferred = _DeferredStatus()


class _CopyStreams(object):
    """A second stream (downloads) and one event per (thread, device) for Call.result_rows, created on first use."""

    def __init__(self):
        import threading
        self._tls = threading.local()

    def get(self):
        dev = _lib.get_device()
        cache = getattr(self._tls, "cache", None)
        if cache is None:
            cache = self._tls.cache = {}
        if dev not in cache:
            st, ev = ctypes.c_void_p(), ctypes.c_void_p()
            _lib.call("af_stream_create", ctypes.byref(st))
            _lib.call("af_event_create", ctypes.byref(ev))
            cache[dev] = (st, ev)
        return cache[dev]


_copy_streams = _CopyStreams()


def _copy_stream():
    return _copy_streams.get()


def check_status(wait=True):
    """Raise ``ValueError`` if a device-mode call made so far met an out-of-range index (waits for those calls'
    streams when ``wait``)."""
    _deferred.poll(wait=wait)


class Call(object):
    """Marshals the arrays of ONE API call; use as a context manager."""

    def __init__(self, *arrays):
        self.torch_device = None
        for a in arrays:
            if a is not None and _is_torch(a):
                if not a.is_cuda:
                    raise ValueError("torch tensors passed to codex_africanus_amd must live on a ROCm device")
                if self.torch_device is None:
                    self.torch_device = a.device
                elif a.device != self.torch_device:
                    raise ValueError("all torch tensors of one call must be on the same device")
        self.device_mode = self.torch_device is not None
        self._keep = []
        self._owned = []
        self._inputs = {}           # (id(array), dtype) -> device pointer: one upload / conversion per distinct input
        self._status = None         # host mode: (device address of an int32 status word, message function)
        self._dev_ctx = None
        self._prev_device = None
        self.stream = None

    def __enter__(self):
        _deferred.poll()            # an earlier device-mode call's index error surfaces here at the latest
        if self.device_mode:
            import torch
            self._dev_ctx = torch.cuda.device(self.torch_device)
            self._dev_ctx.__enter__()
            self.stream = ctypes.c_void_p(torch.cuda.current_stream(self.torch_device).cuda_stream)
        else:
            _lib.load()
            # host mode: the block's device when the call carries a row-block index (placement), then this thread's
            # own stream on it -- concurrent calls from dask worker threads do not meet on the NULL stream
            _, self._prev_device = placement.activate()
            self.stream = _lib.thread_stream() if _USE_POOL else None
        return self

    def __exit__(self, *exc):
        if self._owned and exc and exc[0] is not None:
            # error path: work may still be in flight on buffers about to return to the pool
            try:
                _lib.call("af_stream_synchronize", self.stream)
            except Exception:
                pass
        for b in self._owned:
            b.free()
        self._owned = []
        self._keep = []
        if self._dev_ctx is not None:
            self._dev_ctx.__exit__(*exc)
        if self._prev_device is not None:      # a re-placed call leaves the thread's device as it found it
            placement.restore(self._prev_device)
            self._prev_device = None
        return False

    # ---- inputs -------------------------------------------------------------------
    def inp(self, a, dtype):
        """Device pointer of `a` converted to `dtype`, C-contiguous (None -> NULL)."""
        if a is None:
            return None
        dtype = np.dtype(dtype)
        key = (id(a), dtype.str)
        if key in self._inputs:     # the same object passed twice (dde1_jones is dde2_jones): one buffer, one pointer
            return ctypes.c_void_p(self._inputs[key])
        if self.device_mode:
            import torch
            if _is_torch(a):
                t = a
            else:
                t = torch.from_numpy(np.ascontiguousarray(a)).to(self.torch_device)
            td = _torch_dtype(dtype)
            if t.dtype == torch.bool and td == torch.uint8:
                t = t.contiguous().view(torch.uint8)        # flags: same bytes (True is 1), no conversion pass
            elif t.dtype != td:
                t = t.to(td)
            t = t.contiguous()
            self._keep.append(t)
            self._keep.append(a)    # keeps id(a) unique for the lifetime of the call
            self._inputs[key] = t.data_ptr()
            return ctypes.c_void_p(t.data_ptr())
        if dtype == np.uint8 and getattr(a, "dtype", None) == np.bool_:
            arr = np.ascontiguousarray(a).view(np.uint8)    # flags: same bytes, no conversion pass
        else:
            arr = np.ascontiguousarray(a, dtype=dtype)
        buf = _OwnedBuffer(arr.nbytes)
        self._owned.append(buf)
        self._keep.append(arr)
        self._keep.append(a)
        if arr.nbytes:
            _lib.call("af_memcpy_h2d", buf.ptr, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes, self.stream)
        self._inputs[key] = buf.ptr
        return ctypes.c_void_p(buf.ptr)

    def scratch(self, nbytes):
        nbytes = int(max(nbytes, 256))
        if self.device_mode:
            import torch
            t = torch.empty(nbytes, dtype=torch.uint8, device=self.torch_device)
            if POISON:
                t.fill_(POISON_BYTE)
            self._keep.append(t)
            self._last_scratch = t
            return ctypes.c_void_p(t.data_ptr())
        buf = _OwnedBuffer(nbytes)
        self._owned.append(buf)
        self._last_scratch = buf
        if POISON:
            _lib.call("af_memset", buf.ptr, POISON_BYTE, nbytes, self.stream)
        return ctypes.c_void_p(buf.ptr)

    def watch_status(self, offset_bytes, message):
        """The int32 at `offset_bytes` of the scratch block allocated last is a status word the kernels of this call
        set on an index error; ``message(flags)`` describes it.  Host mode: read with the result (which
        synchronises anyway) and raised there.  Device mode: deferred (_DeferredStatus).  Call after the launch."""
        if self.device_mode:
            _deferred.watch(self._last_scratch, offset_bytes, message)
        else:
            self._status = (self._last_scratch.ptr + offset_bytes, message)

    # ---- outputs --------------------------------------------------------------------
    def out(self, shape, dtype):
        """Allocate the result; returns (device pointer, handle for `result`)."""
        dtype = np.dtype(dtype)
        shape = tuple(int(s) for s in shape)
        if self.device_mode:
            import torch
            t = torch.empty(shape, dtype=_torch_dtype(dtype), device=self.torch_device)
            if POISON:
                t.view(torch.uint8).fill_(POISON_BYTE)
            return ctypes.c_void_p(t.data_ptr()), t
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        buf = _OwnedBuffer(nbytes)
        self._owned.append(buf)
        if POISON and nbytes:
            _lib.call("af_memset", buf.ptr, POISON_BYTE, nbytes, self.stream)
        return ctypes.c_void_p(buf.ptr), (buf, shape, dtype)

    # ---- results produced and downloaded in row chunks ------------------------------------------------
    def result_rows(self, handle, launch, cast=None, edges=None):
        """
        The result of a call whose rows are independent, produced in ROW CHUNKS so that the download of chunk k (on the
        thread's copy stream, behind an event) runs while chunk k + 1 is computed: ``launch(r0, r1, p_out)`` enqueues
        the kernels of rows [r0, r1) on ``self.stream``, writing their rows at device pointer ``p_out``.  What the
        reference gets from dask's thread pool -- one block's copy-out beside the next block's arithmetic
        (africanus/dft/dask.py:37-51) -- for the plain numpy call: BASELINE configs[1] numpy in -> numpy out 95.5 -> 73.7
        ms (copy alone 71.7, transform alone 20.7; tools/bench_d2h_overlap.py, profiles/r05_d2h_overlap_*).
        ``cast`` complex64 / float32: converted ON THE DEVICE chunk by chunk, half the bytes cross PCIe.
        ``edges``: row numbers at which the result may be cut (ascending, from 0 to nrow; e.g. timestep boundaries of
        the fused predict, whose kernels work on whole timesteps) -- chunks then end on the edge nearest to every 256 MB.
        Device mode, small results and AFHIP_D2H_PIPELINE=0: one launch over all rows, then ``result``.
        """
        if self.device_mode:
            launch(0, int(handle.shape[0]), ctypes.c_void_p(handle.data_ptr()))
            return self.result(handle, cast)
        buf, shape, dtype = handle
        nrow = int(shape[0])
        row_bytes = int(np.prod(shape[1:], dtype=np.int64)) * dtype.itemsize
        out_dtype = dtype if cast is None else np.dtype(cast)
        narrow = cast is not None and out_dtype.itemsize * 2 == dtype.itemsize and out_dtype.kind == dtype.kind
        total = nrow * row_bytes
        chunk_bytes = int(float(os.environ.get("AFHIP_D2H_CHUNK_MB", "256")) * (1 << 20))    # read per call (tests use small chunks)
        if (total < 2 * chunk_bytes or os.environ.get("AFHIP_D2H_PIPELINE", "1") == "0"
                or not _USE_POOL or (cast is not None and not narrow)):
            launch(0, nrow, ctypes.c_void_p(buf.ptr))
            return self.result(handle, cast)
        rows = max(256, (chunk_bytes // row_bytes) // 256 * 256)
        if edges is None:
            cuts = list(range(0, nrow, rows)) + [nrow]
        else:
            edges = np.asarray(edges, dtype=np.int64)
            cuts = [0]
            while cuts[-1] < nrow:
                k = int(np.searchsorted(edges, cuts[-1] + rows, side="right")) - 1      # last edge within the chunk size
                nxt = int(edges[k]) if k >= 0 and edges[k] > cuts[-1] else int(edges[np.searchsorted(edges, cuts[-1], side="right")])
                cuts.append(min(nxt, nrow))
        arr = _pinned.array(shape, out_dtype)
        if arr is None:
            arr = np.empty(shape, dtype=out_dtype)
        copy_stream, event = _copy_stream()
        out_row_bytes = row_bytes // 2 if narrow else row_bytes
        nbuf = None
        if narrow:                       # the narrowed copy of the result (pool block, returned with the others)
            nbuf = _OwnedBuffer(nrow * out_row_bytes)
            self._owned.append(nbuf)
        host = arr.ctypes.data
        flags = ctypes.c_int32(0)
        try:
            for r0, r1 in zip(cuts[:-1], cuts[1:]):
                launch(r0, r1, ctypes.c_void_p(buf.ptr + r0 * row_bytes))
                src = buf.ptr + r0 * row_bytes
                if narrow:
                    n = (r1 - r0) * row_bytes // 8       # doubles in, floats out (complex = pairs)
                    _lib.call("af_convert_f64_to_f32", ctypes.c_void_p(src), ctypes.c_void_p(nbuf.ptr + r0 * out_row_bytes), n,
                              self.stream)
                    src = nbuf.ptr + r0 * out_row_bytes
                _lib.call("af_event_record", event, self.stream)
                _lib.call("af_stream_wait_event", copy_stream, event)
                _lib.call("af_memcpy_d2h", ctypes.c_void_p(host + r0 * out_row_bytes), ctypes.c_void_p(src),
                          (r1 - r0) * out_row_bytes, copy_stream)
            if self._status is not None:
                _lib.call("af_memcpy_d2h", ctypes.byref(flags), self._status[0], 4, self.stream)
        finally:
            # also on an error inside the loop: downloads already queued write into `arr` and read `buf` / `nbuf`, which go
            # back to their pools when this call ends -- nothing may still be in flight then
            for st in (copy_stream, self.stream):
                try:
                    _lib.call("af_stream_synchronize", st)
                except Exception:
                    pass
        if self._status is not None and flags.value:
            raise ValueError(self._status[1](flags.value))
        return arr

    def result(self, handle, cast=None):
        """Materialise the result: torch tensor (device mode) or numpy array (host mode)."""
        if self.device_mode:
            return handle if cast is None else handle.to(_torch_dtype(cast))
        buf, shape, dtype = handle
        arr = _pinned.array(shape, dtype) if cast is None else None
        if arr is None:
            arr = np.empty(shape, dtype=dtype)
        if arr.nbytes:
            _lib.call("af_memcpy_d2h", arr.ctypes.data_as(ctypes.c_void_p), buf.ptr, arr.nbytes, self.stream)
        flags = ctypes.c_int32(0)
        if self._status is not None:
            _lib.call("af_memcpy_d2h", ctypes.byref(flags), self._status[0], 4, self.stream)
        _lib.call("af_stream_synchronize", self.stream)
        if self._status is not None and flags.value:
            raise ValueError(self._status[1](flags.value))
        return arr if cast is None else arr.astype(cast, copy=False)
