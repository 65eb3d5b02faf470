"""
Block -> GPU placement of the host (numpy in / numpy out) path on one 8 x MI355X node.

north_star: "dask chunks over the row axis map to GPUs of one node".  The reference's dask wrappers turn every
(row, chan) block into one call of the array-level function (africanus/rime/dask_predict.py:311-369,
africanus/dft/dask.py:37-51) executed by whichever worker thread the scheduler picks; here that call has to land
on A device.  Two rules, both pure arithmetic (tested on CPU):

  * ``block``  -- the wrappers hand every block function its row-block index (an extra, one-element-per-block
                  array riding along the "row" axis); block k runs on device ``k % n_devices``.  Consecutive row
                  blocks, which the scheduler releases together, go to different GPUs, and the same block always
                  lands on the same GPU (deterministic, re-runnable).
  * ``thread`` -- opt-in (``AFHIP_PLACEMENT=thread``): calls take the device of the calling thread, block index or
                  not: threads are numbered in order of first use and thread j is pinned to device ``j % n_devices``,
                  so a pool of N >= n_devices worker threads (e.g. another library's dask wrappers calling ours)
                  drives every GPU.

``AFHIP_PLACEMENT=block|thread|none`` (default ``block``: the block's device when the call carries a block index;
a call WITHOUT one -- a plain function call -- runs on the calling thread's current HIP device, i.e. whatever the
caller's af_set_device / torch.cuda.set_device / HIP_VISIBLE_DEVICES selected: a rank-per-GPU job keeps its rank's
device; ``thread``: as above; ``none``: never touch the device selection).  ``AFHIP_DEVICES=0,2,5``
restricts and orders the devices used.  A re-placed call restores the thread's previous device when it returns
(``_device.Call.__exit__``), so a later torch op on the same thread lands where it did before.  Device-mode calls
(torch ROCm tensors in) are never re-placed: they run where their tensors live.
"""
import contextlib
import os
import threading

_POLICIES = ("block", "thread", "none")
_policy = os.environ.get("AFHIP_PLACEMENT", "block")
if _policy not in _POLICIES:
    raise ValueError("AFHIP_PLACEMENT must be one of %s" % (_POLICIES,))

_lock = threading.Lock()
_thread_slots = {}            # thread ident -> arrival number
_tls = threading.local()      # .block: pending row-block index of the call being made on this thread
_devices = None               # resolved lazily: tuple of device ordinals


def set_policy(policy):
    global _policy
    if policy not in _POLICIES:
        raise ValueError("policy must be one of %s" % (_POLICIES,))
    _policy = policy


def get_policy():
    return _policy


def parse_device_list(text, count):
    """'0,2,5' -> (0, 2, 5), validated against `count` visible devices; empty / None -> all of them."""
    if not text:
        return tuple(range(count))
    devs = tuple(int(t) for t in text.split(",") if t.strip() != "")
    if not devs or any(d < 0 or d >= count for d in devs) or len(set(devs)) != len(devs):
        raise ValueError("AFHIP_DEVICES=%r does not name distinct devices below %d" % (text, count))
    return devs


def devices():
    """Device ordinals placement distributes over (AFHIP_DEVICES or every visible device)."""
    global _devices
    if _devices is None:
        from . import _lib
        with _lock:
            if _devices is None:
                _devices = parse_device_list(os.environ.get("AFHIP_DEVICES"), _lib.device_count())
    return _devices


def set_devices(devs):
    """Override the device list (tests; embedding applications).  None re-reads the environment."""
    global _devices
    _devices = None if devs is None else tuple(int(d) for d in devs)


def device_for_block(block_index, devs):
    """Row block k -> devs[k % len(devs)]."""
    return devs[int(block_index) % len(devs)]


def thread_slot(ident=None):
    """Arrival number of a thread (0 for the first thread that ever asked, 1 for the next, ...)."""
    ident = threading.get_ident() if ident is None else ident
    with _lock:
        slot = _thread_slots.get(ident)
        if slot is None:
            slot = len(_thread_slots)
            _thread_slots[ident] = slot
    return slot


def device_for_thread(devs, ident=None):
    return devs[thread_slot(ident) % len(devs)]


@contextlib.contextmanager
def block(block_index):
    """Calls made inside run on the device of row block `block_index` (None, a python int or a one-element
    array as dask hands it)."""
    prev = getattr(_tls, "block", None)
    if block_index is not None and not isinstance(block_index, int):
        import numpy as np
        block_index = int(np.asarray(block_index).reshape(-1)[0])
    _tls.block = block_index
    try:
        yield
    finally:
        _tls.block = prev


def choose(devs=None, policy=None):
    """The device a host-mode call made now on this thread should be moved to, or None (= leave the thread's current
    device alone: policy ``none``, or the default policy and no block index)."""
    policy = _policy if policy is None else policy
    if policy == "none":
        return None
    blk = getattr(_tls, "block", None)
    if policy == "block" and blk is None:
        return None
    devs = devices() if devs is None else devs
    if not devs:
        return None
    if policy == "block":
        return device_for_block(blk, devs)
    return device_for_thread(devs)


def activate():
    """Select the device of the call about to be made (host mode).  Returns (chosen, previous): the ordinal now
    current (None when the selection was left alone) and the ordinal to restore afterwards (None = nothing to do)."""
    dev = choose()
    if dev is None:
        return None, None
    from . import _lib
    prev = _lib.get_device()
    if prev != dev:
        _lib.set_device(dev)
        return dev, prev
    return dev, None


def restore(prev):
    if prev is not None:
        from . import _lib
        _lib.set_device(prev)
