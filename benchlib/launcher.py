"""`python bench.py --gpus N` without a launcher: start the N rank processes from a GPU-free parent, relay rank 0's
JSON line, fail if any rank fails (pure host logic: tests/test_bench_launcher.py)."""
import os
import socket
import subprocess
import sys
import time

from .common import ROOT


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
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + list(argv)
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

