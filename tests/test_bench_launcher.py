"""
bench.py's launcher logic, on CPU (no GPU call anywhere): the environment `--gpus N` gives the rank processes it
starts itself when no launcher did (it used to fall through to ONE rank and print n_gpus 1), which
extra workloads ride in the default line, the refusal to report an N-GPU number on fewer devices, and the sizing of
the CPU-baseline sample (warm-up, >= 0.5 s single-thread probe, floored all-threads sample).
"""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rank_environments_are_what_torchrun_would_export(bench):
    envs = bench.rank_environments(8, 29511, {"PATH": "/bin", "RANK": "7", "WORLD_SIZE": "1"})
    assert len(envs) == 8
    for r, e in enumerate(envs):
        assert e["RANK"] == str(r) and e["LOCAL_RANK"] == str(r)          # one node: local rank = rank = device
        assert e["WORLD_SIZE"] == "8" and e["LOCAL_WORLD_SIZE"] == "8"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511"
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"                     # dmabuf IPC: RCCL needs it on this pool
        assert e["PATH"] == "/bin"
    assert len({id(e) for e in envs}) == 8                               # separate dicts


def test_extras_ride_only_in_the_default_single_gpu_line(bench):
    a = bench.parse([])
    assert bench.extras_requested(a, 1) == bench.EXTRA_WORKLOADS
    assert bench.extras_requested(a, 2) == ()
    assert bench.extras_requested(bench.parse(["--rows", "5000"]), 1) == ()
    assert bench.extras_requested(bench.parse(["--workload", "degrid"]), 1) == ()
    assert bench.extras_requested(bench.parse(["--extras", "none"]), 1) == ()
    assert bench.extras_requested(bench.parse(["--rows", "5000", "--extras", "degrid,wgrid"]), 1) == ("degrid", "wgrid")
    assert bench.extras_requested(bench.parse(["--workload", "degrid", "--extras", "all"]), 1) == ("dft_complex", "dft_f32", "gauss", "fused_dde", "fused_dde_ant", "fused_dde_ant128", "fused_dde_ant_c64", "fused_dde_c64", "wgrid", "wgrid_f32planes")
    with pytest.raises(SystemExit):
        bench.extras_requested(bench.parse(["--extras", "nonsense"]), 1)


def test_self_launch_refuses_without_devices():
    """No GPU in this container: `--gpus 2` without a launcher must fail loudly (non-zero, no JSON line), never
    print a one-GPU number; same for the threads executor and for a rank whose device does not exist."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "AFHIP_BENCH_DEVICE")}
    for extra in ([], ["--executor", "threads"]):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"] + extra,
                           cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode != 0
        assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    env2 = dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=ROOT,
                       env=env2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode != 0 and b"WORLD_SIZE 4 != --gpus 2" in p.stderr


def test_cpu_sample_sizing(bench):
    """A fake CPU kernel of 1 ms per row on one thread, 8 threads at 50 % parallel efficiency: the probe must run
    >= 0.5 s, the all-threads sample >= the floor, both after a discarded warm-up call."""
    calls = []

    def single(n):
        calls.append(("s", n))
        clock[0] += 1e-3 * n

    def parallel(n):
        calls.append(("p", n))
        return 1e-3 * n / 4.0

    clock = [0.0]
    import time
    real = time.perf_counter
    time.perf_counter = lambda: clock[0]
    try:
        s = bench.sized_cpu_sample(single, parallel, 10 ** 6, 8, 2.0)
    finally:
        time.perf_counter = real
    assert calls[0] == ("s", 16)                                  # warm-up
    assert s["probe_s"] >= 0.5 and abs(s["per_row_s"] - 1e-3) < 1e-9 and s["probe_rows"] >= 500
    assert s["seconds"] >= 2.0 and s["rows"] % 8 == 0 and s["rows"] >= 8000
    assert s["rows"] < 40000                                      # bounded: not minutes of CPU work


def test_a_rank_that_dies_early_ends_the_job(bench):
    """the launcher waited on rank 0 only, so a rank that exited early (device missing, import error) left
    rank 0 blocked in the rendezvous until the timeout.  Now the first non-zero exit stops and reaps the others."""
    import tempfile
    import time
    log = tempfile.TemporaryFile()
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", "import time; time.sleep(120)"], stdout=subprocess.PIPE),
             subprocess.Popen([sys.executable, "-c", "import sys; print('no such device'); sys.exit(7)"], stdout=log,
                              stderr=subprocess.STDOUT)]
    code, out0, message = bench.supervise(procs, [None, log], 100)
    assert code == 7 and time.time() - t0 < 30
    assert "rank 1 exited with code 7" in message and "no such device" in message
    assert all(p.returncode is not None for p in procs)                  # reaped, no zombies
    # the good case: both exit 0, rank 0's output is relayed
    procs = [subprocess.Popen([sys.executable, "-c", "print('{\"ok\": 1}')"], stdout=subprocess.PIPE),
             subprocess.Popen([sys.executable, "-c", "pass"], stdout=subprocess.DEVNULL)]
    code, out0, message = bench.supervise(procs, [None, None], 100)
    assert code == 0 and out0.decode().strip() == '{"ok": 1}' and message == ""
    # the timeout
    procs = [subprocess.Popen([sys.executable, "-c", "import time; time.sleep(120)"], stdout=subprocess.PIPE)]
    code, out0, message = bench.supervise(procs, [None], 1)
    assert code == 1 and "did not finish" in message and procs[0].returncode is not None
