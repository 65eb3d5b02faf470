"""
RCCL on the one-GPU test box: the path's only collective -- the all-reduce of the per-channel
chi^2 vector (north_star "RCCL all-reduce over xGMI") -- had only ever run under gloo, because one device
means world size 1 and a world of one skipped the process group.  Here a child process (started by Popen: never an exec
from this GPU-initialised process) creates a WORLD-SIZE-1 ``nccl`` process group, which loads librccl and builds a
communicator, and reduces the chi^2 vector of a real predict on the device through ``sharding.allreduce_chi2``.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, socket
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from codex_africanus_amd import sharding, dft
from codex_africanus_amd.testing import synthetic_inputs, real_image
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
d = synthetic_inputs(seed=8, nrow=3000, nchan=64, nsrc=30, nant=7)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
vis, c2, (lo, hi) = sharding.predict_shard(0, 1, t(real_image(d)), t(d["uvw"]), t(d["lm"]), t(d["frequency"]))
data = vis + 0.01
vis, c2, (lo, hi) = sharding.predict_shard(0, 1, t(real_image(d)), t(d["uvw"]), t(d["lm"]), t(d["frequency"]), data=data)
torch.cuda.synchronize(dev)
local = sharding.chi2(vis, data)
# (two runs of the chi^2 kernel agree to the order of its atomic adds)
assert (lo, hi) == (0, 3000) and torch.allclose(c2, local, rtol=1e-12, atol=0), "a sum over one rank is the rank's own vector"
x = torch.arange(64, dtype=torch.float64, device=dev)
y = sharding.allreduce_chi2(x.clone())
assert torch.equal(x, y)
img = sharding.allreduce_image(torch.ones(5, 4, 2, dtype=torch.float64, device=dev))
assert float(img.sum()) == 40.0
rccl = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})
dist.destroy_process_group()
print("RCCL_OK %s %.6e" % (",".join(os.path.basename(p) for p in rccl), float(c2.sum())))
"""


def test_world_size_one_nccl_group_reduces_chi2_on_the_device(tmp_path):
    script = tmp_path / "rccl_child.py"
    script.write_text(CHILD)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, str(script), ROOT], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900)
    out, err = p.stdout.decode("utf-8", "replace"), p.stderr.decode("utf-8", "replace")
    assert p.returncode == 0, (out[-2000:], err[-4000:])
    ok = [ln for ln in out.splitlines() if ln.startswith("RCCL_OK")]
    assert len(ok) == 1 and "librccl" in ok[0], out[-2000:]


def test_bench_force_dist_runs_the_collective_through_rccl():
    """`bench.py --gpus 1 --backend nccl --force-dist`: the benchmark's own step with the process group in place"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "AFHIP_BENCH_DEVICE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--force-dist",
                        "--steps", "3", "--warmup", "1", "--rows", "30000", "--sources", "200", "--no-cpu-baseline",
                        "--check-rows", "64", "--extras", "none"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    out, err = p.stdout.decode("utf-8", "replace"), p.stderr.decode("utf-8", "replace")
    assert p.returncode == 0, (out[-2000:], err[-4000:])
    r = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["config"]["rccl_loaded"] is True and "nccl" in r["config"]["collective"]
    assert r["fp64_max_abs_err"] < 1e-8
