"""Perley degridder at configs[4]: blocks numbered one to one against one contiguous eighth of the (uv-tile ordered) blocks per
XCD, interleaved same-box rounds."""
import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(3):
    for xcd in ("0", "1"):
        env = dict(os.environ, AFHIP_DEGRID_XCD=xcd)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "degrid", "--steps", "10", "--warmup", "2",
                              "--extras", "none", "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout
        d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        print("xcd order=%s: step %.2f ms, kernel %.2f ms, max abs err %.3e" % (xcd, d["ms_per_step"], d["roofline"]["kernel_ms"], d["fp64_max_abs_err"]))
