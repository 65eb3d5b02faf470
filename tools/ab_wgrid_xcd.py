"""wgridder `model` at configs[4] (bench.py --workload wgrid / wgrid_f32planes shapes through tools/bench_wgridder.py's set-up):
the tile pass with chunks numbered one to one against one contiguous eighth of the chunk list per XCD, interleaved."""
import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for f32 in ("0", "1"):
        for xcd in ("0", "1"):
            env = dict(os.environ, AFHIP_WGRID_F32=f32, AFHIP_WGRID_XCD=xcd)
            out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "wgrid", "--steps", "5", "--warmup", "2",
                                  "--extras", "none", "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout
            d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
            print("float32 planes=%s xcd order=%s: step %.2f ms, tile pass %.2f ms, max abs err %.3e" % (
                f32, xcd, d["ms_per_step"], d["roofline"]["kernel_ms"], d["fp64_max_abs_err"]))
