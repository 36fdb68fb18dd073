"""Print the fields of a bench.py JSON line that a GPU call's log should show."""
import json
import sys

line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
r = json.loads(line)
print({k: r[k] for k in ("value", "ms_per_step", "n_gpus", "dtype")})
roof = r.get("roofline", {})
print({k: roof[k] for k in roof if k not in ("kernels", "step")})
if "step" in roof:
    print("step:", {k: roof["step"][k] for k in ("achieved", "frac", "traffic") if k in roof["step"]})
if "cpu_baseline" in r:
    print("cpu:", r["cpu_baseline"])
print("full-frame it/s:", r["config"].get("full_frame_attack_iters_per_s"))
for c in r["config"].get("other_configs", []):
    print("other:", c["config"], c["ms_per_iteration"], "ms, igemm", c["roofline"]["frac"] if c.get("roofline") else None)
try:
    for k in roof.get("kernels", []):
        print(f"  {k['kernel'][:70]:70s} {k['ms']:8.4f} ms  {k['frac']:.3f}")
except BrokenPipeError:                 # `| head` in tools/gpu_call.sh
    sys.stderr.close()
