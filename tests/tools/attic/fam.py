"""per-family ms/step of the default bench workload (one line)"""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-exact-f32", "--steps", "20", "--warmup", "5"], capture_output=True, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
print("%.4f" % d["ms_per_step"], {k: round(v, 4) for k, v in d["family_ms_per_step"].items()})
