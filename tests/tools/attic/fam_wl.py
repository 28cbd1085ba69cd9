"""per-family ms/step of one bench workload: python tests/tools/fam_wl.py <workload>"""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-exact-f32", "--steps", "20", "--warmup", "5", "--workload", sys.argv[1]],
                     capture_output=True, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
print(sys.argv[1], "%.4f ms" % d["ms_per_step"], "%.3g atoms*steps/s" % d["value"], {k: round(v, 4) for k, v in d["family_ms_per_step"].items()})
