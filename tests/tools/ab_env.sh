#!/bin/bash
# Same-box A/B of one environment switch: `bash tests/tools/ab_env.sh AIMNET_PREP_FUSED=0 [bench args]`
# runs bench.py in the order A B B A A B B A (A = defaults, B = with the variable set) and prints the mean ms/step of each.
V=$1; shift
python - "$V" "$@" <<'PY'
import json, os, subprocess, sys
v, extra = sys.argv[1], sys.argv[2:]
key, val = v.split("=", 1)
res = {"A": [], "B": []}
for which in "ABBAABBA":
    env = dict(os.environ)
    if which == "B": env[key] = val
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "40", *extra], capture_output=True, text=True, env=env).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    res[which].append(d["ms_per_step"])
for k, name in (("A", "defaults"), ("B", v)):
    print(f"{name:40s} mean {sum(res[k])/len(res[k]):.4f} ms/step   runs {' '.join(f'{x:.4f}' for x in res[k])}")
print(f"B vs A: {100*(sum(res['B'])/sum(res['A'])-1):+.2f} %")
PY
