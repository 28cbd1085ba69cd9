#!/bin/bash
# kernel trace of the pme / ewald steps of tests/tools/pme_bench.py: per-kernel averages for the Coulomb kernels.  GPU box.
# usage: bash tests/tools/pme_prof.sh [rep ...]   -> gpurun_out/pme_prof.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmeprof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pmeprof -- python $R/tests/tools/pme_bench.py ${@:-7,3,5} > /tmp/pmeprof.log 2>&1
tail -3 /tmp/pmeprof.log
python - <<PY | tee $R/gpurun_out/pme_prof.txt
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("/tmp/pmeprof/*/*kernel_trace.csv")[0])))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("aimnet::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if any(t in n for t in ("pme_", "ewald_", "coulomb_")):
        agg[n[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{k:72s} n={len(v):4d} median={v[len(v)//2]:9.1f}us min={v[0]:9.1f} max={v[-1]:9.1f}")
PY
