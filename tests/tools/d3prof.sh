#!/bin/bash
# kernel trace of config 3 with external DFT-D3 on (tests/tools/skin_probe.py's runs): the list / Coulomb / D3 kernels.  GPU box.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trd3
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trd3 -- python $R/tests/tools/skin_probe.py > /tmp/d3.log 2>&1
tail -1 /tmp/d3.log | cut -c1-300
python - <<PY | tee $R/gpurun_out/d3_prof.txt
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("/tmp/trd3/*/*kernel_trace.csv")[0])))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("aimnet::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if any(t in n for t in ("d3", "nlist", "coulomb", "bin_", "scan", "wrap")):
        agg[n[:70] + " grid " + r.get("Grid_Size", r.get("Grid_Size_X", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{k:90s} n={len(v):4d} median={v[len(v)//2]:9.1f}us")
PY
