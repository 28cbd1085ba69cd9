#!/bin/bash
# LDS bank-conflict counters per kernel over the default bench (one rocprofv3 --pmc pass): which kernels lose LDS cycles to conflicts
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/lds_pmc
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ldspmc
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/ldspmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-hessian > /tmp/ldspmc.log 2>&1
cd $R
python - <<PY > gpurun_out/lds_pmc/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob("/tmp/ldspmc/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("aimnet::", "").replace("void ", "").split("(")[0][:52]
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
names = ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_BUSY_CYCLES"]
print("per-dispatch averages; kernel | " + " | ".join(names) + " | conflict/active")
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_LDS_BANK_CONFLICT", [0, 1])[0] / max(1, agg[k].get("SQ_LDS_BANK_CONFLICT", [0, 1])[1])):
    v = {c: agg[k][c][0] / max(1, agg[k][c][1]) for c in names if c in agg[k]}
    print(f"{k:52s} | " + " | ".join(f"{v.get(c, 0):.4g}" for c in names) + f" | {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, v.get('SQ_LDS_IDX_ACTIVE', 1)):.2f}")
PY
head -30 gpurun_out/lds_pmc/summary.txt
