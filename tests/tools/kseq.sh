#!/bin/bash
# the kernel sequence of ONE step of the default bench (names + durations + gaps, from a rocprofv3 kernel trace);
# KSEQ_ARGS="--workload taxol" for another workload
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kseq
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kseq -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-hessian --no-repeat $KSEQ_ARGS > /tmp/kseq.log 2>&1
python - <<PY > $R/gpurun_out/kseq.txt
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/kseq/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one step = from a mol_start_kernel to the next
idx = [k for k, r in enumerate(rows) if ("mol_start_kernel" in r["Kernel_Name"] or "prep_small_kernel" in r["Kernel_Name"])]
a, b = idx[4], idx[5]
prev_end = None
tot = 0.0
for r in rows[a - 2:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = r["Kernel_Name"].replace("aimnet::", "").replace("void ", "").split("(")[0][:60]
    print(f"{name:62s} dur {(e - s) / 1e3:7.1f} us  gap {gap:6.1f} us  grid {r.get('Grid_Size', r.get('Grid_Size_X', ''))}")
    prev_end = e
print("launches per step:", b - a, " span (us):", (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
PY
cat $R/gpurun_out/kseq.txt
