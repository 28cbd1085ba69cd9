#!/bin/bash
# rocprofv3 passes over the default bench (run on the GPU box): kernel trace + two PMC passes, summaries into gpurun_out/$1
out=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$out/trace -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-hessian --no-repeat > $R/gpurun_out/$out/bench_traced.log 2>&1
for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/$out/$tag -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-hessian --no-repeat > $R/gpurun_out/$out/$tag.log 2>&1
done
cd $R
python tests/tools/prof_summary.py $(ls gpurun_out/$out/trace/*/*kernel_trace.csv | head -1) 17 > gpurun_out/$out/kernel_summary.txt
cp $(ls gpurun_out/$out/trace/*/*kernel_stats.csv | head -1) gpurun_out/$out/kernel_stats.csv
python - <<PY > gpurun_out/$out/pmc_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob("gpurun_out/$out/*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("aimnet::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
names = sorted({c for k in agg for c in agg[k]})
print("per-dispatch averages; kernel | " + " | ".join(names))
for k in sorted(agg, key=lambda k: -agg[k].get("GRBM_GUI_ACTIVE", [0, 1])[0]):
    print(f"{k:48s} | " + " | ".join(f"{agg[k][c][0] / max(1, agg[k][c][1]):.4g}" if c in agg[k] else "-" for c in names))
PY
python - <<PY > gpurun_out/$out/pmc.json
# GEMM-family aggregates for bench.py's roofline.traffic (profiles/r2_pmc.json), stamped with the commit (gpurun_in/HEAD)
import csv, glob, collections, json, os
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in sorted(glob.glob("gpurun_out/$out/*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if not any(k in r["Kernel_Name"] for k in ("gemm_nt", "gemm_bf3", "gemm_h2", "gemm_chain", "head_fused")): continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
launches = max(1, n["TCC_EA0_RDREQ_sum"])
rd, wr = tot["TCC_EA0_RDREQ_sum"] * 128 / launches, tot["TCC_EA0_WRREQ_sum"] * 64 / launches
head = open("gpurun_in/HEAD").read().strip() if os.path.exists("gpurun_in/HEAD") else "unknown"
print(json.dumps({"source": "tests/tools/pmc_bench.sh (rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum ..., bench.py --steps 3 --warmup 1)",
                  "commit": head, "gemm_launches_counted": launches, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                  "traffic_bytes_per_launch": rd + wr,
                  "mfma_busy_frac_in_kernel": tot["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, 1024 * tot["GRBM_GUI_ACTIVE"] / 8)}, indent=1))
PY
rm -rf gpurun_out/$out/SQ_VALU_MFMA_BUSY_CYCLES gpurun_out/$out/TCC_EA0_RDREQ_sum gpurun_out/$out/trace
head -30 gpurun_out/$out/kernel_summary.txt; head -24 gpurun_out/$out/pmc_summary.txt
