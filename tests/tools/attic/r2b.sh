#!/bin/bash
# MFMA 4x4x1 rate microbench + PMC counters of the conv kernels (VALU and MFMA forms)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
./gpurun_in/mfma4_rate > gpurun_out/r2b/mfma4_rate.txt 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in 0 3; do
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_MFMA"; do
    tag=$(echo $pass | cut -d' ' -f1)
    AIMNET_CONV_MFMA=$m timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/r2b/m${m}_$tag -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2b/m${m}_$tag.log 2>&1
  done
done
cd $R
python - <<'PY' > gpurun_out/r2b/pmc_summary.txt
import csv, glob, collections
for m in (0, 3):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in sorted(glob.glob(f"gpurun_out/r2b/m{m}_*/*/*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("aimnet::", "").replace("void ", "").split("(")[0][:48]
            if "conv" not in k and "unconcat" not in k: continue
            a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    names = sorted({c for k in agg for c in agg[k]})
    print(f"AIMNET_CONV_MFMA={m}: per-dispatch averages; kernel | " + " | ".join(names))
    for k in sorted(agg, key=lambda k: -agg[k].get("GRBM_GUI_ACTIVE", [0, 1])[0]):
        print(f"{k:48s} | " + " | ".join(f"{agg[k][c][0] / max(1, agg[k][c][1]):.4g}" if c in agg[k] else "-" for c in names))
PY
rm -rf gpurun_out/r2b/m0_* gpurun_out/r2b/m3_*
cat gpurun_out/r2b/mfma4_rate.txt gpurun_out/r2b/pmc_summary.txt
