#!/bin/bash
# PMC passes over one GEMM config (run on the GPU box): usage pmc_gemm.sh <cfg> <outdir>
cfg=$1; out=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
            "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
            "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAVES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
            "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr"; do
  tag=$(echo $pass | cut -d' ' -f1)
  CFGS=$cfg EPI=0 ONESHAPE=1 timeout 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/$out/$tag -- python $R/tests/tune_gemm.py > $R/gpurun_out/$out/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/$out/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "gemm" not in r["Kernel_Name"]: continue
        k = r["Counter_Name"]; agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for k, (v, n) in agg.items(): print(f"{k:32s} per-launch {v/n:16.1f}  (n={n})")
PY
