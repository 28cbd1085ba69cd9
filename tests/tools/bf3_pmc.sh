#!/bin/bash
# PMC passes over the bf3 GEMM bench (one shape): where do the cycles go?  usage: bf3_pmc.sh <outdir> [CFGS]
out=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
export SHAPES=one CFGS=${2:-2351,2152}
[ -f $R/gpurun_out/avail_counters.txt ] || rocprofv3 --list-avail > $R/gpurun_out/avail_counters.txt 2>&1
i=0
PASSES=${PASSES:-1 2 3 4 5}
P1="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"
P2="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
P3="SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"
P4="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"
P5="TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"
for pn in $PASSES; do
  eval pass=\$P$pn
  i=$pn
  rm -rf /tmp/pm$i
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm$i -- python $R/tests/tools/bf3_bench.py > $R/gpurun_out/$out/pass$i.log 2>&1
done
cd $R
python - <<PY > gpurun_out/$out/pmc_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob("/tmp/pm*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("aimnet::", "").replace("void ", "").split("(")[0][:60]
        if "gemm" not in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        print(f"   {c:45s} {agg[k][c][0] / max(1, agg[k][c][1]):.5g}")
PY
cat gpurun_out/$out/pmc_summary.txt
