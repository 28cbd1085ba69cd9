#!/bin/bash
# Where the list-free DSF walk's cycles go (GPU box): PMC passes restricted to coulomb_dsf_walk_kernel over a short default bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/dsf_pmc
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ|TCP|TCC|TA|TD|GRBM)_[A-Z0-9_]+\b" | sort -u > $R/gpurun_out/dsf_pmc/avail.txt
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" \
            "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --kernel-include-regex "coulomb_dsf_walk|nlist_cell|conv_fwd_kernel" --output-format csv -d /tmp/dsfpmc_$tag -- \
    python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-hessian --no-repeat > $R/gpurun_out/dsf_pmc/$tag.log 2>&1
done
cd $R
python - <<'PY' > gpurun_out/dsf_pmc/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob("/tmp/dsfpmc_*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("aimnet::", "").replace("void ", "").split("(")[0][:40]
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        print(f"   {c:28s} {agg[k][c][0] / max(1, agg[k][c][1]):14.5g}  (per dispatch, {agg[k][c][1]} dispatches)")
PY
cat gpurun_out/dsf_pmc/summary.txt
