#!/bin/bash
# bitwise check + timing of the one-launch MLP sweeps, then a kernel trace of the chain-on run: bash tests/tools/chain_prof.sh
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/chain
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/tests/tools/chain_check.py > $R/gpurun_out/chain/check.txt 2>&1
cat $R/gpurun_out/chain/check.txt | tail -8
rm -rf /tmp/kt_chain
CELLS="7,3,5" STEPS=10 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_chain -- python $R/tests/tools/chain_check.py > /tmp/kt_chain.log 2>&1
python $R/tests/tools/prof_summary.py $(ls /tmp/kt_chain/*/*kernel_trace.csv | head -1) 42 > $R/gpurun_out/chain/ktrace.txt
grep -i "chain\|gemm_h2\|head_fused\|total" $R/gpurun_out/chain/ktrace.txt
