#!/bin/bash
# time every probe build under gpurun_in/chain_*.so with tests/tools/chain_sweep.py (chain column only is meaningful)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/chain
for so in "" $(ls $R/gpurun_in/chain_*.so); do
  name=$(basename "${so:-base}" .so)
  echo "== $name"
  AIMNET_HIP_LIB=$so PASSES=${PASSES:-1,2} timeout 300 python $R/tests/tools/chain_sweep.py 2>&1 | grep "pass\|sum" | sed 's/|.*//'
done 2>&1 | tee $R/gpurun_out/chain/variants.txt
