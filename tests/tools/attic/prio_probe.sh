#!/bin/bash
# issue-priority probe of the bf3 GEMM segments (measurement builds): big layer + family time
cd $GRAFT_REPO_ROOT
echo "baseline"; SHAPES=one CFGS=0 python tests/tools/bf3_bench.py 2>&1 | grep "N= 512" | cut -c1-90; python tests/tools/fam.py
for v in "3 0" "0 3" "1 0" "2 0" "3 1"; do set -- $v
  echo "PRIO_L=$1 PRIO_C=$2"
  bash tests/tools/variant.sh gemm_bf3 "-DAIMNET_BF3_PRIO_L=$1 -DAIMNET_BF3_PRIO_C=$2" bash -c "SHAPES=one CFGS=0 python tests/tools/bf3_bench.py 2>&1 | grep 'N= 512' | cut -c1-90; python tests/tools/fam.py"
done
