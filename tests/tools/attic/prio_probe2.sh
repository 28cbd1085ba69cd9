#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo "baseline"; python tests/tools/fam.py | cut -c1-120
  for v in "0 3" "0 1" "0 2"; do set -- $v
    echo "PRIO_L=$1 PRIO_C=$2"
    bash tests/tools/variant.sh gemm_bf3 "-DAIMNET_BF3_PRIO_L=$1 -DAIMNET_BF3_PRIO_C=$2" python tests/tools/fam.py | cut -c1-120
  done
done
