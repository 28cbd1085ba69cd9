#!/bin/bash
# kernel trace of the benchmark step with the cluster conv backward on: per-kernel averages of the cluster kernels
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trc
AIMNET_CONV_CLUSTER=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trc -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/bc.log 2>&1
python $R/tests/tools/prof_summary.py $(ls /tmp/trc/*/*kernel_trace.csv | head -1) 18 | grep -E "total|cluster|conv_|unconcat" 
