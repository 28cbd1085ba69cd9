#!/bin/bash
# cluster conv backward: parity tests, then A/B of the step and the conv_bwd family on the benchmark workload
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cl
(timeout 900 python -m pytest tests/test_gpu_conv_cluster.py -q -x 2>&1 | tail -15) > gpurun_out/cl/tests.txt
cat gpurun_out/cl/tests.txt
for o in 0 1 0 1; do
  AIMNET_CONV_CLUSTER=$o timeout 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cluster $o', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['family_ms_per_step'].items()})"
done
