#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
python -m pytest tests/test_gpu_configs.py tests/test_gpu_calculator.py -x -q 2>&1 | tail -5 > gpurun_out/r4e/tests.txt
for i in 1 2 3; do
  for ps in 1 0; do
    AIMNET_GEMM_PRESPLIT=$ps python bench.py --no-cpu-baseline --no-exact-f32 --no-hessian --steps 40 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('presplit=$ps', d['ms_per_step'], d['roofline']['gemm_ms_per_step'], json.dumps(d['family_ms_per_step']))" >> gpurun_out/r4e/ab.txt
  done
done
CFGS=452,224,234,432,422,223,851 python tests/tools/bf3a_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4e/tiles_gelu.txt
CFGS=452,224,234,432,422,223,851 EPI=3 python tests/tools/bf3a_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4e/tiles_mul.txt
CFGS=452,224,234,432,422,223,851 EPI=0 python tests/tools/bf3a_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4e/tiles_none.txt
cat gpurun_out/r4e/tests.txt gpurun_out/r4e/ab.txt
