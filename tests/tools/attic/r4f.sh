#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
python -m pytest tests/test_gpu_configs.py tests/test_gpu_calculator.py -x -q 2>&1 | tail -15 > gpurun_out/r4f/tests.txt
for i in 1 2 3; do
  for hf in 1 0; do
    AIMNET_HEAD_FUSED=$hf python bench.py --no-cpu-baseline --no-exact-f32 --no-hessian --steps 40 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('head_fused=$hf', d['ms_per_step'], d['roofline']['gemm_ms_per_step'], json.dumps(d['family_ms_per_step']))" >> gpurun_out/r4f/ab.txt
  done
done
AIMNET_GEMM_PRESPLIT=0 python bench.py --no-cpu-baseline --no-exact-f32 --no-hessian --steps 40 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('presplit=0', d['ms_per_step'], d['roofline']['gemm_ms_per_step'], json.dumps(d['family_ms_per_step']))" >> gpurun_out/r4f/ab.txt
cat gpurun_out/r4f/tests.txt gpurun_out/r4f/ab.txt
