#!/bin/bash
# presplit engine integration: parity + A/B against AIMNET_GEMM_PRESPLIT=0 (same library)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
true
for i in 1 2 3; do
  for ps in 1 0; do
    AIMNET_GEMM_PRESPLIT=$ps python bench.py --no-cpu-baseline --no-exact-f32 --no-hessian --steps 40 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('presplit=$ps', d['ms_per_step'], d['roofline']['gemm_ms_per_step'], json.dumps(d['family_ms_per_step']))" >> gpurun_out/r4d/ab.txt
  done
done
cat gpurun_out/r4d/tests.txt gpurun_out/r4d/ab.txt
