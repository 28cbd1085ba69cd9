#!/bin/bash
# round 4, step b: the pre-split-activation GEMM (gemm_bf3a.hip) against the in-kernel-split one, all layer shapes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
python tests/tools/bf3a_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4b/shapes_gelu.txt
EPI=3 python tests/tools/bf3a_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4b/shapes_mul.txt
EPI=0 python tests/tools/bf3a_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4b/shapes_none.txt
SHAPES=one CFGS=452,224,234,432 OUT3=0 python tests/tools/bf3a_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4b/one_f32out.txt
BF3A=1 bash tests/tools/bf3_timing.sh 2>&1 | grep -v amdgpu > gpurun_out/r4b/timing_bf3a.txt
bash tests/tools/bf3_timing.sh 452 2>&1 | grep -v amdgpu > gpurun_out/r4b/timing_bf3.txt
tail -2 gpurun_out/r4b/shapes_gelu.txt; tail -1 gpurun_out/r4b/shapes_mul.txt; tail -1 gpurun_out/r4b/shapes_none.txt; cat gpurun_out/r4b/one_f32out.txt; tail -5 gpurun_out/r4b/timing_bf3a.txt
