#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
python tests/tools/bf3a_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4c/shapes_gelu.txt
EPI=3 python tests/tools/bf3a_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4c/shapes_mul.txt
BF3A=1 bash tests/tools/bf3_timing.sh 2>&1 | grep -v amdgpu > gpurun_out/r4c/timing_bf3a.txt
cat gpurun_out/r4c/shapes_gelu.txt | cut -c1-190; tail -1 gpurun_out/r4c/shapes_mul.txt; tail -3 gpurun_out/r4c/timing_bf3a.txt
