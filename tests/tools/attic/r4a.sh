#!/bin/bash
# round 4, step a: new GEMM prologue (step-1 loads in front of the first wait) against the round-3 prologue
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_gemm_modes.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3 > gpurun_out/r4a/tests.txt
CFGS=0 python tests/tools/bf3_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4a/shapes_new.txt
AIMNET_HIP_LIB=gpurun_in/oldpro.so CFGS=0 python tests/tools/bf3_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4a/shapes_old.txt
bash tests/tools/bf3_timing.sh 452 2>&1 | grep -v amdgpu > gpurun_out/r4a/timing_new.txt
EXTRA=-DAIMNET_BF3_OLD_PROLOGUE bash tests/tools/bf3_timing.sh 452 2>&1 | grep -v amdgpu > gpurun_out/r4a/timing_old.txt
bash tests/tools/ab.sh gpurun_in/oldpro.so --no-exact-f32 --no-hessian 2>&1 | grep -v amdgpu > gpurun_out/r4a/ab.txt
cat gpurun_out/r4a/tests.txt gpurun_out/r4a/ab.txt; tail -1 gpurun_out/r4a/shapes_new.txt; tail -1 gpurun_out/r4a/shapes_old.txt
