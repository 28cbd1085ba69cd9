#!/bin/bash
# 332-seed soak of the randomised parity sweep with the bf16x3-split GEMMs (a) as shipped (mode 1) and (b) forced for every
# batch size (mode 2, tests/test_gpu_gemm_modes.py), + the bitwise-repeatability / memory soak.  Output: gpurun_out/r3_soak.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
echo "== AIMNET_FUZZ_SEEDS=0:332 python -m pytest tests/test_gpu_fuzz.py -q -m gpu   (engine default: gemm_bf3 = 1)"
AIMNET_FUZZ_SEEDS=0:332 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -3
echo "== AIMNET_FUZZ_SEEDS=0:332 python -m pytest tests/test_gpu_gemm_modes.py -q -m gpu -k 'random_configuration and bf3_every_size'"
AIMNET_FUZZ_SEEDS=0:332 python -m pytest tests/test_gpu_gemm_modes.py -q -m gpu -k "random_configuration and bf3_every_size" 2>&1 | tail -3
echo "== python tests/tools/soak.py   (10 080 atoms, 45 s, bitwise repeatability)"
python tests/tools/soak.py 2>&1 | tail -1
} > gpurun_out/r3_soak.txt 2>&1
cat gpurun_out/r3_soak.txt
