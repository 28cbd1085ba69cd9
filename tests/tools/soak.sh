#!/bin/bash
# 332-seed soak of the randomised parity sweep with the split GEMMs (a) as shipped (mode 1: fp16x2 operands above 256 rows), forced for every batch size (b) in the fp16x2 form and (c) in the bf16x3 form
# (gemm_h2.hip / gemm_bf3a.hip + the fused head on every fixture), (d) with the in-loop split (gemm_bf3.hip), + the
# bitwise-repeatability soak with EVERY evaluation compared (a rare wrong tile of a DMA pipeline shows up here).  -> gpurun_out/soak.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
echo "== AIMNET_FUZZ_SEEDS=0:332 python -m pytest tests/test_gpu_fuzz.py -q -m gpu   (engine defaults)"
AIMNET_FUZZ_SEEDS=0:332 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -3
echo "== AIMNET_FUZZ_SEEDS=0:332 ... tests/test_gpu_gemm_modes.py -k 'random_configuration and h2_every_size'   (fp16x2-split kernels on every batch size)"
AIMNET_FUZZ_SEEDS=0:332 python -m pytest tests/test_gpu_gemm_modes.py -q -m gpu -k "random_configuration and h2_every_size" 2>&1 | tail -3
echo "== AIMNET_FUZZ_SEEDS=0:332 ... tests/test_gpu_gemm_modes.py -k 'random_configuration and bf3_every_size'   (bf16x3-split kernels on every batch size)"
AIMNET_FUZZ_SEEDS=0:332 python -m pytest tests/test_gpu_gemm_modes.py -q -m gpu -k "random_configuration and bf3_every_size" 2>&1 | tail -3
echo "== AIMNET_FUZZ_SEEDS=0:332 ... -k 'random_configuration and bf3_split_in_loop'"
AIMNET_FUZZ_SEEDS=0:332 python -m pytest tests/test_gpu_gemm_modes.py -q -m gpu -k "random_configuration and bf3_split_in_loop" 2>&1 | tail -3
echo "== python tests/tools/soak.py   (10 080 atoms, 45 s, every evaluation compared bitwise)"
SOAK_EVERY=1 python tests/tools/soak.py 2>&1 | tail -1
} > gpurun_out/soak.txt 2>&1
cat gpurun_out/soak.txt
