#!/bin/bash
# round-6 profile at HEAD (run on the GPU box): kernel trace + PMC passes of the default bench, the un-profiled default bench (CPU
# baseline incl. one full-size evaluation, parity gates incl. the cold goldens at headline size, exact-fp32 / bf16x3 records, config-4
# Hessian), the other workloads, the A/B of the one-launch MLP sweeps, the per-sweep table, the size sweep, the kernel sequence
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
bash tests/tools/pmc_bench.sh r6 > gpurun_out/r6_stdout.txt 2>&1
python bench.py > gpurun_out/r6/bench.json 2> gpurun_out/r6/bench.err
rm -f gpurun_out/r6/bench_other_workloads.jsonl
for w in batch256 md1024 taxol; do python bench.py --workload $w --no-hessian 2>/dev/null | tail -1 >> gpurun_out/r6/bench_other_workloads.jsonl; done
bash tests/tools/ab_env.sh AIMNET_GEMM_CHAIN=0 --no-exact-f32 --no-hessian --no-repeat > gpurun_out/r6/chain_ab.txt 2>&1
python tests/tools/chain_sweep.py 2>&1 | grep -v amdgpu > gpurun_out/r6/chain_sweeps.txt
M=2304 python tests/tools/chain_sweep.py 2>&1 | grep -v amdgpu >> gpurun_out/r6/chain_sweeps.txt
python tests/tools/size_sweep.py 2>/dev/null > gpurun_out/r6/size_sweep.jsonl
AIMNET_GEMM_CHAIN=0 python tests/tools/size_sweep.py 2>/dev/null > gpurun_out/r6/size_sweep_per_layer.jsonl
bash tests/tools/kseq.sh > /dev/null 2>&1; cp gpurun_out/kseq.txt gpurun_out/r6/kernel_sequence.txt
tail -3 gpurun_out/r6_stdout.txt; tail -c 1200 gpurun_out/r6/bench.json; cat gpurun_out/r6/chain_ab.txt | tail -5
