#!/bin/bash
# round-3 profile at HEAD: kernel trace + PMC passes of the default bench, the un-profiled default bench (with the CPU baseline,
# the parity gate and the exact-fp32 record), the other workloads, MD throughput through the adapters, the bf3 GEMM shape table
cd $GRAFT_REPO_ROOT
bash tests/tools/pmc_bench.sh r3 > gpurun_out/r3_stdout.txt 2>&1
python bench.py > gpurun_out/r3/bench.json 2> gpurun_out/r3/bench.err
for w in batch256 md1024 taxol; do python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r3/bench_other_workloads.jsonl; done
python tests/tools/md_throughput.py 2>/dev/null | tail -1 > gpurun_out/r3/md_throughput.json
python tests/tools/bf3_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r3/gemm_bf3_shapes.txt
M=6400 python tests/tools/bf3_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r3/gemm_bf3_shapes_m6400.txt
python tests/tools/op_bench.py 2>/dev/null | tail -1 > gpurun_out/r3/op_bench.json
tail -3 gpurun_out/r3_stdout.txt; tail -c 1500 gpurun_out/r3/bench.json
# analytic Hessian-vector products: accuracy against the fp64 specification and the reference goldens, cost, kernel trace
python tests/tools/hvp_analytic.py 2>&1 | grep -v amdgpu > gpurun_out/r3/hvp_analytic.txt
python tests/tools/hvp_bench.py 2>/dev/null > gpurun_out/r3/hvp_bench.json
bash tests/tools/hvp_prof.sh 1 > /dev/null 2>&1; cp gpurun_out/hvp_prof.txt gpurun_out/r3/hvp_prof.txt
