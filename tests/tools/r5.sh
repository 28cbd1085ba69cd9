#!/bin/bash
# round-5 profile at HEAD (run on the GPU box): kernel trace + PMC passes of the default bench, the un-profiled default bench (CPU
# baseline incl. one full-size evaluation, parity gates incl. the cold-weight goldens at the reference's literal gates, exact-fp32 and
# bf16x3 records, config-4 Hessian), the other workloads WITH their CPU baseline and parity records, the A/B of the fp16x2-split
# operands, MD throughput through the adapters, the GEMM shape tables (fp16x2 vs bf16x3 vs exact fp32), the literal-gate parity table,
# the weight-seed table, the kernel sequence, the stand-alone ops and the HVP records
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
bash tests/tools/pmc_bench.sh r5 > gpurun_out/r5_stdout.txt 2>&1
python bench.py > gpurun_out/r5/bench.json 2> gpurun_out/r5/bench.err
rm -f gpurun_out/r5/bench_other_workloads.jsonl
for w in batch256 md1024 taxol; do python bench.py --workload $w --no-hessian 2>/dev/null | tail -1 >> gpurun_out/r5/bench_other_workloads.jsonl; done
bash tests/tools/ab_env.sh AIMNET_GEMM_H2=0 --no-exact-f32 --no-hessian --no-repeat > gpurun_out/r5/h2_ab.txt 2>&1
python tests/tools/md_throughput.py 2>/dev/null | tail -1 > gpurun_out/r5/md_throughput.json
python tests/tools/h2_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r5/gemm_h2_shapes_gelu.txt
EPI=3 python tests/tools/h2_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r5/gemm_h2_shapes_mul.txt
EPI=0 python tests/tools/h2_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r5/gemm_h2_shapes_none.txt
STAT=pos EPI=0 python tests/tools/h2_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r5/gemm_h2_shapes_none_positive_operands.txt
STAT=gelu EPI=0 python tests/tools/h2_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r5/gemm_h2_shapes_none_gelu_operands.txt
python tests/tools/parity_literal.py 2>/dev/null > gpurun_out/r5/parity_literal_table.md
bash tests/tools/kseq.sh > /dev/null 2>&1; cp gpurun_out/kseq.txt gpurun_out/r5/kernel_sequence.txt
KSEQ_ARGS="--workload taxol" bash tests/tools/kseq.sh > /dev/null 2>&1; cp gpurun_out/kseq.txt gpurun_out/r5/kernel_sequence_taxol.txt
python tests/tools/op_bench.py 2>/dev/null | tail -1 > gpurun_out/r5/op_bench.json
python tests/tools/hvp_bench.py 2>/dev/null > gpurun_out/r5/hvp_bench.json
rm -f gpurun_out/r5/weight_seeds.jsonl
AIMNET_SEED_TABLE=$GRAFT_REPO_ROOT/gpurun_out/r5/weight_seeds.jsonl python -m pytest tests/test_gpu_weight_seeds.py -q 2>&1 | tail -2 > gpurun_out/r5/weight_seeds_pytest.txt
tail -3 gpurun_out/r5_stdout.txt; tail -c 1500 gpurun_out/r5/bench.json; cat gpurun_out/r5/h2_ab.txt
