#!/bin/bash
# round-4 profile at HEAD (run on the GPU box): kernel trace + PMC passes of the default bench, the un-profiled default bench (CPU
# baseline incl. one full-size evaluation, parity gates, exact-fp32 record, config-4 Hessian), the other workloads, MD throughput
# through the adapters, the GEMM shape tables (in-kernel split vs pre-split activations), the accumulation-bias table, the
# weight-seed table, the LDS conflict counters, the stand-alone ops and the HVP records
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
bash tests/tools/pmc_bench.sh r4 > gpurun_out/r4_stdout.txt 2>&1
python bench.py > gpurun_out/r4/bench.json 2> gpurun_out/r4/bench.err
for w in batch256 md1024 taxol; do python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r4/bench_other_workloads.jsonl; done
for ps in 1 0; do AIMNET_GEMM_PRESPLIT=$ps python bench.py --no-cpu-baseline --no-exact-f32 --no-hessian --steps 40 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'gemm_presplit': $ps, 'ms_per_step': d['ms_per_step'], 'gemm_ms_per_step': d['roofline']['gemm_ms_per_step'], 'family_ms_per_step': d['family_ms_per_step']}))" >> gpurun_out/r4/presplit_ab.jsonl; done
AIMNET_GEMM_PRESPLIT=1 AIMNET_HEAD_FUSED=0 python bench.py --no-cpu-baseline --no-exact-f32 --no-hessian --steps 40 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'gemm_presplit': 1, 'head_fused': 0, 'ms_per_step': d['ms_per_step'], 'gemm_ms_per_step': d['roofline']['gemm_ms_per_step']}))" >> gpurun_out/r4/presplit_ab.jsonl
python tests/tools/md_throughput.py 2>/dev/null | tail -1 > gpurun_out/r4/md_throughput.json
python tests/tools/bf3a_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4/gemm_bf3a_shapes_gelu.txt
EPI=3 python tests/tools/bf3a_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4/gemm_bf3a_shapes_mul.txt
EPI=0 python tests/tools/bf3a_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4/gemm_bf3a_shapes_none.txt
python tests/tools/bf3_bias.py 2>&1 | grep -v amdgpu > gpurun_out/r4/bf3_bias.txt
bash tests/tools/lds_pmc.sh > /dev/null 2>&1; cp gpurun_out/lds_pmc/summary.txt gpurun_out/r4/lds_conflicts.txt
python tests/tools/op_bench.py 2>/dev/null | tail -1 > gpurun_out/r4/op_bench.json
python tests/tools/hvp_bench.py 2>/dev/null > gpurun_out/r4/hvp_bench.json
rm -f gpurun_out/r4/weight_seeds.jsonl
AIMNET_SEED_TABLE=$GRAFT_REPO_ROOT/gpurun_out/r4/weight_seeds.jsonl python -m pytest tests/test_gpu_weight_seeds.py -q 2>&1 | tail -2 > gpurun_out/r4/weight_seeds_pytest.txt
tail -3 gpurun_out/r4_stdout.txt; tail -c 1200 gpurun_out/r4/bench.json; cat gpurun_out/r4/presplit_ab.jsonl
