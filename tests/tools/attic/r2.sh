#!/bin/bash
# round-2 profile at HEAD: kernel trace + PMC passes of the default bench, the un-profiled default bench (with the CPU baseline),
# the other workloads, MD throughput through the adapters
cd $GRAFT_REPO_ROOT
bash tests/tools/pmc_bench.sh r2 > gpurun_out/r2_stdout.txt 2>&1
python bench.py > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err
for w in batch256 md1024 taxol; do python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r2/bench_other_workloads.jsonl; done
python tests/tools/md_throughput.py 2>/dev/null | tail -1 > gpurun_out/r2/md_throughput.json
tail -3 gpurun_out/r2_stdout.txt; tail -c 1500 gpurun_out/r2/bench.json
