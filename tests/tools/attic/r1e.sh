cd $GRAFT_REPO_ROOT
bash tests/tools/pmc_bench.sh r1e > gpurun_out/r1e_stdout.txt 2>&1
python bench.py > gpurun_out/r1e/bench.json 2> gpurun_out/r1e/bench.err
for w in batch256 md1024 taxol; do python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r1e/bench_other_workloads.jsonl; done
tail -3 gpurun_out/r1e_stdout.txt; tail -c 600 gpurun_out/r1e/bench.json
