#!/bin/bash
# Domain decomposition record (GPU box): every case of tests/dd_worker.py, 2 and 3 ranks sharing cuda:0 over gloo -> gpurun_out/dd/*.json
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/dd
port=29560
for spec in "golden2304 2" "golden2304 3" "cube1536 2" "cube1536 3" "cube1536_nocoul 2" "cube1536_nse 2" "cube1536_d3 2" "cube1536_d3rc12 2"; do
  set -- $spec
  port=$((port + 1))
  DD_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$2" --master-addr 127.0.0.1 --master-port $port \
    tests/dd_worker.py "$1" "gpurun_out/dd/$1_w$2.json" > "gpurun_out/dd/$1_w$2.log" 2>&1 || echo "FAILED $spec"
done
DD_BACKEND=gloo DD_SHARE_GPU=1 STEPS=3 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29599 \
  tests/tools/dd_bench.py 2> gpurun_out/dd/dd_bench.err | grep '^{' > gpurun_out/dd/dd_bench_2ranks_shared_gpu.jsonl || echo 'FAILED dd_bench'
cat gpurun_out/dd/*.json
