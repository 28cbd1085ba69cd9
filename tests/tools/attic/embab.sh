#!/bin/bash
# A/B of the embedding bias table (pass-0 first-layer GEMM over 448 instead of 704 columns) + parity suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/emb
(timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -4) > gpurun_out/emb/tests.txt
tail -n 3 gpurun_out/emb/tests.txt
for o in 1 0 1 0; do
  AIMNET_EMB_BIAS=$o timeout 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('emb_bias $o', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['family_ms_per_step'].items()})"
done
