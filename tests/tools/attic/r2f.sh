#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
(timeout 1200 python -m pytest tests -q -x -m gpu 2>&1 | tail -4) > gpurun_out/r2f/tests.txt
for o in 0 1 0 1; do
  AIMNET_OVERLAP_COULOMB=$o python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap $o', round(d['ms_per_step'],4), round(d['roofline']['gemm_ms_per_step'],4), round(d['roofline_e2e']['frac'],4))"
done
for w in batch256 md1024 taxol; do for o in 0 1; do
  AIMNET_OVERLAP_COULOMB=$o python bench.py --no-cpu-baseline --steps 40 --workload $w 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w overlap $o', round(d['ms_per_step'],4))"
done; done
tail -n 3 gpurun_out/r2f/tests.txt
