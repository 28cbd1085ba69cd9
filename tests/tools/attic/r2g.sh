#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
(timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -5) > gpurun_out/r2g/tests.txt
for k in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['family_ms_per_step'].items()})"
done
for w in batch256 md1024 taxol; do
  timeout 300 python bench.py --no-cpu-baseline --steps 40 --workload $w 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],4))"
done
tail -n 4 gpurun_out/r2g/tests.txt
