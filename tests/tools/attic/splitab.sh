#!/bin/bash
# A/B: the 4-waves-per-atom (SPLIT) conv kernels on the 10 080-atom benchmark (work-unit quantisation experiment)
cd $GRAFT_REPO_ROOT
for sm in -1 20000 -1 20000; do
  AIMNET_SPLIT_MAX=$sm timeout 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('split_max $sm', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['family_ms_per_step'].items()})"
done
