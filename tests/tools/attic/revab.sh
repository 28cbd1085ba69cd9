#!/bin/bash
# reverse-pair map: hash tables vs sorted rows + bisection (parity suite with the hash form, A/B, per-kernel times)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rev
(timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -4) > gpurun_out/rev/tests.txt
tail -n 3 gpurun_out/rev/tests.txt
for o in 1 0 1 0; do
  AIMNET_REV_HASH=$o timeout 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['family_ms_per_step']; print('rev_hash $o', round(d['ms_per_step'],4), 'geom', round(f['geom'],3), 'conv_fwd', round(f['conv_fwd'],3), 'conv_bwd', round(f['conv_bwd'],3))"
done
bash tests/tools/ktrace.sh 40 | grep -E "total|pair_|row_sort"
