#!/bin/bash
# family times of the default bench for each variant library in gpurun_in/*.so (ablation / A-B runs)
cd $GRAFT_REPO_ROOT
for v in "" $(ls gpurun_in/*.so 2>/dev/null); do
  AIMNET_HIP_LIB=$v python bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['family_ms_per_step']
print('%-28s step %.4f  ' % ('$v' or 'in-tree', d['ms_per_step']) + '  '.join('%s %.3f' % (k, v) for k, v in f.items() if v > 0))"
done
