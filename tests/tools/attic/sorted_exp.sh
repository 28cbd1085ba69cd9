#!/bin/bash
cd $GRAFT_REPO_ROOT
for srt in "" 1; do for v in "" gpurun_in/b_nomath.so; do
  AIMNET_BENCH_SORT=$srt AIMNET_HIP_LIB=$v python bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['family_ms_per_step']
print('sorted=%-2s %-24s step %.4f  ' % ('$srt', '$v' or 'in-tree', d['ms_per_step']) + '  '.join('%s %.3f' % (k, v) for k, v in f.items() if v > 0))"
done; done
