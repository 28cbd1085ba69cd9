#!/bin/bash
# reverse-pair conv backward as the engine default candidate: the whole GPU suite with it on, A/B on every bench workload
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/xe
(AIMNET_CONV_XE=1 timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -5) > gpurun_out/xe/tests_all.txt
tail -n 3 gpurun_out/xe/tests_all.txt
for w in pbc10k batch256 md1024; do for o in 0 1 0 1; do
  AIMNET_CONV_XE=$o timeout 300 python bench.py --no-cpu-baseline --steps 40 --workload $w 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['family_ms_per_step']; print('$w xe $o', round(d['ms_per_step'],4), 'geom', round(f['geom'],3), 'conv_bwd', round(f['conv_bwd'],3), 'nlist', round(f['nlist'],3))"
done; done
