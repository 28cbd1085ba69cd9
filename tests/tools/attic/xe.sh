#!/bin/bash
# reverse-pair conv backward: parity tests, A/B of the step, per-kernel averages
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/xe
(timeout 900 python -m pytest tests/test_gpu_conv_xe.py -q -x 2>&1 | tail -15) > gpurun_out/xe/tests.txt
cat gpurun_out/xe/tests.txt
for o in 0 1 0 1; do
  AIMNET_CONV_XE=$o timeout 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('xe $o', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['family_ms_per_step'].items()})"
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trx
AIMNET_CONV_XE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trx -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/bx.log 2>&1
python $R/tests/tools/prof_summary.py $(ls /tmp/trx/*/*kernel_trace.csv | head -1) 18 | grep -E "total|pair_|conv_bwd|nlist_cell|row_sort"
