#!/bin/bash
# A/B of the pass-0 element-moment paths (AIMNET_P0_MOMENTS) + parity suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p0
(timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -4) > gpurun_out/p0/tests.txt
tail -n 3 gpurun_out/p0/tests.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trp -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/bp.log 2>&1
python $R/tests/tools/prof_summary.py $(ls /tmp/trp/*/*kernel_trace.csv | head -1) 18 | grep -E "total|conv_fwd"
cd $R
for w in pbc10k batch256 md1024; do
  timeout 300 python bench.py --no-cpu-baseline --steps 40 --workload $w 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['family_ms_per_step']; print('$w', round(d['ms_per_step'],4), 'conv_fwd', round(f['conv_fwd'],3), 'gemm', round(f['gemm'],3))"
done
