#!/bin/bash
# kernel trace of the default bench with pre-split activations on / off: per-kernel averages in the engine's own sequence
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ktrace_ab
cd /tmp && export TMPDIR=/tmp
for ps in 1 0; do
  AIMNET_GEMM_PRESPLIT=$ps timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$ps -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-hessian > /tmp/kt$ps.log 2>&1
  python $R/tests/tools/prof_summary.py $(ls /tmp/kt$ps/*/*kernel_trace.csv | head -1) 17 > $R/gpurun_out/ktrace_ab/summary_ps$ps.txt
done
head -45 $R/gpurun_out/ktrace_ab/summary_ps1.txt
