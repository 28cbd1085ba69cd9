#!/bin/bash
# kernel times of the default bench (+ the DFT-D3 kernels of tests/tools/skin_probe.py) for probe builds of ONE source file compiled with
# -fno-slp-vectorize (gpurun_in/<stem>_noslp.so from tests/tools/obj_variants.sh): a v_pk_*_f32 costs 1.75 scalar instructions on this
# part (profiles/r6_pk_rate.txt), so the compiler's automatic packing - with the moves that build its operand pairs - can lose
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for so in "" $(ls $R/gpurun_in/*_noslp.so 2>/dev/null); do
  tag=$(basename "${so:-shipped}" .so)
  rm -rf /tmp/kt_$tag
  AIMNET_HIP_LIB=$so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-hessian --no-repeat > /tmp/kt_$tag.log 2>&1
  echo "== $tag"
  python $R/tests/tools/prof_summary.py $(ls /tmp/kt_$tag/*/*kernel_trace.csv | head -1) 27 | head -24 | cut -c1-118
done
