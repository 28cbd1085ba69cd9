#!/bin/bash
# conv_fwd / conv_bwd kernels of the shipped library against probe builds (gpurun_in/conv_*.so from tests/tools/obj_variants.sh):
# time per dispatch from a kernel trace of the default bench + the checksum of one config-3 evaluation per library (bit-neutral
# changes give equal checksums) -> gpurun_out/conv_probe.txt
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/conv_probe.txt; : > $out
for so in "" $(ls $R/gpurun_in/conv_*.so 2>/dev/null); do
  tag=$(basename "${so:-shipped}" .so)
  rm -rf /tmp/kt_$tag
  AIMNET_HIP_LIB=$so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-hessian --no-repeat > /tmp/kt_$tag.log 2>&1
  echo "== $tag" >> $out
  python $R/tests/tools/prof_summary.py $(ls /tmp/kt_$tag/*/*kernel_trace.csv | head -1) 27 | grep "conv_fwd_kernel\|conv_bwd_kernel\|total" >> $out
  AIMNET_HIP_LIB=$so python $R/tests/tools/checksum.py 2>/dev/null | tail -1 >> $out
done
cat $out
