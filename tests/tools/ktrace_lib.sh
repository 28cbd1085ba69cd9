#!/bin/bash
# kernel trace of the default bench for one build of the library: bash tests/tools/ktrace_lib.sh <tag> [path/to/lib.so]
R=$GRAFT_REPO_ROOT; tag=$1; lib=$2
mkdir -p $R/gpurun_out/ktrace
cd /tmp && export TMPDIR=/tmp
[ -n "$lib" ] && export AIMNET_HIP_LIB=$R/$lib
rm -rf /tmp/kt_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-hessian > /tmp/kt_$tag.log 2>&1
python $R/tests/tools/prof_summary.py $(ls /tmp/kt_$tag/*/*kernel_trace.csv | head -1) 17 > $R/gpurun_out/ktrace/$tag.txt
