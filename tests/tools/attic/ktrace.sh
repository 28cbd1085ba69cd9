#!/bin/bash
# kernel trace of the default bench: per-kernel averages (top 30)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trk
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trk -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/bk.log 2>&1
python $R/tests/tools/prof_summary.py $(ls /tmp/trk/*/*kernel_trace.csv | head -1) 17 | head -${1:-30}
