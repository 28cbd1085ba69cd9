cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in md1024 batch256 taxol; do
  rm -rf /tmp/tr_$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$w -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-hessian --no-repeat > /tmp/tr_$w.log 2>&1
  echo "== $w"; tail -1 /tmp/tr_$w.log | cut -c1-200
  python $R/tests/tools/prof_summary.py $(ls /tmp/tr_$w/*/*kernel_trace.csv | head -1) 27 | head -32
done
