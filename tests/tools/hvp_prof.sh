cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hp && rocprofv3 --kernel-trace --stats -d /tmp/hp -o hp --output-format csv -- python $GRAFT_REPO_ROOT/tests/tools/hvp_prof.py ${1:-1} > /tmp/hp.log 2>&1
f=$(find /tmp/hp -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/hvp_prof.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms per call: %.3f" % (tot / 4e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
    print("%8.3f ms/call  %5d calls  %s" % (float(r["TotalDurationNs"]) / 4e6, int(r["Calls"]), r["Name"][:110]))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/hvp_prof.txt; tail -3 /tmp/hp.log
