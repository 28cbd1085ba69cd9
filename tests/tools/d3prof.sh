R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/tests/tools/d3_cost.py
rm -rf /tmp/trd3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trd3 -- python $R/tests/tools/d3_cost.py > /tmp/d3.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/trd3/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:40]:
    n=r["Name"].replace("void ","").replace("aimnet::","").split("(")[0]
    if any(k in n for k in ("d3","nlist","coulomb","bin","scan")): print("%-50s calls=%4s avg=%8.1f us"%(n[:50], r["Calls"], float(r["AverageNs"])/1e3))
PY
