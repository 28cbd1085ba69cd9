R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS"; do
  rm -rf /tmp/pm
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm -- python $R/tests/tools/d3_cost.py > /tmp/pm.log 2>&1
  python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for f in glob.glob("/tmp/pm/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("aimnet::","").replace("void ","").split("(")[0][:40]
        if "d3_" not in k and "dsf_walk" not in k: continue
        a=agg[k][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for k in agg:
    print(k, {c: "%.4g"%(v[0]/v[1]) for c,v in agg[k].items()})
PY
done
