R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for bk in 16 32; do
  rm -rf /tmp/tr$bk
  AIMNET_GEMM_BK=$bk timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$bk -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/b$bk.log 2>&1
  echo "== BK $bk"
  python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/tr$bk/*/*kernel_trace.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "gemm" not in n: continue
    key=(n.split("(")[0].replace("void aimnet::",""), r.get("Grid_Size_X", r.get("Grid_Size","")))
    agg[key].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
tot=0
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    print("%-48s grid=%-8s n/step=%4.1f avg=%7.1f us  ms/step=%.4f"%(k[0][:48],k[1],len(v)/12,sum(v)/len(v),sum(v)/12e3)); tot+=sum(v)
print("total gemm ms/step %.4f"%(tot/12e3))
PY
done
