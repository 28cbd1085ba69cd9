R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tax
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tax/trace -- python $R/bench.py --workload taxol --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/tax/bench_traced.log 2>&1
cd $R
python tests/tools/prof_summary.py $(ls gpurun_out/tax/trace/*/*kernel_trace.csv | head -1) 110 > gpurun_out/tax/kernel_summary.txt
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/tax/trace/*/*kernel_trace.csv")[0]
rows=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:50]) for r in csv.DictReader(open(f))))
# take the last 70 kernels: show gaps
tail=rows[-75:]
t0=tail[0][0]
for s,e,n in tail: print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:7.1f} {n}")
PY
rm -rf gpurun_out/tax/trace
cat gpurun_out/tax/kernel_summary.txt | head -50
tail -2 gpurun_out/tax/bench_traced.log
python bench.py --workload taxol --steps 200 --warmup 20 --no-cpu-baseline | tail -1
python tests/tools/host_overhead.py
