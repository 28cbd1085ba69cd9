# kernel time of the list-free DSF walk for variant builds of model.hip (gpurun_in/model_<name>.so from tests/tools/obj_variants.sh): one
# kernel trace of the default bench per library: bash tests/tools/dsf_variants_prof.sh model_nopf nlist_nopf
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in intree "$@"; do
  if [ $v = intree ]; then unset AIMNET_HIP_LIB; else export AIMNET_HIP_LIB=$R/gpurun_in/$v.so; fi
  rm -rf /tmp/tr_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-hessian --no-repeat > /tmp/tr_$v.log 2>&1
  f=$(ls /tmp/tr_$v/*/*kernel_stats.csv | head -1)
  python - $v $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if "coulomb_dsf_walk" in r["Name"] or "nlist_cell" in r["Name"]:
        print(f"{sys.argv[1]:12s} {r['Name'][13:45]:34s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:7.1f} us  min {float(r['MinNs']) / 1e3:7.1f} us")
PY
done
