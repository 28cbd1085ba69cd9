#!/bin/bash
# conv_bwd_kernel (passes 1-2, reverse-pair form) under variants: processing order (bin order vs input order), 3 instead of 4
# waves per SIMD (more registers per wave), 32- instead of 48-pair LDS chunks; kernel time from a trace, fabric reads / L2 hit rate
# from one PMC pass each.  -> gpurun_out/conv_bwd_probe.txt   (variant libraries: tests/tools/build_variant.sh -> gpurun_in/)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/conv_bwd_probe.txt; : > $out
run() {  # tag, env assignments..., library
  tag=$1; shift
  rm -rf /tmp/cbp_$tag /tmp/cbq_$tag
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cbp_$tag -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-hessian > /tmp/cbp_$tag.log 2>&1
  env "$@" timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/cbq_$tag -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-hessian > /tmp/cbq_$tag.log 2>&1
  python - "$tag" <<PY >> $out
import csv, glob, sys, collections
tag = sys.argv[1]
t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(glob.glob(f"/tmp/cbp_{tag}/*/*kernel_trace.csv")[0]))
     if "conv_bwd_kernel<1" in r["Kernel_Name"]]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(glob.glob(f"/tmp/cbq_{tag}/*/*counter_collection.csv")[0])):
    if "conv_bwd_kernel<1" in r["Kernel_Name"]:
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
v = {k: a[0] / max(1, a[1]) for k, a in agg.items()}
hit = v.get("TCC_HIT_sum", 0) / max(1.0, v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0))
print(f"{tag:14s} conv_bwd_kernel<1,...>: avg {sum(t)/len(t):6.1f} us  min {min(t):6.1f} us  fabric reads {v.get('TCC_EA0_RDREQ_sum', 0) * 128 / 1e6:6.0f} MB  L2 hit {hit:.3f}  "
      f"VALU insts {v.get('SQ_INSTS_VALU', 0):.3e}  SQ busy {v.get('SQ_BUSY_CYCLES', 0):.3e}")
PY
}
run default AIMNET_X=0
run input_order AIMNET_SPATIAL_ORDER=0
run occ3 AIMNET_HIP_LIB=$R/gpurun_in/bwd_occ3.so
run chb32 AIMNET_HIP_LIB=$R/gpurun_in/bwd_chb32.so
cat $out
