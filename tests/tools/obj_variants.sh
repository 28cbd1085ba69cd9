#!/bin/bash
# probe builds of ONE source file (one set of -D flags each) linked against the shipped objects -> gpurun_in/<stem>_<name>.so
# usage (build container): bash tests/tools/obj_variants.sh conv deep:-DAIMNET_PROBE_FWD_DEEP "occ3:-DAIMNET_PROBE_FWD_OCC=3"
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/aimnetcentral_amd/csrc
stem=$1; shift
mkdir -p $R/gpurun_in
# the files csrc/Makefile builds without the SLP vectorizer (NOSLP) are built that way here too, so that a probe differs from the
# shipped object only in its -D flags
base=""
case " model conv d3 gemm_chain hvp " in *" $stem "*) base="-fno-slp-vectorize";; esac
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $base $flags -c $C/$stem.hip -o /tmp/${stem}_$name.o || exit 1
  objs=""
  for o in $(ls $C/*.o); do [ $(basename $o) = $stem.o ] && objs="$objs /tmp/${stem}_$name.o" || objs="$objs $o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/gpurun_in/${stem}_$name.so && echo built gpurun_in/${stem}_$name.so
done
