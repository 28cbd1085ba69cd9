#!/bin/bash
# probe builds of gemm_chain.hip (one -D flag each) linked against the shipped objects -> gpurun_in/chain_<name>.so
# usage (build container): bash tests/tools/chain_variants.sh name1:"-DFLAG ..." name2:"..."
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/aimnetcentral_amd/csrc
mkdir -p $R/gpurun_in
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c $C/gemm_chain.hip -o /tmp/chain_$name.o || exit 1
  objs=""
  for o in $(ls $C/*.o); do [ $(basename $o) = gemm_chain.o ] && objs="$objs /tmp/chain_$name.o" || objs="$objs $o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/gpurun_in/chain_$name.so && echo built gpurun_in/chain_$name.so
done
