#!/bin/bash
# measurement build on the GPU box: recompile ONE source with extra -D flags, link it with the shipped objects, run a command
# against the variant library.  usage: variant.sh <source stem> "<-D flags>" <command...>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
stem=$1; flags=$2; shift 2
D=/tmp/variant_$stem
mkdir -p $D /tmp/include && cp $R/aimnetcentral_amd/csrc/*.hip $R/aimnetcentral_amd/csrc/*.h $D/ && cp $R/include/aimnet_hip.h /tmp/include/
cd $D && sed -i 's#../../include/aimnet_hip.h#/tmp/include/aimnet_hip.h#' *.hip *.h
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c $stem.hip -o $stem.o
objs=""
for o in engine hvp gemm gemm_bf3 nlist conv conv_mfma model d3; do
  if [ $o = $stem ]; then objs="$objs $D/$o.o"; else objs="$objs $R/aimnetcentral_amd/csrc/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $D/libaimnet_hip.so
cd $R
AIMNET_HIP_LIB=$D/libaimnet_hip.so "$@"
