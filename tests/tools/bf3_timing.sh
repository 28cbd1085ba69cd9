#!/bin/bash
# measurement build of the bf3 GEMM with s_memtime stamps at the segment boundaries of waves 0 / 4 of block 0 (GPU box)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/aimnetcentral_amd/csrc
mkdir -p /tmp/bf3t && cp *.hip *.h Makefile /tmp/bf3t/ && mkdir -p /tmp/include && cp $R/include/aimnet_hip.h /tmp/include/ 
cd /tmp/bf3t && sed -i 's#../../include/aimnet_hip.h#/tmp/include/aimnet_hip.h#' *.hip *.h Makefile
for f in engine gemm_bf3 gemm_bf3a; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DAIMNET_BF3_TIMING $EXTRA -c $f.hip -o $f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC engine.o gemm_bf3.o gemm_bf3a.o $R/aimnetcentral_amd/csrc/{hvp,gemm,nlist,conv,conv_mfma,model,d3}.o -o /tmp/bf3t/libaimnet_hip.so
cd $R
if [ -n "$BF3A" ]; then STAMPS=1 SHAPES=one AIMNET_HIP_LIB=/tmp/bf3t/libaimnet_hip.so python tests/tools/bf3a_bench.py; else AIMNET_HIP_LIB=/tmp/bf3t/libaimnet_hip.so python tests/tools/bf3_timing.py "$@"; fi
