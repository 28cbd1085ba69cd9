#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for stop in ${STOPS:-0 1 2}; do
mkdir -p /tmp/bf3s && cd $R/aimnetcentral_amd/csrc && cp *.hip *.h /tmp/bf3s/ && mkdir -p /tmp/include && cp $R/include/aimnet_hip.h /tmp/include/
cd /tmp/bf3s && sed -i 's#../../include/aimnet_hip.h#/tmp/include/aimnet_hip.h#' *.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DBF3_DBG_STOP=$stop -c gemm_bf3.hip -o gemm_bf3.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gemm_bf3.o $R/aimnetcentral_amd/csrc/{engine,gemm,nlist,conv,conv_mfma,model,d3}.o -o /tmp/bf3s/libaimnet_hip.so
cd $R
echo "== stop $stop"; AIMNET_HIP_LIB=/tmp/bf3s/libaimnet_hip.so timeout 60 python tests/tools/bf3_small.py 452 2>&1 | grep -v amdgpu.ids | head -4
done
