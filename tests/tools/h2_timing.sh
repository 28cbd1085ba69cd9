#!/bin/bash
# measurement build of the h2 GEMM with s_memtime stamps at the segment boundaries of waves 0 / 4 of block 0 (GPU box)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/aimnetcentral_amd/csrc
rm -rf /tmp/h2t && mkdir -p /tmp/h2t && cp *.hip *.h Makefile /tmp/h2t/ && mkdir -p /tmp/include && cp $R/include/aimnet_hip.h /tmp/include/
cd /tmp/h2t && sed -i 's#../../include/aimnet_hip.h#/tmp/include/aimnet_hip.h#' *.hip *.h Makefile
for f in engine gemm_bf3 gemm_bf3a gemm_h2; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DAIMNET_BF3_TIMING $EXTRA -c $f.hip -o $f.o & done; wait
OTHERS=$(ls $R/aimnetcentral_amd/csrc/*.o | grep -v -E "/(engine|gemm_bf3|gemm_bf3a|gemm_h2)\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC engine.o gemm_bf3.o gemm_bf3a.o gemm_h2.o $OTHERS -o /tmp/h2t/libaimnet_hip.so
cd $R
STAMPS=1 SHAPES=one AIMNET_HIP_LIB=/tmp/h2t/libaimnet_hip.so python tests/tools/h2_bench.py
