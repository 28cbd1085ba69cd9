#!/bin/bash
# measurement builds of the h2 GEMM with the loop's tile requests dropped (wrong results): what the DMA stream costs the main loop
# usage: bash tests/tools/h2_probe_dma.sh   (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
for probe in 0 5 6 7; do
  rm -rf /tmp/h2p && mkdir -p /tmp/h2p && cp $R/aimnetcentral_amd/csrc/*.hip $R/aimnetcentral_amd/csrc/*.h /tmp/h2p/ && mkdir -p /tmp/include && cp $R/include/aimnet_hip.h /tmp/include/
  cd /tmp/h2p && sed -i 's#../../include/aimnet_hip.h#/tmp/include/aimnet_hip.h#' *.hip *.h
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DAIMNET_H2_PROBE=$probe -c gemm_h2.hip -o gemm_h2.o
  OTHERS=$(ls $R/aimnetcentral_amd/csrc/*.o | grep -v -E "/gemm_h2\.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gemm_h2.o $OTHERS -o /tmp/h2p/libaimnet_hip.so
  cd $R
  echo "== probe $probe (bit 2 set: drop; bit 0 activation tiles, bit 1 weight tiles)"
  EPI=0 AIMNET_HIP_LIB=/tmp/h2p/libaimnet_hip.so python tests/tools/h2_bench.py 2>&1 | grep -E "N= 512 K= 736|N= 384 K= 512|sum over" | cut -c1-40,95-260
done
