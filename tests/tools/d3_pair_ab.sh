#!/bin/bash
# d3_pair_kernel with the DSF ride: the erfc polynomial (in-tree) against the library erfcf (gpurun_in/d3_erfcf.so from
# `bash tests/tools/obj_variants.sh d3 "erfcf:-DAIMNET_PROBE_D3_ERFCF"`) or any other variant named on the command line: kernel medians of
# tests/tools/d3prof.sh per library
R=$GRAFT_REPO_ROOT
for v in intree "$@"; do
  if [ $v = intree ]; then unset AIMNET_HIP_LIB; else export AIMNET_HIP_LIB=$R/gpurun_in/$v.so; fi
  echo "== $v"; bash $R/tests/tools/d3prof.sh 2>&1 | grep -E "d3_|nlist_cell|ms_per_step" | cut -c1-400
done
