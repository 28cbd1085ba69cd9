#!/bin/bash
# Build a variant of the library from a patched copy of csrc/ into gpurun_in/<name>.so (for tests/tools/ab.sh):
#   bash tests/tools/build_variant.sh NAME 'sed -i s/old/new/ conv.hip'
set -e
NAME=$1; PATCH=$2
R=$(cd "$(dirname "$0")/../.." && pwd)
D=$(mktemp -d)
mkdir -p $D/aimnetcentral_amd $D/include && cp -r $R/aimnetcentral_amd/csrc $D/aimnetcentral_amd/ && cp $R/include/aimnet_hip.h $D/include/
cd $D/aimnetcentral_amd/csrc && rm -f *.o *.so && eval "$PATCH" && make 2>&1 | grep -E "error|Error" || true
mkdir -p $R/gpurun_in && cp libaimnet_hip.so $R/gpurun_in/$NAME.so && echo built $R/gpurun_in/$NAME.so
