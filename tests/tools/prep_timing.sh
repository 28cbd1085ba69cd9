#!/bin/bash
# measurement build of the single-launch preparation (nlist.hip, prep_small_kernel) with wall-clock stamps per phase (GPU box)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/aimnetcentral_amd/csrc
mkdir -p /tmp/prept && cp *.hip *.h Makefile /tmp/prept/ && mkdir -p /tmp/include && cp $R/include/aimnet_hip.h /tmp/include/
cd /tmp/prept && sed -i 's#../../include/aimnet_hip.h#/tmp/include/aimnet_hip.h#' *.hip *.h Makefile
for f in engine nlist; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DAIMNET_PREP_TIMING -c $f.hip -o $f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC engine.o nlist.o $R/aimnetcentral_amd/csrc/{hvp,gemm,gemm_bf3,gemm_bf3a,gemm_head,conv,conv_mfma,model,d3}.o -o /tmp/prept/libaimnet_hip.so
cd $R
AIMNET_HIP_LIB=/tmp/prept/libaimnet_hip.so python - <<'PY'
import ctypes as C
import numpy as np, torch
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
c, z, cell = workloads.glucose_supercell((7, 3, 5))
dev = eng.device
args = (torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev), torch.zeros(len(z), dtype=torch.int64, device=dev), torch.zeros(1, device=dev))
names = ["zero + cell setup", "mol_start / sanity / species", "bin grid", "sys -> LDS", "wrap + count", "scan", "fill", "order + stream"]
for it in range(3):
    eng.eval(*args, cell=torch.from_numpy(cell.astype(np.float32)).to(dev), forces=True, stress=True, coulomb="dsf", dsf_rc=15.0)
    torch.cuda.synchronize()
    st = (C.c_ulonglong * 16)()
    assert eng.lib.aimnet_debug_prep_stamps(st) == 0
    d = [(st[k + 1] - st[k]) * 0.01 for k in range(8)]
    print(f"eval {it}: total {sum(d):6.2f} us  " + "  ".join(f"{n} {v:.2f}" for n, v in zip(names, d)))
PY
