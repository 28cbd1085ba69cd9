#!/bin/bash
# kernel trace of 20 Ewald evaluations of the 10 080-atom crystal (forces + stress): bash tests/tools/ewald_prof.sh  (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_ewald
cat > /tmp/ewald_run.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, torch
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
c, z, cell = workloads.glucose_supercell((7, 3, 5))
dev = eng.device
a = (torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev), torch.zeros(len(z), dtype=torch.int64, device=dev), torch.zeros(1, device=dev))
cl = torch.from_numpy(cell.astype(np.float32)).to(dev)
for _ in range(22):
    eng.eval(*a, cell=cl, forces=True, stress=True, coulomb="ewald")
torch.cuda.synchronize()
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_ewald -- python /tmp/ewald_run.py > /tmp/kt_ewald.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/kt_ewald/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"]
    if "ewald" in n or "coulomb" in n:
        print(f"{n[:70]:70s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs'])/1e3:8.1f} us")
print(f"all kernels: {tot/22/1e6:.3f} ms per evaluation")
PY
