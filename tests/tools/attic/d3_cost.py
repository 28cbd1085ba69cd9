"""Cost of the external DFT-D3 term on the config-3 workload (10 080 atoms, 15 A)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
t = np.load(os.path.join(ROOT, "tests", "golden", "dftd3_subset.npz"))
eng.set_dftd3_tables({k: t[k] for k in ("c6ab", "cn_ref", "rcov", "r4r2")})
c, z, cell = workloads.glucose_supercell((7, 3, 5))
dev = eng.device
args = (torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev), torch.zeros(len(z), dtype=torch.int32, device=dev), torch.zeros(1, device=dev))
cl = torch.from_numpy(cell.astype(np.float32)).to(dev)
for d3 in (None, dict(s8=0.3908, a1=0.566, a2=3.128)):
    eng.set_profiling(2)
    for _ in range(3): eng.eval(*args, cell=cl, forces=True, stress=True, coulomb="dsf", dftd3=d3)
    torch.cuda.synchronize(); eng.read_profile()
    t0 = time.time()
    for _ in range(10): eng.eval(*args, cell=cl, forces=True, stress=True, coulomb="dsf", dftd3=d3)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    p = eng.read_profile()
    print("dftd3" if d3 else "no d3", f"{dt*1e3:.3f} ms/step", {k: round(v / 10, 3) for k, v in p.items()}, "status", eng.last_status)
