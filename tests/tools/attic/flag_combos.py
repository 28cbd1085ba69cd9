import sys, numpy as np, torch
sys.path.insert(0, ".")
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0), "cuda:0"); dev = eng.device
c, z, cell = workloads.glucose_supercell((7, 3, 5))
a = (torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev), torch.zeros(len(z), dtype=torch.int32, device=dev), torch.zeros(1, device=dev))
cl = torch.from_numpy(cell.astype(np.float32)).to(dev)
r1 = eng.eval(*a, cell=cl, forces=True, stress=True, coulomb="dsf")
r0 = eng.eval(*a, cell=cl, forces=False, stress=False, coulomb="dsf")
r2 = eng.eval(*a, cell=cl, forces=False, stress=True, coulomb="dsf")
r3 = eng.eval(*a, cell=cl, forces=True, stress=False, coulomb="dsf")
print("dE energy-only", float(r1["energy"][0] - r0["energy"][0]), "stress-only dS", float((r2["stress"] - r1["stress"]).abs().max()), "forces-only dF", float((r3["forces"] - r1["forces"]).abs().max()), "dq", float((r0["charges"] - r1["charges"]).abs().max()))
