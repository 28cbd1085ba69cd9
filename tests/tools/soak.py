import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
EVERY = int(os.environ.get('SOAK_EVERY', 500))  # compare every n-th evaluation with the first, bitwise
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
c, z, cell = workloads.glucose_supercell((7, 3, 5))
dev = eng.device
args = (torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev), torch.zeros(len(z), dtype=torch.int32, device=dev), torch.zeros(1, device=dev))
cl = torch.from_numpy(cell.astype(np.float32)).to(dev)
r0 = eng.eval(*args, cell=cl, forces=True, stress=True, coulomb="dsf")
e0, f0 = r0["energy"].clone(), r0["forces"].clone()
m0 = torch.cuda.memory_allocated()
t0 = time.time(); n = 0
while time.time() - t0 < 45:
    r = eng.eval(*args, cell=cl, forces=True, stress=True, coulomb="dsf")
    n += 1
    if n % EVERY == 0:
        assert torch.equal(r["energy"], e0) and torch.equal(r["forces"], f0), f"results drifted at evaluation {n}"
torch.cuda.synchronize()
print(f"{n} evaluations in {time.time()-t0:.1f} s = {(time.time()-t0)/n*1e3:.3f} ms each; bitwise identical (every {EVERY}-th compared); memory {m0} -> {torch.cuda.memory_allocated()} bytes (peak {torch.cuda.max_memory_allocated()})")
