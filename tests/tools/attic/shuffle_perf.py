"""How much does the atom ORDER matter?  Config 3 in file order vs a random permutation of the atoms."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
dev = eng.device
c, z, cell = workloads.glucose_supercell((7, 3, 5))
rng = np.random.default_rng(0)
for tag, perm in (("file order", np.arange(len(z))), ("shuffled", rng.permutation(len(z)))):
    cc, zz = c[perm].astype(np.float32), z[perm]
    args = (torch.from_numpy(cc).to(dev), torch.from_numpy(zz).to(dev), torch.zeros(len(z), dtype=torch.int32, device=dev), torch.zeros(1, device=dev))
    cl = torch.from_numpy(cell.astype(np.float32)).to(dev)
    eng.set_profiling(2)
    for _ in range(3): r = eng.eval(*args, cell=cl, forces=True, stress=True, coulomb="dsf")
    torch.cuda.synchronize(); eng.read_profile()
    t0 = time.time()
    for _ in range(10): r = eng.eval(*args, cell=cl, forces=True, stress=True, coulomb="dsf")
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    p = eng.read_profile()
    print(tag, f"{dt*1e3:.3f} ms/step E={float(r['energy'][0]):.4f}", {k: round(v / 10, 3) for k, v in p.items()})
