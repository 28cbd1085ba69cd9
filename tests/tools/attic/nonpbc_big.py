"""Large NON-periodic system (a 10 080-atom crystal fragment in vacuum): where does the time go?"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
dev = eng.device
for rep in ((7, 3, 5), (14, 6, 5)):
    c, z, cell = workloads.glucose_supercell(rep)
    n = len(z)
    args = (torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(1, device=dev))
    for coul in ("dsf",):
        eng.set_profiling(2)
        for _ in range(2): r = eng.eval(*args, forces=True, coulomb=coul)
        torch.cuda.synchronize(); eng.read_profile()
        t0 = time.time()
        for _ in range(5): r = eng.eval(*args, forces=True, coulomb=coul)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 5
        p = eng.read_profile()
        print(f"n={n} {coul}: {dt*1e3:.2f} ms/step", {k: round(v / 5, 3) for k, v in p.items()}, eng.last_status)
