"""Forces on the first unit cell of glucose supercells of growing size, per Coulomb method: replicated images must carry the same
forces.  GPU box."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
dev = eng.device
for method, kw in (("dsf", {}), ("ewald", dict(ewald_accuracy=1e-6)), ("pme", dict(ewald_accuracy=1e-6)), ("pme", dict(ewald_accuracy=1e-8))):
    ref = None
    for rep in ((4, 2, 3), (7, 3, 5), (14, 6, 5)):
        c, z, cell = workloads.glucose_supercell(rep)
        n = len(z)
        r = eng.eval(torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev), torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(1, device=dev),
                     cell=torch.from_numpy(cell.astype(np.float32)).to(dev), forces=True, coulomb=method, **kw)
        f = r["forces"].cpu().numpy()[:96]; q = r["charges"].cpu().numpy()[:96]
        if ref is None: ref = (f, q)
        print(method, kw, rep, n, "dF_max vs smallest", float(np.abs(f - ref[0]).max()), "dq_max", float(np.abs(q - ref[1]).max()), "status7", int(eng.last_status[7]))
