"""Scale check: energy per cell and forces of a large supercell equal those of the unit cell images (periodicity)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
dev = eng.device
ref = None
METHOD = os.environ.get("COULOMB", "dsf")
for rep in ((7, 3, 5), (14, 6, 10), (21, 9, 15)):
    c, z, cell = workloads.glucose_supercell(rep)
    n = len(z)
    args = (torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(1, device=dev))
    cl = torch.from_numpy(cell.astype(np.float32)).to(dev)
    r = eng.eval(*args, cell=cl, forces=True, stress=True, coulomb=METHOD)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): r = eng.eval(*args, cell=cl, forces=True, stress=True, coulomb=METHOD)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    e_cell = float(r["energy"][0]) / (n / 96)
    f = r["forces"].cpu().numpy()
    if ref is None: ref = (e_cell, np.abs(f).max(), r["stress"].cpu().numpy())
    print(f"{rep} n={n} E/cell={e_cell:.6f} (d {e_cell-ref[0]:+.2e}) |F|max={np.abs(f).max():.4f} dstress={np.abs(r['stress'].cpu().numpy()-ref[2]).max():.2e} "
          f"{dt*1e3:.2f} ms/step {n/dt/1e6:.2f} M atoms*steps/s ws={eng._ws.numel()/2**30:.2f} GiB finite={np.isfinite(f).all()}")
