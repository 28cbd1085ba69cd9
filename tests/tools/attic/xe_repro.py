"""repro of the one_big_many_small batch with the reverse-pair conv backward on (debug)"""
import numpy as np, torch, sys
sys.path.insert(0, ".")
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine
rng = np.random.Generator(np.random.PCG64(98))
sizes = [700] + rng.integers(2, 9, size=200).tolist()
coords, zs, mols = [], [], []
for m, n in enumerate(sizes):
    cm, zm = workloads.random_organic(int(n), rng)
    coords.append(cm); zs.append(zm); mols.append(np.full(int(n), m))
c = np.concatenate(coords).astype(np.float32); z = np.concatenate(zs).astype(np.int64); mol = np.concatenate(mols).astype(np.int64)
q = rng.integers(-1, 2, size=len(sizes)).astype(np.float32)
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
eng.set_option("conv_xe", int(sys.argv[1]))
dev = eng.device
print("atoms", len(z), flush=True)
r = eng.eval(torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.from_numpy(mol).to(dev), torch.from_numpy(q).to(dev), forces=True, coulomb="simple")
print("status", eng.last_status[:8], "max_nb", getattr(eng, "max_nb", None), float(r["energy"].sum()), float(r["forces"].abs().max()), flush=True)
