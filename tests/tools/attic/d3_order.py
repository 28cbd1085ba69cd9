"""How much do the D3 passes depend on the memory order of the atoms?  Same 10 080-atom crystal in file order, shuffled,
and sorted by 5 A bins (z fastest, as the engine's cell list orders them)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
t = np.load(os.path.join(ROOT, "tests", "golden", "dftd3_subset.npz"))
eng.set_dftd3_tables({k: t[k] for k in ("c6ab", "cn_ref", "rcov", "r4r2")})
c, z, cell = workloads.glucose_supercell((7, 3, 5))
frac = (c @ np.linalg.inv(cell)) % 1.0
nb = np.maximum(1, np.floor(np.array([np.linalg.norm(cell[k]) for k in range(3)]) / 5.0)).astype(int)
b = np.minimum((frac * nb).astype(int), nb - 1)
orders = {"file": np.arange(len(z)), "shuffled": np.random.default_rng(0).permutation(len(z)),
          "binned": np.lexsort((b[:, 2], b[:, 1], b[:, 0]))}
dev = eng.device
cl = torch.from_numpy(cell.astype(np.float32)).to(dev)
d3 = dict(s8=0.3908, a1=0.566, a2=3.128)
for name, o in orders.items():
    args = (torch.from_numpy(c[o].astype(np.float32)).to(dev), torch.from_numpy(z[o]).to(dev), torch.zeros(len(z), dtype=torch.int32, device=dev), torch.zeros(1, device=dev))
    for use in (None, d3):
        eng.set_profiling(2)
        for _ in range(3): eng.eval(*args, cell=cl, forces=True, stress=True, coulomb="dsf", dftd3=use)
        torch.cuda.synchronize(); eng.read_profile()
        t0 = time.time()
        for _ in range(10): eng.eval(*args, cell=cl, forces=True, stress=True, coulomb="dsf", dftd3=use)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 10
        p = eng.read_profile()
        print(f"{name:9s} {'d3' if use else '--'} {dt*1e3:.3f} ms/step  nlist {p['nlist']/10:.3f} coulomb+d3 {p['coulomb']/10:.3f}")
