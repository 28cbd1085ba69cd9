"""Where does a single-molecule evaluation spend its wall time: engine (GPU) vs the Python host layers above it?"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aimnetcentral_amd import AIMNet2Calculator, loader
calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "taxol.npz"))
dev = calc.engine.device
def bench(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
c_np, z_np = g["coord"], g["numbers"]
c_t, z_t = torch.from_numpy(c_np).to(dev), torch.from_numpy(z_np).to(dev).int()
mol = torch.zeros(113, dtype=torch.int32, device=dev); q = torch.zeros(1, device=dev)
print("engine.eval, device tensors      %.3f ms" % bench(lambda: calc.engine.eval(c_t, z_t, mol, q, forces=True)))
print("calculator, device tensors       %.3f ms" % bench(lambda: calc({"coord": c_t, "numbers": z_t, "charge": q}, forces=True)))
print("calculator, numpy in             %.3f ms" % bench(lambda: calc({"coord": c_np, "numbers": z_np, "charge": 0.0}, forces=True)))
print("calculator, numpy in + .cpu() out %.3f ms" % bench(lambda: {k: v.cpu() for k, v in calc({"coord": c_np, "numbers": z_np, "charge": 0.0}, forces=True).items()}))

# the ASE adapter's MD step: positions change every call, everything else is cached on the device
from aimnetcentral_amd.aimnet2ase import AIMNet2ASE
class _Atoms:
    def __init__(self, numbers, positions):
        self.numbers, self.positions, self.cell, self.pbc, self.info = numbers, positions, None, np.zeros(3, bool), {}
    def copy(self):
        return _Atoms(self.numbers.copy(), self.positions.copy())
    def __len__(self):
        return len(self.numbers)
ase = AIMNet2ASE(calc)
atoms = _Atoms(z_np.astype(np.int64), c_np.astype(np.float64))
def md_step():
    atoms.positions += 1e-4
    ase.reset()
    ase.calculate(atoms, properties=["energy", "forces"])
print("AIMNet2ASE.calculate (MD step)   %.3f ms" % bench(md_step))
