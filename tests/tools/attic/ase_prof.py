import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aimnetcentral_amd import AIMNet2Calculator, loader, workloads
calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
c, z, cell = workloads.glucose_supercell((7, 3, 5))
calc.set_lrcoulomb_method("dsf", cutoff=15.0, dsf_alpha=0.2)
zt = torch.as_tensor(z.astype(np.int32), device="cuda:0"); qt = torch.zeros(1, device="cuda:0")
def T(f, n=20):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
data = {"coord": c.astype(np.float32), "numbers": zt, "charge": qt, "cell": cell.astype(np.float32), "pbc": np.ones(3, bool)}
print("calc.eval host_out  %.2f ms" % T(lambda: calc.eval(data, forces=True, stress=True, host_out=True)), "status", calc.engine.last_status, "max_nb", calc.engine.max_nb)
print("calc.eval           %.2f ms" % T(lambda: calc.eval(data, forces=True, stress=True)))
ct = torch.as_tensor(c.astype(np.float32), device="cuda:0"); cellt = torch.as_tensor(cell.astype(np.float32), device="cuda:0"); mol = torch.zeros(len(z), dtype=torch.int32, device="cuda:0")
print("engine.eval         %.2f ms" % T(lambda: calc.engine.eval(ct, zt, mol, qt, cell=cellt, forces=True, stress=True, coulomb="dsf")))
print("engine.eval host_out %.2f ms" % T(lambda: calc.engine.eval(ct, zt, mol, qt, cell=cellt, forces=True, stress=True, coulomb="dsf", host_out=True)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): calc.eval(data, forces=True, stress=True, host_out=True)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
