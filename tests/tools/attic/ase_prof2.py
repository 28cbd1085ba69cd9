import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aimnetcentral_amd import AIMNet2Calculator, loader, workloads
calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
c, z, cell = workloads.glucose_supercell((7, 3, 5))
zt = torch.as_tensor(z.astype(np.int32), device="cuda:0"); qt = torch.zeros(1, device="cuda:0")
ct = torch.as_tensor(c.astype(np.float32), device="cuda:0"); cellt = torch.as_tensor(cell.astype(np.float32), device="cuda:0"); mol = torch.zeros(len(z), dtype=torch.int32, device="cuda:0")
eng = calc.engine
import cProfile, pstats
for ho in (False, True):
    for _ in range(3): eng.eval(ct, zt, mol, qt, cell=cellt, forces=True, stress=True, coulomb="dsf", host_out=ho)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): eng.eval(ct, zt, mol, qt, cell=cellt, forces=True, stress=True, coulomb="dsf", host_out=ho)
    pr.disable(); print("host_out", ho); pstats.Stats(pr).sort_stats("tottime").print_stats(6)
