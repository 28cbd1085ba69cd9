import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import loader, synth, workloads
from aimnetcentral_amd.engine import HipEngine
from oracle import aimnet2_oracle as O
c, z, mol, q = workloads.random_batch(128, 50, 50, 5)
sd = synth.synthetic_state_dict(0)
nb, _ = O.neighbor_list_fast(c, 5.0, mol); nbl, _ = O.neighbor_list(c, float("inf"), mol)
r32 = O.evaluate(O.OracleModel(sd, torch.float32), coord=c, numbers=z, charge=q, mol_idx=mol, coulomb="simple", nbmat=nb, nbmat_lr=nbl, forces=False)
r64 = O.evaluate(O.OracleModel(sd, torch.float64), coord=c, numbers=z, charge=q, mol_idx=mol, coulomb="simple", nbmat=nb, nbmat_lr=nbl, forces=False)
rms_ref = np.sqrt(np.mean((r32["energy"] - r64["energy"]) ** 2)); mean_ref = np.mean(r32["energy"] - r64["energy"])
print(f"fp32 oracle vs fp64: rms {rms_ref:.3e} mean {mean_ref:+.3e}")
eng = HipEngine(loader.synthetic_spec(0), "cuda:0"); dev = eng.device
t = [torch.from_numpy(a).to(dev) for a in (c, z, mol, q)]
for mode in (0, 1):
    eng.set_option("gemm_bf3", mode)
    e = eng.eval(*t, forces=True, coulomb="simple")["energy"].cpu().numpy()
    d = e - r64["energy"]
    print(f"flip {os.environ.get('AIMNET_BF3_FLIP', '600')} gemm_bf3={mode}: rms {np.sqrt(np.mean(d**2)):.3e} (ratio {np.sqrt(np.mean(d**2)) / rms_ref:.2f}) mean {d.mean():+.3e}")
