"""Which molecules of the config-5 shard carry the largest fp32 energy errors, engine and fp32 oracle, against the fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aimnetcentral_amd import AIMNet2Calculator, loader, workloads, synth, dist as adist
from oracle import aimnet2_oracle as O

c, z, mol, q = workloads.random_batch(1024, 50, 50, seed=5)
a, b = adist.shard_frames(np.bincount(mol, minlength=1024), 8)[0]
c, z, mol, q = adist.local_batch(c, z, mol, q, a, b)
sd = synth.synthetic_state_dict(0)
r32 = O.evaluate(O.OracleModel(sd, torch.float32), c, z, q, mol, coulomb="simple", return_intermediates=True)
r64 = O.evaluate(O.OracleModel(sd, torch.float64), c, z, q, mol, coulomb="simple", return_intermediates=True)
calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
out = calc({"coord": c, "numbers": z, "mol_idx": mol, "charge": q}, forces=True)
e = out["energy"].cpu().numpy()
ea = calc.engine.debug_view("e_atom").cpu().numpy().ravel()
err, err_ref = np.abs(e - r64["energy"]), np.abs(r32["energy"] - r64["energy"])
print("rms engine %.3e  rms fp32 oracle %.3e" % (np.sqrt((err**2).mean()), np.sqrt((err_ref**2).mean())))
for m in np.argsort(-err)[:6]:
    sel = np.where(np.asarray(mol) == m)[0]
    x = c[sel]
    d = np.linalg.norm(x[:, None] - x[None], axis=-1) + np.eye(len(sel)) * 9
    da = ea[sel] - r64["_e_atom"][sel]
    dr = r32["_e_atom"][sel] - r64["_e_atom"][sel]
    print("mol %3d err_eng %.2e err_ref %.2e  min d %.3f  max|F| %.1f  per-atom eng max %.2e sum %.2e | ref max %.2e sum %.2e | max|e_atom| %.1f" % (
        m, err[m], err_ref[m], d.min(), np.abs(r64["forces"][sel]).max() if "forces" in r64 else -1, np.abs(da).max(), da.sum(), np.abs(dr).max(), dr.sum(),
        np.abs(r64["_e_atom"][sel]).max()))
