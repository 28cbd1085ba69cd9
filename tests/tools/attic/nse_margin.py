import numpy as np, torch, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from aimnetcentral_amd import loader, synth
from aimnetcentral_amd.engine import HipEngine
from oracle import aimnet2_oracle as O
g = np.load('/root/repo/tests/golden/nse.npz')
eng = HipEngine(loader.synthetic_spec(0, num_charge_channels=2), "cuda:0")
dev = eng.device
sd = synth.synthetic_state_dict(0, None, 2)
m32, m64 = O.OracleModel(sd, torch.float32), O.OracleModel(sd, torch.float64)
q, mult, mol = g["b5_charge"], g["b5_mult"], g["b5_mol_idx"]
ch = np.stack([0.5*q + 0.5*(mult-1), 0.5*q - 0.5*(mult-1)], -1).astype(np.float32)
r = eng.eval(torch.from_numpy(g["b5_coord"]).to(dev), torch.from_numpy(g["b5_numbers"]).to(dev), torch.from_numpy(mol).to(dev), torch.from_numpy(ch).to(dev), forces=True)
e = r["energy"].cpu().numpy()
a = O.evaluate(m32, g["b5_coord"], g["b5_numbers"], q, mol, mult=mult)
b = O.evaluate(m64, g["b5_coord"], g["b5_numbers"], q, mol, mult=mult)
print("sizes", np.bincount(mol))
print("hip-golden", e - g["b5_energy"])
print("hip-o32   ", e - a["energy"])
print("hip-64    ", e - b["energy"])
print("o32-64    ", a["energy"] - b["energy"])
print("gold-64   ", g["b5_energy"] - b["energy"])
