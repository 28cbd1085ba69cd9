"""Which is closer to the fp64 truth: the fp32 oracle, the engine with / without the pass-0 embedding fold?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aimnetcentral_amd import loader, synth
from aimnetcentral_amd.engine import HipEngine
from oracle import aimnet2_oracle as O
from conftest import golden
sd = synth.synthetic_state_dict(0)
o32, o64 = O.OracleModel(sd, torch.float32), O.OracleModel(sd, torch.float64)
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
for name, coul in (("batch5", "simple"), ("taxol", "simple")):
    g = golden(name)
    mol = g["mol_idx"] if "mol_idx" in g.files else np.zeros(len(g["numbers"]), dtype=np.int64)
    r32 = O.evaluate(o32, g["coord"], g["numbers"], g["charge"], mol)
    r64 = O.evaluate(o64, g["coord"], g["numbers"], g["charge"], mol)
    dev = eng.device
    res = eng.eval(torch.from_numpy(g["coord"]).to(dev), torch.from_numpy(g["numbers"]).to(dev), torch.from_numpy(mol).to(dev),
                   torch.from_numpy(np.atleast_1d(g["charge"]).astype(np.float32)).to(dev), forces=True, coulomb=coul)
    e = res["energy"].cpu().numpy()
    print(name, "FOLD", os.environ.get("AIMNET_P0_FOLD", "1"))
    print("  hip - o64 :", np.array2string(e - r64["energy"], precision=2))
    print("  o32 - o64 :", np.array2string(r32["energy"] - r64["energy"], precision=2))
    print("  gold- o64 :", np.array2string(np.atleast_1d(g["energy"]) - r64["energy"], precision=2))
    print("  F: hip-o64 %.2e  o32-o64 %.2e" % (np.abs(res["forces"].cpu().numpy() - r64["forces"]).max(), np.abs(r32["forces"] - r64["forces"]).max()))
