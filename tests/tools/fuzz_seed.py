#!/usr/bin/env python
"""One configuration of tests/test_gpu_fuzz.py in detail: per-molecule energy of the engine, the fp32 oracle and the fp64
oracle, split into network and Coulomb parts, plus charges / forces / MLP outputs.   SEED=167 python tests/tools/fuzz_seed.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as F
from aimnetcentral_amd import loader, synth
from aimnetcentral_amd.engine import HipEngine
from oracle import aimnet2_oracle as O

seed = int(os.environ.get("SEED", 4))
case = F.make_case(seed)
c, z, mol, q, mult, nse, kw, okw, d3, label = case
print(label, "sizes", np.bincount(mol))
nq = 2 if nse else 1
sd = synth.synthetic_state_dict(0, None, nq)
m32, m64 = O.OracleModel(sd, torch.float32), O.OracleModel(sd, torch.float64)
eng = HipEngine(loader.synthetic_spec(0, num_charge_channels=nq), "cuda:0")
res = F.run_case(eng, case)
a = O.evaluate(m32, c, z, q, mol, return_intermediates=True, **okw)
b = O.evaluate(m64, c, z, q, mol, return_intermediates=True, **okw)
n, n_mol = len(z), int(mol.max()) + 1
print("E hip-64 ", res["energy"] - b["energy"])
print("E o32-64 ", a["energy"] - b["energy"])
print("q hip-64 max", np.abs(res["charges"] - b["charges"]).max(), " o32-64", np.abs(a["charges"] - b["charges"]).max())
print("F hip-64 max", np.abs(res["forces"] - b["forces"]).max(), " o32-64", np.abs(a["forces"] - b["forces"]).max(), "|F|max", np.abs(b["forces"]).max())
ea = eng.debug_view("e_atom").cpu().numpy().ravel()
for name, v in (("hip", ea), ("o32", a["_e_atom"][:n])):
    d = v.astype(np.float64) - b["_e_atom"][:n]
    per = np.zeros(n_mol); np.add.at(per, mol, d)
    print(f"e_atom {name}-64: max|d| {np.abs(d).max():.3e}  per-molecule sums {per}")
# everything that is not the network term: SAE (exact) + Coulomb (+ D3), per molecule
ec = eng.debug_view("ecoul").cpu().numpy().ravel()
per = np.zeros(n_mol); np.add.at(per, mol, ec)
sae = np.zeros(n_mol); np.add.at(sae, mol, np.asarray(sd["outputs.atomic_shift.shifts.weight"]).reshape(-1)[z])
pe64 = np.zeros(n_mol); np.add.at(pe64, mol, b["_e_atom"][:n])
pe32 = np.zeros(n_mol); np.add.at(pe32, mol, a["_e_atom"][:n].astype(np.float64))
print("pair terms hip - (E64 - sae - e_atom64):", per - (b["energy"] - sae - pe64))
print("pair terms o32 - same                  :", (a["energy"] - sae - pe32) - (b["energy"] - sae - pe64))
if "cell" in kw:
    # the same system entered with the ORACLE's wrapped coordinates: is the difference the rounding of the wrap itself?
    case2 = (a["coord_wrapped"].astype(np.float32),) + case[1:]
    res2 = F.run_case(eng, case2)
    xw = eng.debug_view("xw").cpu().numpy()
    print("engine wrap vs oracle wrap: max |dx| %.3e A" % np.abs(xw - a["coord_wrapped"]).max())
    print("E hip(oracle-wrapped input)-64 ", res2["energy"] - b["energy"], " F max", np.abs(res2["forces"] - b["forces"]).max())
