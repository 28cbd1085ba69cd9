#!/usr/bin/env python
"""One seed of tests/test_gpu_fuzz.py (non-periodic branch) in detail: per-molecule energy of the engine, the fp32 oracle and
the fp64 oracle, charges, e_atom and Coulomb parts.   SEED=4 python tests/tools/fuzz_seed.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aimnetcentral_amd import loader, synth, workloads
from aimnetcentral_amd.engine import HipEngine
from oracle import aimnet2_oracle as O

seed = int(os.environ.get("SEED", 4))
rng = np.random.Generator(np.random.PCG64(1000 + seed))
n_mol = int(rng.integers(1, 6))
c, z, mol, _ = workloads.random_batch(n_mol, 3, 40, seed=int(rng.integers(1 << 30)))
if rng.random() < 0.5:
    kw = dict(coulomb="simple")
else:
    kw = dict(coulomb="dsf", dsf_rc=float(rng.uniform(5.0, 12.0)), dsf_alpha=float(rng.uniform(0.15, 0.3)))
q = rng.integers(-1, 2, size=n_mol).astype(np.float32) if rng.random() < 0.6 else np.zeros(n_mol, dtype=np.float32)
print("sizes", np.bincount(mol), "q", q, kw)
sd = synth.synthetic_state_dict(0)
m32, m64 = O.OracleModel(sd, torch.float32), O.OracleModel(sd, torch.float64)
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
dev = eng.device
r = eng.eval(torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.from_numpy(mol).to(dev), torch.from_numpy(q).to(dev), forces=True, **kw)
res = {k: v.cpu().numpy() for k, v in r.items()}
a = O.evaluate(m32, c, z, q, mol, return_intermediates=True, **kw)
b = O.evaluate(m64, c, z, q, mol, return_intermediates=True, **kw)
print("E hip-64 ", res["energy"] - b["energy"])
print("E o32-64 ", a["energy"] - b["energy"])
print("q hip-64 max", np.abs(res["charges"] - b["charges"]).max(), " o32-64", np.abs(a["charges"] - b["charges"]).max())
print("F hip-64 max", np.abs(res["forces"] - b["forces"]).max(), " o32-64", np.abs(a["forces"] - b["forces"]).max(), "|F|max", np.abs(b["forces"]).max())
ea = eng.debug_view("e_atom").cpu().numpy().ravel()
n = len(z)
for name, v in (("hip", ea), ("o32", a["_e_atom"][:n])):
    d = v.astype(np.float64) - b["_e_atom"][:n]
    per = np.zeros(n_mol); np.add.at(per, mol, d)
    print(f"e_atom {name}-64: max|d| {np.abs(d).max():.3e}  per-molecule sums {per}")
ec = eng.debug_view("ecoul").cpu().numpy().ravel()
per = np.zeros(n_mol); np.add.at(per, mol, ec)
print("hip Coulomb part per molecule", per)
for p in (0, 1):
    qv = eng.debug_view(f"q{p}").cpu().numpy().ravel()[:n]
    print(f"q{p} hip-64 {np.abs(qv - b[f'_q{p}'][:n]).max():.3e}  o32-64 {np.abs(a[f'_q{p}'][:n] - b[f'_q{p}'][:n]).max():.3e}")
for p in (0, 1, 2):
    h = eng.debug_view(f"h{p}_{2 if p < 2 else 3}").cpu().numpy()
    ref = b[f"_mlp{p}_out"][:n]
    print(f"mlp{p}_out hip-64 {np.abs(h[:, :ref.shape[1]] - ref).max():.3e}  o32-64 {np.abs(a[f'_mlp{p}_out'][:n] - ref).max():.3e}  |max| {np.abs(ref).max():.2f}")
