"""How close to their gates are the fixture comparisons?  energy error / gate for engine-vs-golden and engine-vs-fp32-oracle."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import energy_tol, golden
from aimnetcentral_amd import loader, synth
from aimnetcentral_amd.engine import HipEngine
from oracle import aimnet2_oracle as O
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
dev = eng.device
m32 = O.OracleModel(synth.synthetic_state_dict(0), torch.float32)
for name, coul, kw in [("taxol", "simple", {}), ("batch5", "simple", {}), ("dense3x14", "simple", {}), ("pbc96_dsf15", "dsf", dict(dsf_rc=15.0, dsf_alpha=0.2)),
                       ("pbc96_dsf8_wrapped", "dsf", dict(dsf_rc=8.0, dsf_alpha=0.25)), ("pbc2x96_dsf9", "dsf", dict(dsf_rc=9.0, dsf_alpha=0.2)), ("hvp40", "simple", {})]:
    g = golden(name)
    c, z = g["coord"], g["numbers"]
    if c.ndim == 3:
        B, N = c.shape[:2]; mol = np.repeat(np.arange(B), N); c = c.reshape(-1, 3); z = z.reshape(-1)
    else:
        mol = g["mol_idx"] if "mol_idx" in g.files else np.zeros(len(z), dtype=np.int64)
    q = np.atleast_1d(g["charge"]).astype(np.float32)
    cell = g["cell"] if "cell" in g.files else None
    r = eng.eval(torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.from_numpy(mol).to(dev), torch.from_numpy(q).to(dev),
                 cell=None if cell is None else torch.from_numpy(cell).to(dev), forces=True, stress=cell is not None, coulomb=coul, **kw)
    e = r["energy"].cpu().numpy()
    ref = O.evaluate(m32, c, z, q, mol, cell=cell, coulomb=coul, stress=cell is not None, **kw)
    tol = energy_tol(np.bincount(mol))
    ge = np.asarray(g["energy"]).reshape(-1)
    print(f"{name:20s} |hip-golden|/gate {np.abs(e - ge).max() / tol:.2f}   |hip-o32|/gate {np.abs(e - ref['energy']).max() / tol:.2f}   "
          f"forces |hip-golden| / gate {np.abs(r['forces'].cpu().numpy() - g['forces'].reshape(-1, 3)).max() / (1e-5 + 1e-4 * np.abs(g['forces']).max()):.2f}")
