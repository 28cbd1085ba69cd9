import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conftest import golden
from aimnetcentral_amd import AIMNet2Calculator, loader, synth
from oracle import aimnet2_oracle as O
g = golden("relaxed256")
mol = g["mol_idx"].astype(np.int64); sizes = np.bincount(mol)
gate = np.maximum(1e-5, 5e-7 * sizes)
calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
data = {"coord": g["coord"], "numbers": g["numbers"].astype(np.int64), "mol_idx": mol, "charge": g["charge"]}
sd = synth.synthetic_state_dict(0)
r64 = O.evaluate(O.OracleModel(sd, torch.float64), g["coord"], g["numbers"].astype(np.int64), g["charge"], mol, forces=True)
e64, f64 = r64["energy"], r64["forces"].astype(np.float64)
f32o = O.evaluate(O.OracleModel(sd, torch.float32), g["coord"], g["numbers"].astype(np.int64), g["charge"], mol, forces=True)["forces"]
fg = g["forces"].astype(np.float64)
print("forces: golden vs fp64 max %.2e rms %.2e; fp32 oracle vs fp64 max %.2e; gate 1e-5+1e-4 max|F| = %.2e" % (np.abs(fg - f64).max(), np.sqrt(np.mean((fg - f64)**2)), np.abs(f32o - f64).max(), 1e-5 + 1e-4 * np.abs(fg).max()))
print("reference golden vs fp64 oracle: max |dE|/gate %.2f, rms %.2e" % (np.max(np.abs(g["energy"] - e64) / gate), np.sqrt(np.mean((g["energy"] - e64) ** 2))))
for mode in (1, 0, 2):
    calc.engine.set_option("gemm_bf3", mode)
    out = calc(data, forces=True)
    e = out["energy"].cpu().numpy()
    f = out["forces"].cpu().numpy().astype(np.float64)
    df = np.abs(f - f64).max(axis=1)
    print(f"gemm_bf3={mode}: forces vs fp64 max {df.max():.2e} (atom {df.argmax()}, mol {mol[df.argmax()]}), rms {np.sqrt(np.mean((f - f64)**2)):.2e}, atoms above 8e-5: {(df > 8e-5).sum()};  vs golden max {np.abs(f - fg).max():.2e}")
    r = np.abs(e - g["energy"]) / gate
    r64 = np.abs(e - e64) / gate
    print(f"gemm_bf3={mode}: vs golden: worst {r.max():.2f} (mol {r.argmax()}, n={sizes[r.argmax()]}), over gate {(r > 1).sum()}/256, rms dE {np.sqrt(np.mean((e - g['energy'])**2)):.2e};"
          f"  vs fp64: worst {r64.max():.2f}, over {(r64 > 1).sum()}, rms {np.sqrt(np.mean((e - e64)**2)):.2e}, mean signed {np.mean(e - e64):+.2e}")
