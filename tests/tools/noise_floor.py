"""Absolute accuracy of every stage: |hip - o64| next to |o32 - o64| (both fp32 implementations against the fp64 truth)."""
import os, sys, numpy as np, torch
os.environ.setdefault("AIMNET_KEEP_INTERMEDIATES", "1")  # distinct x[p] / hidden-activation buffers: their debug views stay valid
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aimnetcentral_amd import loader, synth
from aimnetcentral_amd.engine import HipEngine
from oracle import aimnet2_oracle as O
sd = synth.synthetic_state_dict(0)
o32, o64 = O.OracleModel(sd, torch.float32), O.OracleModel(sd, torch.float64)
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
for name in ("taxol", "batch5"):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    n = len(g["numbers"]); dev = eng.device
    mol = g["mol_idx"] if "mol_idx" in g.files else np.zeros(n, dtype=np.int64)
    charge = np.atleast_1d(g["charge"]).astype(np.float32)
    r32 = O.evaluate(o32, g["coord"], g["numbers"], charge, mol, return_intermediates=True)
    r64 = O.evaluate(o64, g["coord"], g["numbers"], charge, mol, return_intermediates=True)
    res = eng.eval(torch.from_numpy(g["coord"]).to(dev), torch.from_numpy(g["numbers"]).to(dev), torch.from_numpy(mol).to(dev),
                   torch.from_numpy(charge).to(dev), forces=True, coulomb="simple")
    torch.cuda.synchronize()
    print("==", name)
    def row(label, hip, a32, a64):
        hip, a32, a64 = (np.asarray(v, dtype=np.float64) for v in (hip, a32, a64))
        print(f"  {label:12s} |hip-o64|={np.abs(hip - a64).max():.2e}  |o32-o64|={np.abs(a32 - a64).max():.2e}  rms hip {np.sqrt(((hip-a64)**2).mean()):.2e} o32 {np.sqrt(((a32-a64)**2).mean()):.2e}  scale {np.abs(a64).max():.2e}")
    for p in range(3):
        w = r64[f"_mlp{p}_in"].shape[1]
        row(f"x{p}", eng.debug_view(f"x{p}").cpu().numpy()[:, :w], r32[f"_mlp{p}_in"][:n], r64[f"_mlp{p}_in"][:n])
        nl = len(eng.spec.mlp_dims[p]) - 1
        wo = r64[f"_mlp{p}_out"].shape[1]
        row(f"y{p}", eng.debug_view(f"h{p}_{nl - 1}").cpu().numpy()[:, :wo], r32[f"_mlp{p}_out"][:n], r64[f"_mlp{p}_out"][:n])
        if p < 2:
            row(f"q{p}", eng.debug_view(f"q{p}").cpu().numpy().ravel(), r32[f"_q{p}"][:n], r64[f"_q{p}"][:n])
    row("e_atom", eng.debug_view("e_atom").cpu().numpy().ravel(), r32["_e_atom"][:n], r64["_e_atom"][:n])
    row("energy", res["energy"].cpu().numpy(), r32["energy"], r64["energy"])
    row("forces", res["forces"].cpu().numpy(), r32["forces"], r64["forces"])
