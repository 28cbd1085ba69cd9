#!/usr/bin/env python
"""GPU-side stage-by-stage comparison of the HIP engine against the CPU oracle (debug tool).

    python tests/tools/stagecheck.py            # on a GPU box

Prints max abs error per intermediate for a non-periodic and a periodic system so that one
gpurun call localises a wrong kernel.  Not collected by pytest.
"""
from __future__ import annotations

import os
os.environ.setdefault("AIMNET_KEEP_INTERMEDIATES", "1")  # distinct x[p] / hidden-activation buffers: their debug views stay valid
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from aimnetcentral_amd import loader, synth  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402
from oracle import aimnet2_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def row(name, got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if got.shape != ref.shape:
        print(f"  {name:14s} SHAPE MISMATCH got {got.shape} ref {ref.shape}")
        return
    err = np.abs(got - ref)
    bad = ~np.isfinite(got)
    print(f"  {name:14s} max|d|={np.nanmax(err):.3e}  ref|max|={np.abs(ref).max():.3e}  nonfinite={int(bad.sum())}  "
          f"argmax={np.unravel_index(np.nanargmax(err), err.shape) if err.size else None}")


def check(eng, om, name, g, coulomb, **kw):
    print(f"== {name}")
    dev = eng.device
    mol = g["mol_idx"] if "mol_idx" in g.files else np.zeros(len(g["numbers"]), dtype=np.int64)
    cell = g["cell"] if "cell" in g.files else None
    charge = np.atleast_1d(g["charge"]).astype(np.float32)
    ref = O.evaluate(om, g["coord"], g["numbers"], charge, mol, cell=cell, coulomb=coulomb, stress=cell is not None,
                     return_intermediates=True, **kw)
    res = eng.eval(torch.from_numpy(g["coord"]).to(dev), torch.from_numpy(g["numbers"]).to(dev), torch.from_numpy(mol).to(dev),
                   torch.from_numpy(charge).to(dev), cell=None if cell is None else torch.from_numpy(cell).to(dev),
                   forces=True, stress=cell is not None, coulomb=coulomb, **kw)
    torch.cuda.synchronize()
    n = len(g["numbers"])
    print("  status", eng.last_status)
    cnt = eng.debug_view("nb_cnt").cpu().numpy().ravel()
    ref_cnt = (ref["nbmat"][:n] < n).sum(1)
    row("nb_cnt", cnt, ref_cnt)
    xw = eng.debug_view("xw").cpu().numpy()
    row("xw", xw, ref["coord_wrapped"])
    pg = eng.debug_view("pair_geom").cpu().numpy().reshape(n, -1, 4)
    dref = ref["_d_ij"][:n]
    mx = 0.0
    for i in range(n):
        a = np.sort(pg[i, : cnt[i], 3])
        b = np.sort(dref[i][ref["nbmat"][i] < n])
        if len(a) == len(b) and len(a):
            mx = max(mx, np.abs(a - b).max())
    print(f"  sorted d_ij     max|d|={mx:.3e}")
    for p in range(3):
        x = eng.debug_view(f"x{p}").cpu().numpy()
        w = ref[f"_mlp{p}_in"].shape[1]
        row(f"x{p} (mlp in)", x[:, :w], ref[f"_mlp{p}_in"][:n])
        nl = len(eng.spec.mlp_dims[p]) - 1
        y = eng.debug_view(f"h{p}_{nl - 1}").cpu().numpy()
        wo = ref[f"_mlp{p}_out"].shape[1]
        row(f"y{p} (mlp out)", y[:, :wo], ref[f"_mlp{p}_out"][:n])
        if p < 2:
            row(f"q{p}", eng.debug_view(f"q{p}").cpu().numpy().ravel(), ref[f"_q{p}"][:n])
    row("e_atom", eng.debug_view("e_atom").cpu().numpy().ravel(), ref["_e_atom"][:n])
    row("energy", res["energy"].cpu().numpy(), ref["energy"])
    row("charges", res["charges"].cpu().numpy(), ref["charges"])
    row("forces", res["forces"].cpu().numpy(), ref["forces"])
    if cell is not None:
        row("stress", res["stress"].cpu().numpy(), ref["stress"])
    for k in ("energy", "forces", "charges", "stress"):
        if k in g.files and k in res:
            row(f"{k} vs REF", res[k].cpu().numpy(), g[k])


def main():
    spec = loader.synthetic_spec(0)
    eng = HipEngine(spec, "cuda:0")
    om = O.OracleModel(synth.synthetic_state_dict(0), torch.float32)
    check(eng, om, "taxol / simple", np.load(os.path.join(GOLD, "taxol.npz")), "simple")
    check(eng, om, "batch5 / simple", np.load(os.path.join(GOLD, "batch5.npz")), "simple")
    g = np.load(os.path.join(GOLD, "pbc96_dsf8_wrapped.npz"))
    check(eng, om, "pbc96 dsf8", g, "dsf", dsf_rc=float(g["dsf_rc"]), dsf_alpha=float(g["dsf_alpha"]))
    g = np.load(os.path.join(GOLD, "pbc96_dsf15.npz"))
    check(eng, om, "pbc96 dsf15", g, "dsf", dsf_rc=float(g["dsf_rc"]), dsf_alpha=float(g["dsf_alpha"]))
    g = np.load(os.path.join(GOLD, "pbc2x96_dsf9.npz"))
    check(eng, om, "pbc2x96 dsf9", g, "dsf", dsf_rc=float(g["dsf_rc"]), dsf_alpha=float(g["dsf_alpha"]))


if __name__ == "__main__":
    main()
