#!/usr/bin/env python
"""Force parity at the reference's LITERAL gate - elementwise allclose(rtol 1e-4, atol 1e-5), tests/test_calculator_gpu.py:137,464 -
for every golden the unmodified reference produced: engine vs golden NEXT TO oracle (fp32) vs golden and golden vs the fp64 oracle
(GPU box).  Prints a markdown table (profiles/r5_parity_literal.md keeps it); VERDICT r4 item 2.

Columns per comparison: elements outside the gate / elements, worst |d| / (1e-5 + 1e-4 |ref|), max|dF|; plus max|dE| per molecule."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aimnetcentral_amd import loader, synth  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402
from oracle import aimnet2_oracle as O  # noqa: E402

torch.set_num_threads(16)
dev = torch.device("cuda:0")
GOLD = os.path.join(ROOT, "tests", "golden")


def viol(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    ratio = np.abs(got - ref) / (1e-5 + 1e-4 * np.abs(ref))
    return int((ratio > 1).sum()), ratio.size, float(ratio.max()), float(np.abs(got - ref).max())


def section(g, prefix):
    return {k[len(prefix) + 1:]: g[k] for k in g.files if k.startswith(prefix + "_")}


def plain(g):
    return {k: g[k] for k in g.files}


cases = []  # (label, family, dict)
for n in ("taxol", "batch5", "cold24", "relaxed256", "pbc96_dsf15", "pbc96_dsf8_wrapped", "pbc2x96_dsf9", "hvp40"):
    cases.append((n, "hot", plain(np.load(os.path.join(GOLD, n + ".npz")))))
gn = np.load(os.path.join(GOLD, "nse.npz"))
for pre in ("t40", "b5", "pbc"):
    cases.append(("nse/" + pre, "nse", section(gn, pre)))
gc = np.load(os.path.join(GOLD, "coldw.npz"))
for pre in ("taxol", "batch5", "rand8", "pbc96"):
    cases.append(("coldw/" + pre, "cold", section(gc, pre)))

engines, oracles = {}, {}


def get(family):
    if family not in engines:
        kw = {"hot": {}, "cold": {"cold": True}, "nse": {"num_charge_channels": 2}}[family]
        engines[family] = HipEngine(loader.synthetic_spec(0, **kw), dev)
        sd = synth.synthetic_state_dict(0, None, 2) if family == "nse" else synth.synthetic_state_dict(0, cold=(family == "cold"))
        oracles[family] = (O.OracleModel(sd, torch.float32), O.OracleModel(sd, torch.float64))
    return engines[family], oracles[family]


print("| golden | atoms | max\\|F\\| | engine vs golden: outside / of, worst x gate, max\\|dF\\|, max\\|dE\\| | fp32 oracle vs golden | golden vs fp64 oracle |")
print("|---|---|---|---|---|---|")
for label, fam, g in cases:
    eng, (o32, o64) = get(fam)
    n = len(g["numbers"])
    mol = np.asarray(g.get("mol_idx", np.zeros(n, dtype=np.int64))).astype(np.int64)
    charge = np.atleast_1d(g["charge"]).astype(np.float32) if "charge" in g else np.zeros(1, np.float32)
    okw, ekw = {}, {}
    if "cell" in g:
        rc = float(g["dsf_rc"]) if "dsf_rc" in g else 15.0
        al = float(g["dsf_alpha"]) if "dsf_alpha" in g else 0.2
        okw = dict(cell=g["cell"], coulomb="dsf", stress=True, dsf_rc=rc, dsf_alpha=al)
        ekw = dict(cell=torch.from_numpy(g["cell"]).to(dev), coulomb="dsf", stress=True, dsf_rc=rc, dsf_alpha=al)
        if "pbc" in g:
            okw["pbc"] = g["pbc"]
            ekw["pbc"] = tuple(bool(x) for x in g["pbc"])
    else:
        ekw = dict(coulomb="simple")
    if fam == "nse":
        mult = np.atleast_1d(g["mult"]).astype(np.float32)
        okw["mult"] = g["mult"]
        q = np.atleast_1d(charge)
        charge_e = np.stack([q / 2 + (mult - 1) / 2, q / 2 - (mult - 1) / 2], axis=1).astype(np.float32)
    else:
        charge_e = charge
    r = eng.eval(torch.from_numpy(np.asarray(g["coord"], np.float32)).to(dev), torch.from_numpy(np.asarray(g["numbers"]).astype(np.int64)).to(dev),
                 torch.from_numpy(mol).to(dev), torch.from_numpy(charge_e).to(dev), forces=True, **ekw)
    r = {k: v.cpu().numpy() for k, v in r.items()}
    r32 = O.evaluate(o32, np.asarray(g["coord"], np.float32), np.asarray(g["numbers"]).astype(np.int64), g.get("charge", charge), mol, **okw)
    r64 = O.evaluate(o64, np.asarray(g["coord"], np.float32), np.asarray(g["numbers"]).astype(np.int64), g.get("charge", charge), mol, **okw)
    cols = []
    for a, b, ea, eb in ((r["forces"], g["forces"], r["energy"], g["energy"]), (r32["forces"], g["forces"], r32["energy"], g["energy"]),
                         (g["forces"], r64["forces"], g["energy"], r64["energy"])):
        bad, tot, worst, dmax = viol(a, b)
        cols.append(f"{bad} / {tot}, {worst:.2f}, {dmax:.1e}, {np.abs(np.asarray(ea, np.float64) - eb).max():.1e}")
    print(f"| {label} | {n} | {np.abs(g['forces']).max():.2f} | " + " | ".join(cols) + " |", flush=True)
