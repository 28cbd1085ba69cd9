#!/usr/bin/env python
"""Distribution of the energy error of the randomised sweep (tests/test_gpu_fuzz.py) over its statistical gate, per GEMM mode.

For every seed: max over the molecules of |E_hip - E_fp64| / (reference gate + the fp32 oracle's own distance from fp64 + 3 sigma of
the random walk of its per-atom errors) - the quantity the test asserts <= 1.  Modes: exact-fp32 kernels, split kernels with the
in-loop split (gemm_bf3.hip, sign-flipped second phase at 0.56 K) and with pre-split activations (gemm_bf3a.hip + gemm_head.hip,
interleaved accumulator sets), all forced onto every batch size.  Env: SEEDS=lo:hi (default 0:332)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as Z  # noqa: E402
from conftest import energy_tol  # noqa: E402
from aimnetcentral_amd import loader, synth  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402
from oracle import aimnet2_oracle as O  # noqa: E402

lo, hi = (int(v) for v in os.environ.get("SEEDS", "0:332").split(":"))
eng = {False: HipEngine(loader.synthetic_spec(0), "cuda:0"), True: HipEngine(loader.synthetic_spec(0, num_charge_channels=2), "cuda:0")}
orc = {False: (O.OracleModel(synth.synthetic_state_dict(0), torch.float32), O.OracleModel(synth.synthetic_state_dict(0), torch.float64)),
       True: (O.OracleModel(synth.synthetic_state_dict(0, None, 2), torch.float32), O.OracleModel(synth.synthetic_state_dict(0, None, 2), torch.float64))}
modes = {"exact_f32": (0, 1), "bf3_split_in_loop": (2, 0), "bf3_presplit": (2, 1)}
ratios = {m: [] for m in modes}
for seed in range(lo, hi):
    case = Z.make_case(seed)
    c, z, mol, q, mult, nse, kw, okw, d3, label = case
    o32, o64 = orc[nse]
    ref = O.evaluate(o32, c, z, q, mol, return_intermediates=True, **okw)
    ref64 = O.evaluate(o64, c, z, q, mol, return_intermediates=True, **dict(okw, forces=False, stress=False))
    sizes = np.bincount(mol)
    d = (ref["_e_atom"][: len(mol)].astype(np.float64) - ref64["_e_atom"][: len(mol)]) ** 2
    walk = np.zeros(len(sizes))
    np.add.at(walk, mol, d)
    gate = energy_tol(sizes) + np.abs(ref["energy"] - ref64["energy"]) + 3.0 * np.sqrt(walk)
    for name, (bf3, ps) in modes.items():
        e = eng[nse]
        e.set_option("gemm_bf3", bf3)
        e.set_option("gemm_presplit", ps)
        cc = (ref["coord_wrapped"].astype(np.float32),) + case[1:] if "cell" in kw else case
        r = Z.run_case(e, cc)
        ratios[name].append(float(np.max(np.abs(r["energy"] - ref64["energy"]) / gate)))
for name, v in ratios.items():
    v = np.array(v)
    print(f"{name:18s} seeds {lo}:{hi}  median {np.median(v):.3f}  p90 {np.percentile(v, 90):.3f}  p99 {np.percentile(v, 99):.3f}  max {v.max():.3f}  "
          f"> 1: {int((v > 1).sum())} (seeds {[lo + int(k) for k in np.nonzero(v > 1)[0]]})")
