#!/usr/bin/env python
"""Energy of the bench's 2 304-atom parity sample: engine in its GEMM modes, the fp32 oracle and the fp64 oracle (GPU box).
Who is how far from fp64?  (the fp32-vs-fp32 gate 5e-7 eV x atoms = 1.152e-3 eV sits at the noise of either side on these hot weights)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from aimnetcentral_amd import loader, synth  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402
from oracle import aimnet2_oracle as O  # noqa: E402

torch.set_num_threads(16)
dev = torch.device("cuda:0")
smp = bench.parity_samples()["pbc2304"]
eng = HipEngine(loader.synthetic_spec(0), dev)
args = (torch.from_numpy(smp["coord"]).to(dev), torch.from_numpy(np.asarray(smp["numbers"])).to(dev), torch.from_numpy(np.asarray(smp["mol_idx"])).to(dev),
        torch.from_numpy(smp["charge"]).to(dev))
cell = torch.from_numpy(smp["cell"]).to(dev)
E = {}
for label, opts in (("h2", {}), ("bf16x3", {"gemm_h2": 0}), ("bf3 split in loop", {"gemm_presplit": 0}), ("exact fp32", {"gemm_bf3": 0})):
    for k, v in opts.items():
        eng.set_option(k, v)
    r = eng.eval(*args, cell=cell, forces=True, stress=True, coulomb="dsf")
    E[label] = float(r["energy"][0])
    for k in opts:
        eng.set_option(k, 1)
sd = synth.synthetic_state_dict(0)
for nt in (16, 1):
    torch.set_num_threads(nt)
    E[f"oracle fp32 ({nt} threads)"] = float(smp["step"]()["energy"][0])
torch.set_num_threads(16)
o64 = O.OracleModel(sd, torch.float64)
mol = np.asarray(smp["mol_idx"])
xw = O.wrap_into_cell(smp["coord"], smp["cell"], mol, np.ones(3, dtype=bool))
nb, sh = O.neighbor_list_fast(xw, 5.0, mol, smp["cell"].astype(np.float64), np.ones(3, dtype=bool))
nbl, shl = O.neighbor_list_fast(xw, 15.0, mol, smp["cell"].astype(np.float64), np.ones(3, dtype=bool))
e64 = float(O.evaluate(o64, coord=xw, numbers=smp["numbers"], charge=np.zeros(1, np.float32), mol_idx=mol, cell=smp["cell"], coulomb="dsf",
                       forces=False, nbmat=nb, shifts=sh, nbmat_lr=nbl, shifts_lr=shl)["energy"][0])
print("fp64 oracle E = %.6f eV; gate (5e-7 eV x 2304 atoms) = 1.152e-3 eV" % e64)
for k, v in E.items():
    print(f"{k:28s} E - E64 = {v - e64:+.3e} eV")
