#!/usr/bin/env python
"""Bitwise repeatability probe (GPU box): the 2 304-atom crystal evaluated repeatedly with fresh workspaces under several switch
settings; prints which outputs differ between repeats of the SAME setting."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import loader, workloads  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402

dev = torch.device("cuda:0")
eng = HipEngine(loader.synthetic_spec(0), dev)
reps = tuple(int(x) for x in os.environ.get("REPS", "2,3,4").split(","))
c, z, cell = workloads.glucose_supercell(reps)
args = (torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev), torch.zeros(len(z), dtype=torch.int64, device=dev),
        torch.zeros(1, device=dev))
cell_t = torch.from_numpy(cell.astype(np.float32)).to(dev)


def go():
    r = eng.eval(*args, cell=cell_t, forces=True, stress=True, coulomb="dsf", dsf_rc=9.0)
    out = {k: v.cpu().numpy().copy() for k, v in r.items()}
    out["e_atom"] = eng.debug_view("e_atom").cpu().numpy().copy()
    return out


for label, opts in (("h2", {}), ("h2 energy_rides=0", {"energy_rides": 0}), ("bf3", {"gemm_h2": 0}), ("bf3 energy_rides=0", {"gemm_h2": 0, "energy_rides": 0}),
                    ("exact", {"gemm_bf3": 0})):
    for k, v in opts.items():
        eng.set_option(k, v)
    runs = []
    for i in range(6):
        eng._ws = None  # fresh workspace
        junk = torch.randn(50_000_000, device=dev)  # stir the allocator
        del junk
        runs.append(go())
    diffs = {}
    for r in runs[1:]:
        for k in r:
            if not np.array_equal(r[k], runs[0][k]):
                d = np.abs(r[k].astype(np.float64) - runs[0][k].astype(np.float64))
                diffs.setdefault(k, []).append((int((d > 0).sum()), float(d.max())))
    print(label, "E=%.10f" % runs[0]["energy"][0], "differing:", diffs or "none", flush=True)
    for k in opts:
        eng.set_option(k, 1)
ref = None
