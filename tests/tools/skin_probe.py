#!/usr/bin/env python
"""What a Verlet skin would buy the 15 A neighbour MATRIX (external DFT-D3 on; VERDICT r4 item 8 / SURVEY 8f next-2), measured
without building the feature (GPU box): the 10 080-atom crystal with the D3 list at cutoff rc and at rc + skin.

  t(rc)            step time, list rebuilt every step (what the engine does)
  nlist(rc)        the list-building family's share of it (per-family events), minus the same family with D3 off = the D3 list build
  t(rc + skin)     step time with rows wide enough for the skin, list still rebuilt every step
 => with the list kept for k steps:  t_reuse(k) ~ t(rc + skin) - build(rc + skin) (1 - 1/k) + t_check
    (t_check: a displacement check + the re-wrap into the build-time frame, ~5 us: one floor-level launch)
Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import loader, workloads  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402

dev = torch.device("cuda:0")
eng = HipEngine(loader.synthetic_spec(0), dev)
g, t = (np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", f + ".npz")) for f in ("dftd3", "dftd3_subset"))
tables = {k: t[k] for k in ("c6ab", "cn_ref", "rcov", "r4r2")}
eng.set_dftd3_tables(tables)
c, z, cell = workloads.glucose_supercell((7, 3, 5))
rng = np.random.default_rng(0)
c = (c + rng.normal(0, 0.02, c.shape)).astype(np.float32)
args = (torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.zeros(len(z), dtype=torch.int64, device=dev), torch.zeros(1, device=dev))
cell_t = torch.from_numpy(cell.astype(np.float32)).to(dev)
skin = float(os.environ.get("SKIN", 0.5))


def run(d3_rc, steps=40):
    par = None if d3_rc is None else dict(s6=float(g["s6"]), s8=float(g["s8"]), a1=float(g["a1"]), a2=float(g["a2"]), cutoff=d3_rc,
                                          smoothing_fraction=0.2 * 15.0 / d3_rc)  # the same switching window in Angstrom

    def step(sync):
        return eng.eval(*args, cell=cell_t, forces=True, stress=True, coulomb="dsf", dsf_rc=15.0, dftd3=par, sync=sync)

    for _ in range(3):
        step(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    eng.set_profiling(2)
    for _ in range(5):
        step(True)
    torch.cuda.synchronize()
    fam = {k: v / 5 for k, v in eng.read_profile().items() if k != "evals"}
    eng.set_profiling(0)
    return ms, fam, int(eng.last_status[4])


ms0, fam0, _ = run(None)
ms1, fam1, rows1 = run(15.0)
ms2, fam2, rows2 = run(15.0 + skin)
build1, build2 = fam1["nlist"] - fam0["nlist"], fam2["nlist"] - fam0["nlist"]
out = {"atoms": len(z), "skin_A": skin, "ms_per_step": {"d3_off": ms0, "d3_rc15": ms1, "d3_rc15_plus_skin": ms2},
       "longest_d3_row": {"rc15": rows1, "rc15_plus_skin": rows2},
       "nlist_family_ms": {"d3_off": fam0["nlist"], "d3_rc15": fam1["nlist"], "d3_rc15_plus_skin": fam2["nlist"]},
       "d3_list_build_ms": {"rc15": build1, "rc15_plus_skin": build2},
       "wider_rows_cost_ms": ms2 - ms1 - (build2 - build1),
       "estimated_ms_per_step_with_reuse": {f"k={k}": ms2 - build2 * (1 - 1 / k) + 0.005 for k in (5, 10, 20)},
       "family_ms_d3_rc15": fam1}
print(json.dumps(out))
