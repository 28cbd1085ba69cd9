#!/usr/bin/env python
"""Step time with kept neighbour matrices (aimnetcentral_amd/verlet.py) against the engine's per-step rebuild (GPU box): the
10 080-atom crystal with DSF (the engine walks the grid - the kept form has to read a 15 A matrix), without the Coulomb term
(only the 5 A list is at stake) and a batch of 256 molecules.  Deferred mode on both sides (no host read per step), atoms on a
random walk of 0.003 A per step.  Prints one JSON line per case."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import loader, workloads  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402
from aimnetcentral_amd.verlet import VerletSkinLists  # noqa: E402

dev = torch.device("cuda:0")
eng = HipEngine(loader.synthetic_spec(0), dev)
STEPS = int(os.environ.get("STEPS", 60))


def bench(tag, c, z, mol, q, cell, **kw):
    zt, mt, qt = (torch.as_tensor(a, device=dev) for a in (z, mol, q))
    ct = None if cell is None else torch.as_tensor(cell, dtype=torch.float32, device=dev)
    x0 = torch.as_tensor(c, dtype=torch.float32, device=dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    steps = [x0 + 0.003 * np.sqrt(k) * torch.randn(x0.shape, device=dev, generator=gen) for k in range(STEPS)]
    vl = VerletSkinLists(eng, skin=0.5, rebuild_every=20)
    out = {"case": tag, "atoms": int(x0.shape[0]), "steps": STEPS, "skin_A": 0.5}
    for name, fn, chk in (("rebuild_every_step", eng.eval, eng.check_deferred), ("kept_matrices", vl.eval, vl.check_deferred)):
        for x in steps[:5]:
            fn(x, zt, mt, qt, cell=ct, forces=True, sync=False, defer=True, **kw)
        chk()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for x in steps:
            fn(x, zt, mt, qt, cell=ct, forces=True, sync=False, defer=True, **kw)
        torch.cuda.synchronize()
        out[name + "_ms"] = (time.perf_counter() - t0) / STEPS * 1e3
        chk()
    out["builds"], out["reuses"] = vl.builds, vl.reuses
    out["lr_width"] = int(vl._lists["nbmat_lr"].shape[1]) if "nbmat_lr" in vl._lists else 0
    print(json.dumps(out), flush=True)


c, z, cell = workloads.glucose_supercell((7, 3, 5))
mol, q = np.zeros(len(z), np.int64), np.zeros(1, np.float32)
bench("config3_dsf15", c, z, mol, q, cell, coulomb="dsf", dsf_rc=15.0)
bench("config3_no_coulomb", c, z, mol, q, cell, coulomb="none")
cb, zb, mb, qb = workloads.random_batch(256, 20, 60, seed=0)
bench("config2_batch256_simple", cb, zb, mb, qb, None, coulomb="simple")
