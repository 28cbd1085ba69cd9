"""Single-launch preparation (engine option "prep_fused", nlist.hip prep_small_kernel) against the separate kernels, per system size:
periodic glucose supercells, DSF, forces + stress, same process, interleaved A B B A.  GPU box: python tests/tools/prep_ab.py"""
import time

import numpy as np
import torch

import os, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from aimnetcentral_amd import loader, workloads  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402

eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
dev = eng.device
for reps in ((1, 1, 1), (2, 2, 2), (3, 2, 3), (3, 3, 4), (4, 4, 4), (7, 3, 5)):
    c, z, cell = workloads.glucose_supercell(reps)
    args = (torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev), torch.zeros(len(z), dtype=torch.int64, device=dev),
            torch.zeros(1, device=dev))
    cl = torch.from_numpy(cell.astype(np.float32)).to(dev)
    t = {0: [], 1: []}
    for v in (1, 0, 0, 1, 1, 0, 0, 1):
        eng.set_option("prep_fused", v)
        for _ in range(20):
            eng.eval(*args, cell=cl, forces=True, stress=True, coulomb="dsf", dsf_rc=9.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            eng.eval(*args, cell=cl, forces=True, stress=True, coulomb="dsf", dsf_rc=9.0)
        torch.cuda.synchronize()
        t[v].append((time.perf_counter() - t0) / 200 * 1e3)
    a, b = np.mean(t[1]), np.mean(t[0])
    print(f"{len(z):6d} atoms: fused {a:.4f} ms  separate {b:.4f} ms  ({100 * (a / b - 1):+.2f} %)")

# molecules (no cell): status memset + molecule offsets + coordinate copy as one launch
g_c, g_z = workloads.random_organic(113, np.random.default_rng(1))
for name, (c, z, mol, q) in (("113-atom molecule", (g_c.astype(np.float32), g_z, np.zeros(113, dtype=np.int64), np.zeros(1, np.float32))),
                             ("64 molecules of 20-60 atoms", workloads.random_batch(64, 20, 60, seed=2))):
    args = (torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.from_numpy(mol).to(dev), torch.from_numpy(q).to(dev))
    t = {0: [], 1: []}
    for v in (1, 0, 0, 1, 1, 0, 0, 1):
        eng.set_option("prep_fused", v)
        for _ in range(20):
            eng.eval(*args, forces=True, coulomb="simple")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            eng.eval(*args, forces=True, coulomb="simple")
        torch.cuda.synchronize()
        t[v].append((time.perf_counter() - t0) / 200 * 1e3)
    a, b = np.mean(t[1]), np.mean(t[0])
    print(f"{name} ({len(z)} atoms): fused {a:.4f} ms  separate {b:.4f} ms  ({100 * (a / b - 1):+.2f} %)")
