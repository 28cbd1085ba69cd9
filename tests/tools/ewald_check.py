"""Ewald summation (csrc/ewald.hip) at size: parity against the fp64 oracle on the 2 304-atom periodic sample, and the cost of the
method against DSF on config 3's 10 080-atom crystal (forces + stress).  GPU box: python tests/tools/ewald_check.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from aimnetcentral_amd import loader, synth, workloads  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402
from oracle import aimnet2_oracle as O  # noqa: E402

eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
dev = eng.device


def run(c, z, cell, **kw):
    a = (torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.zeros(len(z), dtype=torch.int64, device=dev), torch.zeros(1, device=dev))
    return lambda: eng.eval(*a, cell=torch.from_numpy(cell).to(dev), forces=True, stress=True, **kw)


c, z, cell = workloads.glucose_supercell((2, 3, 4))
rng = np.random.default_rng(7)
c = (c + rng.normal(0.0, 0.02, c.shape)).astype(np.float32)
cell = cell.astype(np.float32)
res = {k: v.cpu().numpy() for k, v in run(c, z, cell, coulomb="ewald")().items()}
t0 = time.time()
ref = O.evaluate(O.OracleModel(synth.synthetic_state_dict(0), torch.float64), c, z, np.zeros(1, np.float32), cell=cell, coulomb="ewald", stress=True)
print(f"2 304 atoms vs the fp64 oracle ({time.time() - t0:.1f} s): dE {res['energy'][0] - ref['energy'][0]:+.3e} eV  "
      f"max|dF| {np.abs(res['forces'] - ref['forces']).max():.3e} eV/A (max|F| {np.abs(ref['forces']).max():.2f})  "
      f"max|dstress| {np.abs(res['stress'] - ref['stress']).max():.3e}  k entries {int(eng.last_status[7])}")

c, z, cell = workloads.glucose_supercell((7, 3, 5))
c, cell = c.astype(np.float32), cell.astype(np.float32)
for name, kw in (("dsf 15 A", dict(coulomb="dsf", dsf_rc=15.0)), ("ewald 1e-6", dict(coulomb="ewald")), ("ewald 1e-8", dict(coulomb="ewald", ewald_accuracy=1e-8))):
    f = run(c, z, cell, **kw)
    for _ in range(5):
        r = f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        r = f()
    torch.cuda.synchronize()
    print(f"10 080 atoms {name:12s}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step  E = {float(r['energy'][0]):.4f} eV  k entries {int(eng.last_status[7])}")
