"""100-seed sweep of the analytic Hessian-vector products against the fp64 specification (the generator and the gate of
tests/test_gpu_hvp.py::test_hvp_fuzz_random_molecules).  python tests/tools/hvp_soak.py [n_seeds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from aimnetcentral_amd import loader, synth, workloads  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402
from oracle import aimnet2_oracle as O  # noqa: E402
from test_gpu_hvp import _spec_and_engine  # noqa: E402

eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
om = O.OracleModel(synth.synthetic_state_dict(0), torch.float64)
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
worst, fails = 0.0, 0
for seed in range(n_seeds):
    rng = np.random.default_rng(1000 + seed)
    n_mol = 1 + seed % 3
    cs, zs, ms = [], [], []
    for m in range(n_mol):
        c, z = workloads.random_organic(int(rng.integers(8, 48)), rng)
        cs.append(c); zs.append(z); ms.append(np.full(len(z), m))
    coord, numbers, mol = np.concatenate(cs).astype(np.float32), np.concatenate(zs), np.concatenate(ms)
    charge = rng.integers(-1, 2, size=n_mol).astype(np.float32)
    V = rng.standard_normal((2, len(numbers), 3)).astype(np.float32)
    spec, hv, f = _spec_and_engine(eng, om, coord, numbers, charge, mol, V)
    err, top = np.abs(hv - spec["hv"]).max(), np.abs(spec["hv"]).max()
    rel = err / (1e-5 + 3e-5 * top)
    worst = max(worst, rel)
    fails += rel > 1.0
print(f"{n_seeds} seeds: {fails} outside the gate 1e-5 + 3e-5 max|Hv|; worst error / gate = {worst:.3f}")
