"""Step time against system size (glucose supercells, DSF 15 A, E + F + stress): per-family ms from the engine's profile marks.
GPU box.  usage: python tests/tools/size_sweep.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from aimnetcentral_amd import loader, workloads  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402


def main():
    eng = HipEngine(loader.synthetic_spec(0, cold=True), device="cuda:0")
    dev = eng.device
    for rep in [(1, 1, 1), (2, 1, 1), (2, 2, 1), (2, 2, 2), (4, 2, 2), (4, 2, 3), (4, 3, 4), (7, 3, 5)]:
        c, z, cell = workloads.glucose_supercell(rep)
        rng = np.random.default_rng(0)
        c = (c + rng.normal(0, 0.02, c.shape)).astype(np.float32)
        n = len(z)
        args = (torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(1, device=dev))
        kw = dict(cell=torch.from_numpy(cell.astype(np.float32)).to(dev), forces=True, stress=True, coulomb="dsf")
        for _ in range(3):
            eng.eval(*args, **kw)
        torch.cuda.synchronize()
        steps = 50
        t = time.perf_counter()
        for _ in range(steps):
            eng.eval(*args, sync=False, **kw)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / steps * 1e3
        eng.set_profiling(2)
        eng.read_profile()
        for _ in range(10):
            eng.eval(*args, **kw)
        prof = eng.read_profile()
        eng.set_profiling(0)
        ev = max(1.0, prof.get("evals", 1.0))
        fam = {k: round(v / ev, 4) for k, v in prof.items() if k != "evals"}
        print(json.dumps({"atoms": n, "ms_per_step": round(ms, 4), "atoms_steps_per_s": round(n / ms * 1e3), "family_ms": fam}), flush=True)


main()
