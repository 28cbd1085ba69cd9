"""Where a step of the ASE adapter spends its host time (cProfile over the loop of md_throughput.py; Atoms stand-in)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
from md_throughput import Atoms  # noqa: E402

from aimnetcentral_amd import AIMNet2Calculator, loader, workloads  # noqa: E402
from aimnetcentral_amd.aimnet2ase import AIMNet2ASE  # noqa: E402

calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "taxol.npz"))
c, z, cell = workloads.glucose_supercell((7, 3, 5))
for name, atoms, props in (("taxol", Atoms(g["numbers"], g["coord"]), ["energy", "forces"]),
                           ("pbc10k", Atoms(z, c, cell=cell, pbc=(True, True, True)), ["energy", "forces", "stress"])):
    if name == "pbc10k":
        calc.set_lrcoulomb_method("dsf", cutoff=15.0, dsf_alpha=0.2)
    ase = AIMNet2ASE(calc, charge=0)
    rng = np.random.default_rng(0)
    frames = [atoms.positions + rng.normal(scale=0.005, size=atoms.positions.shape) for _ in range(8)]

    def step(k):
        atoms.positions = frames[k % 8]
        ase.reset()
        ase.calculate(atoms, properties=props)

    for k in range(10):
        step(k)
    n = 200 if name == "taxol" else 60
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        step(k)
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per step")
    pr = cProfile.Profile()
    pr.enable()
    for k in range(n):
        step(k)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
