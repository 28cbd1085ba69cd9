"""Where an ASE MD step of the 10 080-atom crystal spends its host time (cProfile over the get_forces loop of md_throughput.py)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ase import Atoms  # noqa: E402

from aimnetcentral_amd import AIMNet2Calculator, loader, workloads  # noqa: E402
from aimnetcentral_amd.aimnet2ase import AIMNet2ASE  # noqa: E402

calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
c, z, cell = workloads.glucose_supercell((7, 3, 5))
atoms = Atoms(numbers=z, positions=c, cell=cell, pbc=True)
atoms.calc = AIMNet2ASE(calc)
rng = np.random.default_rng(0)


def step():
    atoms.positions += rng.normal(scale=1e-4, size=atoms.positions.shape)
    return atoms.get_forces()


for _ in range(5):
    step()
t0 = time.perf_counter()
for _ in range(50):
    step()
print("ms per step: %.3f" % ((time.perf_counter() - t0) / 50 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
