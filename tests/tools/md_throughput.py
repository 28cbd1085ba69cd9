#!/usr/bin/env python
"""MD-loop throughput THROUGH the adapters, PCIe included: steps/s of
  * AIMNet2ASE.calculate on taxol (113 atoms) and on the 10 080-atom crystal (E+F[+stress]; positions change on the host every step,
    one pinned D2H copy brings status + every output back), and
  * AIMNet2TorchSim.forward on the config-5 shard (128 frames x 50 atoms) with a device-resident velocity-Verlet-like update,
    status read every step (K = 1, the reference's behaviour) vs verified every K = 25 steps (no host read in between)."""
import json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from aimnetcentral_amd import AIMNet2Calculator, AIMNet2TorchSim, loader, workloads
from aimnetcentral_amd.aimnet2ase import AIMNet2ASE


class Atoms:
    def __init__(self, numbers, positions, cell=None, pbc=(False, False, False)):
        self.numbers, self.positions = np.asarray(numbers), np.asarray(positions, dtype=float)
        self.cell, self.pbc, self.info = cell, np.asarray(pbc), {}

    def copy(self):
        return Atoms(self.numbers.copy(), self.positions.copy(), None if self.cell is None else np.array(self.cell), self.pbc.copy())

    def __len__(self):
        return len(self.numbers)


class State:
    def __init__(self, positions, numbers, system_idx, n_systems):
        self.positions, self.atomic_numbers, self.system_idx = positions, numbers, system_idx
        self.row_vector_cell, self.pbc, self.n_systems = torch.zeros(n_systems, 3, 3, device=positions.device), False, n_systems
        self.device, self.dtype = positions.device, torch.float32
        self.charge = torch.zeros(n_systems, device=positions.device)


def ase_loop(calc, atoms, props, steps):
    ase = AIMNet2ASE(calc, charge=0)
    rng = np.random.default_rng(0)
    x0 = atoms.positions.copy()
    # eight pre-drawn displaced frames, cycled: drawing 3 N normal deviates per step costs 0.4 ms at 10 080 atoms - the harness, not
    # the adapter (an integrator updates the positions in place for a few microseconds)
    frames = [x0 + rng.normal(scale=0.005, size=x0.shape) for _ in range(8)]
    for k in range(steps + 5):
        if k == 5:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        atoms.positions = frames[k % 8]
        ase.reset()
        ase.calculate(atoms, properties=props)
        f = ase.results["forces"]
    dt = time.perf_counter() - t0
    assert np.isfinite(f).all()
    return steps / dt


def torchsim_loop(calc, every, steps):
    dev = torch.device("cuda:0")
    c, z, mol, q = workloads.random_batch(128, 50, 50, seed=5)
    st = State(torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.from_numpy(mol).to(dev), 128)
    model = AIMNet2TorchSim(calc, compute_forces=True, status_check_every=every)
    x0 = st.positions.clone()
    for k in range(steps + 5):
        if k == 5:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        out = model(st)
        st.positions = x0 + 1e-6 * out["forces"].clamp(-100, 100)  # device-resident "integrator"
    if every > 1:
        calc.check_status()
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)


def main():
    calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "golden", "taxol.npz"))
    res = {}
    res["ase_taxol_steps_per_s"] = ase_loop(calc, Atoms(g["numbers"], g["coord"]), ["energy", "forces"], 300)
    c, z, cell = workloads.glucose_supercell((7, 3, 5))
    calc.set_lrcoulomb_method("dsf", cutoff=15.0, dsf_alpha=0.2)
    res["ase_pbc10k_steps_per_s"] = ase_loop(calc, Atoms(z, c, cell=cell, pbc=(True, True, True)), ["energy", "forces", "stress"], 60)
    calc.set_lrcoulomb_method("simple")
    res["torchsim_md128x50_steps_per_s_check_every_1"] = torchsim_loop(calc, 1, 200)
    res["torchsim_md128x50_steps_per_s_check_every_25"] = torchsim_loop(calc, 25, 200)
    res["atoms"] = {"taxol": 113, "pbc10k": int(len(z)), "md128x50": 6400}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
