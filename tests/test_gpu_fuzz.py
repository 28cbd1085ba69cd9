"""Randomised parity sweep: the HIP engine against the oracle on seeded random systems that mix everything the path
supports - flat batches of ragged molecules, periodic cells with random strain / vacancies / partial periodicity /
per-system cells, charges, simple / DSF Coulomb with random (cutoff, alpha), external DFT-D3 with its own cutoff, stress,
and the open-shell NSE family - at the reference's own gates (conftest.py).  The random geometries are "hot" (forces up to
~80 eV/A, activations up to 30 with the synthetic weights), so fp32 arithmetic leaves per-atom energy errors of ~5e-6 eV
that add up like a random walk over a molecule (tests/tools/fuzz_seed.py: every intermediate of the engine is as close
to the fp64 oracle as the fp32 oracle's, only the signs of the per-atom errors differ).  The energy gate is therefore taken
against the fp64 oracle and widened by three standard deviations of that walk, measured on the fp32 oracle's own per-atom
errors, plus the fp32 oracle's distance from fp64 - the engine may not be farther from the truth than the reference's
precision allows on top of the reference's gate."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import CHARGE_ATOL, STRESS_ATOL, assert_forces_close, energy_tol, golden
from aimnetcentral_amd import workloads
from oracle import aimnet2_oracle as O

pytestmark = pytest.mark.gpu


def _d3_tables():
    t = golden("dftd3_subset")
    g = golden("dftd3")
    return {k: t[k] for k in ("c6ab", "cn_ref", "rcov", "r4r2")}, dict(s6=float(g["s6"]), s8=float(g["s8"]), a1=float(g["a1"]), a2=float(g["a2"]))


def _periodic_case(rng):
    """The glucose cell (96 atoms), strained, jittered, with vacancies; some atoms moved out of the box by lattice vectors."""
    c, z, cell = workloads.glucose_cell()
    frac = c @ np.linalg.inv(cell)
    strain = np.eye(3) + rng.uniform(-0.04, 0.04, size=(3, 3))
    cell = cell @ strain
    keep = rng.random(len(z)) > rng.uniform(0.0, 0.25)
    keep[:4] = True
    c = (frac @ cell)[keep] + rng.normal(scale=0.03, size=(int(keep.sum()), 3))
    c = c + rng.integers(-1, 2, size=(len(c), 3)) @ cell
    return c.astype(np.float32), z[keep], cell.astype(np.float32)


def _compare(res, ref, ref64, mol, what, nse, energy=None):
    """`energy`: engine energies to hold against the fp64 oracle (periodic cases: those of a second evaluation that starts
    from the oracle's wrapped coordinates - see test_random_configuration); default res["energy"]."""
    sizes = np.bincount(mol)
    assert np.isfinite(res["energy"]).all(), what
    err = np.abs((res["energy"] if energy is None else energy) - ref64["energy"])
    d = (ref["_e_atom"][: len(mol)].astype(np.float64) - ref64["_e_atom"][: len(mol)]) ** 2
    walk = np.zeros(len(sizes))
    np.add.at(walk, mol, d)
    floor = np.abs(ref["energy"] - ref64["energy"]) + 3.0 * np.sqrt(walk)
    assert (err <= energy_tol(sizes) + floor).all(), f"{what}: energy |hip - fp64| {err.max():.3e}, fp32 floor {floor.max():.3e}"
    assert np.abs(res["charges"] - ref["charges"]).max() <= CHARGE_ATOL, what
    assert_forces_close(res["forces"], ref["forces"], what)
    if "stress" in res:
        assert np.abs(res["stress"] - ref["stress"]).max() <= STRESS_ATOL, what
    if nse:
        assert np.abs(res["spin_charges"] - ref["spin_charges"]).max() <= CHARGE_ATOL, what


import os  # noqa: E402

# AIMNET_FUZZ_SEEDS="lo:hi" widens the sweep for a soak run (default: the 32 seeds that are part of the suite).  The energy
# gate is statistical (reference gate + fp32 oracle's own distance + a 3-sigma random walk of its per-atom errors): in the
# 1 000-seed soak of round 4 one configuration exceeded it (seed 671: two periodic open-shell systems, 6.7e-5 eV from fp64 against
# a gate of 5.5e-5 - in EVERY GEMM mode, the exact-fp32 kernels by the most: 1.25 / 1.17 / 1.08 x the gate, tests/tools/
# fuzz_margins.py; round 3's soak had seed 535 over by 2.5 %); nothing else failed (profiles/r4_soak1000.txt).
_LO, _HI = (int(v) for v in os.environ.get("AIMNET_FUZZ_SEEDS", "0:32").split(":"))


def make_case(seed: int, dev=None):
    """Inputs of random configuration `seed`: (c, z, mol, q, mult, nse, engine kwargs, oracle kwargs, d3 options, label)."""
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    nse = bool(seed % 4 == 3)
    periodic = bool(seed % 2)
    kw, okw = {}, {}
    if periodic:
        two = bool(rng.random() < 0.35)
        c, z, cell = _periodic_case(rng)
        mol = np.zeros(len(z), dtype=np.int64)
        if two:
            c2, z2, cell2 = _periodic_case(rng)
            c, z = np.concatenate([c, c2]), np.concatenate([z, z2])
            mol = np.concatenate([mol, np.ones(len(z2), dtype=np.int64)])
            cell = np.stack([cell, cell2])
        pbc = (True, True, True) if (two or rng.random() < 0.6) else tuple(bool(b) for b in rng.permutation([True, True, False]))
        rc, alpha = float(rng.uniform(6.0, 11.0)), float(rng.uniform(0.15, 0.3))
        kw = dict(cell=cell, pbc=pbc, coulomb="dsf", dsf_rc=rc, dsf_alpha=alpha, stress=True)
        okw = dict(cell=cell, pbc=np.array(pbc), coulomb="dsf", dsf_rc=rc, dsf_alpha=alpha, stress=True)
    else:
        n_mol = int(rng.integers(1, 6))
        c, z, mol, _ = workloads.random_batch(n_mol, 3, 40, seed=int(rng.integers(1 << 30)))
        if rng.random() < 0.5:
            kw = dict(coulomb="simple")
            okw = dict(coulomb="simple")
        else:
            rc, alpha = float(rng.uniform(5.0, 12.0)), float(rng.uniform(0.15, 0.3))
            kw = dict(coulomb="dsf", dsf_rc=rc, dsf_alpha=alpha)
            okw = dict(coulomb="dsf", dsf_rc=rc, dsf_alpha=alpha)
    n_mol = int(mol.max()) + 1
    q = rng.integers(-1, 2, size=n_mol).astype(np.float32) if rng.random() < 0.6 else np.zeros(n_mol, dtype=np.float32)
    mult = (1.0 + rng.integers(0, 3, size=n_mol)).astype(np.float32)
    d3 = None
    if rng.random() < 0.4:
        tables, par = _d3_tables()
        d3_rc = kw.get("dsf_rc", 9.0) if rng.random() < 0.5 else float(rng.uniform(6.0, 10.0))
        d3 = dict(par, cutoff=float(d3_rc), smoothing_fraction=float(rng.uniform(0.1, 0.3)))
        okw["dftd3"] = dict(d3, **tables)
    if nse:
        okw["mult"] = mult
    label = (f"seed {seed}: {'pbc' + str(kw.get('pbc')) if periodic else 'molecules'} n={len(z)} n_mol={n_mol} {kw.get('coulomb')} "
             f"d3={d3 is not None} nse={nse} q={q.tolist()}")
    return c, z, mol, q, mult, nse, kw, okw, d3, label


def run_case(eng, case):
    c, z, mol, q, mult, nse, kw, okw, d3, label = case
    dev = eng.device
    if d3 is not None:
        eng.set_dftd3_tables(_d3_tables()[0])
    charge_t = np.stack([0.5 * q + 0.5 * (mult - 1.0), 0.5 * q - 0.5 * (mult - 1.0)], axis=-1).astype(np.float32) if nse else q
    ekw = dict(kw)
    if "cell" in ekw:
        ekw["cell"] = torch.from_numpy(ekw["cell"]).to(dev)
    r = eng.eval(torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.from_numpy(mol).to(dev), torch.from_numpy(charge_t).to(dev),
                 forces=True, dftd3=d3, **ekw)
    return {k: v.cpu().numpy() for k, v in r.items()}


@pytest.mark.parametrize("seed", range(_LO, _HI))
def test_random_configuration(seed, hip_engine, hip_engine_nse, oracle32, oracle32_nse, oracle64, oracle64_nse):
    case = make_case(seed)
    c, z, mol, q, mult, nse, kw, okw, d3, label = case
    eng, orc, orc64 = (hip_engine_nse, oracle32_nse, oracle64_nse) if nse else (hip_engine, oracle32, oracle64)
    res = run_case(eng, case)
    ref = O.evaluate(orc, c, z, q, mol, return_intermediates=True, **okw)
    ref64 = O.evaluate(orc64, c, z, q, mol, return_intermediates=True, **dict(okw, forces=False, stress=False))
    energy = None
    if "cell" in kw:
        # Wrapping atoms into the cell in fp32 is not unique to the last bit: the engine's wrapped coordinates differ from
        # the oracle's (and the reference's) by up to 2e-6 A, which these hot systems (|F| ~ 20 eV/A) turn into ~1e-4 eV
        # (tests/tools/fuzz_seed.py).  The energy gate is therefore applied to an evaluation that starts from the oracle's
        # wrapped coordinates; the raw-coordinate evaluation must agree with it within that sensitivity, 4 |F|_2 x 2e-6 A,
        # and passes the force / charge / stress gates (relative, hence insensitive to it) as it is.
        wrapped = run_case(eng, (ref["coord_wrapped"].astype(np.float32),) + case[1:])
        energy = wrapped["energy"]
        f2 = np.zeros(len(energy))
        np.add.at(f2, mol, (ref["forces"].astype(np.float64) ** 2).sum(-1))
        assert (np.abs(res["energy"] - energy) <= 4.0 * np.sqrt(f2) * 2e-6 + energy_tol(np.bincount(mol))).all(), label
    _compare(res, ref, ref64, mol, label, nse, energy)


def tiny_cells(seed: int):
    """(coord, numbers, mol_idx, cell, n_sys, rng) of test_tiny_cells_many_images."""
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    n_sys = 1 if seed % 2 == 0 else 3
    coords, zs, mols, cells = [], [], [], []
    for m in range(n_sys):
        cell = np.diag(rng.uniform(3.0, 4.5, size=3)) + rng.uniform(-0.5, 0.5, size=(3, 3))
        n = int(rng.integers(2, 10))
        pts = []
        while len(pts) < n:  # random fractional positions at least 0.95 A apart under the minimum-image convention
            f = rng.random(3)
            ok = True
            for g in pts:
                df = f - g
                df -= np.round(df)
                best = min(np.linalg.norm((df + np.array(im) - 1) @ cell) for im in np.ndindex(3, 3, 3))
                if best < 0.95:
                    ok = False
                    break
            if ok:
                pts.append(f)
        coords.append(np.array(pts) @ cell)
        zs.append(rng.choice([1, 6, 7, 8], size=n, p=[0.4, 0.3, 0.15, 0.15]))
        mols.append(np.full(n, m))
        cells.append(cell)
    c = np.concatenate(coords).astype(np.float32)
    z = np.concatenate(zs).astype(np.int64)
    mol = np.concatenate(mols).astype(np.int64)
    cell = (cells[0] if n_sys == 1 else np.stack(cells)).astype(np.float32)
    return c, z, mol, cell, n_sys, rng


@pytest.mark.parametrize("seed", range(6))
def test_tiny_cells_many_images(seed, hip_engine, oracle32, oracle64):
    """Cells much smaller than the cutoffs (3-4.5 A edges, 2-9 atoms, triclinic): every atom sees many periodic images of
    every other atom and of itself (lattice shifts up to +-4 at a 9 A DSF cutoff), plus many single-atom / few-atom
    neighbour rows.  Also a batch of such cells with per-system cell matrices."""
    c, z, mol, cell, n_sys, rng = tiny_cells(seed)
    rc, alpha = float(rng.uniform(7.0, 9.0)), 0.25
    dev = hip_engine.device
    r = hip_engine.eval(torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.from_numpy(mol).to(dev),
                        torch.zeros(n_sys, device=dev), cell=torch.from_numpy(cell).to(dev), forces=True, stress=True, coulomb="dsf",
                        dsf_rc=rc, dsf_alpha=alpha)
    res = {k: v.cpu().numpy() for k, v in r.items()}
    okw = dict(cell=cell, coulomb="dsf", dsf_rc=rc, dsf_alpha=alpha, stress=True)
    ref = O.evaluate(oracle32, c, z, np.zeros(n_sys, np.float32), mol, return_intermediates=True, **okw)
    ref64 = O.evaluate(oracle64, c, z, np.zeros(n_sys, np.float32), mol, return_intermediates=True, **dict(okw, forces=False, stress=False))
    _compare(res, ref, ref64, mol, f"tiny cells seed {seed}: n={len(z)} systems={n_sys} rc={rc:.2f}", False)


@pytest.mark.parametrize("case", ["many_tiny", "one_big_many_small"])
def test_extreme_batch_shapes(case, hip_engine, oracle32, oracle64):
    """Ragged extremes of the flat layout: 1 500 molecules of 1-3 atoms (single atoms have empty neighbour rows), and one
    700-atom molecule followed by 200 small ones (per-molecule reductions of very different lengths in one launch)."""
    rng = np.random.Generator(np.random.PCG64(99 if case == "many_tiny" else 98))
    coords, zs, mols = [], [], []
    sizes = rng.integers(1, 4, size=1500).tolist() if case == "many_tiny" else [700] + rng.integers(2, 9, size=200).tolist()
    for m, n in enumerate(sizes):
        cm, zm = workloads.random_organic(int(n), rng)
        coords.append(cm)
        zs.append(zm)
        mols.append(np.full(int(n), m))
    c = np.concatenate(coords).astype(np.float32)
    z = np.concatenate(zs).astype(np.int64)
    mol = np.concatenate(mols).astype(np.int64)
    q = rng.integers(-1, 2, size=len(sizes)).astype(np.float32)
    dev = hip_engine.device
    r = hip_engine.eval(torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.from_numpy(mol).to(dev), torch.from_numpy(q).to(dev),
                        forces=True, coulomb="simple")
    res = {k: v.cpu().numpy() for k, v in r.items()}
    ref = O.evaluate(oracle32, c, z, q, mol, return_intermediates=True)
    ref64 = O.evaluate(oracle64, c, z, q, mol, forces=False, return_intermediates=True)
    _compare(res, ref, ref64, mol, f"{case}: n={len(z)} n_mol={len(sizes)}", False)
    # (per-molecule charge conservation is NOT exact for 1-3 atom molecules: ops.nse divides by sum f + 1e-6, and the
    # reference leaves the same residue - the charges above are compared with the oracle's, atom by atom)
