"""Non-periodic DSF Coulomb (aimnet/modules/lr.py:559-615) on systems large enough for the engine's bounding-box cell grid
(>= 1 500 atoms per molecule): the list-free walk over that grid (default, engine option `dsf_np_walk`) against the
neighbour-matrix form it replaces - same energies, forces and charges at the reference's gates, on one big cluster, on a batch of
two, and next to a DFT-D3 matrix with a cutoff of its own; below the size threshold nothing changes."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import CHARGE_ATOL, elementwise_violations, energy_tol, golden

pytestmark = pytest.mark.gpu


def _cluster(reps, seed):
    from aimnetcentral_amd import workloads

    c, z, _ = workloads.glucose_supercell(reps)
    c = c + np.random.default_rng(seed).normal(0.0, 0.02, c.shape)
    return c.astype(np.float32), z


def _both(eng, c, z, mol, n_mol, **kw):
    dev = eng.device
    args = (torch.as_tensor(c, device=dev), torch.as_tensor(z, device=dev), torch.as_tensor(mol, device=dev), torch.zeros(n_mol, device=dev))
    out = {}
    for mode in (1, 0):
        eng.set_option("dsf_np_walk", mode)
        try:
            out[mode] = {k: v.cpu().numpy() for k, v in eng.eval(*args, forces=True, coulomb="dsf", **kw).items()}
        finally:
            eng.set_option("dsf_np_walk", 1)
    return out[1], out[0]


def _same(a, b, n_per_mol):
    assert (np.abs(a["energy"] - b["energy"]) <= energy_tol(n_per_mol)).all(), np.abs(a["energy"] - b["energy"])
    v, _, worst = elementwise_violations(a["forces"], b["forces"])
    assert v == 0, (v, worst)
    assert np.abs(a["charges"] - b["charges"]).max() <= CHARGE_ATOL


def test_one_large_cluster_and_a_batch_of_two(hip_engine_cold):
    c, z = _cluster((2, 3, 4), 1)  # 2 304 atoms, no cell: a finite crystallite
    a, b = _both(hip_engine_cold, c, z, np.zeros(len(z), np.int32), 1)
    _same(a, b, len(z))
    assert abs(a["energy"][0]) > 1.0 and not np.array_equal(a["forces"], b["forces"])  # (two different summation orders did run)
    c2, z2 = _cluster((4, 2, 2), 2)  # 1 536 atoms each
    cc = np.concatenate([c2, c2[::-1] + np.array([100.0, 0.0, 0.0], np.float32)])
    zz = np.concatenate([z2, z2[::-1]])
    mol = np.repeat(np.arange(2, dtype=np.int32), len(z2))
    a, b = _both(hip_engine_cold, cc, zz, mol, 2, dsf_rc=12.0)
    _same(a, b, len(z2))
    assert abs(a["energy"][0] - a["energy"][1]) < 1e-3  # the same cluster twice (atom order reversed, translated)


def test_with_a_dftd3_matrix_of_its_own_cutoff(hip_engine):
    g, t = golden("dftd3"), golden("dftd3_subset")
    hip_engine.set_dftd3_tables({k: t[k] for k in ("c6ab", "cn_ref", "rcov", "r4r2")})
    par = dict(s6=float(g["s6"]), s8=float(g["s8"]), a1=float(g["a1"]), a2=float(g["a2"]), cutoff=10.0, smoothing_fraction=0.2)
    c, z = _cluster((4, 2, 2), 3)
    a, b = _both(hip_engine, c, z, np.zeros(len(z), np.int32), 1, dsf_rc=13.0, dftd3=par)
    # hot weights: two fp32 summation orders of the Coulomb sums
    assert abs(a["energy"][0] - b["energy"][0]) <= 5e-3 and np.abs(a["forces"] - b["forces"]).max() <= 1e-4 * max(1.0, np.abs(b["forces"]).max())
    assert np.abs(a["charges"] - b["charges"]).max() <= CHARGE_ATOL


def test_small_molecules_keep_the_matrix_form(hip_engine_cold):
    from aimnetcentral_amd import workloads

    c, z, mol, q = workloads.random_batch(4, 30, 40, seed=9)
    a, b = _both(hip_engine_cold, c, z, mol.astype(np.int32), 4, dsf_rc=9.0)
    for k in a:
        assert np.array_equal(a[k], b[k]), k  # below the threshold the switch changes nothing: bitwise
