"""Verlet-skin reuse of the neighbour matrices across MD steps (aimnetcentral_amd/verlet.py; the
reference's static-geometry counterpart is StaticInputCache, aimnet/calculators/neighbors.py:150-250): along a random walk the
evaluations through kept matrices (cutoff + skin, pairs cut at the true cutoff by the kernels) agree with evaluations that
rebuild every list, at the reference's gates; the matrices are rebuilt exactly when an atom has left the skin."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import CHARGE_ATOL, STRESS_ATOL, elementwise_violations, energy_tol, golden

pytestmark = pytest.mark.gpu


def _walk(vl, eng, c0, z, mol, q, cell, steps, step_sigma, seed, **kw):
    from aimnetcentral_amd.engine import HipEngine  # noqa: F401

    dev = eng.device
    rng = np.random.default_rng(seed)
    zt, mt, qt = (torch.as_tensor(a, device=dev) for a in (z, mol, q))
    ct = None if cell is None else torch.as_tensor(cell, dtype=torch.float32, device=dev)
    c = np.array(c0, dtype=np.float64)
    worst = dict(dE=0.0, viol=0, ratio=0.0, dq=0.0, ds=0.0)
    for _ in range(steps):
        x = torch.as_tensor(c.astype(np.float32), device=dev)
        a = vl.eval(x, zt, mt, qt, cell=ct, forces=True, **kw)
        b = eng.eval(x, zt, mt, qt, cell=ct, forces=True, **kw)
        n_per = np.bincount(np.asarray(mol), minlength=len(np.atleast_1d(q)))
        worst["dE"] = max(worst["dE"], float((a["energy"] - b["energy"]).abs().max() / energy_tol(n_per)))
        v, _, r = elementwise_violations(a["forces"].cpu().numpy(), b["forces"].cpu().numpy())
        worst["viol"] += v
        worst["ratio"] = max(worst["ratio"], r)
        worst["dq"] = max(worst["dq"], float((a["charges"] - b["charges"]).abs().max()))
        if "stress" in a:
            worst["ds"] = max(worst["ds"], float((a["stress"] - b["stress"]).abs().max()))
        c = c + rng.normal(0.0, step_sigma, c.shape)
    return worst


def test_periodic_dsf_with_stress_along_a_walk(hip_engine_cold):
    from aimnetcentral_amd import workloads
    from aimnetcentral_amd.verlet import VerletSkinLists

    c, z, cell = workloads.glucose_supercell((2, 1, 1))
    c = c + np.array([3.0, -20.0, 7.5])  # atoms start outside the cell: the build-time wrap offsets matter
    vl = VerletSkinLists(hip_engine_cold, skin=0.6)
    w = _walk(vl, hip_engine_cold, c, z, np.zeros(len(z), np.int64), np.zeros(1, np.float32), cell, 12, 0.03, 1, coulomb="dsf",
              dsf_rc=9.0, stress=True)
    # (atoms that start outside the cell: the kept matrices see them in the build-time wrap frame, the fresh evaluation re-wraps them
    # on the device in fp32 - positions that differ by an ulp; over 12 steps x 576 force components at most a couple may touch the
    # literal gate, none beyond 1.5 x)
    assert w["dE"] <= 1.0 and w["viol"] <= 2 and w["ratio"] <= 1.5 and w["dq"] <= CHARGE_ATOL and w["ds"] <= STRESS_ATOL, w
    assert vl.builds + vl.reuses == 12 and 1 <= vl.builds <= 6 and vl.reuses >= 6, (vl.builds, vl.reuses)


def test_molecule_batch_simple_coulomb_and_forced_rebuild(hip_engine_cold):
    from aimnetcentral_amd import workloads
    from aimnetcentral_amd.verlet import VerletSkinLists

    c, z, mol, q = workloads.random_batch(12, 20, 40, seed=4)
    vl = VerletSkinLists(hip_engine_cold, skin=0.5)
    w = _walk(vl, hip_engine_cold, c, z, mol, q, None, 8, 0.02, 2, coulomb="simple")
    assert w["dE"] <= 1.0 and w["viol"] == 0 and w["dq"] <= CHARGE_ATOL, w
    assert vl.builds < 8
    # one atom jumps by more than skin / 2: the very next evaluation rebuilds
    dev = hip_engine_cold.device
    b0 = vl.builds
    x = torch.as_tensor(c, device=dev)
    args = (torch.as_tensor(z, device=dev), torch.as_tensor(mol, device=dev), torch.as_tensor(q, device=dev))
    vl.eval(x, *args, forces=True)
    b1 = vl.builds
    x2 = x.clone()
    x2[5, 0] += 0.3
    a = vl.eval(x2, *args, forces=True)
    assert vl.builds == b1 + 1 and b1 >= b0
    b = hip_engine_cold.eval(x2, *args, forces=True)
    assert elementwise_violations(a["forces"].cpu().numpy(), b["forces"].cpu().numpy())[0] == 0


def test_dftd3_and_dsf_share_one_kept_matrix(hip_engine):
    from aimnetcentral_amd.verlet import VerletSkinLists

    g, t = golden("dftd3"), golden("dftd3_subset")
    hip_engine.set_dftd3_tables({k: t[k] for k in ("c6ab", "cn_ref", "rcov", "r4r2")})
    rc = float(g["pbc_cutoff"])
    par = dict(s6=float(g["s6"]), s8=float(g["s8"]), a1=float(g["a1"]), a2=float(g["a2"]), cutoff=rc, smoothing_fraction=0.2)
    vl = VerletSkinLists(hip_engine, skin=0.5)
    for kw in (dict(coulomb="dsf", dsf_rc=rc, dftd3=par), dict(coulomb="dsf", dsf_rc=rc - 2.0, dftd3=par)):
        w = _walk(vl, hip_engine, g["pbc_coord"], g["pbc_numbers"], np.zeros(96, np.int64), np.zeros(1, np.float32), g["pbc_cell"], 5,
                  0.02, 3, **kw)
        # hot weights: two fp32 evaluations in different pair orders (the engine's own lists are bin-ordered, imported ones are not)
        assert w["dE"] <= 3.0 and w["ratio"] <= 10.0 and w["dq"] <= CHARGE_ATOL, (kw, w)
    assert vl.builds >= 2 and vl.reuses >= 4


def test_deferred_mode_flags_an_atom_that_left_the_skin(hip_engine_cold):
    from aimnetcentral_amd import workloads
    from aimnetcentral_amd.engine import NeighborOverflowError
    from aimnetcentral_amd.verlet import VerletSkinLists

    c, z, mol, q = workloads.random_batch(6, 20, 30, seed=7)
    dev = hip_engine_cold.device
    args = (torch.as_tensor(z, device=dev), torch.as_tensor(mol, device=dev), torch.as_tensor(q, device=dev))
    vl = VerletSkinLists(hip_engine_cold, skin=0.4, rebuild_every=50)
    x = torch.as_tensor(c, device=dev)
    for k in range(4):
        vl.eval(x + 0.01 * k, *args, forces=True, sync=False, defer=True)
    vl.check_deferred()  # all inside the skin: passes, one build
    assert vl.builds == 1 and vl.reuses == 3
    vl.eval(x + 0.5, *args, forces=True, sync=False, defer=True)
    with pytest.raises(NeighborOverflowError):
        vl.check_deferred()
    vl.eval(x + 0.5, *args, forces=True, sync=False, defer=True)  # invalidated: rebuilt
    vl.check_deferred()
    assert vl.builds == 2
