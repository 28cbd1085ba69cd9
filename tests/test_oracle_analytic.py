"""The hand-derived analytic backward (the specification of the HIP kernels) against the autograd
oracle, in fp64 where both must agree to round-off."""
from __future__ import annotations

import numpy as np
import pytest

from conftest import golden
from oracle import aimnet2_analytic as AN
from oracle import aimnet2_oracle as O


@pytest.mark.parametrize("name,kw", [
    ("batch5", {}),
    ("pbc96_dsf8_wrapped", {"coulomb": "dsf", "dsf_rc": 8.0, "dsf_alpha": 0.25}),
    ("pbc2x96_dsf9", {"coulomb": "dsf", "dsf_rc": 9.0, "dsf_alpha": 0.2}),
])
def test_analytic_matches_autograd_fp64(oracle64, name, kw):
    g = golden(name)
    mol = g["mol_idx"] if "mol_idx" in g.files else np.zeros(len(g["numbers"]), dtype=np.int64)
    cell = g["cell"] if "cell" in g.files else None
    ref = O.evaluate(oracle64, g["coord"], g["numbers"], g["charge"], mol, cell=cell, stress=cell is not None,
                     return_intermediates=True, **kw)
    xw = ref["coord_wrapped"]
    if cell is None:
        nbl, shl = O.neighbor_list(xw, float("inf"), mol)
        coul = "simple"
    else:
        nbl, shl = O.neighbor_list(xw, kw["dsf_rc"], mol, cell, np.ones(3, bool))
        coul = "dsf"
    a = AN.evaluate(oracle64, xw, g["numbers"], g["charge"], mol, ref["nbmat"], ref.get("shifts"), cell, coulomb=coul,
                    nbmat_lr=nbl, shifts_lr=shl, stress=cell is not None,
                    **{k: v for k, v in kw.items() if k != "coulomb"})
    assert np.abs(a["energy"] - ref["energy"]).max() < 1e-9
    assert np.abs(a["charges"] - ref["charges"]).max() < 1e-12
    assert np.abs(a["forces"] - ref["forces"]).max() < 1e-10
    if cell is not None:
        assert np.abs(a["stress"] - ref["stress"]).max() < 1e-12


def test_forces_are_energy_gradient_fd(oracle64):
    """central finite difference of the fp64 oracle energy (the reference's own FD check is
    tests/test_pbc.py:1054-1107 at 5e-2; fp64 lets us be much tighter)."""
    g = golden("batch5")
    sel = g["mol_idx"] == 0
    c, z = g["coord"][sel].astype(np.float64), g["numbers"][sel]
    r0 = O.evaluate(oracle64, c, z, 0.0)
    h = 1e-4
    for (i, k) in [(0, 0), (3, 1), (5, 2)]:
        cp, cm = c.copy(), c.copy()
        cp[i, k] += h
        cm[i, k] -= h
        ep = O.evaluate(oracle64, cp.astype(np.float32).astype(np.float64), z, 0.0, forces=False)["energy"][0]
        em = O.evaluate(oracle64, cm.astype(np.float32).astype(np.float64), z, 0.0, forces=False)["energy"][0]
        hp = float(cp.astype(np.float32)[i, k]) - float(cm.astype(np.float32)[i, k])
        assert abs(-(ep - em) / hp - r0["forces"][i, k]) < 5e-4


@pytest.mark.parametrize("case", ["b5", "pbc"])
def test_analytic_matches_autograd_fp64_two_charge_channels(oracle64_nse, case):
    """The same for the open-shell NSE family: both channels through the charge convolution, the NSE adjoint and the
    Coulomb seed (which sees alpha + beta and feeds every channel alike)."""
    g = golden("nse")
    if case == "b5":
        c, z, mol, q, mult, cell, kw = g["b5_coord"], g["b5_numbers"], g["b5_mol_idx"], g["b5_charge"], g["b5_mult"], None, {}
    else:
        c, z, mol, q, mult, cell = g["pbc_coord"], g["pbc_numbers"], np.zeros(96, dtype=np.int64), np.zeros(1, np.float32), g["pbc_mult"], g["pbc_cell"]
        kw = {"coulomb": "dsf", "dsf_rc": 9.0, "dsf_alpha": 0.2}
    ref = O.evaluate(oracle64_nse, c, z, q, mol, cell=cell, stress=cell is not None, mult=mult, return_intermediates=True, **kw)
    xw = ref["coord_wrapped"]
    if cell is None:
        nbl, shl = O.neighbor_list(xw, float("inf"), mol)
        coul = "simple"
    else:
        nbl, shl = O.neighbor_list(xw, 9.0, mol, cell, np.ones(3, bool))
        coul = "dsf"
    a = AN.evaluate(oracle64_nse, xw, z, q, mol, ref["nbmat"], ref.get("shifts"), cell, coulomb=coul, nbmat_lr=nbl, shifts_lr=shl,
                    stress=cell is not None, mult=mult, **{k: v for k, v in kw.items() if k != "coulomb"})
    assert np.abs(a["energy"] - ref["energy"]).max() < 1e-9
    assert np.abs(a["charges"] - ref["charges"]).max() < 1e-12 and np.abs(a["spin_charges"] - ref["spin_charges"]).max() < 1e-12
    assert np.abs(a["forces"] - ref["forces"]).max() < 1e-10
    if cell is not None:
        assert np.abs(a["stress"] - ref["stress"]).max() < 1e-12


def _lists(oracle, g, kw):
    mol = g["mol_idx"] if "mol_idx" in g.files else np.zeros(len(g["numbers"]), dtype=np.int64)
    cell = g["cell"] if "cell" in g.files else None
    ref = O.evaluate(oracle, g["coord"], g["numbers"], g["charge"], mol, cell=cell, return_intermediates=True, forces=False, **kw)
    xw = ref["coord_wrapped"]
    if cell is None:
        nbl, shl = O.neighbor_list(xw, float("inf"), mol)
        coul = "simple"
    else:
        nbl, shl = O.neighbor_list(xw, kw["dsf_rc"], mol, cell, np.ones(3, bool))
        coul = "dsf"
    args = (xw, g["numbers"], g["charge"], mol, ref["nbmat"])
    kws = dict(shifts=ref.get("shifts"), cell=cell, coulomb=coul, nbmat_lr=nbl, shifts_lr=shl,
               **{k: v for k, v in kw.items() if k != "coulomb"})
    return args, kws


def test_tangent_sweep_matches_autograd_hessian_fp64(oracle64):
    """The hand-derived tangent sweep (spec of csrc/hvp.hip) against the autograd Hessian of the oracle, config 4's molecule."""
    g = golden("hvp40")
    H = O.evaluate(oracle64, g["coord"], g["numbers"], g["charge"], hessian=True, forces=False)["hessian"].reshape(120, 120)
    args, kws = _lists(oracle64, g, {})
    V = np.concatenate([g["v1"][None], g["v4"]]).astype(np.float64)
    r = AN.evaluate_hvp(oracle64, *args, V, **kws)
    assert np.abs(r["hv"].reshape(5, 120) - V.reshape(5, 120) @ H).max() < 1e-9
    f = AN.evaluate(oracle64, *args, **kws)["forces"]
    assert np.abs(r["forces"] - f).max() < 1e-12


@pytest.mark.parametrize("name,kw", [
    ("batch5", {}),
    ("pbc96_dsf8_wrapped", {"coulomb": "dsf", "dsf_rc": 8.0, "dsf_alpha": 0.25}),
])
def test_tangent_sweep_matches_central_differences_fp64(oracle64, name, kw):
    """... and against central differences of the analytic fp64 forces on a batch and on a periodic DSF cell (lists held fixed)."""
    g = golden(name)
    args, kws = _lists(oracle64, g, kw)
    n = len(g["numbers"])
    rng = np.random.default_rng(5)
    V = rng.standard_normal((2, n, 3))
    r = AN.evaluate_hvp(oracle64, *args, V, **kws)
    h = 1e-5
    for k in range(2):
        fp = AN.evaluate(oracle64, args[0] + h * V[k], *args[1:], **kws)["forces"]
        fm = AN.evaluate(oracle64, args[0] - h * V[k], *args[1:], **kws)["forces"]
        fd = -(fp - fm) / (2 * h)
        assert np.abs(r["hv"][k] - fd).max() < 2e-6 * max(1.0, np.abs(fd).max()), (k, np.abs(r["hv"][k] - fd).max())


def test_tangent_sweep_two_charge_channels_fp64(oracle64_nse):
    """The NSE family (two charge channels) through the tangent sweep: central differences of the analytic fp64 forces."""
    g = golden("nse")
    c, z, mol, q, mult = g["b5_coord"], g["b5_numbers"], g["b5_mol_idx"], g["b5_charge"], g["b5_mult"]
    ref = O.evaluate(oracle64_nse, c, z, q, mol, mult=mult, return_intermediates=True, forces=False)
    nbl, shl = O.neighbor_list(ref["coord_wrapped"], float("inf"), mol)
    args = (ref["coord_wrapped"], z, q, mol, ref["nbmat"])
    kws = dict(coulomb="simple", nbmat_lr=nbl, shifts_lr=shl, mult=mult)
    V = np.random.default_rng(7).standard_normal((1, len(z), 3))
    r = AN.evaluate_hvp(oracle64_nse, *args, V, **kws)
    h = 1e-5
    fd = -(AN.evaluate(oracle64_nse, args[0] + h * V[0], *args[1:], **kws)["forces"]
           - AN.evaluate(oracle64_nse, args[0] - h * V[0], *args[1:], **kws)["forces"]) / (2 * h)
    assert np.abs(r["hv"][0] - fd).max() < 2e-6 * max(1.0, np.abs(fd).max())
