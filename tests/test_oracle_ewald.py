"""The oracle's Ewald restatement (oracle/aimnet2_oracle.py, ewald_*) against known answers: the Madelung constants of rock salt
and caesium chloride (published values), independence of the splitting parameter, a charged cell against the same cell doubled,
and the autograd forces / stress against finite differences.  The reference's own arithmetic for this method lives in
nvalchemiops (absent): these are the pins this path has."""
from __future__ import annotations

import math

import numpy as np
import torch

from oracle import aimnet2_oracle as O


def _ewald_point_charges(pos, q, cell, accuracy=None, alpha=None, rc=None, kc=None):
    """Total electrostatic energy / k_e of point charges in a periodic cell through the oracle's pieces (fp64)."""
    pos, q, cell = np.asarray(pos, float), np.asarray(q, float), np.asarray(cell, float)
    if alpha is None:
        alpha, rc, kc = O.ewald_parameters(len(q), abs(np.linalg.det(cell)), accuracy)
    nh = O.ewald_kvectors(cell, kc)
    e_rec = float(O.ewald_reciprocal(torch.tensor(pos), torch.tensor(q), torch.tensor(cell), nh, alpha))
    heights = abs(np.linalg.det(cell)) / np.array([np.linalg.norm(np.cross(cell[(a + 1) % 3], cell[(a + 2) % 3])) for a in range(3)])
    R = [int(math.ceil(rc / h)) for h in heights]
    e_real = 0.0
    for sx in range(-R[0], R[0] + 1):
        for sy in range(-R[1], R[1] + 1):
            for sz in range(-R[2], R[2] + 1):
                d = pos[None, :, :] + np.array([sx, sy, sz]) @ cell - pos[:, None, :]
                r = np.sqrt((d * d).sum(-1))
                m = (r > 1e-9) & (r < rc)
                rr = np.where(m, r, 1.0)
                e_real += 0.5 * (q[:, None] * q[None, :] * np.where(m, torch.erfc(torch.tensor(alpha * rr)).numpy() / rr, 0.0)).sum()
    return e_real + e_rec - alpha / math.sqrt(math.pi) * (q * q).sum()


def test_madelung_rock_salt_and_caesium_chloride():
    pos = np.array([[i, j, k] for i in range(2) for j in range(2) for k in range(2)], float)
    q = np.array([1.0 if (i + j + k) % 2 == 0 else -1.0 for i in range(2) for j in range(2) for k in range(2)])
    e = _ewald_point_charges(pos, q, 2.0 * np.eye(3), accuracy=1e-10)
    assert abs(e / 4.0 - (-1.7475645946)) < 1e-9  # per ion pair, nearest-neighbour distance 1
    e6 = _ewald_point_charges(pos, q, 2.0 * np.eye(3), accuracy=1e-6)
    assert abs(e6 / 4.0 - (-1.7475645946)) < 2e-6
    a = 2.0 / math.sqrt(3.0)  # CsCl: nearest-neighbour distance a sqrt(3) / 2 = 1
    e = _ewald_point_charges(np.array([[0.0, 0.0, 0.0], [a / 2, a / 2, a / 2]]), np.array([1.0, -1.0]), a * np.eye(3), accuracy=1e-10)
    assert abs(e - (-1.76267477)) < 1e-8


def test_independent_of_the_splitting_and_of_the_cell_choice():
    rng = np.random.default_rng(0)
    cell = np.array([[7.0, 0.3, -0.2], [0.5, 6.0, 0.4], [-0.3, 0.2, 8.0]])
    pos = rng.random((12, 3)) @ cell
    q = rng.normal(size=12)
    q -= q.mean()
    ref = _ewald_point_charges(pos, q, cell, accuracy=1e-12)
    for al, rc, kc in ((0.35, 22.0, 4.5), (0.6, 13.0, 7.5)):
        assert abs(_ewald_point_charges(pos, q, cell, alpha=al, rc=rc, kc=kc) - ref) < 1e-9
    # a charged cell (neutralising background) against the same cell doubled along a: twice the energy
    qc = q + 0.25
    e1 = _ewald_point_charges(pos, qc, cell, accuracy=1e-12)
    cell2 = cell.copy()
    cell2[0] *= 2.0
    e2 = _ewald_point_charges(np.concatenate([pos, pos + cell[0]]), np.concatenate([qc, qc]), cell2, accuracy=1e-12)
    assert abs(e2 - 2.0 * e1) < 1e-8


def test_reciprocal_space_derivatives_by_autograd_match_finite_differences():
    rng = np.random.default_rng(1)
    cell = torch.tensor([[6.0, 0.4, 0.0], [0.0, 5.0, 0.3], [0.2, 0.0, 7.0]], dtype=torch.float64)
    x = torch.tensor(rng.random((6, 3)), dtype=torch.float64) @ cell
    q = torch.tensor(rng.normal(size=6), dtype=torch.float64)
    al, _, kc = O.ewald_parameters(6, float(torch.linalg.det(cell)), 1e-10)
    nh = O.ewald_kvectors(cell.numpy(), kc)
    xg = x.clone().requires_grad_(True)
    scal = torch.eye(3, dtype=torch.float64, requires_grad=True)
    e = O.ewald_reciprocal(xg @ scal, q, cell @ scal, nh, al)
    gx, gs = torch.autograd.grad(e, [xg, scal])
    h = 1e-5
    for i, c in ((0, 0), (3, 2)):
        xp, xm = x.clone(), x.clone()
        xp[i, c] += h
        xm[i, c] -= h
        fd = (O.ewald_reciprocal(xp, q, cell, nh, al) - O.ewald_reciprocal(xm, q, cell, nh, al)) / (2 * h)
        assert abs(float(fd - gx[i, c])) < 1e-7
    for a, b in ((0, 0), (1, 2)):
        sp, sm = torch.eye(3, dtype=torch.float64), torch.eye(3, dtype=torch.float64)
        sp[a, b] += h
        sm[a, b] -= h
        fd = (O.ewald_reciprocal(x @ sp, q, cell @ sp, nh, al) - O.ewald_reciprocal(x @ sm, q, cell @ sm, nh, al)) / (2 * h)
        assert abs(float(fd - gs[a, b])) < 1e-7


def test_evaluate_ewald_runs_and_rejects_open_boundaries(oracle64):
    from conftest import golden

    g = golden("pbc96_dsf15")
    r = O.evaluate(oracle64, g["coord"], g["numbers"], np.zeros(1, np.float32), cell=g["cell"], coulomb="ewald", stress=True)
    d = O.evaluate(oracle64, g["coord"], g["numbers"], np.zeros(1, np.float32), cell=g["cell"], coulomb="dsf", dsf_rc=15.0, stress=True)
    assert np.isfinite(r["energy"]).all() and abs(r["energy"][0] - d["energy"][0]) < 0.5  # DSF approximates the same sum
    import pytest

    with pytest.raises(ValueError):
        O.evaluate(oracle64, g["coord"], g["numbers"], np.zeros(1, np.float32), cell=g["cell"], pbc=(True, True, False), coulomb="ewald")


def test_against_the_references_in_tree_ewald_twin():
    """tests/golden/ewald_matrix.npz: Coulomb matrices of `aimnet.ops.coulomb_matrix_ewald` (ops.py:196-276, the reference's own
    pure-PyTorch Ewald, fp32; generated by tests/golden/make_golden.py --only-ewald): E = 1/2 q^T J q of a neutral triclinic cell of
    12 charges and of rock salt, and the potential J q, against this oracle's pieces.  Same splitting parameters by construction
    (the formula of calculator.py:660-667 is the one of ops.py:207-214); agreement to the fp32 rounding of the reference's matrix."""
    from conftest import golden

    g = golden("ewald_matrix")
    for acc, key in ((1e-6, "1e-06"), (1e-8, "1e-08")):
        e = _ewald_point_charges(g["coord"], g["q"], g["cell"], accuracy=acc)
        assert abs(e - float(g["E_" + key])) < 3e-6 * abs(e), (acc, e, float(g["E_" + key]))
    e = _ewald_point_charges(g["nacl_coord"], g["nacl_q"], g["nacl_cell"], accuracy=1e-8)
    assert abs(e - float(g["nacl_E"])) < 3e-6 * abs(e)
    # the potential phi = dE/dq = J q (neutral cell: no background term): reciprocal part by autograd + real part + self term
    al, rc, kc = O.ewald_parameters(12, abs(np.linalg.det(g["cell"])), 1e-8)
    nh = O.ewald_kvectors(g["cell"], kc)
    q = torch.tensor(g["q"], requires_grad=True)
    x, cell = torch.tensor(g["coord"]), torch.tensor(g["cell"])
    e_rec = O.ewald_reciprocal(x, q, cell, nh, al)
    (phi_rec,) = torch.autograd.grad(e_rec, q)
    heights = abs(np.linalg.det(g["cell"])) / np.array([np.linalg.norm(np.cross(g["cell"][(a + 1) % 3], g["cell"][(a + 2) % 3])) for a in range(3)])
    R = [int(math.ceil(rc / h)) for h in heights]
    phi_real = np.zeros(12)
    for sx in range(-R[0], R[0] + 1):
        for sy in range(-R[1], R[1] + 1):
            for sz in range(-R[2], R[2] + 1):
                d = g["coord"][None, :, :] + np.array([sx, sy, sz]) @ g["cell"] - g["coord"][:, None, :]
                r = np.sqrt((d * d).sum(-1))
                m = (r > 1e-9) & (r < rc)
                rr = np.where(m, r, 1.0)
                phi_real += (np.where(m, torch.erfc(torch.tensor(al * rr)).numpy() / rr, 0.0) * g["q"][None, :]).sum(-1)
    phi = phi_rec.detach().numpy() + phi_real - 2.0 * al / math.sqrt(math.pi) * g["q"]
    ref = g["J_1e-08"].astype(np.float64) @ g["q"]
    assert np.abs(phi - ref).max() < 5e-6 * np.abs(ref).max()
