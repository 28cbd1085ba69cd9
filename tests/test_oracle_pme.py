"""oracle/pme.py (the CPU twin of csrc/pme.hip) against the exact structure-factor sum it must converge to, and its own building
blocks.  The exact sum is the numpy twin of oracle.aimnet2_oracle.ewald_reciprocal (pinned to the reference's in-tree torch
Ewald, test_oracle_golden.py); nvalchemiops, whose particle_mesh_ewald the reference calls (lr.py:752-775), is not in the
reference tree, so this is as far as the mesh can be pinned."""
import math

import numpy as np
import pytest
import torch

from oracle import aimnet2_oracle as O
from oracle import pme


def _system(rep=(2, 1, 1), seed=0, neutral=True):
    rng = np.random.default_rng(seed)
    base = np.array([[4.982, 0.0, 0.0], [0.0, 12.562, 0.0], [-0.233, 0.0, 11.814]])
    cell = base * np.array(rep)[:, None]
    n = 96 * rep[0] * rep[1] * rep[2]
    x = rng.random((n, 3)) @ cell
    q = rng.normal(0.0, 0.3, n)
    if neutral:
        q -= q.mean()
    return x, q, cell


def test_bspline_partition_of_unity_and_derivative():
    d = np.linspace(0.0, 0.999, 37)
    for p in (4, 6, 8):
        w, dw = pme.bspline(d, p)
        assert np.abs(w.sum(1) - 1.0).max() < 1e-14 and np.abs(dw.sum(1)).max() < 1e-13
        h = 1e-6
        wp, _ = pme.bspline(d + h, p)
        wm, _ = pme.bspline(d - h + (d < h) * h, p)
        fd = (wp - wm) / np.where(d < h, h, 2 * h)[:, None]
        assert np.abs(fd - dw).max() < 1e-6
    w, _ = pme.bspline(np.zeros(1), 8)
    assert abs(w[0, 7]) < 1e-300 and abs(w[0, 3] - 2416.0 / 5040.0) < 1e-15  # M8(0) = 0, M8(4) = 2416 / 7!


def test_exact_twin_equals_the_pinned_oracle_sum():
    x, q, cell = _system((1, 1, 1), 3, neutral=False)
    alpha, _, kc = O.ewald_parameters(len(x), abs(np.linalg.det(cell)), 1e-6)
    ex = pme.exact_reciprocal(x, q, cell, alpha, kc)
    ref = O.ewald_reciprocal(torch.from_numpy(x), torch.from_numpy(q), torch.from_numpy(cell), O.ewald_kvectors(cell, kc), alpha)
    assert abs(ex["e"] - float(ref)) < 1e-12 * max(1.0, abs(float(ref)))
    assert abs(0.5 * (q * ex["phi"]).sum() - ex["e"]) < 1e-12


@pytest.mark.parametrize("acc", [1e-4, 1e-6, 1e-8])
@pytest.mark.parametrize("rep,neutral", [((1, 1, 1), True), ((3, 1, 1), False)])
def test_mesh_converges_to_the_exact_sum_within_the_accuracy(acc, rep, neutral):
    x, q, cell = _system(rep, 1, neutral)
    alpha, rc, mesh = pme.pme_parameters(len(x), cell, acc)
    assert rc <= pme.PME_RC_MAX + 1e-12 and all(k >= 8 and k % 2 == 0 for k in mesh)
    kc = math.sqrt(2.0) * math.sqrt(-2.0 * math.log(acc)) * alpha
    ex = pme.exact_reciprocal(x, q, cell, alpha, 1.2 * kc)
    r = pme.pme_reciprocal(x, q, cell, alpha, mesh)
    f_rms = np.sqrt(((q[:, None] * ex["grad"]) ** 2).sum(1).mean())
    df_rms = np.sqrt(((q[:, None] * (r["grad"] - ex["grad"])) ** 2).sum(1).mean())
    assert df_rms < acc * f_rms
    assert abs(r["e"] - ex["e"]) < acc * max(abs(ex["e"]), 1.0)
    assert np.abs(r["phi"] - ex["phi"]).max() < 3.0 * acc * np.abs(ex["phi"]).max()
    assert np.abs(r["strain"] - ex["strain"]).max() < 30.0 * acc * np.abs(ex["strain"]).max()
    assert abs(0.5 * (q * r["phi"]).sum() - r["e"]) < 1e-12 * max(1.0, abs(r["e"]))  # E = 1/2 sum q phi on the mesh too


def test_direct_axis_transforms_equal_the_fft_and_forces_are_the_gradient():
    x, q, cell = _system((1, 1, 1), 2)
    alpha, _, mesh = pme.pme_parameters(len(x), cell, 1e-5)
    a = pme.pme_reciprocal(x, q, cell, alpha, mesh)
    b = pme.pme_reciprocal(x, q, cell, alpha, mesh, dft="direct")
    assert abs(a["e"] - b["e"]) < 1e-12 and np.abs(a["grad"] - b["grad"]).max() < 1e-12 and np.abs(a["phi"] - b["phi"]).max() < 1e-12
    # q_i grad phi_i is the exact derivative of the MESH energy (what makes mesh dynamics conservative)
    h = 1e-5
    for i, c in ((0, 0), (17, 1), (50, 2)):
        xp, xm = x.copy(), x.copy()
        xp[i, c] += h
        xm[i, c] -= h
        fd = (pme.pme_reciprocal(xp, q, cell, alpha, mesh)["e"] - pme.pme_reciprocal(xm, q, cell, alpha, mesh)["e"]) / (2 * h)
        assert abs(fd - q[i] * a["grad"][i, c]) < 1e-8
    # and the strain derivative that of a homogeneous deformation (row-vector strain: x -> x (1 + eps), cell -> cell (1 + eps))
    eps = np.zeros((3, 3))
    eps[0, 1] = eps[1, 0] = 0.5e-5
    eps[2, 2] = 1e-5
    ep = pme.pme_reciprocal(x @ (np.eye(3) + eps), q, cell @ (np.eye(3) + eps), alpha, mesh)["e"]
    em = pme.pme_reciprocal(x @ (np.eye(3) - eps), q, cell @ (np.eye(3) - eps), alpha, mesh)["e"]
    assert abs((ep - em) / 2.0 - (a["strain"] * eps).sum()) < 1e-10
