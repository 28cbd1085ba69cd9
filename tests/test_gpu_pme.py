"""Particle-mesh Ewald (engine method "pme", csrc/pme.hip + the real-space term on the cell-grid walk).

Two anchors.  (1) The mesh kernels alone (aimnet_debug_pme_recip) against their CPU twin oracle/pme.py on the same mesh: double
against double.  (2) The whole evaluation against the oracle's EXACT Ewald sum (oracle/aimnet2_oracle.py, pinned to the
reference's in-tree torch Ewald): a converged Ewald energy does not depend on the splitting or on how the reciprocal sum is
taken, so "pme" must land on it to the requested accuracy.  Unpinned against nvalchemiops' particle_mesh_ewald (lr.py:752-775),
which is not in the reference tree."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import pytest
import torch

import test_gpu_parity as P
from conftest import STRESS_ATOL, assert_forces_close, energy_tol, golden
from oracle import aimnet2_oracle as O
from oracle import pme as OP

pytestmark = pytest.mark.gpu


def _run(eng, coord, numbers, mol, charge, cell, method="pme", **kw):
    dev = eng.device
    r = eng.eval(torch.from_numpy(coord).to(dev), torch.from_numpy(numbers).to(dev), torch.from_numpy(mol).to(dev),
                 torch.from_numpy(np.atleast_1d(charge).astype(np.float32)).to(dev), cell=torch.from_numpy(cell).to(dev), forces=True, stress=True,
                 coulomb=method, **kw)
    return {k: v.cpu().numpy() for k, v in r.items()}


def _recip(eng, x, q, cell, acc, max_mesh=1 << 20, order=None):
    lib, dev = eng.lib, eng.device
    n = len(x)
    xd = torch.from_numpy(x.astype(np.float32)).to(dev).contiguous()
    qd = torch.from_numpy(q.astype(np.float32)).to(dev).contiguous()
    cd = torch.from_numpy(cell.astype(np.float32)).to(dev).contiguous()
    e = torch.zeros(n, dtype=torch.float64, device=dev)
    qb = torch.zeros(n, dtype=torch.float32, device=dev)
    fg = torch.zeros(n, 3, dtype=torch.float32, device=dev)
    va = torch.zeros(n, 9, dtype=torch.float32, device=dev)
    info = (C.c_double * 8)()
    od = None if order is None else torch.from_numpy(np.ascontiguousarray(order, dtype=np.int32)).to(dev)
    torch.cuda.synchronize()
    rc = lib.aimnet_debug_pme_recip(xd.data_ptr(), qd.data_ptr(), None if od is None else od.data_ptr(), cd.data_ptr(), float(q.astype(np.float32).sum(dtype=np.float64)), n, acc,
                                    max_mesh, e.data_ptr(), qb.data_ptr(), fg.data_ptr(), va.data_ptr(), info, None)
    assert rc == 0
    return e.cpu().numpy(), qb.cpu().numpy(), fg.cpu().numpy(), va.cpu().numpy(), list(info)


def _random_cell(rep, seed, neutral=True):
    rng = np.random.default_rng(seed)
    base = np.array([[4.982, 0.0, 0.0], [0.0, 12.562, 0.0], [-0.233, 0.0, 11.814]])
    cell = (base * np.array(rep)[:, None]).astype(np.float32)
    n = 96 * rep[0] * rep[1] * rep[2]
    x = ((rng.random((n, 3)) * 3.0 - 1.0) @ cell).astype(np.float32)  # images outside the cell as well
    q = rng.normal(0.0, 0.3, n).astype(np.float32)
    if neutral:
        q -= q.mean()
    return x, q, cell


@pytest.mark.parametrize("rep,acc,neutral", [((1, 1, 1), 1e-6, True), ((2, 1, 1), 1e-4, False), ((2, 2, 1), 1e-8, True), ((3, 2, 2), 1e-6, False)])
def test_mesh_kernels_equal_their_cpu_twin(hip_engine, rep, acc, neutral):
    """Same mesh, same splitting: potential (double) to 1e-10, gradient / strain (stored fp32) to fp32 rounding."""
    x, q, cell = _random_cell(rep, 11, neutral)
    e, qb, fg, va, info = _recip(hip_engine, x, q, cell, acc)
    alpha, rc, mesh = OP.pme_parameters(len(x), cell.astype(np.float64), acc)
    assert [int(v) for v in info[2:5]] == list(mesh) and int(info[5]) == mesh[0] * mesh[1] * mesh[2]
    assert abs(info[0] - alpha) < 1e-6 * alpha and abs(info[1] - rc) < 1e-6 * rc
    # the engine keeps 1 / (4 alpha^2) and phi_bg as fp32 (EwaldSystem): evaluate the twin at the alpha that fp32 value stands for;
    # phi_bg's fp32 rounding enters the tolerance below
    alpha_eff = math.sqrt(1.0 / (4.0 * float(np.float32(1.0 / (4.0 * alpha * alpha)))))
    ref = OP.pme_reciprocal(x.astype(np.float64), q.astype(np.float64), cell.astype(np.float64), alpha_eff, mesh)
    qphi = q.astype(np.float64) * ref["phi"]
    scale = np.abs(ref["phi"]).max()
    assert np.abs(e - qphi).max() < (1e-10 * max(1.0, scale) + 2e-7 * abs(info[6])) * np.abs(q).max()
    assert np.abs(qb - 2.0 * ref["phi"]).max() < 3e-7 * max(1.0, scale)
    gref = 2.0 * q[:, None].astype(np.float64) * ref["grad"]
    assert np.abs(fg - gref).max() < 3e-7 * max(1.0, np.abs(gref).max())
    assert np.abs(va.sum(0).reshape(3, 3) - 2.0 * ref["strain"]).max() < 2e-6 * max(1.0, np.abs(ref["strain"]).max())


def test_mesh_kernels_repeat_bitwise_and_report_their_capacity(hip_engine):
    x, q, cell = _random_cell((2, 2, 1), 5)
    a = _recip(hip_engine, x, q, cell, 1e-6)
    b = _recip(hip_engine, x, q, cell, 1e-6)
    for u, v in zip(a[:4], b[:4]):
        assert np.array_equal(u, v)  # integer charge assignment, fixed-order sums
    # atoms grouped by 4 A boxes: the charge assignment sums each group in an LDS tile first - the same integers, the same bits
    frac = (x.astype(np.float64) @ np.linalg.inv(cell.astype(np.float64))) % 1.0
    box = np.floor(frac * np.maximum(1, np.floor(np.linalg.norm(cell, axis=1) / 4.0))).astype(np.int64)
    order = np.lexsort((box[:, 2], box[:, 1], box[:, 0]))
    c = _recip(hip_engine, x, q, cell, 1e-6, order=order)
    for u, v in zip(a[:4], c[:4]):
        assert np.array_equal(u, v)
    e, qb, fg, va, info = _recip(hip_engine, x, q, cell, 1e-6, max_mesh=512)
    assert int(info[5]) > 512 and not e.any() and not fg.any()  # too small: nothing computed, the need reported


@pytest.mark.parametrize("name,charge,acc", [("pbc96_dsf15", 0.0, 1e-8), ("pbc96_dsf8_wrapped", -1.0, 1e-8), ("pbc96_dsf15", 2.0, 1e-7)])
def test_cell_vs_exact_ewald_oracle(hip_engine, oracle32, oracle64, name, charge, acc):
    """Neutral and charged cells at a tight accuracy: the mesh evaluation against the oracle's exact sum at the gates of the Ewald
    tests (energy, charges, forces, stress)."""
    g = golden(name)
    mol = np.zeros(96, dtype=np.int64)
    q = np.array([charge], dtype=np.float32)
    res = _run(hip_engine, g["coord"], g["numbers"], mol, q, g["cell"], ewald_accuracy=acc)
    okw = dict(cell=g["cell"], coulomb="ewald", ewald_accuracy=acc, stress=True)
    ref = O.evaluate(oracle32, g["coord"], g["numbers"], q, mol, **okw)
    e64 = O.evaluate(oracle64, g["coord"], g["numbers"], q, mol, **dict(okw, forces=False, stress=False))["energy"]
    P.compare(res, ref, 96, f"{name} q={charge} pme/exact-ewald oracle", e64)
    assert abs(res["energy"][0] - e64[0]) <= energy_tol(96) + abs(ref["energy"][0] - e64[0])


def test_default_accuracy_lands_on_the_ewald_method(hip_engine):
    """1e-6 (the reference's default, calculator.py:643): engine "pme" against engine "ewald" - different alpha (capped real-space
    cutoff), different reciprocal sums, same energy / forces / stress to the accuracy."""
    from aimnetcentral_amd import workloads

    c, z, cell = workloads.glucose_supercell((2, 2, 2))
    rng = np.random.default_rng(5)
    c = (c + rng.normal(0.0, 0.02, c.shape)).astype(np.float32)
    cell = cell.astype(np.float32)
    mol = np.zeros(len(z), dtype=np.int64)
    q = np.zeros(1, dtype=np.float32)
    hip_engine._pme_max_mesh = 512  # starts too small: grows to what the engine reports
    a = _run(hip_engine, c, z, mol, q, cell)
    assert hip_engine._pme_max_mesh >= int(hip_engine.last_status[7]) > 512
    b = _run(hip_engine, c, z, mol, q, cell, method="ewald")
    assert abs(a["energy"][0] - b["energy"][0]) < 2e-4  # 768 atoms: ~1e-6 of the reciprocal energy (~10^2 eV)
    assert np.abs(a["charges"] - b["charges"]).max() < 2e-6
    assert np.abs(a["forces"] - b["forces"]).max() < 2e-5 + 1e-4 * np.abs(b["forces"]).max()
    assert np.abs(a["stress"] - b["stress"]).max() < STRESS_ATOL
    a2 = _run(hip_engine, c, z, mol, q, cell)
    for k in ("energy", "forces", "stress", "charges"):
        assert np.array_equal(a[k], a2[k])  # bitwise repeatable


def test_two_systems_with_different_cells(hip_engine, oracle32, oracle64):
    """Per-system splitting and mesh: two cells of different volume and atom count in one batch, one of them charged."""
    g = golden("pbc2x96_dsf9")
    keep = np.ones(192, dtype=bool)
    keep[100:130] = False
    coord, numbers, mol = g["coord"][keep], g["numbers"][keep], g["mol_idx"][keep]
    cell = g["cell"].copy()
    cell[1] = cell[1] * 1.07
    q = np.array([0.0, 1.0], dtype=np.float32)
    res = _run(hip_engine, coord, numbers, mol, q, cell, ewald_accuracy=1e-8)
    okw = dict(cell=cell, coulomb="ewald", ewald_accuracy=1e-8, stress=True)
    ref = O.evaluate(oracle32, coord, numbers, q, mol, **okw)
    e64 = O.evaluate(oracle64, coord, numbers, q, mol, **dict(okw, forces=False, stress=False))["energy"]
    P.compare(res, ref, np.bincount(mol), "two cells pme/exact-ewald oracle", e64)


def test_through_the_calculator_with_forces_as_the_energy_gradient():
    """set_lrcoulomb_method("pme") reaches the mesh kernels; the forces are the derivative of the energy the engine returns
    (analytic spline derivatives), checked by central differences along a random direction."""
    from aimnetcentral_amd import AIMNet2Calculator, loader

    calc = AIMNet2Calculator(loader.synthetic_spec(0, cold=True), device="cuda:0")
    calc.set_lrcoulomb_method("pme", ewald_accuracy=1e-7)
    g = golden("pbc96_dsf15")
    data = dict(coord=g["coord"], numbers=g["numbers"], charge=0.0, cell=g["cell"])
    out = calc(data, forces=True, stress=True)
    assert calc.engine.last_status[7] > 0
    f = out["forces"].cpu().numpy().astype(np.float64)
    rng = np.random.default_rng(3)
    v = rng.normal(size=(96, 3))
    v /= np.linalg.norm(v)
    h = 2e-3
    ep = float(calc(dict(data, coord=(g["coord"] + h * v).astype(np.float32)))["energy"])
    em = float(calc(dict(data, coord=(g["coord"] - h * v).astype(np.float32)))["energy"])
    assert abs(-(ep - em) / (2 * h) - (f * v).sum()) < 2e-3 * max(1.0, abs((f * v).sum()))
    with pytest.raises(ValueError, match="requires a periodic 'cell'"):
        calc(dict(coord=g["coord"], numbers=g["numbers"], charge=0.0))


def test_with_dftd3_and_with_two_charge_channels(hip_engine, hip_engine_nse):
    """The mesh beside the other long-range pieces: external DFT-D3 (its own 15 A list) and an open-shell NSE model (two charge
    channels, the Coulomb sum sees their total) - each against the same evaluation with the exact Ewald sum at 1e-8."""
    g = golden("dftd3")
    par, tables = P._d3(12.0)
    hip_engine.set_dftd3_tables(tables)
    mol = np.zeros(96, dtype=np.int64)
    a, a0 = P._run_d3(hip_engine, g["pbc_coord"], g["pbc_numbers"], mol, "pme", par, cell=g["pbc_cell"], stress=True, ewald_accuracy=1e-8)
    b, b0 = P._run_d3(hip_engine, g["pbc_coord"], g["pbc_numbers"], mol, "ewald", par, cell=g["pbc_cell"], stress=True, ewald_accuracy=1e-8)
    assert abs(a["energy"][0] - a0["energy"][0]) > 1e-3  # the dispersion term is there
    for x, y in ((a, b), (a0, b0)):
        assert abs(x["energy"][0] - y["energy"][0]) < 2e-6
        assert np.abs(x["forces"] - y["forces"]).max() < 2e-6 * max(1.0, np.abs(y["forces"]).max())
        assert np.abs(x["stress"] - y["stress"]).max() < 1e-7  # (fp32 outputs: one ulp at 0.18 is 1.5e-8)
    p = golden("nse")
    dev = hip_engine_nse.device
    args = (torch.from_numpy(p["pbc_coord"]).to(dev), torch.from_numpy(p["pbc_numbers"]).to(dev), torch.zeros(96, dtype=torch.int64, device=dev),
            torch.from_numpy(P._nse_charge(1.0, np.array([2.0], dtype=np.float32))).to(dev))  # a doublet cation
    kw = dict(cell=torch.from_numpy(p["pbc_cell"]).to(dev), forces=True, stress=True, ewald_accuracy=1e-8)
    u = {k: v.cpu().numpy() for k, v in hip_engine_nse.eval(*args, coulomb="pme", **kw).items()}
    v = {k: v.cpu().numpy() for k, v in hip_engine_nse.eval(*args, coulomb="ewald", **kw).items()}
    assert abs(u["energy"][0] - v["energy"][0]) < 5e-6
    assert np.abs(u["forces"] - v["forces"]).max() < 3e-6 * max(1.0, np.abs(v["forces"]).max())
    assert np.abs(u["spin_charges"] - v["spin_charges"]).max() < 1e-6 and np.abs(u["charges"] - v["charges"]).max() < 1e-6


def test_deferred_status_grows_the_mesh_capacity():
    """Device-resident stepping with the mesh method: a capacity that is too small is noticed by check_status(), grown to what the
    engine reported, and the repeated evaluation is the synchronous one's."""
    from aimnetcentral_amd import AIMNet2Calculator, loader
    from aimnetcentral_amd.engine import NeighborOverflowError

    calc = AIMNet2Calculator(loader.synthetic_spec(0, cold=True), device="cuda:0")
    calc.set_lrcoulomb_method("pme")
    g = golden("pbc96_dsf15")
    data = dict(coord=g["coord"], numbers=g["numbers"], charge=0.0, cell=g["cell"])
    ref = calc(data, forces=True)
    need = int(calc.engine.last_status[7])
    calc.engine._pme_max_mesh = 512
    assert need > 512
    calc.eval(data, forces=True, defer_status=True)
    with pytest.raises(NeighborOverflowError):
        calc.check_status()
    assert calc.engine._pme_max_mesh >= need
    out = calc.eval(data, forces=True, defer_status=True)
    calc.check_status()
    assert float(out["energy"]) == float(ref["energy"]) and torch.equal(out["forces"], ref["forces"])


def test_sheared_cell_and_tiny_cell_mesh_kernels(hip_engine):
    """A strongly sheared triclinic cell (60 degree angles) and a cell smaller than the spline support (5 A: every atom's 8 x 8 x 8
    stencil wraps around the 8-point minimum mesh) against the CPU twin."""
    rng = np.random.default_rng(21)
    for cell, n in ((np.array([[9.0, 0.0, 0.0], [4.5, 7.794, 0.0], [4.5, 2.598, 7.348]], np.float32), 150),
                    (np.eye(3, dtype=np.float32) * 5.0, 12)):
        x = ((rng.random((n, 3)) * 2.0 - 0.5) @ cell).astype(np.float32)
        q = rng.normal(0.0, 0.4, n).astype(np.float32)
        e, qb, fg, va, info = _recip(hip_engine, x, q, cell, 1e-6)
        alpha, rc, mesh = OP.pme_parameters(n, cell.astype(np.float64), 1e-6)
        assert [int(v) for v in info[2:5]] == list(mesh)
        alpha_eff = math.sqrt(1.0 / (4.0 * float(np.float32(1.0 / (4.0 * alpha * alpha)))))
        ref = OP.pme_reciprocal(x.astype(np.float64), q.astype(np.float64), cell.astype(np.float64), alpha_eff, mesh)
        scale = max(1.0, np.abs(ref["phi"]).max())
        assert np.abs(e - q.astype(np.float64) * ref["phi"]).max() < (1e-10 * scale + 2e-7 * abs(info[6])) * np.abs(q).max()
        gref = 2.0 * q[:, None].astype(np.float64) * ref["grad"]
        assert np.abs(fg - gref).max() < 3e-7 * max(1.0, np.abs(gref).max())
        assert np.abs(va.sum(0).reshape(3, 3) - 2.0 * ref["strain"]).max() < 2e-6 * max(1.0, np.abs(ref["strain"]).max())


def test_batch_of_many_cells_each_with_its_own_mesh(hip_engine):
    """24 periodic systems of 96 atoms with cells scaled 0.97 ... 1.20 (different alpha, mesh and capacity needs) in one batch:
    every system equals its single-system evaluation (to the difference between the split-operand GEMMs of the batch and the exact-fp32
    ones a 96-atom system takes), and the batch lands on the Ewald method."""
    g = golden("pbc96_dsf15")
    nsys = 24
    rng = np.random.default_rng(8)
    scale = np.linspace(0.97, 1.20, nsys).astype(np.float32)
    coords = [(g["coord"] * s + rng.normal(0.0, 0.01, g["coord"].shape)).astype(np.float32) for s in scale]
    cells = np.stack([g["cell"] * s for s in scale]).astype(np.float32)
    c, z = np.concatenate(coords), np.tile(g["numbers"], nsys)
    mol, q = np.repeat(np.arange(nsys), 96), np.zeros(nsys, np.float32)
    a = _run(hip_engine, c, z, mol, q, cells)
    b = _run(hip_engine, c, z, mol, q, cells, method="ewald")
    assert np.abs(a["energy"] - b["energy"]).max() < 5e-5
    assert np.abs(a["forces"] - b["forces"]).max() < 2e-5 + 1e-4 * np.abs(b["forces"]).max()
    for k in (0, 11, 23):
        s1 = _run(hip_engine, coords[k], g["numbers"], np.zeros(96, np.int64), np.zeros(1, np.float32), cells[k])
        assert abs(s1["energy"][0] - a["energy"][k]) < 1e-4
        assert np.abs(s1["forces"] - a["forces"][96 * k:96 * (k + 1)]).max() < 2e-5 + 1e-4 * np.abs(s1["forces"]).max()
