"""Ewald summation (engine method "ewald", csrc/ewald.hip + the real-space term on the cell-grid walk) against the oracle's
restatement (oracle/aimnet2_oracle.py, ewald_*: pinned in tests/test_oracle_ewald.py to golden matrices of the reference's in-tree
pure-PyTorch Ewald `ops.coulomb_matrix_ewald`, to Madelung constants and to a direct lattice sum; unpinned against nvalchemiops'
kernel, which is not in the reference tree).  Gates: the ones the
periodic DSF fixtures use (tests/test_gpu_parity.py), energies additionally against the fp64 oracle."""
from __future__ import annotations

import numpy as np
import pytest
import torch

import test_gpu_parity as P
from conftest import STRESS_ATOL, assert_forces_close, energy_tol, golden
from oracle import aimnet2_oracle as O

pytestmark = pytest.mark.gpu


def _run(eng, coord, numbers, mol, charge, cell, **kw):
    dev = eng.device
    r = eng.eval(torch.from_numpy(coord).to(dev), torch.from_numpy(numbers).to(dev), torch.from_numpy(mol).to(dev),
                 torch.from_numpy(np.atleast_1d(charge).astype(np.float32)).to(dev), cell=torch.from_numpy(cell).to(dev), forces=True, stress=True,
                 coulomb="ewald", **kw)
    return {k: v.cpu().numpy() for k, v in r.items()}


@pytest.mark.parametrize("name,charge,acc", [("pbc96_dsf15", 0.0, 1e-6), ("pbc96_dsf8_wrapped", -1.0, 1e-6), ("pbc96_dsf15", 2.0, 1e-8)])
def test_cell_vs_oracle(hip_engine, oracle32, oracle64, name, charge, acc):
    """Neutral and charged (neutralising background) cells: energy, charges, forces, stress."""
    g = golden(name)
    mol = np.zeros(96, dtype=np.int64)
    q = np.array([charge], dtype=np.float32)
    res = _run(hip_engine, g["coord"], g["numbers"], mol, q, g["cell"], ewald_accuracy=acc)
    okw = dict(cell=g["cell"], coulomb="ewald", ewald_accuracy=acc, stress=True)
    ref = O.evaluate(oracle32, g["coord"], g["numbers"], q, mol, **okw)
    e64 = O.evaluate(oracle64, g["coord"], g["numbers"], q, mol, **dict(okw, forces=False, stress=False))["energy"]
    P.compare(res, ref, 96, f"{name} q={charge} ewald/oracle", e64)
    assert abs(res["energy"][0] - e64[0]) <= energy_tol(96) + abs(ref["energy"][0] - e64[0])


def test_two_systems_with_different_cells(hip_engine, oracle32, oracle64):
    """Per-system parameters: two cells of different volume and atom count in one batch, one of them charged."""
    g = golden("pbc2x96_dsf9")
    keep = np.ones(192, dtype=bool)
    keep[100:130] = False  # the second system loses 30 atoms: its own (alpha, rc, kc)
    coord, numbers, mol = g["coord"][keep], g["numbers"][keep], g["mol_idx"][keep]
    cell = g["cell"].copy()
    cell[1] = cell[1] * 1.07
    q = np.array([0.0, 1.0], dtype=np.float32)
    res = _run(hip_engine, coord, numbers, mol, q, cell)
    okw = dict(cell=cell, coulomb="ewald", stress=True)
    ref = O.evaluate(oracle32, coord, numbers, q, mol, **okw)
    e64 = O.evaluate(oracle64, coord, numbers, q, mol, **dict(okw, forces=False, stress=False))["energy"]
    P.compare(res, ref, np.bincount(mol), "two cells ewald/oracle", e64)


def test_accuracy_parameter_moves_the_split_not_the_energy(hip_engine):
    """Converged sums do not depend on the splitting: 1e-6 and 1e-9 agree to the looser accuracy; "pme" (the mesh, test_gpu_pme.py)
    lands on the same energy to that accuracy."""
    g = golden("pbc96_dsf15")
    mol = np.zeros(96, dtype=np.int64)
    q = np.zeros(1, dtype=np.float32)
    a = _run(hip_engine, g["coord"], g["numbers"], mol, q, g["cell"], ewald_accuracy=1e-6)
    b = _run(hip_engine, g["coord"], g["numbers"], mol, q, g["cell"], ewald_accuracy=1e-9)
    assert abs(a["energy"][0] - b["energy"][0]) < 5e-5
    assert np.abs(a["forces"] - b["forces"]).max() < 2e-5 + 1e-4 * np.abs(a["forces"]).max()
    assert np.abs(a["stress"] - b["stress"]).max() < 1e-6
    assert int(hip_engine.last_status[7]) > 0
    dev = hip_engine.device
    c = hip_engine.eval(torch.from_numpy(g["coord"]).to(dev), torch.from_numpy(g["numbers"]).to(dev), torch.from_numpy(mol).to(dev),
                        torch.from_numpy(q).to(dev), cell=torch.from_numpy(g["cell"]).to(dev), forces=True, stress=True, coulomb="pme")
    assert abs(c["energy"].cpu().numpy()[0] - a["energy"][0]) < 5e-5


def test_supercell_vs_oracle_and_k_capacity(hip_engine, oracle32, oracle64):
    """768 atoms (many k vectors, several bins per axis); the k arrays start too small and grow to what the engine reports."""
    from aimnetcentral_amd import workloads

    c, z, cell = workloads.glucose_supercell((2, 2, 2))
    rng = np.random.default_rng(5)
    c = (c + rng.normal(0.0, 0.02, c.shape)).astype(np.float32)
    cell = cell.astype(np.float32)
    mol = np.zeros(len(z), dtype=np.int64)
    q = np.zeros(1, dtype=np.float32)
    hip_engine._ewald_max_k = 64
    res = _run(hip_engine, c, z, mol, q, cell)
    assert hip_engine._ewald_max_k >= int(hip_engine.last_status[7]) > 64
    okw = dict(cell=cell, coulomb="ewald", stress=True)
    ref = O.evaluate(oracle32, c, z, q, mol, **okw)
    e64 = O.evaluate(oracle64, c, z, q, mol, **dict(okw, forces=False, stress=False))["energy"]
    P.compare(res, ref, len(z), "768-atom supercell ewald/oracle", e64)


def test_rejections(hip_engine):
    g = golden("taxol")
    dev = hip_engine.device
    args = (torch.from_numpy(g["coord"]).to(dev), torch.from_numpy(g["numbers"]).to(dev), torch.zeros(113, dtype=torch.int64, device=dev),
            torch.zeros(1, device=dev))
    with pytest.raises(ValueError, match="periodic cell"):
        hip_engine.eval(*args, coulomb="ewald")
    p = golden("pbc96_dsf15")
    pargs = (torch.from_numpy(p["coord"]).to(dev), torch.from_numpy(p["numbers"]).to(dev), torch.zeros(96, dtype=torch.int64, device=dev),
             torch.zeros(1, device=dev))
    from aimnetcentral_amd._lib import HipLibraryError

    with pytest.raises(HipLibraryError, match="periodic along all three axes"):
        hip_engine.eval(*pargs, cell=torch.from_numpy(p["cell"]).to(dev), pbc=(True, True, False), coulomb="ewald")


def test_through_the_calculator_and_hvp_by_force_differences(oracle64):
    """set_lrcoulomb_method("ewald") -> the engine's method; Hessian-vector products fall back to differences of the analytic forces
    (the tangent sweep covers the pair-wise methods only), checked against differences of the fp64 oracle's forces."""
    from aimnetcentral_amd import AIMNet2Calculator, loader

    calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
    calc.set_lrcoulomb_method("ewald", ewald_accuracy=1e-7)
    g = golden("pbc96_dsf15")
    data = dict(coord=g["coord"], numbers=g["numbers"], charge=0.0, cell=g["cell"])
    out = calc(data, forces=True, stress=True)
    ref = O.evaluate(oracle64, g["coord"], g["numbers"], np.zeros(1, np.float32), cell=g["cell"], coulomb="ewald", ewald_accuracy=1e-7, stress=True)
    assert abs(float(out["energy"]) - ref["energy"][0]) < 1e-4
    assert_forces_close(out["forces"].cpu().numpy(), ref["forces"], "calculator ewald")
    assert np.abs(out["stress"].cpu().numpy() - ref["stress"]).max() <= STRESS_ATOL
    rng = np.random.default_rng(2)
    v = rng.normal(size=(1, 96, 3)).astype(np.float32)
    v /= np.abs(v).max()
    hv = calc.hessian_vector_product(data, v)
    hv = (hv["hvp"] if isinstance(hv, dict) else hv).cpu().numpy().reshape(96, 3)
    h = 1e-3
    xw = O.evaluate(oracle64, g["coord"], g["numbers"], np.zeros(1, np.float32), cell=g["cell"], coulomb="ewald", ewald_accuracy=1e-7,
                    return_intermediates=True)["coord_wrapped"].astype(np.float64)

    def f64(x):
        return O.evaluate(oracle64, x.astype(np.float32), g["numbers"], np.zeros(1, np.float32), cell=g["cell"], coulomb="ewald",
                          ewald_accuracy=1e-7)["forces"].astype(np.float64)

    fd = (f64(xw - h * v[0]) - f64(xw + h * v[0])) / (2 * h)
    assert np.abs(hv - fd).max() < 2e-3 * max(1.0, np.abs(fd).max())
    with pytest.raises(ValueError, match="requires a periodic 'cell'"):
        calc(dict(coord=g["coord"], numbers=g["numbers"], charge=0.0))


@pytest.mark.parametrize("seed", [1, 3, 5, 7, 9, 11, 13, 15, 19, 23, 27, 31])
def test_random_cells(seed, hip_engine, hip_engine_nse, oracle32, oracle32_nse, oracle64, oracle64_nse):
    """The periodic cases of the randomised sweep (tests/test_gpu_fuzz.py: strained triclinic cells, vacancies, atoms outside the
    box, one or two systems, charged, open-shell NSE models, with and without the external DFT-D3 term) with the Coulomb method
    switched to Ewald; gates as there."""
    import test_gpu_fuzz as Z

    case = Z.make_case(seed)
    c, z, mol, q, mult, nse, kw, okw, d3, label = case
    if not all(kw["pbc"]):
        pytest.skip("slab geometry: Ewald summation needs three periodic axes")
    acc = 1e-7
    kw = dict({k: v for k, v in kw.items() if not k.startswith("dsf")}, coulomb="ewald", ewald_accuracy=acc)
    okw = dict({k: v for k, v in okw.items() if not k.startswith("dsf")}, coulomb="ewald", ewald_accuracy=acc)
    case = (c, z, mol, q, mult, nse, kw, okw, d3, label + " [ewald]")
    eng, orc, orc64 = (hip_engine_nse, oracle32_nse, oracle64_nse) if nse else (hip_engine, oracle32, oracle64)
    res = Z.run_case(eng, case)
    ref = O.evaluate(orc, c, z, q, mol, return_intermediates=True, **okw)
    ref64 = O.evaluate(orc64, c, z, q, mol, return_intermediates=True, **dict(okw, forces=False, stress=False))
    wrapped = Z.run_case(eng, (ref["coord_wrapped"].astype(np.float32),) + case[1:])
    Z._compare(res, ref, ref64, mol, case[-1], nse, wrapped["energy"])


@pytest.mark.parametrize("seed", range(6))
def test_tiny_cells(seed, hip_engine, oracle32, oracle64):
    """Cells of 3-4.5 A with 2-9 atoms (tests/test_gpu_fuzz.py, tiny_cells): the real-space cutoff spans several images of the
    cell, the k boxes are a few dozen entries, every system of the batch has its own parameters."""
    import test_gpu_fuzz as Z

    c, z, mol, cell, n_sys, _ = Z.tiny_cells(seed)
    q = np.zeros(n_sys, dtype=np.float32)
    res = _run(hip_engine, c, z, mol, q, cell, ewald_accuracy=1e-7)
    okw = dict(cell=cell, coulomb="ewald", ewald_accuracy=1e-7, stress=True)
    ref = O.evaluate(oracle32, c, z, q, mol, return_intermediates=True, **okw)
    ref64 = O.evaluate(oracle64, c, z, q, mol, return_intermediates=True, **dict(okw, forces=False, stress=False))
    Z._compare(res, ref, ref64, mol, f"tiny cells + ewald seed {seed}", False)


def test_batch_of_many_small_cells(hip_engine, oracle32, oracle64):
    """Fourteen cells of 2-9 atoms in one batch: fourteen parameter sets and k-box slices (the offsets are a scan over the batch)."""
    import test_gpu_fuzz as Z

    cs, zs, mols, cells = [], [], [], []
    n_sys = 0
    for seed in range(6):
        c, z, mol, cell, n, _ = Z.tiny_cells(seed)
        cs.append(c)
        zs.append(z)
        mols.append(mol + n_sys)
        cells.append(cell.reshape(-1, 3, 3))
        n_sys += n
    c, z, mol, cell = np.concatenate(cs), np.concatenate(zs), np.concatenate(mols), np.concatenate(cells)
    q = np.zeros(n_sys, dtype=np.float32)
    q[3] = 1.0
    res = _run(hip_engine, c, z, mol, q, cell)
    okw = dict(cell=cell, coulomb="ewald", stress=True)
    ref = O.evaluate(oracle32, c, z, q, mol, return_intermediates=True, **okw)
    ref64 = O.evaluate(oracle64, c, z, q, mol, return_intermediates=True, **dict(okw, forces=False, stress=False))
    Z._compare(res, ref, ref64, mol, f"{n_sys} tiny cells + ewald", False)
    assert int(hip_engine.last_status[7]) % 8 == 0
