"""BASELINE.json configurations at FULL size against the oracle / reference goldens:
config 2 (256 molecules of 20-60 atoms, flat and padded-dense), config 4 on the aimnet2_rxn architecture, config 5 (one
rank's shard: 128 frames x 50 atoms), the ASE / TorchSim adapters through the real engine, and a "cold" fixture on which
the reference's un-widened 1e-5 eV energy gate holds."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import CHARGE_ATOL, STRESS_ATOL, assert_forces_close, energy_tol, golden
from aimnetcentral_amd import workloads
from oracle import aimnet2_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def calc():
    from aimnetcentral_amd import AIMNet2Calculator, loader

    return AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")


def npy(out):
    return {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in out.items()}


def check_batch(out_e, out_f, out_q, ref, ref64, mol, what):
    """Forces and charges at the reference's gates against the fp32 oracle.  Energies: these random geometries are hot
    (contacts of 0.9 A, max|F| ~ 800 eV/A, per-atom energies up to 85 eV on the synthetic weights); the fp32 ORACLE itself
    sits up to 1.2e-3 eV from the fp64 energy of a 50-atom molecule, and its per-atom errors are one-signed within a molecule,
    not a random walk (tests/tools/cfg5_diag.py: engine and fp32 oracle both -3e-4 .. -1.2e-3 eV on the worst frames).
    So the engine is held against the fp64 oracle (i) in the rms over the batch - its fp32 noise may exceed the reference
    path's own by at most a factor 1.5 (measured 1.07) - and (ii) per molecule at the reference's gate + the fp32 oracle's
    own distance from fp64 + twice the coherent sum of the fp32 oracle's per-atom errors."""
    mol = np.asarray(mol)
    sizes = np.bincount(mol)
    assert np.isfinite(out_e).all(), what
    err = np.abs(out_e - ref64["energy"])
    err_ref = np.abs(ref["energy"] - ref64["energy"])
    rms, rms_ref = np.sqrt(np.mean(err**2)), np.sqrt(np.mean(err_ref**2))
    assert rms <= 1.5 * rms_ref + 1e-5, f"{what}: rms energy error {rms:.3e} vs the fp32 oracle's {rms_ref:.3e}"
    d = np.abs(ref["_e_atom"][: len(mol)].astype(np.float64) - ref64["_e_atom"][: len(mol)])
    l1 = np.zeros(len(sizes))
    np.add.at(l1, mol, d)
    gate = energy_tol(sizes) + err_ref + 2.0 * l1
    assert (err <= gate).all(), f"{what}: energy |hip - fp64| / gate = {np.max(err / gate):.2f} (molecule {int(np.argmax(err / gate))})"
    assert_forces_close(out_f, ref["forces"], what)
    assert np.abs(out_q - ref["charges"]).max() <= CHARGE_ATOL, what


def test_config2_full_size_flat_and_padded_dense(calc, oracle32, oracle64):
    """256 random neutral organics of 20-60 atoms (10.5 k atoms): flat mol_idx input and the (256, 60, 3) zero-padded dense
    input of the GPU configuration, both against the oracle on the flat layout (the engine compacts the padding away)."""
    c, z, mol, q = workloads.random_batch(256, 20, 60, seed=2)
    ref = O.evaluate(oracle32, c, z, q, mol, coulomb="simple", return_intermediates=True)
    e64 = O.evaluate(oracle64, c, z, q, mol, coulomb="simple", forces=False, return_intermediates=True)
    flat = npy(calc({"coord": c, "numbers": z, "mol_idx": mol, "charge": q}, forces=True))
    check_batch(flat["energy"], flat["forces"], flat["charges"], ref, e64, mol, "config 2 flat")
    cp, zp = workloads.pad_batch(c, z, mol, 256)
    assert cp.shape == (256, 60, 3)
    dense = npy(calc({"coord": cp, "numbers": zp, "charge": q}, forces=True))
    assert dense["forces"].shape == (256, 60, 3) and dense["energy"].shape == (256,)
    real = zp > 0
    check_batch(dense["energy"], dense["forces"][real], dense["charges"][real], ref, e64, mol, "config 2 padded dense")
    assert (dense["forces"][~real] == 0).all() and (dense["charges"][~real] == 0).all()


def test_config5_one_rank_shard(calc, oracle32, oracle64):
    """1024 frames x 50 atoms over 8 ranks: rank 0's 128 frames, through the same sharding helpers bench.py uses."""
    from aimnetcentral_amd import dist as adist

    c, z, mol, q = workloads.random_batch(1024, 50, 50, seed=5)
    sizes = np.bincount(mol, minlength=1024)
    a, b = adist.shard_frames(sizes, 8)[0]
    assert b - a == 128
    c, z, mol, q = adist.local_batch(c, z, mol, q, a, b)
    ref = O.evaluate(oracle32, c, z, q, mol, coulomb="simple", return_intermediates=True)
    e64 = O.evaluate(oracle64, c, z, q, mol, coulomb="simple", forces=False, return_intermediates=True)
    out = npy(calc({"coord": c, "numbers": z, "mol_idx": mol, "charge": q}, forces=True))
    check_batch(out["energy"], out["forces"], out["charges"], ref, e64, np.asarray(mol), "config 5 shard")


def test_config4_on_the_rxn_architecture():
    """aimnet2_rxn.yaml (Dipole / Quadrupole output modules next to the energy head) through the loader, then Hessian and
    Hessian-vector products against the reference's double backward on the same artifact (tests/golden/hvp40_rxn.npz)."""
    from aimnetcentral_amd import AIMNet2Calculator, loader

    from aimnetcentral_amd import synth

    assert "aimnet.modules.Dipole" in synth.rxn_yaml() and "aimnet.modules.Quadrupole" in synth.rxn_yaml()
    spec = loader.synthetic_spec(0, rxn=True)
    calc = AIMNet2Calculator(spec, device="cuda:0")
    g = golden("hvp40_rxn")
    data = {"coord": g["coord"], "numbers": g["numbers"], "charge": 0.0}
    out = npy(calc(data, forces=True, hessian=True))
    assert abs(out["energy"][0] - g["energy"][0]) <= energy_tol(40)
    assert_forces_close(out["forces"], g["forces"], "rxn forces")
    H = out["hessian"].reshape(120, 120)
    # Hessian and H v (analytic tangent sweep) at the reference's own gate for its double backward, elementwise
    # allclose(rtol=1e-3, atol=1e-3) (tests/test_hvp.py:75), and at the 1e-4 eV/A^2 of two fp32 evaluations of one operator
    Href = g["hessian"].reshape(120, 120)
    assert np.allclose(H, Href, rtol=1e-3, atol=1e-3) and np.abs(H - Href).max() <= 1e-4
    hv1 = calc.hessian_vector_product(data, g["v1"]).cpu().numpy()
    hv4 = calc.hessian_vector_product(data, g["v4"]).cpu().numpy()
    assert hv1.shape == (40, 3) and hv4.shape == (4, 40, 3)
    for got, ref in ((hv1, g["hv1"]), (hv4, g["hv4"])):
        assert np.allclose(got, ref, rtol=1e-3, atol=1e-3)
        assert np.abs(got - ref).max() <= 1e-4 + 1e-5 * np.abs(ref).max(), np.abs(got - ref).max()


def test_cold_fixture_holds_the_unwidened_energy_gate(calc):
    """A geometry relaxed on the synthetic surface (max|F| = 0.3 eV/A): engine vs the reference golden at the reference's own
    |dE| < 1e-5 eV (tests/test_calculator_gpu.py:445) with NO fp64-anchored slack, forces to 1e-5 + 1e-4 max|F|."""
    g = golden("cold24")
    assert np.abs(g["forces"]).max() < 0.5
    out = npy(calc({"coord": g["coord"], "numbers": g["numbers"], "charge": 0.0}, forces=True))
    assert abs(out["energy"][0] - g["energy"][0]) < 1e-5
    assert np.abs(out["forces"] - g["forces"]).max() <= 1e-5 + 1e-4 * np.abs(g["forces"]).max()
    assert np.abs(out["charges"] - g["charges"]).max() <= CHARGE_ATOL


# ---- adapters through the real engine (duck-typed ase.Atoms / torch_sim SimState: neither package is installed) ----------
class _Atoms:
    def __init__(self, numbers, positions, cell=None, pbc=(False, False, False), info=None):
        self.numbers, self.positions = np.asarray(numbers), np.asarray(positions, dtype=float)
        self.cell, self.pbc, self.info = cell, np.asarray(pbc), dict(info or {})

    def copy(self):
        return _Atoms(self.numbers.copy(), self.positions.copy(), None if self.cell is None else np.array(self.cell), self.pbc.copy(), self.info)

    def __len__(self):
        return len(self.numbers)


class _State:
    def __init__(self, positions, numbers, system_idx, cell, pbc, n_systems, **extras):
        self.positions, self.atomic_numbers, self.system_idx = positions, numbers, system_idx
        self.row_vector_cell, self.pbc, self.n_systems = cell, pbc, n_systems
        self.device, self.dtype = positions.device, torch.float32
        for k, v in extras.items():
            setattr(self, k, v)


def test_ase_adapter_on_the_gpu_matches_the_reference_goldens(calc):
    """AIMNet2ASE.calculate (aimnet2ase.py:238-263 of the reference) on taxol and on the periodic DSF cell."""
    from aimnetcentral_amd.aimnet2ase import AIMNet2ASE

    g = golden("taxol")
    ase_calc = AIMNet2ASE(calc, charge=0)
    atoms = _Atoms(g["numbers"], g["coord"])
    ase_calc.calculate(atoms, properties=["energy", "forces"])
    r = ase_calc.results
    assert abs(r["energy"] - g["energy"][0]) <= energy_tol(113)
    assert_forces_close(r["forces"], g["forces"], "ase taxol")
    assert np.abs(r["charges"] - g["charges"]).max() <= CHARGE_ATOL
    # an MD-like second step: moved positions, cached device inputs, results change
    moved = atoms.copy()
    moved.positions = moved.positions + 0.01 * np.sin(np.arange(339).reshape(113, 3))
    ase_calc.reset()
    ase_calc.calculate(moved, properties=["energy", "forces"])
    assert abs(ase_calc.results["energy"] - r["energy"]) > 1e-6
    p = golden("pbc96_dsf15")
    calc.set_lrcoulomb_method("dsf", cutoff=15.0, dsf_alpha=0.2)
    try:
        patoms = _Atoms(p["numbers"], p["coord"], cell=p["cell"], pbc=(True, True, True))
        pc = AIMNet2ASE(calc, charge=0)
        pc.calculate(patoms, properties=["energy", "forces", "stress"])
        rr = pc.results
        assert abs(rr["energy"] - p["energy"][0]) <= 1e-4
        assert_forces_close(rr["forces"], p["forces"], "ase pbc96")
        s = np.asarray(rr["stress"])
        ref_s = p["stress"].reshape(3, 3)
        if s.shape == (6,):  # Voigt
            ref_s = np.array([ref_s[0, 0], ref_s[1, 1], ref_s[2, 2], ref_s[1, 2], ref_s[0, 2], ref_s[0, 1]])
        assert np.abs(s - ref_s).max() <= STRESS_ATOL
    finally:
        calc.set_lrcoulomb_method("simple")


def test_torchsim_adapter_on_the_gpu(calc, oracle32, oracle64):
    """AIMNet2TorchSim.forward (aimnet2torchsim.py:106-144): a flat multi-system state of ragged molecules, and a periodic one."""
    from aimnetcentral_amd import AIMNet2TorchSim

    dev = torch.device("cuda:0")
    c, z, mol, q = workloads.random_batch(5, 9, 23, seed=4)
    st = _State(torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.from_numpy(mol).to(dev), torch.zeros(5, 3, 3, device=dev),
                False, 5, charge=torch.from_numpy(q).to(dev))
    model = AIMNet2TorchSim(calc, compute_forces=True)
    out = model(st)
    ref = O.evaluate(oracle32, c, z, q, mol, coulomb="simple", return_intermediates=True)
    e64 = O.evaluate(oracle64, c, z, q, mol, coulomb="simple", forces=False, return_intermediates=True)
    check_batch(out["energy"].cpu().numpy(), out["forces"].cpu().numpy(), out["charges"].cpu().numpy(), ref, e64, mol, "torchsim batch")
    p = golden("pbc96_dsf15")
    calc.set_lrcoulomb_method("dsf", cutoff=15.0, dsf_alpha=0.2)
    try:
        stp = _State(torch.from_numpy(p["coord"]).to(dev), torch.from_numpy(p["numbers"]).to(dev), torch.zeros(96, dtype=torch.int64, device=dev),
                     torch.from_numpy(p["cell"]).to(dev).view(1, 3, 3), True, 1)
        pm = AIMNet2TorchSim(calc, compute_forces=True, compute_stress=True)
        o = pm(stp)
        assert abs(float(o["energy"].cpu()[0]) - p["energy"][0]) <= 1e-4
        assert_forces_close(o["forces"].cpu().numpy(), p["forces"], "torchsim pbc96")
        assert np.abs(o["stress"].cpu().numpy().reshape(3, 3) - p["stress"].reshape(3, 3)).max() <= STRESS_ATOL
    finally:
        calc.set_lrcoulomb_method("simple")


def test_deferred_status_and_capacity_hysteresis(oracle32):
    """Device-resident stepping: evaluations enqueued with defer_status=True do no host read; check_status()
    verifies them in one go, grows the row capacity and raises on a neighbour overflow; the synchronous path shrinks an
    under-used capacity by the AdaptiveNeighborList rule (neighbors.py:135-139)."""
    from aimnetcentral_amd import AIMNet2Calculator, loader
    from aimnetcentral_amd.engine import NeighborOverflowError

    calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
    eng = calc.engine
    g = golden("taxol")
    data = {"coord": torch.from_numpy(g["coord"]).cuda(), "numbers": torch.from_numpy(g["numbers"]).cuda(), "charge": torch.zeros(1).cuda()}
    w = npy(calc({"coord": [[0.0, 0.0, 0.1173], [0.0, 0.7572, -0.4692], [0.0, -0.7572, -0.4692]], "numbers": [8, 1, 1], "charge": 0.0}))
    assert np.isfinite(w["energy"]).all() and eng.max_nb == 16  # water: 2 neighbours -> 112 shrinks to the floor of 16
    sync = npy(calc(data, forces=True))  # taxol, up to 62 neighbours: grows back by the x1.5 retry rule (16 -> 32 -> 48 -> 80)
    assert 62 <= eng.max_nb <= 112
    outs = [calc.eval(data, forces=True, defer_status=True) for _ in range(5)]
    assert len(eng.pending_status) == 5 and "status" not in sync
    calc.check_status()
    assert not eng.pending_status
    assert np.array_equal(outs[-1]["forces"].cpu().numpy(), sync["forces"])  # same kernels, same (shrunk) capacity
    eng.max_nb = 16  # too small for taxol: the deferred evaluation overflows, nobody notices until the check
    calc.eval(data, forces=True, defer_status=True)
    with pytest.raises(NeighborOverflowError, match="repeat them"):
        calc.check_status()
    assert eng.max_nb >= 24
    again = npy(calc(data, forces=True))  # the synchronous path retries by itself and lands on the same answer
    assert np.array_equal(again["forces"], sync["forces"])


def test_per_system_pbc_flags(oracle32, oracle64):
    """normalize_pbc (neighbors.py:309-321): `pbc` of shape (B, 3) - two copies of the glucose cell, one fully periodic, one a slab
    (periodic in x and z only) - against the oracle with the same per-system flags, forces and stress included."""
    from aimnetcentral_amd import AIMNet2Calculator, loader

    calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
    c, z, cell = workloads.glucose_cell()
    cc = np.concatenate([c, c + 0.01]).astype(np.float32)
    zz = np.concatenate([z, z])
    mol = np.repeat(np.arange(2), len(z))
    cells = np.stack([cell, cell]).astype(np.float32)
    pbc = np.array([[True, True, True], [True, False, True]])
    calc.set_lrcoulomb_method("dsf", cutoff=9.0, dsf_alpha=0.25)
    out = npy(calc({"coord": cc, "numbers": zz, "mol_idx": mol, "charge": np.zeros(2, np.float32), "cell": cells, "pbc": pbc},
                   forces=True, stress=True))
    kw = dict(cell=cells, pbc=pbc, coulomb="dsf", dsf_rc=9.0, dsf_alpha=0.25)
    ref = O.evaluate(oracle32, cc, zz, np.zeros(2, np.float32), mol, stress=True, **kw)
    e64 = O.evaluate(oracle64, cc, zz, np.zeros(2, np.float32), mol, forces=False, **kw)["energy"]
    tol = energy_tol(96) + np.abs(ref["energy"] - e64)
    assert (np.abs(out["energy"] - ref["energy"]) <= tol).all()
    assert_forces_close(out["forces"], ref["forces"], "per-system pbc")
    assert np.abs(out["charges"] - ref["charges"]).max() <= CHARGE_ATOL
    assert np.abs(out["stress"] - ref["stress"]).max() <= STRESS_ATOL
    same = npy(calc({"coord": c.astype(np.float32), "numbers": z, "charge": 0.0, "cell": cell.astype(np.float32)}, forces=True))
    assert abs(same["energy"][0] - out["energy"][0]) < 1e-4 and abs(out["energy"][1] - out["energy"][0]) > 1e-3


def test_periodic_2304_atoms_engine_vs_oracle(oracle32):
    """The largest direct engine-vs-oracle comparison of the periodic path: the (2,3,4) supercell of the config-3 crystal with a
    thermal jitter, 2 304 atoms (the one-wave-per-atom conv kernels, the reverse-pair backward, the bf16x3-split GEMMs on their
    large tiles, the cell-walk lists and the list-free DSF), energy + forces + charges + stress at the plain gates - config 3 at
    10 080 atoms is otherwise only held to periodic-image properties of the 96-atom golden."""
    from aimnetcentral_amd import loader, workloads
    from aimnetcentral_amd.engine import HipEngine
    from oracle import aimnet2_oracle as O

    c, z, cell = workloads.glucose_supercell((2, 3, 4))
    rng = np.random.default_rng(11)
    c = (c + rng.normal(0.0, 0.02, c.shape)).astype(np.float32)
    cell32 = cell.astype(np.float32)
    mol = np.zeros(len(z), dtype=np.int64)
    pbc = np.ones(3, dtype=bool)
    xw = O.wrap_into_cell(c, cell32, mol, pbc)
    nb, sh = O.neighbor_list_fast(xw, 5.0, mol, cell, pbc)
    nbl, shl = O.neighbor_list_fast(xw, 15.0, mol, cell, pbc)
    ref = O.evaluate(oracle32, coord=xw, numbers=z, charge=np.zeros(1, np.float32), mol_idx=mol, cell=cell32, coulomb="dsf", stress=True,
                     nbmat=nb, shifts=sh, nbmat_lr=nbl, shifts_lr=shl)
    eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
    dev = eng.device
    r = eng.eval(torch.from_numpy(xw).to(dev), torch.from_numpy(z).to(dev), torch.from_numpy(mol).to(dev), torch.zeros(1, device=dev),
                 cell=torch.from_numpy(cell32).to(dev), forces=True, stress=True, coulomb="dsf")
    r = {k: v.cpu().numpy() for k, v in r.items()}
    assert abs(r["energy"][0] - ref["energy"][0]) <= energy_tol(len(z))
    assert_forces_close(r["forces"], ref["forces"], "pbc2304")
    assert np.abs(r["charges"] - ref["charges"]).max() <= CHARGE_ATOL
    assert np.abs(r["stress"] - ref["stress"]).max() <= 1e-5


def test_config2_cold_variant_holds_the_unwidened_gates(calc, oracle64):
    """BASELINE config 2 on geometries where the headline tolerance means something: 256 molecules of 20-60 atoms (15 relaxed
    fragments in 256 rigid placements, tests/golden/make_golden.py G12; outputs of the unmodified reference), one flat batch of
    10 k atoms - so the MLP GEMMs take the bf16x3-split kernels on their large tiles.
    EVERY molecule has to hold the reference's gate |dE| <= max(1e-5, 5e-7 n) eV, un-widened, against the fp64 oracle (measured:
    worst 0.80 of the gate, rms 3.9e-6 eV; the reference's own fp32 golden: 0.75, 5.3e-6; the exact-fp32 GEMM kernels: 1.05,
    6.6e-6 - tests/tools/relaxed256_probe.py).  Against the reference's fp32 golden - two fp32 computations, each that far from the
    truth - the gate is widened by the golden's own distance from fp64 and nothing else (worst measured 1.24 plain gates, one
    molecule of 256 above 1), and the rms difference must stay below the smallest gate.  Charges 1e-4.  Forces: on a relaxed set
    max|F| is 0.7 eV/A, so the reference's 1e-5 + 1e-4 max|F| = 8e-5 eV/A lies BELOW the fp32 noise of every implementation (the
    reference's golden is 1.1e-4 from the fp64 forces, the fp32 oracle 1.0e-4, the engine 9.8e-5, its exact-fp32 GEMM mode 1.5e-4):
    the engine's distance from the fp64 forces may exceed the golden's own by at most 1.5x, in the maximum and in the rms, and its
    distance from the golden that gate plus the golden's own error.  The same for the 3-D padded (dense) layout."""
    g = golden("relaxed256")
    mol = g["mol_idx"].astype(np.int64)
    numbers = g["numbers"].astype(np.int64)
    sizes = np.bincount(mol)
    assert len(sizes) == 256 and sizes.min() >= 20 and sizes.max() <= 60 and np.abs(g["forces"]).max() < 1.0
    r64 = O.evaluate(oracle64, g["coord"], numbers, g["charge"], mol, forces=True)
    e64, f64, fg = r64["energy"], r64["forces"].astype(np.float64), g["forces"].astype(np.float64)
    fg_max, fg_rms = np.abs(fg - f64).max(), np.sqrt(np.mean((fg - f64) ** 2))
    f_gate = 1e-5 + 1e-4 * np.abs(fg).max()

    def check_forces(f, f64_, fg_, what):
        f = f.astype(np.float64)
        assert np.abs(f - f64_).max() <= 1.5 * fg_max and np.sqrt(np.mean((f - f64_) ** 2)) <= 1.5 * fg_rms, \
            f"{what}: |hip - fp64| max {np.abs(f - f64_).max():.2e} rms {np.sqrt(np.mean((f - f64_) ** 2)):.2e} vs the golden's {fg_max:.2e} / {fg_rms:.2e}"
        assert np.abs(f - fg_).max() <= f_gate + fg_max, what

    gate = np.maximum(1e-5, 5e-7 * sizes)
    gold_off = np.abs(g["energy"] - e64)
    assert (gold_off <= gate).all()  # the fixture itself is inside the gate

    def check(energy, what):
        err64, errg = np.abs(energy - e64), np.abs(energy - g["energy"])
        assert (err64 <= gate).all(), f"{what}: worst |hip - fp64| / gate = {np.max(err64 / gate):.2f} (molecule {int(np.argmax(err64 / gate))})"
        assert (errg <= gate + gold_off).all(), f"{what}: worst |hip - golden| / (gate + |golden - fp64|) = {np.max(errg / (gate + gold_off)):.2f}"
        assert np.sqrt(np.mean(errg**2)) <= 1e-5, what

    out = npy(calc({"coord": g["coord"], "numbers": numbers, "mol_idx": mol, "charge": g["charge"]}, forces=True))
    check(out["energy"], "flat")
    check_forces(out["forces"], f64, fg, "relaxed256 flat")
    assert np.abs(out["charges"] - g["charges"]).max() <= CHARGE_ATOL
    # the dense (B, Nmax, 3) layout of the same molecules
    nmax = int(sizes.max())
    starts = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    c3 = np.zeros((256, nmax, 3), dtype=np.float32)
    z3 = np.zeros((256, nmax), dtype=np.int64)
    for m in range(256):
        c3[m, : sizes[m]] = g["coord"][starts[m] : starts[m] + sizes[m]]
        z3[m, : sizes[m]] = numbers[starts[m] : starts[m] + sizes[m]]
    o3 = npy(calc({"coord": c3, "numbers": z3, "charge": g["charge"]}, forces=True))
    check(o3["energy"], "dense")
    f3 = np.concatenate([o3["forces"][m, : sizes[m]] for m in range(256)])
    check_forces(f3, f64, fg, "relaxed256 dense")
