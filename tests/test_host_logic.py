"""Host-side calculator logic (layouts, validation, warnings, Coulomb switching) with a recording
fake in place of the HIP engine - the reference does the same with TinyLegacyModel /
RecordingExternalCoulomb fakes (tests/test_calculator.py:15-196).  CPU only."""
from __future__ import annotations

import warnings

import numpy as np
import pytest
import torch

from aimnetcentral_amd import calculator as calc_mod
from aimnetcentral_amd import loader, workloads

import pathlib

GOLDEN = pathlib.Path(__file__).parent / "golden"


class FakeEngine:
    """Records eval() calls and returns recognisable tensors; stands in for HipEngine."""

    def __init__(self, spec, device):
        self.spec, self.device, self.calls, self.d3_tables = spec, torch.device("cpu"), [], None
        self.nq = int(getattr(spec, "num_charge_channels", 1))

    def set_dftd3_tables(self, tables):
        self.d3_tables = tables

    def eval(self, coord, numbers, mol_idx, charge, cell=None, pbc=(True, True, True), forces=False, stress=False,
             coulomb="simple", dsf_rc=15.0, dsf_alpha=0.2, ewald_accuracy=1e-6, sync=True, dftd3=None, host_out=False, **lists):
        self.calls.append(dict(lists=lists, host_out=host_out, n=coord.shape[0], n_mol=charge.shape[0], coulomb=coulomb, dsf_rc=dsf_rc, dsf_alpha=dsf_alpha, dftd3=dftd3,
                               ewald_accuracy=ewald_accuracy,
                               pbc=pbc, cell=None if cell is None else tuple(cell.shape), mol_idx=mol_idx.clone(),
                               numbers=numbers.clone(), charge=charge.clone()))
        n = coord.shape[0]
        out = {"energy": torch.arange(charge.shape[0], dtype=torch.float64), "charges": torch.arange(n, dtype=torch.float32) + 1}
        if self.nq == 2:
            assert charge.ndim == 2 and charge.shape[1] == 2
            out["spin_charges"] = -(torch.arange(n, dtype=torch.float32) + 1)
        if forces:
            out["forces"] = torch.ones(n, 3) * (torch.arange(n, dtype=torch.float32) + 1).unsqueeze(-1)
        if stress:
            out["stress"] = torch.zeros(3, 3) if cell is not None and cell.ndim == 2 else torch.zeros(charge.shape[0], 3, 3)
        return out


@pytest.fixture()
def calc(monkeypatch):
    monkeypatch.setattr(calc_mod, "HipEngine", FakeEngine)
    monkeypatch.setattr(torch, "as_tensor", _as_tensor_cpu(torch.as_tensor))
    c = calc_mod.AIMNet2Calculator(loader.synthetic_spec(0), device="cuda")
    c.device = "cpu"
    return c


def _as_tensor_cpu(orig):
    def f(data, dtype=None, device=None):
        return orig(data, dtype=dtype, device="cpu")

    return f


WATER = dict(coord=[[0.0, 0.0, 0.1173], [0.0, 0.7572, -0.4692], [0.0, -0.7572, -0.4692]], numbers=[8, 1, 1], charge=0.0)


def test_cpu_device_is_refused():
    from aimnetcentral_amd import HipLibraryError

    with pytest.raises(HipLibraryError):
        calc_mod.AIMNet2Calculator(loader.synthetic_spec(0), device="cpu")


def test_missing_keys_raise_keyerror(calc):
    for k in ("coord", "numbers", "charge"):
        d = dict(WATER)
        d.pop(k)
        with pytest.raises(KeyError, match=f"Missing key {k} in the input data"):
            calc(d, validate_species=False)


def test_flat_single_molecule(calc):
    out = calc(WATER, forces=True)
    call = calc.engine.calls[-1]
    assert call["n"] == 3 and call["n_mol"] == 1 and call["coulomb"] == "simple" and call["cell"] is None
    assert out["energy"].shape == (1,) and out["energy"].dtype == torch.float64
    assert out["charges"].shape == (3,) and out["forces"].shape == (3, 3)
    assert set(out) == {"energy", "charges", "forces"}


def test_dense_padded_batch_is_compacted_and_restored(calc):
    c, z, mol, q = workloads.random_batch(3, 4, 7, seed=1)
    cp, zp = workloads.pad_batch(c, z, mol, 3)
    out = calc({"coord": cp, "numbers": zp, "charge": q}, forces=True)
    call = calc.engine.calls[-1]
    assert call["n"] == len(z) and call["n_mol"] == 3
    assert torch.equal(call["mol_idx"].long(), torch.from_numpy(mol))
    assert torch.equal(call["numbers"].long(), torch.from_numpy(z))
    B, N = zp.shape
    assert out["forces"].shape == (B, N, 3) and out["charges"].shape == (B, N)
    pad = torch.from_numpy(zp == 0)
    assert (out["charges"][pad] == 0).all() and (out["forces"][pad] == 0).all()
    assert (out["charges"][~pad] > 0).all()


def test_species_and_charge_validation(calc):
    bad = dict(WATER, numbers=[8, 1, 2])
    with pytest.raises(ValueError, match=r"Atomic numbers \[2\]"):
        calc(bad)
    calc(bad, validate_species=False)
    calc._metadata["supports_charged_systems"] = False
    with pytest.raises(ValueError, match="net-charged"):
        calc(dict(WATER, charge=1.0))
    calc(dict(WATER, charge=1.0), validate_species=False)


def test_mult_warning_once(calc):
    with pytest.warns(UserWarning, match="mult"):
        calc(dict(WATER, mult=2.0))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        calc(dict(WATER, mult=2.0))


def test_pbc_switches_to_dsf_for_one_eval_and_restores(calc):
    cell = np.eye(3, dtype=np.float32) * 12.0
    with pytest.warns(UserWarning, match="Switching to DSF Coulomb for PBC"):
        out = calc(dict(WATER, cell=cell), forces=True, stress=True)
    call = calc.engine.calls[-1]
    assert call["coulomb"] == "dsf" and call["dsf_rc"] == 15.0 and call["cell"] == (3, 3)
    assert out["stress"].shape == (3, 3)
    assert calc.coulomb_method == "simple" and calc.coulomb_cutoff == float("inf")
    calc.set_lrcoulomb_method("dsf", cutoff=9.0, dsf_alpha=0.25)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        calc(dict(WATER, cell=cell), forces=True)
    call = calc.engine.calls[-1]
    assert call["dsf_rc"] == 9.0 and call["dsf_alpha"] == 0.25 and calc.coulomb_method == "dsf"


def test_ewald_methods_need_a_cell_and_carry_their_accuracy(calc):
    """set_lrcoulomb_method("ewald" | "pme") (calculator.py:638-727): no fixed cutoff, the accuracy travels to the engine; without a
    cell the evaluation raises the reference's ValueError (calculator.py:1063-1068)."""
    calc.set_lrcoulomb_method("ewald", cutoff=9.0, ewald_accuracy=1e-7)
    assert calc.coulomb_method == "ewald" and calc.coulomb_cutoff is None and calc.cutoff_lr is None
    with pytest.raises(ValueError, match="requires a periodic 'cell'"):
        calc(WATER)
    cell = np.eye(3, dtype=np.float32) * 12.0
    calc(dict(WATER, cell=cell), forces=True)
    call = calc.engine.calls[-1]
    assert call["coulomb"] == "ewald" and call["ewald_accuracy"] == 1e-7
    calc.set_lrcoulomb_method("pme")  # served by the mesh kernels of csrc/pme.hip (round 5); no fixed cutoff either
    assert calc.coulomb_method == "pme" and calc.coulomb_cutoff is None
    calc(dict(WATER, cell=cell))
    assert calc.engine.calls[-1]["coulomb"] == "pme" and calc.engine.calls[-1]["ewald_accuracy"] == 1e-6
    calc.set_lrcoulomb_method("simple")
    assert calc.coulomb_cutoff == float("inf")


def test_setters_and_unsupported_paths(calc):
    with pytest.raises(ValueError, match="Invalid method"):
        calc.set_lrcoulomb_method("nope")
    with pytest.raises(AssertionError):
        calc(WATER, stress=True)
    x = torch.tensor(WATER["coord"], requires_grad=True)
    with pytest.raises(NotImplementedError, match="requires grad"):
        calc(dict(WATER, coord=x))
    assert calc.has_external_coulomb and not calc.has_external_dftd3 and not calc.is_nse
    assert calc.cutoff == 5.0 and calc.metadata["coulomb_mode"] == "sr_embedded"
    with pytest.raises(TypeError):
        calc.metadata["cutoff"] = 1.0


def test_dispersion_wiring(monkeypatch, tmp_path):
    """needs_dispersion metadata -> external DFT-D3 state, table lookup rules and per-eval options
    (calculator.py:234-247,752-783 of the reference)."""
    monkeypatch.setattr(calc_mod, "HipEngine", FakeEngine)
    monkeypatch.delenv("AIMNET_DFTD3_DATA", raising=False)
    spec = loader.synthetic_spec(0)
    spec.metadata = dict(spec.metadata, needs_dispersion=True, d3_params={"s8": 0.39, "a1": 0.57, "a2": 3.1, "s6": 1.0})
    with pytest.raises(FileNotFoundError, match="DFT-D3 reference tables"):
        calc_mod.AIMNet2Calculator(spec, device="cuda")            # no table file anywhere: loud, no silent skip
    c0 = calc_mod.AIMNet2Calculator(spec, device="cuda", needs_dispersion=False)
    assert not c0.has_external_dftd3 and c0.engine.d3_tables is None
    tables = dict(np.load(str(GOLDEN / "dftd3_subset.npz")))
    c1 = calc_mod.AIMNet2Calculator(spec, device="cuda", dftd3_data=tables)
    assert c1.has_external_dftd3 and c1.engine.d3_tables["c6ab"].shape == (18, 18, 5, 5)
    assert (c1.external_dftd3.s8, c1.external_dftd3.smoothing_on, c1.external_dftd3.smoothing_off) == (0.39, 12.0, 15.0)
    monkeypatch.setattr(torch, "as_tensor", _as_tensor_cpu(torch.as_tensor))
    c1.device = "cpu"
    c1(WATER)
    assert c1.engine.calls[-1]["dftd3"] == {"s6": 1.0, "s8": 0.39, "a1": 0.57, "a2": 3.1, "cutoff": 15.0, "smoothing_fraction": 0.2}
    c1.set_dftd3_cutoff(9.0, 0.25)
    c1(WATER)
    assert c1.engine.calls[-1]["dftd3"]["cutoff"] == 9.0 and c1.external_dftd3.smoothing_on == 6.75 and c1.dftd3_cutoff == 9.0
    c1.set_dftd3_cutoff()
    assert c1.dftd3_cutoff == 15.0 and c1.external_dftd3.smoothing_on == 12.0
    # the packed legacy layout [Z,Z,5,5,3] and the env-var / file route
    packed = np.stack([tables["c6ab"], tables["cn_ref"], tables["cn_ref"]], axis=-1)
    path = tmp_path / "d3.npz"
    np.savez(path, c6ab=packed, rcov=tables["rcov"], r4r2=tables["r4r2"])
    monkeypatch.setenv("AIMNET_DFTD3_DATA", str(path))
    t2 = loader.load_dftd3_tables()
    assert np.array_equal(t2["cn_ref"], tables["cn_ref"]) and t2["c6ab"].shape == (18, 18, 5, 5)
    with pytest.raises(ValueError, match="malformed"):
        loader.load_dftd3_tables({"c6ab": tables["c6ab"][:5], "cn_ref": tables["cn_ref"], "rcov": tables["rcov"], "r4r2": tables["r4r2"]})
    spec.metadata = dict(spec.metadata, d3_params=None)
    with pytest.raises(ValueError, match="d3_params"):
        calc_mod.AIMNet2Calculator(spec, device="cuda", needs_dispersion=True, dftd3_data=tables)


def test_workloads_are_deterministic_and_sized():
    c, z, cell = workloads.glucose_supercell((7, 3, 5))
    assert c.shape == (10080, 3) and sorted(set(z.tolist())) == [1, 6, 8]
    assert abs(abs(np.linalg.det(cell)) - 739.36 * 105) / (739.36 * 105) < 1e-3
    c1 = workloads.random_batch(4, 20, 60, seed=2)
    c2 = workloads.random_batch(4, 20, 60, seed=2)
    assert all(np.array_equal(a, b) for a, b in zip(c1, c2))
    d = np.linalg.norm(c1[0][:, None] - c1[0][None], axis=-1) + np.eye(len(c1[0])) * 9
    same = c1[2][:, None] == c1[2][None]
    assert d[same].min() >= 0.9


class _Atoms:
    def __init__(self, numbers, positions, cell=None, pbc=(False, False, False), info=None):
        self.numbers, self.positions = np.asarray(numbers), np.asarray(positions, dtype=float)
        self.cell, self.pbc, self.info = cell, np.asarray(pbc), dict(info or {})

    def copy(self):
        return _Atoms(self.numbers.copy(), self.positions.copy(), None if self.cell is None else np.array(self.cell), self.pbc.copy(), self.info)

    def __len__(self):
        return len(self.numbers)


def test_ase_adapter_contract(calc):
    from aimnetcentral_amd.aimnet2ase import AIMNet2ASE

    ase_calc = AIMNet2ASE(calc, charge=0)
    atoms = _Atoms(WATER["numbers"], WATER["coord"], info={"charge": 1})
    ase_calc.calculate(atoms, properties=["energy", "forces"])
    call = calc.engine.calls[-1]
    assert call["n"] == 3 and call["n_mol"] == 1 and call["cell"] is None
    assert ase_calc.charge == 1  # atoms.info["charge"] wins (aimnet2ase.py:112-135)
    r = ase_calc.results
    assert isinstance(r["energy"], float) and r["forces"].shape == (3, 3) and r["charges"].shape == (3,)
    assert np.allclose(r["dipole_moment"], (r["charges"][:, None] * atoms.positions).sum(0))
    assert "stress" not in r
    # an MD step uploads coordinates only: numbers / charge stay the cached device tensors, outputs come back with the status copy
    assert call["host_out"] is True
    t_numbers = ase_calc._numbers[1]
    moved = atoms.copy()
    moved.positions = moved.positions + 0.01
    ase_calc.reset()  # what ASE does when the positions change
    ase_calc.calculate(moved, properties=["energy", "forces"])
    assert ase_calc._numbers[1] is t_numbers and calc.engine.calls[-1]["numbers"].tolist() == [8, 1, 1]
    with pytest.raises(ValueError, match="not implemented"):
        ase_calc.set_atoms(_Atoms([8, 1, 2], WATER["coord"]))
    # periodic: flat input with cell + pbc, DSF auto-switch warning from the calculator
    patoms = _Atoms(WATER["numbers"], WATER["coord"], cell=np.eye(3) * 12.0, pbc=(True, True, True))
    with pytest.warns(UserWarning, match="Switching to DSF"):
        ase_calc.calculate(patoms, properties=["energy", "forces", "stress"])
    assert calc.engine.calls[-1]["cell"] == (3, 3) and ase_calc.results["stress"].shape == (3, 3)
    assert sorted(AIMNet2ASE.implemented_properties) == sorted(["energy", "forces", "free_energy", "charges", "stress", "dipole_moment"])


def test_caller_supplied_neighbour_matrices_reach_the_engine(calc):
    """`nbmat` (+ `nbmat_lr`, shifts) skip the list builder in the reference (calculator.py:1069-1071) and go to the model as they
    are; here they go to the engine in the same role: padded (N + 1, M) or plain (N, M) rows, float shifts -> int32."""
    nb = np.array([[1, 2], [0, 2], [0, 1], [3, 3]])          # the reference's layout: padding row last, sentinel N = 3
    calc(dict(WATER, nbmat=nb, nbmat_lr=nb[:3]))
    lists = calc.engine.calls[-1]["lists"]
    assert set(lists) == {"nbmat", "nbmat_lr"} and lists["nbmat"].shape == (3, 2) and lists["nbmat"].dtype == torch.int32
    assert lists["nbmat_lr"].tolist() == nb[:3].tolist()
    with pytest.raises(KeyError, match="nbmat_lr"):       # the external Coulomb term reads the _lr matrix (nbops.resolve_suffix)
        calc(dict(WATER, nbmat=nb))
    with pytest.raises(ValueError, match="only read together"):
        calc(dict(WATER, nbmat_lr=nb))
    with pytest.raises(ValueError, match="must have shape"):
        calc(dict(WATER, nbmat=nb[:2], nbmat_lr=nb))
    with pytest.raises(NotImplementedError, match="flat"):
        calc(dict(coord=np.zeros((1, 3, 3)), numbers=[[8, 1, 1]], charge=[0.0], nbmat=nb, nbmat_lr=nb))
    with pytest.raises(NotImplementedError, match="caller-supplied"):
        calc(dict(WATER, nbmat=nb, nbmat_lr=nb), hessian=True)
    # periodic: shifts are required, float multiples become integers; simple -> DSF switch as without lists
    sh = np.zeros((4, 2, 3), dtype=np.float32)
    sh[0, 0] = [1.0, 0.0, -1.0]
    with pytest.warns(UserWarning, match="Switching to DSF"):
        calc(dict(WATER, cell=np.eye(3) * 9.0, nbmat=nb, shifts=sh, nbmat_lr=nb, shifts_lr=sh))
    lists = calc.engine.calls[-1]["lists"]
    assert lists["shifts"].dtype == torch.int32 and lists["shifts"][0, 0].tolist() == [1, 0, -1] and lists["shifts_lr"].shape == (3, 2, 3)
    with pytest.raises(KeyError, match="shifts"), pytest.warns(UserWarning):
        calc(dict(WATER, cell=np.eye(3) * 9.0, nbmat=nb, nbmat_lr=nb))


# ---- finite-difference Hessian / HVP over the engine forces (calculator.py:904-910,1753-1989 of the reference) ------
class QuadraticEngine(FakeEngine):
    """E = 1/2 x^T A x per molecule (same symmetric A for every 3-atom molecule): forces = -A x, Hessian = A exactly,
    so the 4th-order stencil must reproduce A to fp32 rounding."""

    A = None

    def eval(self, coord, numbers, mol_idx, charge, **kw):
        out = super().eval(coord, numbers, mol_idx, charge, **kw)
        n_mol = charge.shape[0]
        x = coord.double().view(n_mol, -1)
        out["forces"] = (-(x @ self.A.double())).float().view(-1, 3)
        return out

    def hvp(self, coord, numbers, mol_idx, charge, vectors, **kw):  # the analytic operator of the real engine: H v exactly
        self.hvp_calls = getattr(self, "hvp_calls", [])
        self.hvp_calls.append(dict(K=int(vectors.shape[0]), charge=charge.clone(), coulomb=kw.get("coulomb"), cell=kw.get("cell"),
                                   dftd3=kw.get("dftd3")))
        v = vectors.double().reshape(vectors.shape[0], -1)
        return {"hv": (v @ self.A.double()).float().view_as(vectors)}


@pytest.fixture()
def qcalc(monkeypatch):
    monkeypatch.setattr(calc_mod, "HipEngine", QuadraticEngine)
    monkeypatch.setattr(torch, "as_tensor", _as_tensor_cpu(torch.as_tensor))
    c = calc_mod.AIMNet2Calculator(loader.synthetic_spec(0), device="cuda")
    c.device = "cpu"
    g = torch.Generator().manual_seed(3)
    a = torch.randn(9, 9, generator=g, dtype=torch.float64)
    QuadraticEngine.A = a + a.T
    return c


def test_analytic_hessian_and_hvp_go_through_the_engine_sweep(qcalc):
    """Default operator: every direction in ONE call of the engine's tangent sweep (HipEngine.hvp); the Hessian is its 3N unit
    directions, symmetrised; `eps` is ignored like in the reference (calculator.py:1770-1773)."""
    A = QuadraticEngine.A
    assert qcalc.hvp_method == "analytic"
    out = qcalc(WATER, forces=True, hessian=True)
    assert (out["hessian"].reshape(9, 9).double() - A).abs().max() < 1e-5 and "forces" in out
    assert [c["K"] for c in qcalc.engine.hvp_calls] == [9] and qcalc.engine.hvp_calls[0]["coulomb"] == "simple"
    v = torch.randn(5, 3, 3, generator=torch.Generator().manual_seed(4))
    hv = qcalc.hessian_vector_product(WATER, v, eps=0.3)
    assert hv.shape == (5, 3, 3) and (hv.reshape(5, 9).double() - v.reshape(5, 9).double() @ A).abs().max() < 1e-4
    assert qcalc.hessian_vector_product(WATER, v[0]).shape == (3, 3) and qcalc.engine.hvp_calls[-1]["K"] == 1
    n_eval = len(qcalc.engine.calls)
    qcalc.hessian_vector_product(WATER, v)
    assert len(qcalc.engine.calls) == n_eval  # no force evaluations: the sweep carries its own primal pass
    qcalc.hvp_method = "newton"
    with pytest.raises(ValueError, match="hvp_method"):
        qcalc.hessian_vector_product(WATER, v)


def test_fd_hessian_and_hvp_recover_a_quadratic_model(qcalc):
    A = QuadraticEngine.A
    qcalc.hvp_method = "fd"  # the cross-check operator: central differences of the engine's forces
    out = qcalc(WATER, hessian=True)
    assert set(out) == {"energy", "charges", "hessian"} and out["hessian"].shape == (3, 3, 3, 3)
    assert (out["hessian"].reshape(9, 9).double() - A).abs().max() < 2e-3  # fp32 forces of O(10) / h = 5e-3
    # one base evaluation, then ONE batched evaluation of the 4 x 9 displaced copies
    assert [c["n_mol"] for c in qcalc.engine.calls[-2:]] == [1, 36]
    assert "forces" in qcalc(WATER, forces=True, hessian=True)
    g = torch.Generator().manual_seed(4)
    v = torch.randn(5, 3, 3, generator=g)
    hv = qcalc.hessian_vector_product(WATER, v)
    assert hv.shape == (5, 3, 3)
    assert (hv.reshape(5, 9).double() - v.reshape(5, 9).double() @ A).abs().max() < 5e-3
    hv1 = qcalc.hessian_vector_product(WATER, v[0])
    assert hv1.shape == (3, 3) and torch.allclose(hv1, hv[0], atol=1e-4)
    assert (qcalc.hessian_vector_product(WATER, torch.zeros(3, 3)) == 0).all()


def test_hessian_batched_inputs_follow_the_reference_contract(qcalc):
    c3 = np.stack([np.asarray(WATER["coord"]), np.asarray(WATER["coord"]) + 0.05]).astype(np.float32)
    z3 = np.array([[8, 1, 1], [8, 1, 1]])
    out = qcalc({"coord": c3, "numbers": z3, "charge": [0.0, 0.0]}, hessian=True)
    assert out["hessian"].shape == (2, 3, 3, 3, 3) and out["energy"].shape[0] == 2      # 3D batch: stacked
    flat = {"coord": c3.reshape(6, 3), "numbers": z3.reshape(6), "charge": [0.0, 0.0], "mol_idx": [0, 0, 0, 1, 1, 1]}
    out = qcalc(flat, hessian=True)
    assert isinstance(out["hessian"], list) and len(out["hessian"]) == 2 and out["hessian"][0].shape == (3, 3, 3, 3)
    with pytest.raises(NotImplementedError, match="single structure"):
        qcalc.hessian_vector_product({"coord": c3, "numbers": z3, "charge": [0.0, 0.0]}, torch.zeros(3, 3))
    with pytest.raises(NotImplementedError, match="single structure"):
        qcalc.hessian_vector_product(flat, torch.zeros(6, 3))
    with pytest.raises(NotImplementedError, match="create_graph"):
        qcalc.hessian_vector_product(WATER, torch.zeros(3, 3), create_graph=True)
    with pytest.raises(ValueError, match="vectors must have shape"):
        qcalc.hessian_vector_product(WATER, torch.zeros(4, 3))


# ---- TorchSim adapter (aimnet2torchsim.py:41-175 of the reference) with a duck-typed SimState ---------------------
class _State:
    def __init__(self, positions, numbers, system_idx, cell, pbc, n_systems, **extras):
        self.positions, self.atomic_numbers, self.system_idx = positions, numbers, system_idx
        self.row_vector_cell, self.pbc, self.n_systems = cell, pbc, n_systems
        self.device, self.dtype = torch.device("cpu"), torch.float32
        for k, v in extras.items():
            setattr(self, k, v)


def test_torchsim_adapter_maps_a_flat_multi_system_state(calc):
    from aimnetcentral_amd import AIMNet2TorchSim

    c, z, mol, q = workloads.random_batch(3, 4, 7, seed=1)
    st = _State(torch.from_numpy(c), torch.from_numpy(z), torch.from_numpy(mol), torch.zeros(3, 3, 3), False, 3, charge=torch.tensor([0.0, 1.0, -1.0]))
    model = AIMNet2TorchSim(calc, compute_forces=True)
    model._device = torch.device("cpu")
    assert model.implemented_properties == ["energy", "forces", "charges", "partial_charges"]
    out = model(st)
    call = calc.engine.calls[-1]
    assert call["n"] == len(z) and call["n_mol"] == 3 and call["cell"] is None and torch.equal(call["mol_idx"].long(), torch.from_numpy(mol))
    assert set(out) == {"energy", "charges", "forces", "partial_charges"} and torch.equal(out["partial_charges"], out["charges"])
    model.compute_forces, model.compute_stress = False, True
    assert model.implemented_properties == ["energy", "stress", "charges", "partial_charges"]
    with pytest.raises(ValueError, match="periodic TorchSim state"):
        model(st)
    cell = torch.eye(3).repeat(3, 1, 1) * 14.0
    stp = _State(torch.from_numpy(c), torch.from_numpy(z), torch.from_numpy(mol), cell, True, 3)
    with pytest.warns(UserWarning, match="Switching to DSF"):
        out = model(stp)
    assert calc.engine.calls[-1]["cell"] == (3, 3, 3) and "stress" in out and "forces" not in out
    with pytest.raises(ValueError, match="one value per system"):
        model(_State(torch.from_numpy(c), torch.from_numpy(z), torch.from_numpy(mol), cell, True, 3, charge=[0.0, 1.0]))


# ---- open-shell NSE models: mult -> (alpha, beta) charges, spin_charges output (calculator.py:418-451,473-476) ------------------
@pytest.fixture()
def nse_calc(monkeypatch):
    monkeypatch.setattr(calc_mod, "HipEngine", FakeEngine)
    monkeypatch.setattr(torch, "as_tensor", _as_tensor_cpu(torch.as_tensor))
    c = calc_mod.AIMNet2Calculator(loader.synthetic_spec(0, num_charge_channels=2), device="cuda")
    c.device = "cpu"
    return c


def test_nse_calculator_channels_and_outputs(nse_calc, calc):
    import warnings as w

    assert nse_calc.is_nse and not calc.is_nse
    with pytest.raises(ValueError, match="mult key is required"):
        nse_calc(WATER)
    with w.catch_warnings():
        w.simplefilter("error")  # no "mult is ignored" warning from an NSE model
        out = nse_calc(dict(WATER, charge=1.0, mult=2.0), forces=True)
    assert set(out) == {"energy", "charges", "spin_charges", "forces"} and out["spin_charges"].shape == (3,)
    # (alpha, beta) = Q/2 +- (mult - 1)/2 (aimnet2.py:94-100)
    assert nse_calc.engine.calls[-1]["charge"].tolist() == [[1.0, 0.0]]
    # flat batch: one (charge, mult) per molecule; a scalar mult is broadcast
    data = dict(coord=torch.randn(6, 3), numbers=[8, 1, 1, 8, 1, 1], mol_idx=[0, 0, 0, 1, 1, 1], charge=[0.0, -1.0], mult=[3.0, 2.0])
    nse_calc(data)
    assert nse_calc.engine.calls[-1]["charge"].tolist() == [[1.0, -1.0], [0.0, -1.0]]
    nse_calc(dict(data, mult=1.0))
    assert nse_calc.engine.calls[-1]["charge"].tolist() == [[0.0, 0.0], [-0.5, -0.5]]
    # 3D batches un-flatten spin_charges like charges
    out = nse_calc(dict(coord=torch.randn(2, 3, 3), numbers=[[8, 1, 1], [8, 1, 0]], charge=[0.0, 0.0], mult=[1.0, 3.0]))
    assert out["spin_charges"].shape == (2, 3) and out["spin_charges"][1, 2] == 0
    # closed-shell calculators still warn once that mult is ignored
    with pytest.warns(UserWarning, match="is ignored"):
        calc(dict(WATER, mult=3.0))


def test_nse_fd_hessian_carries_both_channels(monkeypatch):
    monkeypatch.setattr(calc_mod, "HipEngine", QuadraticEngine)
    monkeypatch.setattr(torch, "as_tensor", _as_tensor_cpu(torch.as_tensor))
    c = calc_mod.AIMNet2Calculator(loader.synthetic_spec(0, num_charge_channels=2), device="cuda")
    c.device = "cpu"
    a = torch.randn(9, 9, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    QuadraticEngine.A = a + a.T
    out = c(dict(WATER, mult=3.0), hessian=True)  # analytic sweep: one molecule, both channels
    assert (out["hessian"].reshape(9, 9).double() - QuadraticEngine.A).abs().max() < 1e-5
    assert c.engine.hvp_calls[-1]["charge"].tolist() == [[1.0, -1.0]]
    c.hvp_method = "fd"
    out = c(dict(WATER, mult=3.0), hessian=True)
    assert (out["hessian"].reshape(9, 9).double() - QuadraticEngine.A).abs().max() < 2e-3
    ch = c.engine.calls[-1]["charge"]
    assert ch.shape == (36, 2) and ch[0].tolist() == [1.0, -1.0] and bool((ch == ch[0]).all())


def test_nse_ase_adapter_spin(nse_calc, calc):
    """aimnet2ase.py:62-63,100-105,136-142,158-161: spin_charges property, info['mult'|'spin'] precedence and cache invalidation."""
    from aimnetcentral_amd.aimnet2ase import AIMNet2ASE, PropertyNotImplementedError

    a = AIMNet2ASE(nse_calc, charge=0, mult=1)
    assert "spin_charges" in a.implemented_properties and "spin_charges" not in AIMNet2ASE.implemented_properties
    atoms = _Atoms(WATER["numbers"], WATER["coord"], info={"charge": 1, "spin": 2})
    a.calculate(atoms, properties=["energy"])
    assert a.mult == 2 and nse_calc.engine.calls[-1]["charge"].tolist() == [[1.0, 0.0]]
    assert a.get_spin_charges().shape == (3,)
    atoms2 = atoms.copy()
    atoms2.info = {"charge": 1, "mult": 4}
    assert "info" in a.check_state(atoms2)  # a changed multiplicity invalidates the cache of an NSE calculator
    closed = AIMNet2ASE(calc)
    closed.calculate(atoms, properties=["energy"])
    assert "info" not in closed.check_state(atoms2) and closed.mult == 1
    with pytest.raises(PropertyNotImplementedError):
        closed.get_spin_charges()


def test_unsorted_mol_idx_is_refused(calc):
    data = dict(coord=torch.randn(4, 3), numbers=[8, 1, 1, 1], mol_idx=[0, 1, 0, 1], charge=[0.0, 0.0])
    with pytest.raises(ValueError, match="mol_idx must be sorted"):
        calc(data)
    calc(dict(data, mol_idx=[0, 0, 1, 1]))


def test_torchsim_adapter_nse_spin(nse_calc):
    """aimnet2torchsim.py:87-88,114-115 of the reference: NSE models read the per-system `mult` / `spin` extra and return spin_charges."""
    from aimnetcentral_amd import AIMNet2TorchSim

    c, z, mol, _ = workloads.random_batch(2, 4, 6, seed=2)
    st = _State(torch.from_numpy(c), torch.from_numpy(z), torch.from_numpy(mol), torch.zeros(2, 3, 3), False, 2, charge=torch.tensor([1.0, 0.0]))
    st.spin = torch.tensor([2.0, 3.0])
    model = AIMNet2TorchSim(nse_calc, compute_forces=True)
    model._device = torch.device("cpu")
    assert "spin_charges" in model.implemented_properties
    out = model(st)
    assert nse_calc.engine.calls[-1]["charge"].tolist() == [[1.0, 0.0], [1.0, -1.0]]
    assert out["spin_charges"].shape == (len(z),)


def test_species_cache_is_not_fooled_by_a_recycled_tensor_id(calc):
    """calculator.py:806-823 of the reference: the validation cache holds a weak reference to the validated tensor, so a new
    tensor that happens to be allocated where a freed, already-validated one lived (same id(), _version 0) is validated again."""
    good = torch.tensor([8, 1, 1], dtype=torch.int64)
    calc(dict(WATER, numbers=good))
    key = calc._species_cache[0]
    assert calc._species_cache[1]() is good
    calc(dict(WATER, numbers=good))  # same live tensor: cache hit, nothing changes
    assert calc._species_cache[0] == key
    # emulate the id recycling: forge the cache entry of a DIFFERENT tensor with the new tensor's id / version
    bad = torch.tensor([8, 1, 2], dtype=torch.int64)
    calc._species_cache = ((id(bad), bad._version, key[2]), calc._species_cache[1])  # weakref still points at `good`
    with pytest.raises(ValueError, match=r"Atomic numbers \[2\]"):
        calc(dict(WATER, numbers=bad))
    del good
    import gc

    gc.collect()
    for _ in range(50):  # the real thing: free / reallocate in a loop, every fresh tensor has to be validated
        t = torch.tensor([8, 1, 2], dtype=torch.int64)
        with pytest.raises(ValueError, match=r"Atomic numbers \[2\]"):
            calc(dict(WATER, numbers=t))
        del t
        calc(dict(WATER, numbers=torch.tensor([8, 1, 1], dtype=torch.int64)))


def test_molecule_count_is_checked_against_the_charges(calc):
    """n_mol comes from `charge`: a batch with fewer charges than molecules must fail loudly (the reference fails in
    mol_sum / index_add), not index the per-molecule buffers out of bounds."""
    c3 = torch.randn(2, 3, 3)
    z3 = torch.tensor([[8, 1, 1], [8, 1, 1]])
    with pytest.raises(ValueError, match="one entry per molecule"):
        calc(dict(coord=c3, numbers=z3, charge=0.0))
    calc(dict(coord=c3, numbers=z3, charge=[0.0, 0.0]))
    flat = dict(coord=torch.randn(4, 3), numbers=[8, 1, 1, 1], mol_idx=[0, 0, 1, 2], charge=[0.0, 0.0])
    with pytest.raises(ValueError, match=r"mol_idx must lie in \[0, 2\)"):
        calc(flat)
    with pytest.raises(ValueError, match=r"mol_idx must lie in \[0, 2\)"):
        calc(dict(flat, mol_idx=[-1, 0, 0, 1]))
    calc(dict(flat, mol_idx=[0, 0, 1, 1]))


def test_external_coulomb_subtract_sr_for_models_without_embedded_srcoulomb(monkeypatch):
    """calculator.py:218-230 of the reference: needs_coulomb with coulomb_mode != 'sr_embedded' -> LRCoulomb(subtract_sr=True);
    the engine then carries the SR subtraction (same sum as the embedded SRCoulomb) with the metadata's rc / envelope."""
    import dataclasses

    monkeypatch.setattr(calc_mod, "HipEngine", FakeEngine)
    base = loader.synthetic_spec(0)
    md = dict(base.metadata, coulomb_mode="none", needs_coulomb=True, coulomb_sr_rc=4.2, coulomb_sr_envelope="cosine")
    spec = dataclasses.replace(base, sr_coulomb=False, metadata=md)
    c = calc_mod.AIMNet2Calculator(spec, device="cuda")
    assert c.external_coulomb.subtract_sr is True
    assert c.engine.spec.sr_coulomb and c.engine.spec.sr_rc == pytest.approx(4.2) and c.engine.spec.sr_envelope == "cosine"
    c2 = calc_mod.AIMNet2Calculator(base, device="cuda")
    assert c2.external_coulomb.subtract_sr is False
    with pytest.raises(NotImplementedError, match="subtracted twice"):
        calc_mod.AIMNet2Calculator(dataclasses.replace(base, metadata=md), device="cuda")
