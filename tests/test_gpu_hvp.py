"""Analytic Hessian-vector products (csrc/hvp.hip, aimnet_engine_hvp through the C ABI) against the hand-derived fp64 tangent
sweep of the oracle (oracle/aimnet2_analytic.py::evaluate_hvp, itself pinned to the autograd Hessian in
tests/test_oracle_analytic.py), plus the size-independent properties of the operator.  Tolerance: the sweep is fp32 arithmetic on
second derivatives of up to several hundred eV/A^2 - 3e-5 relative to the largest element of the product plus 1e-5 absolute
(measured 2-3e-6 relative on the fixtures; worst of a 100-seed sweep of random molecules 1.3e-5, tests/tools/hvp_soak.py), well inside the reference's allclose(1e-3, 1e-3) gate for its own operator
(tests/test_hvp.py:75)."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import assert_forces_close, golden

pytestmark = pytest.mark.gpu


def _t(a, dt=torch.float32):
    return torch.as_tensor(np.asarray(a)).to(dt).cuda()


def _spec_and_engine(eng, om64, coord, numbers, charge, mol, V, cell=None, kw=None, mult=None):
    from oracle import aimnet2_analytic as AN
    from oracle import aimnet2_oracle as O

    kw = kw or {}
    n = len(numbers)
    ref = O.evaluate(om64, coord, numbers, charge, mol, cell=cell, return_intermediates=True, forces=False, mult=mult, **kw)
    xw = ref["coord_wrapped"]
    if cell is None:
        nbl, shl = O.neighbor_list(xw, float("inf"), mol)
        coul = "simple"
    else:
        nbl, shl = O.neighbor_list(xw, kw["dsf_rc"], mol, cell, np.ones(3, bool))
        coul = "dsf"
    spec = AN.evaluate_hvp(om64, xw, numbers, charge, mol, ref["nbmat"], V, shifts=ref.get("shifts"), cell=cell, coulomb=coul,
                           nbmat_lr=nbl, shifts_lr=shl, mult=mult, **{k: v for k, v in kw.items() if k != "coulomb"})
    q = np.atleast_1d(np.asarray(charge, dtype=np.float32))
    if eng.nq == 2:
        mt = np.ones_like(q) if mult is None else np.atleast_1d(np.asarray(mult, dtype=np.float32))
        q = np.stack([0.5 * q + 0.5 * (mt - 1), 0.5 * q - 0.5 * (mt - 1)], -1)
    out = eng.hvp(_t(coord), _t(numbers, torch.int32), _t(mol, torch.int32), _t(q), _t(V), cell=None if cell is None else _t(cell),
                  coulomb=coul, dsf_rc=kw.get("dsf_rc", 15.0), dsf_alpha=kw.get("dsf_alpha", 0.2), want_forces=True)
    assert out["hv"].shape == (len(V), n, 3)
    return spec, out["hv"].cpu().numpy().astype(np.float64), out["forces"].cpu().numpy()


def _close(hv, ref, what):
    err, top = np.abs(hv - ref).max(), np.abs(ref).max()
    assert err <= 1e-5 + 3e-5 * top, f"{what}: max|d(Hv)| = {err:.3e} on max|Hv| = {top:.3e}"


@pytest.mark.parametrize("name,kw", [
    ("batch5", {}),
    ("taxol", {}),
    ("pbc96_dsf8_wrapped", {"coulomb": "dsf", "dsf_rc": 8.0, "dsf_alpha": 0.25}),
    ("pbc2x96_dsf9", {"coulomb": "dsf", "dsf_rc": 9.0, "dsf_alpha": 0.2}),
])
def test_hvp_matches_the_fp64_tangent_sweep(hip_engine, oracle64, name, kw):
    """molecule, ragged batch (the engine takes any number of molecules), periodic DSF cell, two cells with their own DSF lists"""
    g = golden(name)
    n = len(g["numbers"])
    mol = g["mol_idx"] if "mol_idx" in g.files else np.zeros(n, dtype=np.int64)
    cell = g["cell"] if "cell" in g.files else None
    V = np.random.default_rng(11).standard_normal((3, n, 3)).astype(np.float32)
    spec, hv, f = _spec_and_engine(hip_engine, oracle64, g["coord"], g["numbers"], g["charge"], mol, V, cell, kw)
    _close(hv, spec["hv"], name)
    assert_forces_close(f, spec["forces"], name + " (forces of the sweep)")
    assert_forces_close(f, g["forces"], name + " (forces of the sweep vs the reference golden)")


def test_hvp_two_charge_channels(hip_engine_nse, oracle64_nse):
    g = golden("nse")
    n = len(g["b5_numbers"])
    V = np.random.default_rng(12).standard_normal((2, n, 3)).astype(np.float32)
    spec, hv, f = _spec_and_engine(hip_engine_nse, oracle64_nse, g["b5_coord"], g["b5_numbers"], g["b5_charge"], g["b5_mol_idx"], V,
                                   mult=g["b5_mult"])
    _close(hv, spec["hv"], "nse b5")
    assert_forces_close(f, g["b5_forces"], "nse b5 (forces of the sweep)")


def test_hvp_two_charge_channels_periodic(hip_engine_nse, oracle64_nse):
    """NSE family in a periodic cell with DSF Coulomb (both channels through the charge convolution, one Coulomb seed for both)."""
    g = golden("nse")
    V = np.random.default_rng(13).standard_normal((2, 96, 3)).astype(np.float32)
    spec, hv, f = _spec_and_engine(hip_engine_nse, oracle64_nse, g["pbc_coord"], g["pbc_numbers"], np.zeros(1, np.float32),
                                   np.zeros(96, dtype=np.int64), V, cell=g["pbc_cell"],
                                   kw={"coulomb": "dsf", "dsf_rc": 9.0, "dsf_alpha": 0.2}, mult=g["pbc_mult"])
    _close(hv, spec["hv"], "nse pbc")
    assert_forces_close(f, g["pbc_forces"], "nse pbc (forces of the sweep)")


def test_hvp_operator_properties(hip_engine):
    """Linearity in v, symmetry of the dense Hessian, translation invariance (H . uniform shift = 0), bitwise repeatability, and
    independence of how the directions are split into sweeps."""
    g = golden("hvp40")
    args = (_t(g["coord"]), _t(g["numbers"], torch.int32), torch.zeros(40, dtype=torch.int32, device="cuda"), _t([0.0]))
    eye = torch.eye(120, device="cuda").view(120, 40, 3)
    H = hip_engine.hvp(*args, eye)["hv"].view(120, 120)
    assert (H - H.T).abs().max().item() < 1e-5          # computed column by column: symmetric to round-off
    shift = torch.zeros(3, 40, 3, device="cuda")
    for c in range(3):
        shift[c, :, c] = 1.0
    assert hip_engine.hvp(*args, shift)["hv"].abs().max().item() < 2e-4
    v = torch.randn(4, 40, 3, generator=torch.Generator().manual_seed(2)).cuda()
    hv = hip_engine.hvp(*args, v)["hv"]
    assert torch.equal(hv, hip_engine.hvp(*args, v)["hv"])
    assert (hv.view(4, 120) - v.view(4, 120) @ H).abs().max().item() < 2e-4
    lin = hip_engine.hvp(*args, (2.0 * v[0] - 0.5 * v[1]).unsqueeze(0))["hv"][0]
    assert (lin - (2.0 * hv[0] - 0.5 * hv[1])).abs().max().item() < 2e-4
    budget = hip_engine.HVP_BYTES_BUDGET
    try:  # one direction per sweep: identical numbers (a direction never sees another one)
        hip_engine.HVP_BYTES_BUDGET = 1
        assert torch.equal(hv, hip_engine.hvp(*args, v)["hv"])
    finally:
        hip_engine.HVP_BYTES_BUDGET = budget


def test_hvp_grows_the_rows_on_overflow(hip_engine):
    g = golden("taxol")
    n = len(g["numbers"])
    args = (_t(g["coord"]), _t(g["numbers"], torch.int32), torch.zeros(n, dtype=torch.int32, device="cuda"), _t([0.0]))
    v = torch.randn(1, n, 3, generator=torch.Generator().manual_seed(3)).cuda()
    want = hip_engine.hvp(*args, v)["hv"]
    old = hip_engine.max_nb
    hip_engine.max_nb = 16
    got = hip_engine.hvp(*args, v)["hv"]
    assert hip_engine.max_nb > 16 and torch.equal(got, want)
    hip_engine.max_nb = max(old, hip_engine.max_nb)


def test_hvp_rejects_what_it_does_not_carry(hip_engine):
    from aimnetcentral_amd._lib import HipLibraryError

    g = golden("pbc96_dsf8_wrapped")
    args = (_t(g["coord"]), _t(g["numbers"], torch.int32), torch.zeros(96, dtype=torch.int32, device="cuda"), _t([0.0]))
    v = torch.zeros(1, 96, 3, device="cuda")
    with pytest.raises(HipLibraryError, match="simple"):
        hip_engine.hvp(*args, v, cell=_t(g["cell"]), coulomb="simple")


def test_calculator_periodic_hvp_and_dftd3_fallback(oracle64):
    """Through the calculator: a periodic cell (simple -> DSF switch) uses the analytic sweep and agrees with the finite-difference
    operator; with the external DFT-D3 term attached the finite-difference operator is what runs (the sweep does not carry D3)."""
    import warnings

    from aimnetcentral_amd import AIMNet2Calculator, loader

    g = golden("pbc96_dsf8_wrapped")
    calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        calc.set_lrcoulomb_method("dsf", cutoff=8.0, dsf_alpha=0.25)
    data = {"coord": g["coord"], "numbers": g["numbers"], "charge": 0.0, "cell": g["cell"]}
    v = np.random.default_rng(4).standard_normal((2, 96, 3)).astype(np.float32)
    hv = calc.hessian_vector_product(data, v).cpu().numpy()
    calc.hvp_method = "fd"
    hv_fd = calc.hessian_vector_product(data, v).cpu().numpy()
    assert np.abs(hv - hv_fd).max() <= 5e-3 + 1e-3 * np.abs(hv).max(), np.abs(hv - hv_fd).max()
    # the same cell with the external DFT-D3 term (periodic D3 list, 15 A): sweep + D3 block against the finite-difference operator
    gd, t = golden("dftd3"), golden("dftd3_subset")
    spec = loader.synthetic_spec(0)
    spec.metadata = dict(spec.metadata, needs_dispersion=True,
                         d3_params={k: float(gd[k]) for k in ("s6", "s8", "a1", "a2")})
    cd3 = AIMNet2Calculator(spec, device="cuda:0", dftd3_data={k: t[k] for k in ("c6ab", "cn_ref", "rcov", "r4r2")})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cd3.set_lrcoulomb_method("dsf", cutoff=8.0, dsf_alpha=0.25)
    hv3 = cd3.hessian_vector_product(data, v).cpu().numpy()
    cd3.hvp_method = "fd"
    hv3_fd = cd3.hessian_vector_product(data, v).cpu().numpy()
    assert np.abs(hv3 - hv3_fd).max() <= 5e-3 + 1e-3 * np.abs(hv3).max(), np.abs(hv3 - hv3_fd).max()
    assert np.abs(hv3 - hv).max() > 1e-3  # the term is there


def test_hvp_with_external_dftd3(oracle64):
    """The dispersion block (central difference of the D3 gradient alone, inside the same call) against the autograd Hessian of
    the fp64 oracle WITH the D3 term, on config 4's molecule: the block is up to 0.63 eV/A^2 here, the whole operator must stay
    inside the reference's allclose(1e-3, 1e-3) gate and within 2e-4 eV/A^2."""
    from aimnetcentral_amd import AIMNet2Calculator, loader
    from oracle import aimnet2_oracle as O

    gd, t = golden("dftd3"), golden("dftd3_subset")
    par = dict(s6=float(gd["s6"]), s8=float(gd["s8"]), a1=float(gd["a1"]), a2=float(gd["a2"]), cutoff=15.0, smoothing_fraction=0.2)
    tables = {k: t[k] for k in ("c6ab", "cn_ref", "rcov", "r4r2")}
    g = golden("hvp40")
    H64 = O.evaluate(oracle64, g["coord"], g["numbers"], 0.0, hessian=True, forces=False, dftd3=dict(par, **tables))["hessian"].reshape(120, 120)
    H64_no = O.evaluate(oracle64, g["coord"], g["numbers"], 0.0, hessian=True, forces=False)["hessian"].reshape(120, 120)
    assert np.abs(H64 - H64_no).max() > 0.3  # the term matters on this geometry
    spec = loader.synthetic_spec(0)
    spec.metadata = dict(spec.metadata, needs_dispersion=True, d3_params={k: par[k] for k in ("s6", "s8", "a1", "a2")})
    calc = AIMNet2Calculator(spec, device="cuda:0", dftd3_data=tables)
    data = {"coord": g["coord"], "numbers": g["numbers"], "charge": 0.0}
    out = calc(data, forces=True, hessian=True)
    H = out["hessian"].cpu().numpy().reshape(120, 120).astype(np.float64)
    assert np.allclose(H, H64, rtol=1e-3, atol=1e-3) and np.abs(H - H64).max() <= 2e-4, np.abs(H - H64).max()
    hv = calc.hessian_vector_product(data, g["v4"]).cpu().numpy().reshape(4, 120)
    ref = g["v4"].reshape(4, 120).astype(np.float64) @ H64
    assert np.abs(hv - ref).max() <= 2e-4 + 1e-5 * np.abs(ref).max(), np.abs(hv - ref).max()
    # the forces returned next to the Hessian carry the term too
    f64 = O.evaluate(oracle64, g["coord"], g["numbers"], 0.0, dftd3=dict(par, **tables))["forces"]
    assert_forces_close(out["forces"].cpu().numpy(), f64, "hvp40 + d3")


@pytest.mark.parametrize("seed", range(6))
def test_hvp_fuzz_random_molecules(hip_engine, oracle64, seed):
    """Seeded random organic geometries (the generator of BASELINE config 2: contacts down to 0.9 A, |H v| of several hundred
    eV/A^2), charged and neutral, one to three molecules per call: the sweep against the fp64 specification."""
    from aimnetcentral_amd import workloads

    rng = np.random.default_rng(100 + seed)
    n_mol = 1 + seed % 3
    cs, zs, ms = [], [], []
    for m in range(n_mol):
        c, z = workloads.random_organic(int(rng.integers(8, 40)), rng)
        cs.append(c)
        zs.append(z)
        ms.append(np.full(len(z), m))
    coord, numbers, mol = np.concatenate(cs).astype(np.float32), np.concatenate(zs), np.concatenate(ms)
    charge = rng.integers(-1, 2, size=n_mol).astype(np.float32)
    V = rng.standard_normal((2, len(numbers), 3)).astype(np.float32)
    spec, hv, f = _spec_and_engine(hip_engine, oracle64, coord, numbers, charge, mol, V)
    _close(hv, spec["hv"], f"fuzz seed {seed}")
    assert_forces_close(f, spec["forces"], f"fuzz seed {seed} (forces of the sweep)")


def test_cosine_srcoulomb_envelope_eval_and_hvp():
    """The SRCoulomb block with the cosine envelope (rc = 4.2 A) - the branch the shipped YAMLs do not take - through the engine:
    energy / forces / charges of taxol and the 40-atom Hessian + products against the reference's outputs (tests/golden/srcos.npz)."""
    from aimnetcentral_amd import AIMNet2Calculator, loader
    from conftest import CHARGE_ATOL, energy_tol

    g = golden("srcos")
    spec = loader.synthetic_spec(0, sr_envelope="cosine", sr_rc=4.2)
    assert spec.sr_envelope == "cosine" and abs(spec.sr_rc - 4.2) < 1e-6
    calc = AIMNet2Calculator(spec, device="cuda:0")
    out = calc({"coord": g["coord"], "numbers": g["numbers"], "charge": 0.0}, forces=True)
    assert abs(float(out["energy"][0]) - g["energy"][0]) <= energy_tol(113)
    assert_forces_close(out["forces"].cpu().numpy(), g["forces"], "taxol, cosine SR envelope")
    assert np.abs(out["charges"].cpu().numpy() - g["charges"]).max() <= CHARGE_ATOL
    data = {"coord": g["coord"][:40], "numbers": g["numbers"][:40], "charge": 0.0}
    res = calc(data, forces=True, hessian=True)
    H = res["hessian"].cpu().numpy().reshape(120, 120)
    Href = g["hessian40"].reshape(120, 120)
    assert np.allclose(H, Href, rtol=1e-3, atol=1e-3) and np.abs(H - Href).max() <= 1e-4, np.abs(H - Href).max()
    hv4 = calc.hessian_vector_product(data, g["v4"]).cpu().numpy()
    assert np.abs(hv4 - g["hv4"]).max() <= 1e-4 + 1e-5 * np.abs(g["hv4"]).max(), np.abs(hv4 - g["hv4"]).max()
    assert_forces_close(res["forces"].cpu().numpy(), g["forces40"], "40 atoms, cosine SR envelope")


def test_hvp_announces_calls_that_would_run_for_minutes(hip_engine):
    """A dense Hessian of a 10 k-atom crystal is 30 240 directions x 7.5 ms: `HipEngine.hvp` warns (the reference runs such requests
    too, slowly - ADVICE r4) and refuses only with HVP_ON_LONG = 'raise'; the threshold is a class attribute.  (The direction tensor
    of the refused call is a stride-0 view: nothing is materialised.)"""
    from aimnetcentral_amd import workloads

    c, z, cell = workloads.glucose_supercell((7, 3, 5))
    dev = hip_engine.device
    n = len(z)
    v = torch.zeros(1, n, 3, device=dev).expand(3 * n, n, 3)
    args = (torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev), torch.zeros(n, dtype=torch.int32, device=dev),
            torch.zeros(1, device=dev))
    kw = dict(cell=torch.from_numpy(cell.astype(np.float32)).to(dev), coulomb="dsf")
    hip_engine.HVP_ON_LONG = "raise"
    try:
        with pytest.raises(ValueError, match="would take"):
            hip_engine.hvp(*args, v, **kw)
    finally:
        del hip_engine.HVP_ON_LONG
    hip_engine.HVP_MAX_SECONDS = 1e-9  # one direction on 10 080 atoms: announced, and computed
    try:
        with pytest.warns(RuntimeWarning, match="will take"):
            out = hip_engine.hvp(*args, torch.ones(1, n, 3, device=dev), **kw)
    finally:
        del hip_engine.HVP_MAX_SECONDS
    assert torch.isfinite(out["hv"]).all()
