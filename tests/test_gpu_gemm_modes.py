"""The parity fixtures under the two non-default GEMM modes (engine option "gemm_bf3", csrc/gemm_bf3.hip).

Default (1): MLP GEMMs of batches above 256 rows run with split operands on the 16-bit matrix pipe (fp16x2, gemm_h2.hip), smaller
ones on the exact-fp32 skinny kernel - so the small golden fixtures never see the split kernels in the other test modules.  Here
every parity check runs with mode 2 (split kernels for EVERY batch size; pre-split activations in the fp16x2 form = gemm_h2.hip
+ gemm_head.hip, in the bf16x3 form = gemm_bf3a.hip, and the in-loop split of gemm_bf3.hip) and mode 0 (exact-fp32 MFMA kernels everywhere, the fallback the
bench reports as `exact_f32`), at the same, unchanged gates: goldens of the unmodified reference, the un-widened 1e-5 eV gate on
the cold fixture, the 32-seed randomised sweep, both charge-channel families."""
from __future__ import annotations

import numpy as np
import pytest

import test_gpu_fuzz as Z
import test_gpu_parity as P
from conftest import golden
from oracle import aimnet2_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[(2, 1, 1), (2, 1, 0), (2, 0, 1), (0, 1, 1)], ids=["h2_every_size", "bf3_every_size", "bf3_split_in_loop", "exact_f32"])
def gemm_mode(request, hip_engine, hip_engine_nse, hip_engine_cold):
    """(gemm_bf3, gemm_presplit, gemm_h2): split kernels for every batch size with pre-split activations in the fp16x2 form
    (gemm_h2.hip + the fused head: what large systems run by default) and in the bf16x3 form (gemm_bf3a.hip, the fallback for
    activations beyond fp16's range), the same with fp32 activations split inside the GEMM loop (gemm_bf3.hip, the tangent sweep's
    kernels), and the exact-fp32 kernels."""
    engines = (hip_engine, hip_engine_nse, hip_engine_cold)
    for e in engines:
        e.set_option("gemm_bf3", request.param[0])
        e.set_option("gemm_presplit", request.param[1])
        e.set_option("gemm_h2", request.param[2])
    yield request.param
    for e in engines:
        e.set_option("gemm_bf3", 1)
        e.set_option("gemm_presplit", 1)
        e.set_option("gemm_h2", 1)


def test_taxol(gemm_mode, hip_engine, oracle32):
    P.test_taxol_vs_oracle_and_reference_golden(hip_engine, oracle32)


def test_ragged_charged_batch(gemm_mode, hip_engine, oracle32, oracle64):
    P.test_ragged_charged_batch(hip_engine, oracle32, oracle64)


@pytest.mark.parametrize("name", ["pbc96_dsf15", "pbc96_dsf8_wrapped", "pbc2x96_dsf9"])
def test_periodic_dsf_forces_stress(gemm_mode, hip_engine, oracle32, oracle64, name):
    P.test_periodic_dsf_forces_stress(hip_engine, oracle32, oracle64, name)


def test_closer_to_fp64_truth_than_tolerance(gemm_mode, hip_engine, oracle64):
    P.test_closer_to_fp64_truth_than_tolerance(hip_engine, oracle64)


def test_cold_fixture_unwidened_gate(gemm_mode, hip_engine):
    """cold24 (relaxed on the synthetic surface): |dE| < 1e-5 eV against the reference golden, no fp64-anchored slack."""
    g = golden("cold24")
    res, _ = P.run(hip_engine, g, "simple")
    P.compare(res, g, 24, "cold24/reference golden")


@pytest.mark.parametrize("name", ["taxol", "batch5", "rand8", "pbc96"])
def test_cold_weights_at_the_reference_literal_gates(gemm_mode, hip_engine_cold, name):
    """|dE| < 1e-5 eV and zero force components outside allclose(1e-4, 1e-5) against the reference's cold-weight goldens, with the
    split kernels forced onto these small batches."""
    P.test_cold_weights_at_the_reference_literal_gates(hip_engine_cold, name)


def test_nse_molecule_and_batch(gemm_mode, hip_engine_nse, oracle32_nse, oracle64_nse):
    P.test_nse_molecule_vs_oracle_and_reference_golden(hip_engine_nse, oracle32_nse)
    P.test_nse_ragged_batch_mixed_multiplicities(hip_engine_nse, oracle32_nse, oracle64_nse)


def test_bitwise_repeatability(gemm_mode, hip_engine):
    P.test_bitwise_repeatability(hip_engine)


@pytest.mark.parametrize("seed", range(Z._LO, Z._HI))
def test_random_configuration(gemm_mode, seed, hip_engine, hip_engine_nse, oracle32, oracle32_nse, oracle64, oracle64_nse):
    Z.test_random_configuration(seed, hip_engine, hip_engine_nse, oracle32, oracle32_nse, oracle64, oracle64_nse)


def test_modes_agree_on_a_large_batch(hip_engine, oracle32):
    """One 2 304-atom periodic evaluation per mode: the split kernels and the exact kernels give the same energies, forces and
    stress to the fp32 noise of the MLP stack (a direct A/B on a batch that takes the big tiles)."""
    import torch

    from aimnetcentral_amd import workloads

    c, z, cell = workloads.glucose_supercell((2, 3, 4))
    rng = np.random.default_rng(7)
    c = (c + rng.normal(0.0, 0.02, c.shape)).astype(np.float32)
    dev = hip_engine.device
    out = {}
    try:
        for mode in (0, 1, 2):  # exact fp32, fp16x2-split (default), bf16x3-split
            hip_engine.set_option("gemm_bf3", min(mode, 1))
            hip_engine.set_option("gemm_h2", 0 if mode == 2 else 1)
            r = hip_engine.eval(torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.zeros(len(z), dtype=torch.int64, device=dev),
                                torch.zeros(1, device=dev), cell=torch.from_numpy(cell.astype(np.float32)).to(dev), forces=True, stress=True,
                                coulomb="dsf", dsf_rc=9.0)
            out[mode] = {k: v.cpu().numpy() for k, v in r.items()}
    finally:
        hip_engine.set_option("gemm_bf3", 1)
        hip_engine.set_option("gemm_h2", 1)
    n = len(z)
    for m in (1, 2):
        assert abs(out[0]["energy"][0] - out[m]["energy"][0]) <= max(1e-5, 5e-7 * n)
        fmax = np.abs(out[0]["forces"]).max()
        assert np.abs(out[0]["forces"] - out[m]["forces"]).max() <= 1e-5 + 1e-4 * fmax
        assert np.abs(out[0]["charges"] - out[m]["charges"]).max() <= 1e-4
        assert np.abs(out[0]["stress"] - out[m]["stress"]).max() <= 1e-5


def test_fp16_range_fallback():
    """An activation beyond fp16's range (|x| >= 65504) cannot be held by the fp16x2-split operands: the outputs turn non-finite,
    `HipEngine.eval` repeats the call with the bf16x3-split operands, warns, and stays there.  Weights: seed 0 with the first layer of
    pass 0 scaled by 2e4 (the weights themselves stay inside fp16's range - otherwise the engine would not start in the h2 form at
    all - its pre-activations do not; finite garbage in fp32 arithmetic)."""
    import copy

    import torch

    from aimnetcentral_amd import loader, workloads
    from aimnetcentral_amd.engine import HipEngine

    spec = copy.deepcopy(loader.synthetic_spec(0))
    spec.weights["mlps.0.0.weight"] = (spec.weights["mlps.0.0.weight"] * np.float32(2e4)).astype(np.float32)
    c, z, mol, q = workloads.random_batch(12, 30, 40, seed=1)  # > 256 atoms: the split kernels run by default
    dev = torch.device("cuda:0")
    args = (torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.from_numpy(mol).to(dev), torch.from_numpy(q).to(dev))
    ref_eng = HipEngine(spec, "cuda:0")
    ref_eng.set_option("gemm_h2", 0)
    ref = ref_eng.eval(*args, forces=True)
    assert torch.isfinite(ref["energy"]).all() and torch.isfinite(ref["forces"]).all()
    eng = HipEngine(spec, "cuda:0")
    assert eng.get_option("gemm_h2") == 1
    with pytest.warns(RuntimeWarning, match="exceeded fp16's range"):
        res = eng.eval(*args, forces=True)
    assert eng.get_option("gemm_h2") == 0
    assert torch.equal(res["energy"], ref["energy"]) and torch.equal(res["forces"], ref["forces"])
    # the deferred path (no host read per step): the check every K steps finds the non-finite energies, switches and asks for a repeat
    from aimnetcentral_amd.engine import ActivationRangeError

    eng2 = HipEngine(spec, "cuda:0")
    for _ in range(3):
        eng2.eval(*args, forces=True, sync=False, defer=True)
    with pytest.raises(ActivationRangeError, match="repeat them"):
        eng2.check_deferred()
    assert eng2.get_option("gemm_h2") == 0
    assert int(eng2.last_status[6]) & 32  # raised on the device by the kernels that write the energies / forces (include/aimnet_hip.h)
    again = eng2.eval(*args, forces=True)
    assert not int(eng2.last_status[6]) & 32
    assert torch.equal(again["energy"], ref["energy"])
