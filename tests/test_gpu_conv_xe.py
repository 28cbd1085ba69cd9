"""The reverse-pair (XE) form of the conv backward (csrc/conv.hip, conv_bwd_kernel<.., XE>: every ordered pair evaluates only its Y half, F1 goes through a pair buffer and the reverse-pair map)
against the oracle, the reference goldens and the default packed-FMA kernels.  They are an engine option
(`set_option("conv_xe", 1)`); `split_max = 0` sends even the small fixtures through the one-wave-per-atom kernels that only
systems above 1 024 atoms take by default."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import golden
from oracle import aimnet2_oracle as O
from test_gpu_parity import compare, run

pytestmark = pytest.mark.gpu


@pytest.fixture()
def engines(synth_sd):
    from aimnetcentral_amd import loader
    from aimnetcentral_amd.engine import HipEngine

    spec = loader.synthetic_spec(0)
    valu, xe = HipEngine(spec, "cuda:0"), HipEngine(spec, "cuda:0")
    valu.set_option("conv_xe", 0)  # the combined-adjoint kernel as the baseline
    for e in (valu, xe):
        e.set_option("split_max", 0)  # per engine: both run the large-system kernels on every fixture
    xe.set_option("conv_xe", 1)
    yield valu, xe


def test_taxol_and_batch(engines, oracle32, oracle64):
    valu, xe = engines
    g = golden("taxol")
    r, _ = run(xe, g, "simple")
    compare(r, O.evaluate(oracle32, g["coord"], g["numbers"], g["charge"]), 113, "taxol xe/oracle")
    compare(r, g, 113, "taxol xe/reference golden")
    r0, _ = run(valu, g, "simple")
    assert np.abs(r["forces"] - r0["forces"]).max() < 2e-5 and abs(r["energy"][0] - r0["energy"][0]) < 1e-5
    g = golden("batch5")
    r, mol = run(xe, g, "simple")
    e64 = O.evaluate(oracle64, g["coord"], g["numbers"], g["charge"], mol, forces=False)["energy"]
    compare(r, g, np.bincount(mol), "batch5 xe/reference golden", e64)


@pytest.mark.parametrize("name", ["pbc96_dsf15", "pbc2x96_dsf9"])
def test_periodic_stress(engines, oracle32, oracle64, name):
    valu, xe = engines
    g = golden(name)
    kw = dict(dsf_rc=float(g["dsf_rc"]), dsf_alpha=float(g["dsf_alpha"]))
    r, mol = run(xe, g, "dsf", stress=True, **kw)
    e64 = O.evaluate(oracle64, g["coord"], g["numbers"], g["charge"], mol, cell=g["cell"], coulomb="dsf", forces=False, **kw)["energy"]
    compare(r, g, 96, name + " xe/reference golden", e64)
    r0, _ = run(valu, g, "dsf", stress=True, **kw)
    assert np.abs(r["stress"] - r0["stress"]).max() < 2e-6 and np.abs(r["forces"] - r0["forces"]).max() < 2e-5


def test_repeatability_and_non_multiple_of_four(engines, oracle32):
    """two evaluations are bitwise identical; a batch whose atom count is not a multiple of four (wave tails)."""
    valu, xe = engines
    g = golden("pbc96_dsf15")
    kw = dict(dsf_rc=15.0, dsf_alpha=0.2)
    ref, _ = run(xe, g, "dsf", stress=True, **kw)
    again, _ = run(xe, g, "dsf", stress=True, **kw)
    assert np.array_equal(ref["forces"], again["forces"]) and np.array_equal(ref["energy"], again["energy"])
    g = golden("taxol")  # 113 atoms
    r, _ = run(xe, g, "simple")
    r0, _ = run(valu, g, "simple")
    assert np.abs(r["forces"] - r0["forces"]).max() < 2e-5


def test_nse_two_channel_model(synth_sd_nse, oracle32_nse):
    from aimnetcentral_amd import loader
    from aimnetcentral_amd.engine import HipEngine

    eng = HipEngine(loader.synthetic_spec(0, num_charge_channels=2), "cuda:0")
    eng.set_option("split_max", 0)
    eng.set_option("conv_xe", 1)
    try:
        c, z, mol, q = __import__("aimnetcentral_amd.workloads", fromlist=["x"]).random_batch(4, 10, 18, seed=11)
        mult = np.array([1.0, 2.0, 3.0, 2.0], dtype=np.float32)
        qq = q + np.array([0, 1, 0, -1], dtype=np.float32)
        dev = eng.device
        ab = np.stack([0.5 * qq + 0.5 * (mult - 1), 0.5 * qq - 0.5 * (mult - 1)], axis=-1).astype(np.float32)
        res = eng.eval(torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.from_numpy(mol).to(dev),
                       torch.from_numpy(ab).to(dev), forces=True, coulomb="simple")
        ref = O.evaluate(oracle32_nse, c, z, qq, mol, coulomb="simple", mult=mult)
        assert np.abs(res["energy"].cpu().numpy() - ref["energy"]).max() < 5e-5
        assert np.abs(res["forces"].cpu().numpy() - ref["forces"]).max() < 1e-5 + 1e-4 * np.abs(ref["forces"]).max()
        assert np.abs(res["charges"].cpu().numpy() - ref["charges"]).max() < 1e-4
    finally:
        eng.set_option("split_max", -1)


def test_embedding_bias_table_switch(synth_sd):
    """Pass 0's first GEMM over the 448 conv columns with the embedding block folded into a per-element bias table
    (engine.hip, emb_bias0; default) against the full-width GEMM (`emb_bias = 0`)."""
    from aimnetcentral_amd import loader
    from aimnetcentral_amd.engine import HipEngine

    on, off = HipEngine(loader.synthetic_spec(0), "cuda:0"), HipEngine(loader.synthetic_spec(0), "cuda:0")
    off.set_option("emb_bias", 0)
    for name, coul, kw in (("taxol", "simple", {}), ("pbc96_dsf15", "dsf", dict(dsf_rc=15.0, dsf_alpha=0.2))):
        g = golden(name)
        r1, _ = run(on, g, coul, stress=coul == "dsf", **kw)
        r0, _ = run(off, g, coul, stress=coul == "dsf", **kw)
        assert abs(r1["energy"][0] - r0["energy"][0]) < 2e-5, name
        assert np.abs(r1["forces"] - r0["forces"]).max() < 2e-5, name
        assert np.abs(r1["charges"] - r0["charges"]).max() < 2e-6, name


def test_pass0_element_moments_switch(synth_sd):
    """Pass 0 through per-element moments (conv_fwd P0M + the species-moment backward; default) against the generic
    row-gather kernels (`p0_moments = 0`), on the one-wave-per-atom kernels (`split_max = 0`)."""
    from aimnetcentral_amd import loader
    from aimnetcentral_amd.engine import HipEngine

    on, off = HipEngine(loader.synthetic_spec(0), "cuda:0"), HipEngine(loader.synthetic_spec(0), "cuda:0")
    on.set_option("split_max", 0)
    off.set_option("split_max", 0)
    off.set_option("p0_moments", 0)
    try:
        for name, coul, kw in (("taxol", "simple", {}), ("pbc96_dsf15", "dsf", dict(dsf_rc=15.0, dsf_alpha=0.2))):
            g = golden(name)
            r1, _ = run(on, g, coul, stress=coul == "dsf", **kw)
            r0, _ = run(off, g, coul, stress=coul == "dsf", **kw)
            assert abs(r1["energy"][0] - r0["energy"][0]) < 2e-5, name
            assert np.abs(r1["forces"] - r0["forces"]).max() < 2e-5, name
            if coul == "dsf":
                assert np.abs(r1["stress"] - r0["stress"]).max() < 2e-6, name
    finally:
        on.set_option("split_max", -1)
