"""The MFMA forms of the conv kernels (csrc/conv_mfma.hip: forward, T-layout unconcat, backward on v_mfma_f32_4x4x1_16B_f32)
against the oracle, the reference goldens and the default packed-FMA kernels.  They are an engine option
(`set_option("conv_mfma", 3)`); `split_max = 0` sends even the small fixtures through the one-wave-per-atom kernels that only
systems above 1 024 atoms take by default.

These kernels are north-star evidence (the contractions on the matrix pipe, measured not faster: profiles/r2_conv_mfma.md), not a
path anything runs by default: the module is collected only with AIMNET_TEST_CONV_MFMA=1."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import golden
from oracle import aimnet2_oracle as O
from test_gpu_parity import compare, run

import os

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("AIMNET_TEST_CONV_MFMA") != "1",
                                                  reason="optional kernels: set AIMNET_TEST_CONV_MFMA=1 to test csrc/conv_mfma.hip")]


@pytest.fixture()
def engines(synth_sd):
    from aimnetcentral_amd import loader
    from aimnetcentral_amd.engine import HipEngine

    spec = loader.synthetic_spec(0)
    valu, mfma = HipEngine(spec, "cuda:0"), HipEngine(spec, "cuda:0")
    for e in (valu, mfma):
        e.set_option("split_max", 0)  # per engine: both run the large-system kernels on every fixture
    mfma.set_option("conv_mfma", 3)
    yield valu, mfma


def test_taxol_and_batch(engines, oracle32, oracle64):
    valu, mfma = engines
    g = golden("taxol")
    r, _ = run(mfma, g, "simple")
    compare(r, O.evaluate(oracle32, g["coord"], g["numbers"], g["charge"]), 113, "taxol mfma/oracle")
    compare(r, g, 113, "taxol mfma/reference golden")
    r0, _ = run(valu, g, "simple")
    assert np.abs(r["forces"] - r0["forces"]).max() < 2e-5 and abs(r["energy"][0] - r0["energy"][0]) < 1e-5
    g = golden("batch5")
    r, mol = run(mfma, g, "simple")
    e64 = O.evaluate(oracle64, g["coord"], g["numbers"], g["charge"], mol, forces=False)["energy"]
    compare(r, g, np.bincount(mol), "batch5 mfma/reference golden", e64)


@pytest.mark.parametrize("name", ["pbc96_dsf15", "pbc2x96_dsf9"])
def test_periodic_stress(engines, oracle32, oracle64, name):
    valu, mfma = engines
    g = golden(name)
    kw = dict(dsf_rc=float(g["dsf_rc"]), dsf_alpha=float(g["dsf_alpha"]))
    r, mol = run(mfma, g, "dsf", stress=True, **kw)
    e64 = O.evaluate(oracle64, g["coord"], g["numbers"], g["charge"], mol, cell=g["cell"], coulomb="dsf", forces=False, **kw)["energy"]
    compare(r, g, 96, name + " mfma/reference golden", e64)
    r0, _ = run(valu, g, "dsf", stress=True, **kw)
    assert np.abs(r["stress"] - r0["stress"]).max() < 2e-6 and np.abs(r["forces"] - r0["forces"]).max() < 2e-5


def test_each_kernel_alone_and_repeatability(engines):
    """forward-only / backward-only MFMA (bits 1, 2) agree with both-on; two evaluations are bitwise identical."""
    valu, mfma = engines
    g = golden("pbc96_dsf15")
    kw = dict(dsf_rc=15.0, dsf_alpha=0.2)
    ref, _ = run(mfma, g, "dsf", stress=True, **kw)
    again, _ = run(mfma, g, "dsf", stress=True, **kw)
    assert np.array_equal(ref["forces"], again["forces"]) and np.array_equal(ref["energy"], again["energy"])
    for bits in (1, 2):
        mfma.set_option("conv_mfma", bits)
        r, _ = run(mfma, g, "dsf", stress=True, **kw)
        assert np.abs(r["forces"] - ref["forces"]).max() < 2e-5, bits
    mfma.set_option("conv_mfma", 3)


def test_nse_two_channel_model(synth_sd_nse, oracle32_nse):
    from aimnetcentral_amd import loader
    from aimnetcentral_amd.engine import HipEngine

    eng = HipEngine(loader.synthetic_spec(0, num_charge_channels=2), "cuda:0")
    eng.set_option("split_max", 0)
    eng.set_option("conv_mfma", 3)
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
