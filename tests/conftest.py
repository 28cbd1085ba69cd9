"""pytest configuration: `-m gpu` tests need an MI355X (run via gpurun / the driver), everything
else runs on CPU in the build container."""
from __future__ import annotations

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

# tolerances the reference states for itself (tests/conftest.py:162-165, test_calculator_gpu.py:445,464)
ENERGY_ATOL = 1e-5
ENERGY_PER_ATOM = 5e-7  # fp32 summation-order noise allowance for large molecules
FORCE_ATOL, FORCE_RTOL = 1e-5, 1e-4
CHARGE_ATOL = 1e-4
STRESS_ATOL = 1e-5


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); run with `-m gpu` on the GPU box")


def golden(name: str):
    return np.load(os.path.join(GOLD, name + ".npz"))


def energy_tol(n_atoms_per_mol) -> float:
    return max(ENERGY_ATOL, ENERGY_PER_ATOM * float(np.max(n_atoms_per_mol)))


def assert_forces_close(got, ref, what=""):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    tol = FORCE_ATOL + FORCE_RTOL * np.abs(ref).max()
    err = np.abs(got - ref).max()
    assert err <= tol, f"{what} forces: max|d|={err:.3e} > {tol:.3e}"


def elementwise_violations(got, ref, rtol: float = FORCE_RTOL, atol: float = FORCE_ATOL) -> tuple[int, int, float]:
    """The reference's LITERAL force gate, torch.allclose / np.testing.assert_allclose(rtol=1e-4, atol=1e-5)
    (tests/test_calculator_gpu.py:137,464): (number of elements with |got - ref| > atol + rtol |ref|, number of elements, the
    largest |got - ref| / (atol + rtol |ref|))."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    ratio = np.abs(got - ref) / (atol + rtol * np.abs(ref))
    return int((ratio > 1.0).sum()), int(ratio.size), float(ratio.max()) if ratio.size else 0.0


def golden_section(g, prefix: str) -> dict:
    """The arrays `prefix_*` of a multi-fixture golden file (coldw.npz) as a dict keyed without the prefix."""
    return {k[len(prefix) + 1:]: g[k] for k in g.files if k.startswith(prefix + "_")}


COLDW_SECTIONS = ("taxol", "batch5", "rand8", "pbc96")


@pytest.fixture(scope="session")
def synth_sd_cold():
    from aimnetcentral_amd import synth

    return synth.synthetic_state_dict(0, cold=True)


@pytest.fixture(scope="session")
def oracle32_cold(synth_sd_cold):
    import torch

    from oracle import aimnet2_oracle as O

    return O.OracleModel(synth_sd_cold, torch.float32)


@pytest.fixture(scope="session")
def hip_engine_cold():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from aimnetcentral_amd import loader
    from aimnetcentral_amd.engine import HipEngine

    return HipEngine(loader.synthetic_spec(0, cold=True), "cuda:0")


@pytest.fixture(scope="session")
def synth_sd():
    from aimnetcentral_amd import synth

    return synth.synthetic_state_dict(0)


@pytest.fixture(scope="session")
def oracle32(synth_sd):
    import torch

    from oracle import aimnet2_oracle as O

    return O.OracleModel(synth_sd, torch.float32)


@pytest.fixture(scope="session")
def oracle64(synth_sd):
    import torch

    from oracle import aimnet2_oracle as O

    return O.OracleModel(synth_sd, torch.float64)


@pytest.fixture(scope="session")
def synth_sd_nse():
    from aimnetcentral_amd import synth

    return synth.synthetic_state_dict(0, None, 2)


@pytest.fixture(scope="session")
def oracle32_nse(synth_sd_nse):
    import torch

    from oracle import aimnet2_oracle as O

    return O.OracleModel(synth_sd_nse, torch.float32)


@pytest.fixture(scope="session")
def oracle64_nse(synth_sd_nse):
    import torch

    from oracle import aimnet2_oracle as O

    return O.OracleModel(synth_sd_nse, torch.float64)


@pytest.fixture(scope="session")
def hip_engine_nse():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from aimnetcentral_amd import loader
    from aimnetcentral_amd.engine import HipEngine

    return HipEngine(loader.synthetic_spec(0, num_charge_channels=2), "cuda:0")


@pytest.fixture(scope="session")
def hip_engine():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from aimnetcentral_amd import loader
    from aimnetcentral_amd.engine import HipEngine

    return HipEngine(loader.synthetic_spec(0), "cuda:0")
