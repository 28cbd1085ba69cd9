"""The C-ABI shared library loads and exports every symbol include/aimnet_hip.h declares
(no compute calls: there is no GPU in the build container)."""
from __future__ import annotations

import os
import re

import pytest

from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "aimnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(aimnet_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from aimnetcentral_amd import _lib

    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 11
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/aimnet_hip.h but not exported"
    assert set(declared) == set(_lib.EXPORTED_SYMBOLS)
    assert lib.aimnet_abi_version() == _lib.ABI_VERSION == 11


def test_struct_layouts_match_header():
    """ctypes mirrors of the header structs: sizes follow from the declared field lists."""
    import ctypes as C

    from aimnetcentral_amd import _lib

    assert C.sizeof(_lib.EvalOptions) == 24 + 8 * 4 + 3 * 4  # + dftd3, s6, s8, a1, a2, cutoff, smoothing_on, max_nb_d3; + ewald_accuracy, ewald_max_k, pme_max_mesh
    assert C.sizeof(_lib.DftD3Tables) == 8 + 4 * 8
    assert C.sizeof(_lib.Inputs) == 8 + 5 * 8 + 4 + 12 + 8 + 3 * 24  # + pbc_sys; + three optional caller-supplied matrices
    assert C.sizeof(_lib.Outputs) == 6 * 8      # + spin_charges
    n_arch = 4 + 4 + 4 * 7 + 4 + 1 + 7 + 2 + 32 + 3 + 1  # + n_charge_channels
    assert C.sizeof(_lib.Arch) == 4 * n_arch


def test_invalid_arguments_are_error_codes_not_crashes():
    from aimnetcentral_amd import _lib

    lib = _lib.load()
    assert lib.aimnet_engine_create(None, None, 0, None) == _lib.E_INVALID
    assert lib.aimnet_engine_workspace_bytes(None, 10, 1, 0, None) == 0
    assert lib.aimnet_neighbor_list_workspace_bytes(0, 1, 16) == 0
    assert lib.aimnet_neighbor_list_workspace_bytes(100, 1, 16) > 0
    assert lib.aimnet_engine_set_profiling(None, 1) == _lib.E_INVALID
    assert lib.aimnet_engine_set_dftd3(None, None) == _lib.E_INVALID


def test_engine_refuses_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from aimnetcentral_amd import HipLibraryError, loader
    from aimnetcentral_amd.engine import HipEngine

    with pytest.raises(HipLibraryError):
        HipEngine(loader.synthetic_spec(0), "cuda:0")


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under aimnetcentral_amd/ may reference it."""
    pkg = os.path.join(ROOT, "aimnetcentral_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
