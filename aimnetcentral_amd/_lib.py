"""ctypes binding of libaimnet_hip.so (C ABI in include/aimnet_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C aimnetcentral_amd/csrc`.
There is NO fallback: if the shared object is missing or fails to load, every product entry
point raises `HipLibraryError` - the engine never silently routes through PyTorch or the oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from functools import lru_cache

MAX_PASS, MAX_LAYERS, MAX_SHIFTS = 4, 6, 32
ABI_VERSION = 11  # AIMNET_ABI_VERSION of include/aimnet_hip.h this binding was written against
FORCES, STRESS = 1, 2
COULOMB_NONE, COULOMB_SIMPLE, COULOMB_DSF, COULOMB_EWALD, COULOMB_PME = 0, 1, 2, 3, 4
E_INVALID, E_HIP, E_WORKSPACE = -1, -2, -3
PROF_FAMILIES = ("nlist", "geom", "conv_fwd", "gemm", "pointwise", "coulomb", "unconcat", "conv_bwd", "other")

# AIMNET_HIP_LIB names another build of the same library (A/B runs of kernel variants on one box); default: the in-tree build
LIB_PATH = os.environ.get("AIMNET_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libaimnet_hip.so")

# every symbol include/aimnet_hip.h declares (checked by tests/test_abi.py without a GPU)
EXPORTED_SYMBOLS = (
    "aimnet_abi_version",
    "aimnet_engine_create",
    "aimnet_engine_destroy",
    "aimnet_last_error",
    "aimnet_engine_workspace_bytes",
    "aimnet_engine_eval",
    "aimnet_engine_hvp_workspace_bytes",
    "aimnet_engine_hvp",
    "aimnet_engine_debug_view",
    "aimnet_engine_set_profiling",
    "aimnet_engine_set_profile_sampling",
    "aimnet_engine_profile_read",
    "aimnet_debug_gemm",
    "aimnet_debug_split_bf3",
    "aimnet_debug_gemm_bf3",
    "aimnet_debug_gemm_bf3a",
    "aimnet_debug_split_h2",
    "aimnet_debug_gemm_h2",
    "aimnet_engine_debug_mlp_sweep",
    "aimnet_debug_pme_recip",
    "aimnet_debug_mfma4_probe",
    "aimnet_engine_set_option",
    "aimnet_engine_get_option",
    "aimnet_engine_set_dftd3",
    "aimnet_engine_set_dd",
    "aimnet_neighbor_list",
    "aimnet_neighbor_list_workspace_bytes",
    "aimnet_conv_sv_2d_sp_fwd",
    "aimnet_conv_sv_2d_sp_bwd",
    "aimnet_conv_sv_2d_sp_bwd_bwd",
)


# aimnet_dd_exchange_fn of include/aimnet_hip.h (domain decomposition, aimnetcentral_amd/dd.py)
DD_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p)


class HipLibraryError(RuntimeError):
    pass


class Arch(C.Structure):
    _fields_ = [
        ("nfeature", C.c_int32),
        ("nshifts", C.c_int32),
        ("ncomb_v", C.c_int32),
        ("n_pass", C.c_int32),
        ("n_layers", C.c_int32 * MAX_PASS),
        ("layer_dims", (C.c_int32 * (MAX_LAYERS + 1)) * MAX_PASS),
        ("last_linear", C.c_int32 * MAX_PASS),
        ("head_n_layers", C.c_int32),
        ("head_dims", C.c_int32 * (MAX_LAYERS + 1)),
        ("rc", C.c_float),
        ("eta", C.c_float),
        ("shifts", C.c_float * MAX_SHIFTS),
        ("sr_coulomb", C.c_int32),
        ("sr_envelope", C.c_int32),
        ("sr_rc", C.c_float),
        ("n_charge_channels", C.c_int32),
    ]


class Weights(C.Structure):
    _fields_ = [
        ("afv", C.c_void_p),
        ("agh_a", C.c_void_p),
        ("agh_q", C.c_void_p),
        ("mlp_w", (C.c_void_p * MAX_LAYERS) * MAX_PASS),
        ("mlp_b", (C.c_void_p * MAX_LAYERS) * MAX_PASS),
        ("head_w", C.c_void_p * MAX_LAYERS),
        ("head_b", C.c_void_p * MAX_LAYERS),
        ("sae", C.c_void_p),
    ]


class Inputs(C.Structure):
    _fields_ = [
        ("n_atoms", C.c_int32),
        ("n_mol", C.c_int32),
        ("coord", C.c_void_p),
        ("numbers", C.c_void_p),
        ("mol_idx", C.c_void_p),
        ("charge", C.c_void_p),
        ("cell", C.c_void_p),
        ("n_cell", C.c_int32),
        ("pbc", C.c_int32 * 3),
        ("pbc_sys", C.c_void_p),
        # optional caller-supplied neighbour matrices (include/aimnet_hip.h): all NULL = the engine builds its lists
        ("nbmat", C.c_void_p),
        ("shifts", C.c_void_p),
        ("nbmat_width", C.c_int32),
        ("nbmat_lr", C.c_void_p),
        ("shifts_lr", C.c_void_p),
        ("nbmat_lr_width", C.c_int32),
        ("nbmat_d3", C.c_void_p),
        ("shifts_d3", C.c_void_p),
        ("nbmat_d3_width", C.c_int32),
    ]


class EvalOptions(C.Structure):
    _fields_ = [
        ("flags", C.c_uint32),
        ("coulomb", C.c_int32),
        ("dsf_rc", C.c_float),
        ("dsf_alpha", C.c_float),
        ("max_nb", C.c_int32),
        ("max_nb_lr", C.c_int32),
        ("dftd3", C.c_int32),
        ("d3_s6", C.c_float),
        ("d3_s8", C.c_float),
        ("d3_a1", C.c_float),
        ("d3_a2", C.c_float),
        ("d3_cutoff", C.c_float),
        ("d3_smoothing_on", C.c_float),
        ("max_nb_d3", C.c_int32),
        ("ewald_accuracy", C.c_float),
        ("ewald_max_k", C.c_int32),
        ("pme_max_mesh", C.c_int32),
    ]


class DftD3Tables(C.Structure):
    """aimnet_dftd3_tables: host pointers, Z-indexed."""

    _fields_ = [
        ("n_z", C.c_int32),
        ("c6ab", C.c_void_p),
        ("cn_ref", C.c_void_p),
        ("rcov", C.c_void_p),
        ("r4r2", C.c_void_p),
    ]


class Outputs(C.Structure):
    _fields_ = [
        ("energy", C.c_void_p),
        ("charges", C.c_void_p),
        ("forces", C.c_void_p),
        ("stress", C.c_void_p),
        ("status", C.c_void_p),
        ("spin_charges", C.c_void_p),
    ]


@lru_cache(maxsize=1)
def load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C aimnetcentral_amd/csrc`). The HIP engine has no CPU/PyTorch fallback."
        )
    # torch first: its wheel carries its own libamdhip64 / libhsa-runtime64, and the device pointers it hands us are only valid in
    # THAT runtime.  Loaded before torch, this library would pull /opt/rocm's copies into the process instead and a second HSA
    # runtime finds "no ROCm-capable device" (seen with `python __graft_entry__.py smoke`, where build() loads the library first).
    import torch  # noqa: F401

    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # missing ROCm runtime etc.
        raise HipLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    vp, i32, sz = C.c_void_p, C.c_int32, C.c_size_t
    lib.aimnet_abi_version.restype = C.c_int
    got = lib.aimnet_abi_version()
    if got != ABI_VERSION:  # the struct layouts below would silently mismatch (e.g. AIMNET_HIP_LIB naming an older build)
        raise HipLibraryError(f"{LIB_PATH} has ABI version {got}, this package binds version {ABI_VERSION}: rebuild the library")
    lib.aimnet_last_error.restype = C.c_char_p
    lib.aimnet_engine_create.restype = C.c_int
    lib.aimnet_engine_create.argtypes = [C.POINTER(Arch), C.POINTER(Weights), C.c_int, C.POINTER(vp)]
    lib.aimnet_engine_destroy.restype = None
    lib.aimnet_engine_destroy.argtypes = [vp]
    lib.aimnet_engine_workspace_bytes.restype = sz
    lib.aimnet_engine_workspace_bytes.argtypes = [vp, i32, i32, i32, C.POINTER(EvalOptions)]
    lib.aimnet_engine_eval.restype = C.c_int
    lib.aimnet_engine_eval.argtypes = [vp, C.POINTER(Inputs), C.POINTER(EvalOptions), C.POINTER(Outputs), vp, sz, vp]
    lib.aimnet_engine_hvp_workspace_bytes.restype = sz
    lib.aimnet_engine_hvp_workspace_bytes.argtypes = [vp, i32, i32, i32, C.POINTER(EvalOptions)]
    lib.aimnet_engine_hvp.restype = C.c_int
    lib.aimnet_engine_hvp.argtypes = [vp, C.POINTER(Inputs), C.POINTER(EvalOptions), vp, i32, vp, vp, vp, vp, sz, vp]
    lib.aimnet_engine_debug_view.restype = C.c_int
    lib.aimnet_engine_debug_view.argtypes = [vp, C.c_char_p, C.POINTER(sz), C.POINTER(sz), C.POINTER(i32), C.POINTER(i32)]
    lib.aimnet_engine_set_profiling.restype = C.c_int
    lib.aimnet_engine_set_profiling.argtypes = [vp, C.c_int]
    lib.aimnet_engine_set_profile_sampling.restype = C.c_int
    lib.aimnet_engine_set_profile_sampling.argtypes = [vp, C.c_int]
    lib.aimnet_engine_profile_read.restype = C.c_int
    lib.aimnet_engine_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.c_int, C.c_int]
    lib.aimnet_engine_set_dftd3.restype = C.c_int
    lib.aimnet_engine_set_dftd3.argtypes = [vp, C.POINTER(DftD3Tables)]
    lib.aimnet_engine_set_dd.restype = C.c_int
    lib.aimnet_engine_set_dd.argtypes = [vp, vp, DD_EXCHANGE_FN, vp]
    lib.aimnet_debug_gemm.restype = C.c_int
    lib.aimnet_debug_gemm.argtypes = [C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp]
    lib.aimnet_debug_split_bf3.restype = C.c_int
    lib.aimnet_debug_split_bf3.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]
    lib.aimnet_debug_gemm_bf3.restype = C.c_int
    lib.aimnet_debug_gemm_bf3.argtypes = [C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp]
    lib.aimnet_debug_gemm_bf3a.restype = C.c_int
    lib.aimnet_debug_gemm_bf3a.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int,
                                          vp, C.c_int, C.c_int, vp]
    lib.aimnet_engine_set_option.restype = C.c_int
    lib.aimnet_engine_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    lib.aimnet_engine_get_option.restype = C.c_int
    lib.aimnet_engine_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
    lib.aimnet_debug_split_h2.restype = C.c_int
    lib.aimnet_debug_split_h2.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]
    lib.aimnet_debug_gemm_h2.restype = C.c_int
    lib.aimnet_debug_gemm_h2.argtypes = list(lib.aimnet_debug_gemm_bf3a.argtypes)
    lib.aimnet_engine_debug_mlp_sweep.restype = C.c_int
    lib.aimnet_engine_debug_mlp_sweep.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.POINTER(vp), C.POINTER(vp),
                                                  C.POINTER(vp), C.POINTER(C.c_int), vp]
    lib.aimnet_debug_pme_recip.restype = C.c_int
    lib.aimnet_debug_pme_recip.argtypes = [vp, vp, vp, vp, C.c_float, C.c_int, C.c_float, C.c_int, vp, vp, vp, vp, C.POINTER(C.c_double), vp]
    lib.aimnet_debug_mfma4_probe.restype = C.c_int
    lib.aimnet_debug_mfma4_probe.argtypes = [vp, vp]
    lib.aimnet_neighbor_list_workspace_bytes.restype = sz
    lib.aimnet_neighbor_list_workspace_bytes.argtypes = [i32, i32, i32]
    lib.aimnet_neighbor_list.restype = C.c_int
    lib.aimnet_neighbor_list.argtypes = [vp, vp, i32, i32, vp, i32, C.POINTER(i32 * 3), C.c_float, i32, i32, vp, vp, vp,
                                         vp, vp, vp, sz, vp]
    lib.aimnet_conv_sv_2d_sp_fwd.restype = C.c_int
    lib.aimnet_conv_sv_2d_sp_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.aimnet_conv_sv_2d_sp_bwd.restype = C.c_int
    lib.aimnet_conv_sv_2d_sp_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.aimnet_conv_sv_2d_sp_bwd_bwd.restype = C.c_int
    lib.aimnet_conv_sv_2d_sp_bwd_bwd.argtypes = [vp] * 9 + [i32, i32, i32, i32, vp]
    return lib


def last_error() -> str:
    return load().aimnet_last_error().decode(errors="replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise HipLibraryError(f"{what} failed with code {rc}: {last_error()}")
