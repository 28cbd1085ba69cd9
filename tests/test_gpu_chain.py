"""The one-launch MLP sweeps of csrc/gemm_chain.hip (engine option "gemm_chain", default on) against the per-layer launches of
csrc/gemm_h2.hip they replace (the GEMM + GELU chain of aimnet/modules/core.py:11-46, call site aimnet/models/aimnet2.py:166).

Same products, same accumulation order, same epilogue arithmetic: every output is compared BITWISE - GELU' of every layer, the last
layer's output, the input adjoint of the backward sweep - for panel heights 16 / 32 / 48 (chosen by the row count), ragged last
panels, both pass-0 backward variants, and through the whole evaluation (energies, forces, charges, stress)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest
import torch

from aimnetcentral_amd import _lib, workloads

pytestmark = pytest.mark.gpu

LAYER_DIMS = [[704, 512, 380, 258], [733, 512, 380, 258], [733, 512, 380, 380, 256]]  # synthetic_spec(0) = the shipped architecture


def pad32(n):
    return (n + 31) // 32 * 32


def _split2(eng, x, mode=1):
    m, k = x.shape
    out = torch.zeros(m, 2 * pad32(k), dtype=torch.int16, device=x.device)
    st = torch.cuda.current_stream(x.device).cuda_stream
    assert eng.lib.aimnet_debug_split_h2(x.data_ptr(), k, m, k, out.data_ptr(), 2 * pad32(k), mode, st) == 0, _lib.last_error()
    return out


def _ptrs(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() if t is not None else None for t in ts])


def _sweeps(eng, p, M, seed):
    dev = eng.device
    st = torch.cuda.current_stream(dev).cuda_stream
    d = LAYER_DIMS[p]
    nl, kp = len(d) - 1, [pad32(v) for v in d]
    g = torch.Generator(device="cpu").manual_seed(seed)
    numbers = torch.tensor([1, 6, 7, 8], dtype=torch.int32)[torch.randint(0, 4, (M,), generator=g)].to(dev)
    x2 = _split2(eng, torch.randn(M, d[0], generator=g).to(dev))
    zbar2 = _split2(eng, torch.randn(M, d[nl], generator=g).to(dev))
    out = {}
    for chain in (0, 1):
        H = [torch.zeros(M, kp[l + 1], device=dev) for l in range(nl)]
        D = [torch.zeros(M, kp[l + 1], device=dev) for l in range(nl)]
        rc = eng.lib.aimnet_engine_debug_mlp_sweep(eng._h, p, 0, chain, 0, x2.data_ptr(), M, numbers.data_ptr(), _ptrs(H), _ptrs(D), None, None, st)
        assert rc == 0, _lib.last_error()
        res = {"D": D, "out": H[nl - 1]}
        for flag in ((1, 0) if p == 0 else (0,)):
            zb = [torch.zeros(M * 2 * max(kp), dtype=torch.int16, device=dev) for _ in range(2)]
            zb[0][: M * 2 * kp[nl]] = zbar2.view(-1)  # dense rows, row stride 2 * k_out of the last layer
            which = C.c_int(-1)
            rc = eng.lib.aimnet_engine_debug_mlp_sweep(eng._h, p, 1, chain, flag, zb[0].data_ptr(), M, numbers.data_ptr(), None, _ptrs(out[0]["D"] if chain else D),
                                                       _ptrs(zb), C.byref(which), st)
            assert rc == 0, _lib.last_error()
            xbar = zb[which.value].view(torch.float32)[: M * kp[0]].view(M, kp[0]).clone()
            res[f"xbar{flag}"] = xbar[:, 256:] if flag else xbar
        torch.cuda.synchronize()
        out[chain] = res
    return out, nl


@pytest.mark.parametrize("M", [300, 1000, 2311, 5000, 10080, 12400])
@pytest.mark.parametrize("p", [0, 1, 2])
def test_sweeps_bitwise_equal_to_the_per_layer_launches(hip_engine, p, M):
    """panel heights 16 (M <= 4096), 32 (<= 8192), 48 rows; ragged last panels; 12 400 rows = more panels than CUs"""
    out, nl = _sweeps(hip_engine, p, M, seed=17 * p + M)
    a, b = out[0], out[1]
    for l in range(nl):
        if l < nl - 1 or p == 2:  # (the last layer of passes 0 / 1 is linear: no GELU')
            assert torch.equal(a["D"][l], b["D"][l]), f"GELU' of layer {l}"
            assert torch.isfinite(b["D"][l]).all()
    assert torch.equal(a["out"], b["out"]) and torch.isfinite(b["out"]).all()
    for key in a:
        if key.startswith("xbar"):
            assert torch.equal(a[key], b[key]), key
            assert torch.isfinite(b[key]).all() and b[key].abs().max() > 0


@pytest.mark.parametrize("rep", [(1, 2, 2), (2, 3, 4)])
def test_evaluation_bitwise_equal(hip_engine, rep):
    """energy, forces, charges and stress of a periodic DSF evaluation with and without the one-launch sweeps"""
    eng, dev = hip_engine, hip_engine.device
    c, z, cell = workloads.glucose_supercell(rep)
    rng = np.random.default_rng(3)
    c = torch.from_numpy((c + rng.normal(0, 0.02, c.shape)).astype(np.float32)).to(dev)
    z = torch.from_numpy(z).to(dev)
    cell = torch.from_numpy(cell.astype(np.float32)).to(dev)
    res = {}
    try:
        for mode in (0, 1):
            eng.set_option("gemm_chain", mode)
            r = eng.eval(c, z, torch.zeros(len(z), dtype=torch.int64, device=dev), torch.zeros(1, device=dev), cell=cell, forces=True, stress=True,
                         coulomb="dsf", dsf_rc=15.0)
            res[mode] = {k: v.clone() for k, v in r.items()}
    finally:
        eng.set_option("gemm_chain", 1)
    for k in ("energy", "forces", "charges", "stress"):
        assert torch.equal(res[0][k], res[1][k]), k
