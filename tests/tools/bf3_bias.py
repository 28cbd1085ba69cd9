#!/usr/bin/env python
"""Signed error statistics of the bf16x3-split GEMMs against fp64: is the bf16 MFMA accumulation biased, and what cancels it?

  fp32 mfma      exact-fp32 MFMA chain (gemm.hip)
  bf3            gemm_bf3.hip, one-signed accumulation
  bf3 flip@x     gemm_bf3.hip, second accumulation phase sign-flipped from x K on (round 3; the engine used 0.56 - a fitted constant)
  bf3a sum       gemm_bf3a.hip, two accumulator sets added (weights with their own signs): same bias as `bf3`
  bf3a alt       gemm_bf3a.hip, odd k-blocks negated, even - odd (round 4, the engine's form: no tunable)
Operands: random sign, all positive, and a growing / decaying magnitude profile along k (the flip point that balances the two phases
depends on how |acc| grows over k; the interleaved sets do not care)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
M, N, K = 6400, 512, 736
ramp = torch.linspace(0.05, 2.0, K, device=dev)
cases = (("random sign", lambda *s: torch.randn(*s, device=dev)), ("positive", lambda *s: torch.rand(*s, device=dev) + 0.5),
         ("growing |a_k|", lambda *s: torch.randn(*s, device=dev) * ramp), ("decaying |a_k|", lambda *s: torch.randn(*s, device=dev) * ramp.flip(0)))


def split(x, neg):
    m, k = x.shape
    out = torch.empty(m, 3 * k, dtype=torch.int16, device=dev)
    assert lib.aimnet_debug_split_bf3(x.data_ptr(), k, m, k, out.data_ptr(), 3 * k, neg, stream) == 0, _lib.last_error()
    return out


for name, mk in cases:
    A = mk(M, K); Bt = torch.randn(N, K, device=dev) * 0.05 if "a_k" in name else mk(N, K) * 0.05
    z = A.double() @ Bt.double().T
    res = {}
    C32 = torch.empty(M, N, device=dev)
    assert lib.aimnet_debug_gemm(0, 0, A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, None, C32.data_ptr(), None, N, stream) == 0
    res["fp32 mfma"] = C32
    for frac in (1.0, 0.5, 0.56, 0.6, 2 / 3, 0.75):
        KNEG = int(round(frac * (K // 32))) if frac < 1 else -1
        Cx = torch.empty(M, N, device=dev)
        assert lib.aimnet_debug_gemm_bf3(0, 0, A.data_ptr(), K, split(Bt, KNEG if KNEG >= 0 else 1 << 30).data_ptr(), 3 * K, M, N, K, None,
                                         Cx.data_ptr(), None, N, KNEG, stream) == 0
        res["bf3" if frac >= 1 else f"bf3 flip@{frac:.2f}"] = Cx
    A3 = split(A, 1 << 30)
    for tag, neg, alt in (("bf3a sum", 1 << 30, 0), ("bf3a alt", -2, 1)):
        Cx = torch.empty(M, N, device=dev)
        assert lib.aimnet_debug_gemm_bf3a(0, 0, 0, A3.data_ptr(), 3 * K, split(Bt, neg).data_ptr(), 3 * K, M, N, K, None, Cx.data_ptr(), None, 0,
                                          None, N, alt, stream) == 0, _lib.last_error()
        res[tag] = Cx
    torch.cuda.synchronize()
    for tag, C in res.items():
        e = (C.double() - z) / z.abs().mean()
        print(f"{name:14s} {tag:14s} mean rel err {e.mean().item():+.3e}  rms {e.pow(2).mean().sqrt().item():.3e}  max {e.abs().max().item():.3e}")
