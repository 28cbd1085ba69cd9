#!/usr/bin/env python
"""Signed error statistics of the bf3 GEMM vs the exact-fp32 MFMA GEMM against fp64 (is the bf16 MFMA accumulation biased?)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
M, N, K = 6400, 512, 736
for name, mk in (("random sign", lambda *s: torch.randn(*s, device=dev)), ("positive", lambda *s: torch.rand(*s, device=dev) + 0.5)):
    A = mk(M, K); Bt = mk(N, K) * 0.05
    z = A.double() @ Bt.double().T
    B3 = torch.empty(N, 3 * K, dtype=torch.int16, device=dev)
    C3s = {}
    for frac in (1.0, 0.5, 0.6, 2 / 3, 0.75):
        KNEG = int(round(frac * (K // 32))) if frac < 1 else -1
        assert lib.aimnet_debug_split_bf3(Bt.data_ptr(), K, N, K, B3.data_ptr(), 3 * K, KNEG if KNEG >= 0 else 1 << 30, stream) == 0
        Cx = torch.empty(M, N, device=dev)
        assert lib.aimnet_debug_gemm_bf3(0, 0, A.data_ptr(), K, B3.data_ptr(), 3 * K, M, N, K, None, Cx.data_ptr(), None, N, KNEG, stream) == 0
        C3s[f"bf3 flip@{frac:.2f}"] = Cx
    KNEG = -1
    assert lib.aimnet_debug_split_bf3(Bt.data_ptr(), K, N, K, B3.data_ptr(), 3 * K, 1 << 30, stream) == 0
    C32 = torch.empty(M, N, device=dev); C3 = torch.empty(M, N, device=dev)
    assert lib.aimnet_debug_gemm(0, 0, A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, None, C32.data_ptr(), None, N, stream) == 0
    assert lib.aimnet_debug_gemm_bf3(0, 0, A.data_ptr(), K, B3.data_ptr(), 3 * K, M, N, K, None, C3.data_ptr(), None, N, KNEG, stream) == 0
    torch.cuda.synchronize()
    for tag, C in (("fp32 mfma", C32), ("bf3", C3), *C3s.items()):
        e = (C.double() - z) / z.abs().mean()
        print(f"{name:12s} {tag:14s} mean rel err {e.mean().item():+.3e}  rms {e.pow(2).mean().sqrt().item():.3e}  max {e.abs().max().item():.3e}")
