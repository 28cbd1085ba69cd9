#!/usr/bin/env python
"""One h2 GEMM launch per (shape, tile) with a correctness check (GPU box; bisecting tool).  Env: SHAPE=N,K  M  CFG  EPI  OUT."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
M = int(os.environ.get("M", 10080))
N, K = (int(x) for x in os.environ.get("SHAPE", "512,736").split(","))
cfg, EPI, OUT = int(os.environ.get("CFG", 0)), int(os.environ.get("EPI", 0)), int(os.environ.get("OUT", 0))
stream = torch.cuda.current_stream(dev).cuda_stream
pad32 = lambda n: (n + 31) // 32 * 32  # noqa: E731
A, Bt = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05
bias, Dm = torch.randn(N, device=dev), torch.rand(M, N, device=dev)


def split2(x, mode):
    m, k = x.shape
    out = torch.zeros(m + 64, 2 * pad32(k), dtype=torch.int16, device=dev)  # (slack rows: an over-read shows as a wrong result, not a fault)
    assert lib.aimnet_debug_split_h2(x.data_ptr(), k, m, k, out.data_ptr(), 2 * pad32(k), mode, stream) == 0
    return out


A2, B2 = split2(A, 1), split2(Bt, 2)
C, C2 = torch.zeros(M, N, device=dev), torch.zeros(M, 2 * pad32(N), dtype=torch.int16, device=dev)
print("launching", N, K, M, cfg, flush=True)
rc = lib.aimnet_debug_gemm_h2(cfg, EPI, OUT, A2.data_ptr(), 2 * pad32(K), B2.data_ptr(), 2 * K, M, N, K, bias.data_ptr(), C.data_ptr(),
                              C2.data_ptr(), 2 * pad32(N), Dm.data_ptr(), N, 1, stream)
assert rc == 0, _lib.last_error()
torch.cuda.synchronize()
ref = A.double() @ Bt.double().T
print("ok; max err", (C.double() - ref).abs().max().item(), flush=True)
