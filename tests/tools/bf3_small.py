#!/usr/bin/env python
"""bf3 GEMM correctness over K (number of 32-k steps) and M for one tile id."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 452
stream = torch.cuda.current_stream(dev).cuda_stream
for M, N in ((160, 128), (1000, 288)):
    for nk in (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 23):
        K = 32 * nk
        g = torch.Generator().manual_seed(nk)
        A = torch.randn(M, K, generator=g).to(dev)
        Bt = (torch.randn(N, K, generator=g) * 0.1).to(dev)
        KNEG = (2 * nk + 1) // 3 if nk > 1 else -1
        B3 = torch.empty(N, 3 * K, dtype=torch.int16, device=dev)
        assert lib.aimnet_debug_split_bf3(Bt.data_ptr(), K, N, K, B3.data_ptr(), 3 * K, KNEG if KNEG >= 0 else 1 << 30, stream) == 0
        Cm = torch.full((M, N), float("nan"), device=dev)
        rc = lib.aimnet_debug_gemm_bf3(cfg, 0, A.data_ptr(), K, B3.data_ptr(), 3 * K, M, N, K, None, Cm.data_ptr(), None, N, KNEG, stream)
        assert rc == 0, _lib.last_error()
        torch.cuda.synchronize()
        err = (Cm.double() - A.double() @ Bt.double().T).abs()
        bad = (err > 1e-4) | err.isnan()
        rows = bad.any(dim=1).nonzero().flatten().tolist()
        print(f"cfg {cfg} M={M} N={N} nk={nk}: max err {err.nan_to_num(9e9).max().item():.2e}  bad rows {len(rows)} {rows[:12]}", flush=True)
