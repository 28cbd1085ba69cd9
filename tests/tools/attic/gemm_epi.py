#!/usr/bin/env python
"""What the fused epilogues cost: the engine's own tile choice (cfg 0) per MLP layer shape with every epilogue
(0 plain, 1 bias, 2 bias + GELU + GELU' store, 3 multiply by D), warm buffers, 20 back-to-back launches."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
M = int(os.environ.get("M", 10080))
stream = torch.cuda.current_stream(dev).cuda_stream
for (N, K) in [(512, 448), (512, 736), (384, 512), (288, 384), (736, 512), (512, 384), (384, 288), (128, 256)]:
    A = torch.randn(M, K, device=dev); Bt = torch.randn(N, K, device=dev) * 0.05; bias = torch.randn(N, device=dev)
    line = f"N={N:4d} K={K:4d}: "
    for epi, with_d in ((0, 0), (1, 0), (2, 1), (2, 0), (3, 1)):
        Cm = torch.empty(M, N, device=dev); D = torch.rand(M, N, device=dev)
        def run():
            rc = lib.aimnet_debug_gemm(0, epi, A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), Cm.data_ptr(), D.data_ptr() if with_d else None, N, stream)
            assert rc == 0, _lib.last_error()
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line += f" epi{epi}{'+D' if with_d else '  '}: {us:6.1f}us {2*M*N*K/us/1e6:6.1f}TF |"
    print(line)
