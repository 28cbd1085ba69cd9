#!/usr/bin/env python
"""fixed overhead vs steady-state rate of the GEMM kernels: time vs K at fixed M x N."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
M, N = int(os.environ.get("M", 10080)), int(os.environ.get("N", 512))
stream = torch.cuda.current_stream(dev).cuda_stream
EPI = int(os.environ.get('EPI', 0))
for cfg in [int(c) for c in os.environ.get('CFGS', '5,152,351').split(',')]:
    line = f"cfg{cfg:5d} M={M} N={N}: "
    for K in (32, 128, 256, 512, 736, 1472, 2944):
        A = torch.randn(M, K, device=dev); Bt = torch.randn(N, K, device=dev) * 0.05
        C = torch.empty(M, N, device=dev); D = torch.empty(M, N, device=dev); bias = torch.randn(N, device=dev)
        def run():
            assert lib.aimnet_debug_gemm(cfg, EPI, A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), C.data_ptr(), D.data_ptr(), N, stream) == 0
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        line += f"K={K}:{e0.elapsed_time(e1)/30*1e3:6.1f}us "
    print(line)
