#!/usr/bin/env python
"""GEMM tuning / validation harness (GPU box): times aimnet_debug_gemm tile configs on the MLP shapes."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
M = int(os.environ.get("M", 10080))
shapes = [(512, 736), (384, 512), (256, 384), (736, 512), (512, 384), (128, 256)]
if os.environ.get("ONESHAPE"): shapes = shapes[:1]
cfgs = [int(c) for c in os.environ.get("CFGS", "5,152,153,122,351,331").split(",")]
EPI = int(os.environ.get("EPI", 2))
stream = torch.cuda.current_stream(dev).cuda_stream
for (N, K) in shapes:
    A = torch.randn(M, K, device=dev)
    Bt = torch.randn(N, K, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    ref = torch.nn.functional.gelu(A.double() @ Bt.double().T + bias.double())
    line = f"N={N:4d} K={K:4d}: "
    for cfg in cfgs:
        Cm = torch.empty(M, N, device=dev); D = torch.empty(M, N, device=dev)
        def run():
            rc = lib.aimnet_debug_gemm(cfg, EPI, A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), Cm.data_ptr(), D.data_ptr(), N, stream)
            assert rc == 0, _lib.last_error()
        run(); torch.cuda.synchronize()
        err = (Cm.double() - ref).abs().max().item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line += f" cfg{cfg}: {us:6.1f}us {2*M*N*K/us/1e6:6.1f}TF (err {err:.1e}) |"
    print(line)
