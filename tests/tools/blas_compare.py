"""Calibration only: the vendor fp32 GEMM (torch.mm -> hipBLASLt / rocBLAS) on the MLP layer shapes next to the engine's
plain-epilogue kernel (EPI_NONE), same M.  Not used by the product."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
stream = torch.cuda.current_stream(dev).cuda_stream
for M in (6400, 10080, 25600):
    for (N, K) in [(512, 736), (384, 512), (288, 384), (736, 512), (512, 384), (128, 256)]:
        A = torch.randn(M, K, device=dev); Bt = torch.randn(N, K, device=dev) * 0.05
        Cm = torch.empty(M, N, device=dev)
        def t(f, n=20):
            for _ in range(3): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): f()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        us_v = t(lambda: torch.mm(A, Bt.t(), out=Cm))
        us_e = t(lambda: lib.aimnet_debug_gemm(0, 0, A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, None, Cm.data_ptr(), None, N, stream))
        fl = 2 * M * N * K
        print(f"M={M:6d} N={N:4d} K={K:4d}  vendor {us_v:7.1f} us {fl/us_v/1e6:6.1f} TF   engine {us_e:7.1f} us {fl/us_e/1e6:6.1f} TF")
