#!/usr/bin/env python
"""Every panel tile id on ONE layer shape (env N, K, EPI; M = 10080): which tile the chooser should pick."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
M, N, K, EPI = int(os.environ.get("M", 10080)), int(os.environ.get("N", 512)), int(os.environ.get("K", 448)), int(os.environ.get("EPI", 2))
stream = torch.cuda.current_stream(dev).cuda_stream
A = torch.randn(M, K, device=dev); Bt = torch.randn(N, K, device=dev) * 0.05; bias = torch.randn(N, device=dev)
Cm = torch.empty(M, N, device=dev); D = torch.empty(M, N, device=dev)
ids = [152, 142, 132, 122, 153, 143, 223, 213, 222, 412, 411, 410, 409, 381, 371, 361, 351, 341, 331, 321, 233]
res = []
for base in [0] + ids + [1000 + i for i in ids]:
    def run():
        return lib.aimnet_debug_gemm(base, EPI, A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), Cm.data_ptr(), D.data_ptr(), N, stream)
    if run() != 0:
        continue
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    res.append((e0.elapsed_time(e1) / 20 * 1e3, base))
res.sort()
print(f"N={N} K={K} EPI={EPI}: chooser {[r for r in res if r[1] == 0][0][0]:.1f} us; best " + "  ".join(f"{b}:{t:.1f}" for t, b in res[:8]))
