#!/usr/bin/env python
"""Segment timeline of one block of the bf3 GEMM (measurement build, tests/tools/bf3_timing.sh)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
M, N, K = 10080, 512, 736
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 452
A = torch.randn(M, K, device=dev)
Bt = torch.randn(N, K, device=dev) * 0.05
bias = torch.randn(N, device=dev)
B3 = torch.empty(N, 3 * K, dtype=torch.int16, device=dev)
KNEG = (2 * (K // 32) + 1) // 3
stream = torch.cuda.current_stream(dev).cuda_stream
assert lib.aimnet_debug_split_bf3(Bt.data_ptr(), K, N, K, B3.data_ptr(), 3 * K, KNEG, stream) == 0
Cm = torch.empty(M, N, device=dev)
D = torch.empty(M, N, device=dev)
for _ in range(3):
    rc = lib.aimnet_debug_gemm_bf3(cfg, 2, A.data_ptr(), K, B3.data_ptr(), 3 * K, M, N, K, bias.data_ptr(), Cm.data_ptr(), D.data_ptr(), N, KNEG, stream)
    assert rc == 0, _lib.last_error()
torch.cuda.synchronize()
ref = torch.nn.functional.gelu(A.double() @ Bt.double().T + bias.double())
print("max err vs fp64", (Cm.double() - ref).abs().max().item())
if not hasattr(lib, "aimnet_debug_bf3_stamps"):
    sys.exit(0)
buf = (C.c_ulonglong * 1024)()
lib.aimnet_debug_bf3_stamps.argtypes = [C.c_void_p]
assert lib.aimnet_debug_bf3_stamps(buf) == 0
t = np.array(buf[:], dtype=np.int64).reshape(2, 512)
t0 = t[0][0]
for g in range(2):
    n = int((t[g] > 0).sum())
    rel = t[g][:n] - t0
    print(f"group {g}: {n} stamps; deltas:", " ".join(str(int(d)) for d in np.diff(rel)))
    print(f"   first {rel[0]} last {rel[-1]}")
