#!/usr/bin/env python
"""fp16x2-split GEMM (csrc/gemm_h2.hip) against the bf16x3-split one (gemm_bf3a.hip) on the MLP layer shapes (GPU box): time,
error against fp64 (max, rms, MEAN = accumulation bias) of both and of the exact-fp32 MFMA kernel.

Env: M, CFGS (tile ids, 0 = automatic), EPI (2 GELU / 3 chain rule / 0 / 1), OUT (1: split output), SHAPES=all|one, STAT=randn|pos|gelu,
STAMPS=1 (timing build).
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
M = int(os.environ.get("M", 10080))
shapes = [(512, 736), (512, 448), (384, 512), (288, 384), (384, 384), (256, 384), (128, 256), (128, 128),
          (736, 512), (448, 512), (512, 384), (384, 288), (384, 256), (256, 128)]
if os.environ.get("SHAPES", "all") == "one":
    shapes = shapes[:1]
cfgs = [int(c) for c in os.environ.get("CFGS", "0").split(",")]
EPI = int(os.environ.get("EPI", 2))
OUT = int(os.environ.get("OUT", 1 if EPI in (2, 3) else 0))
STAT = os.environ.get("STAT", "randn")
stream = torch.cuda.current_stream(dev).cuda_stream


def pad32(n):
    return (n + 31) // 32 * 32


def split3(x, neg=1 << 30):
    m, k = x.shape
    out = torch.empty(m, 3 * pad32(k), dtype=torch.int16, device=dev)
    assert lib.aimnet_debug_split_bf3(x.data_ptr(), k, m, k, out.data_ptr(), 3 * pad32(k), neg, stream) == 0, _lib.last_error()
    return out


def split2(x, mode):
    m, k = x.shape
    out = torch.empty(m, 2 * pad32(k), dtype=torch.int16, device=dev)
    assert lib.aimnet_debug_split_h2(x.data_ptr(), k, m, k, out.data_ptr(), 2 * pad32(k), mode, stream) == 0, _lib.last_error()
    return out


def unsplit3(c3, n):
    m = c3.shape[0]
    v = c3.view(m, -1)[:, : 3 * pad32(n)].reshape(m, pad32(n) // 32, 3, 32).to(torch.int32) << 16
    return v.view(torch.float32).double().sum(dim=2).reshape(m, pad32(n))[:, :n]


def unsplit2(c2, n):
    m = c2.shape[0]
    v = c2.view(m, -1)[:, : 2 * pad32(n)].reshape(m, pad32(n) // 32, 2, 32).view(torch.float16).double()
    sign = torch.where(torch.arange(pad32(n) // 32, device=dev) % 2 == 1, -1.0, 1.0).double().view(1, -1, 1)
    return (v[:, :, 0] + sign * v[:, :, 1] / 4096.0).reshape(m, pad32(n))[:, :n]


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def stats(got, ref):
    d = got - ref
    sc = ref.abs().mean().item()
    return d.abs().max().item() / sc, d.pow(2).mean().sqrt().item() / sc, d.mean().item() / sc


tot3, tot2 = 0.0, {c: 0.0 for c in cfgs}
for (N, K) in shapes:
    torch.manual_seed(N * 1000 + K)
    if STAT == "pos":
        A = torch.rand(M, K, device=dev)
        Bt = torch.rand(N, K, device=dev) * 0.05
    elif STAT == "gelu":
        A = torch.nn.functional.gelu(torch.randn(M, K, device=dev) * 1.5)
        Bt = torch.randn(N, K, device=dev) * 0.05
    else:
        A = torch.randn(M, K, device=dev)
        Bt = torch.randn(N, K, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    Dm = torch.rand(M, N, device=dev)
    z = A.double() @ Bt.double().T
    ref = {0: z, 1: z + bias.double(), 2: torch.nn.functional.gelu(z + bias.double()), 3: z * Dm.double()}[EPI]
    A3, B3a = split3(A), split3(Bt, -2)
    A2, B2 = split2(A, 1), split2(Bt, 2)
    # exact-fp32 MFMA kernel
    Cx, Dx = torch.empty(M, N, device=dev), Dm.clone()
    assert lib.aimnet_debug_gemm(0, EPI, A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), Cx.data_ptr(), Dx.data_ptr(), N, stream) == 0
    sx = stats(Cx.double(), ref)
    C3f, C3 = torch.zeros(M, N, device=dev), torch.zeros(M, 3 * pad32(N), dtype=torch.int16, device=dev)
    D3 = Dm.clone()

    def run3():
        rc = lib.aimnet_debug_gemm_bf3a(0, EPI, OUT, A3.data_ptr(), 3 * pad32(K), B3a.data_ptr(), 3 * K, M, N, K, bias.data_ptr(),
                                        C3f.data_ptr(), C3.data_ptr(), 3 * pad32(N), D3.data_ptr(), N, 1, stream)
        assert rc == 0, _lib.last_error()

    us3 = timeit(run3)
    tot3 += us3
    s3 = stats(unsplit3(C3, N) if OUT else C3f.double(), ref)
    line = (f"N={N:4d} K={K:4d}: f32 max {sx[0]:.1e} rms {sx[1]:.1e} mean {sx[2]:+.1e} | bf3a {us3:6.1f}us max {s3[0]:.1e} rms {s3[1]:.1e} "
            f"mean {s3[2]:+.1e} |")
    for cfg in cfgs:
        C2f, C2 = torch.zeros(M, N, device=dev), torch.zeros(M, 2 * pad32(N), dtype=torch.int16, device=dev)
        D2 = Dm.clone()

        def run2():
            rc = lib.aimnet_debug_gemm_h2(cfg, EPI, OUT, A2.data_ptr(), 2 * pad32(K), B2.data_ptr(), 2 * K, M, N, K, bias.data_ptr(),
                                          C2f.data_ptr(), C2.data_ptr(), 2 * pad32(N), D2.data_ptr(), N, 1, stream)
            assert rc == 0, _lib.last_error()

        us = timeit(run2)
        tot2[cfg] += us
        s2 = stats(unsplit2(C2, N) if OUT else C2f.double(), ref)
        line += f" h2[{cfg}] {us:6.1f}us {2*M*N*K/us/1e6:5.0f}TF max {s2[0]:.1e} rms {s2[1]:.1e} mean {s2[2]:+.1e} |"
        if EPI == 2:
            line += f" dD {(D2 - D3).abs().max().item():.1e} |"
    print(line, flush=True)
print("sum over shapes (us): bf3a", round(tot3, 1), "h2", {c: round(v, 1) for c, v in tot2.items()})

if os.environ.get("STAMPS") and hasattr(lib, "aimnet_debug_h2_stamps"):
    buf = (C.c_ulonglong * 1024)()
    lib.aimnet_debug_h2_stamps.argtypes = [C.c_void_p]
    assert lib.aimnet_debug_h2_stamps(buf) == 0
    t = np.array(buf[:], dtype=np.int64).reshape(2, 512)
    t0 = t[0][0]
    for g in range(2):
        n = int((t[g] > 0).sum())
        rel = t[g][:n] - t0
        print(f"group {g}: {n} stamps; first {rel[0]} last {rel[-1]}; deltas:", " ".join(str(int(d)) for d in np.diff(rel)))
