#!/usr/bin/env python
"""Pre-split-activation GEMM (csrc/gemm_bf3a.hip) against the in-kernel-split one (gemm_bf3.hip) on the MLP layer shapes (GPU box).

Env: M, CFGS (tile ids, 0 = automatic), EPI (2 GELU / 3 chain rule / 0 / 1), OUT3 (1: bf3 output), SHAPES=all|one, STAMPS=1 (timing build).
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
OUT3 = int(os.environ.get("OUT3", 1 if EPI in (2, 3) else 0))
stream = torch.cuda.current_stream(dev).cuda_stream


def pad32(n):
    return (n + 31) // 32 * 32


def split(x, neg=1 << 30):
    m, k = x.shape
    out = torch.empty(m, 3 * pad32(k), dtype=torch.int16, device=dev)
    rc = lib.aimnet_debug_split_bf3(x.data_ptr(), k, m, k, out.data_ptr(), 3 * pad32(k), neg, stream)
    assert rc == 0, _lib.last_error()
    return out


def unsplit(c3, n):
    m = c3.shape[0]
    v = c3.view(m, -1)[:, : 3 * pad32(n)].reshape(m, pad32(n) // 32, 3, 32).to(torch.int32) << 16
    return v.view(torch.float32).double().sum(dim=2).reshape(m, pad32(n))[:, :n]


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


tot_old, tot = 0.0, {c: 0.0 for c in cfgs}
for (N, K) in shapes:
    A = torch.randn(M, K, device=dev)
    Bt = torch.randn(N, K, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    Dm = torch.rand(M, N, device=dev)
    z = A.double() @ Bt.double().T
    ref = {0: z, 1: z + bias.double(), 2: torch.nn.functional.gelu(z + bias.double()), 3: z * Dm.double()}[EPI]
    KNEG = (2 * (K // 32) + 1) // 3
    B3 = split(Bt, KNEG)
    B3a = split(Bt, -2)  # every odd k-block negated: the operand form of gemm_bf3a.hip
    A3 = split(A)
    Cold, Dold = torch.empty(M, N, device=dev), Dm.clone()

    def run_old():
        rc = lib.aimnet_debug_gemm_bf3(0, EPI, A.data_ptr(), K, B3.data_ptr(), 3 * K, M, N, K, bias.data_ptr(), Cold.data_ptr(),
                                       Dold.data_ptr(), N, KNEG, stream)
        assert rc == 0, _lib.last_error()

    us_old = timeit(run_old)
    tot_old += us_old
    rms_old = (Cold.double() - ref).pow(2).mean().sqrt().item()
    line = f"N={N:4d} K={K:4d}: in-kernel split {us_old:6.1f}us rms {rms_old:.1e} |"
    for cfg in cfgs:
        Cn = torch.zeros(M, N, device=dev)
        C3 = torch.zeros(M, 3 * pad32(N), dtype=torch.int16, device=dev)
        Dn = Dm.clone()

        def run():
            rc = lib.aimnet_debug_gemm_bf3a(cfg, EPI, OUT3, A3.data_ptr(), 3 * pad32(K), B3a.data_ptr(), 3 * K, M, N, K, bias.data_ptr(),
                                            Cn.data_ptr(), C3.data_ptr(), 3 * pad32(N), Dn.data_ptr(), N, 1, stream)
            assert rc == 0, _lib.last_error()

        us = timeit(run)
        tot[cfg] += us
        got = unsplit(C3, N) if OUT3 else Cn.double()
        err = (got - ref).abs().max().item()
        rms = (got - ref).pow(2).mean().sqrt().item()
        same = (got - Cold.double()).abs().max().item()
        line += f" {cfg}: {us:6.1f}us {2*M*N*K/us/1e6:5.0f}TF err {err:.1e} rms {rms:.1e} |old-new| {same:.1e} |"
        if EPI == 2:
            line += f" dD {(Dn - Dold).abs().max().item():.1e} |"
    print(line, flush=True)
print("sum over shapes (us): in-kernel split", round(tot_old, 1), "pre-split", {c: round(v, 1) for c, v in tot.items()})

if os.environ.get("STAMPS") and hasattr(lib, "aimnet_debug_bf3a_stamps"):
    buf = (C.c_ulonglong * 1024)()
    lib.aimnet_debug_bf3a_stamps.argtypes = [C.c_void_p]
    assert lib.aimnet_debug_bf3a_stamps(buf) == 0
    t = np.array(buf[:], dtype=np.int64).reshape(2, 512)
    t0 = t[0][0]
    for g in range(2):
        n = int((t[g] > 0).sum())
        rel = t[g][:n] - t0
        print(f"group {g}: {n} stamps; first {rel[0]} last {rel[-1]}; deltas:", " ".join(str(int(d)) for d in np.diff(rel)))
