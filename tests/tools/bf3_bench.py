#!/usr/bin/env python
"""bf16x3-split GEMM (csrc/gemm_bf3.hip) validation + tuning harness (GPU box).

For every MLP layer shape: split A / Bt on the device, run aimnet_debug_gemm_bf3 for each tile id, compare with an fp64
product (and with the exact-fp32 MFMA GEMM's own error), time 20 launches.  Env: M, CFGS (comma list), EPI, OUTF, SHAPES=all|one.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
M = int(os.environ.get("M", 10080))
# (N, K) of the forward and backward layers of the synthetic aimnet2 MLPs (padded to 32)
shapes = [(512, 736), (512, 448), (384, 512), (288, 384), (384, 384), (256, 384), (128, 256), (128, 128),
          (736, 512), (448, 512), (512, 384), (384, 288), (384, 256), (256, 128)]
if os.environ.get("SHAPES", "all") == "one":
    shapes = shapes[:1]
cfgs = [int(c) for c in os.environ.get("CFGS", "0,452,224,432,422,223,851,234,861,871,891").split(",")]
EPI = int(os.environ.get("EPI", 2))
FLIP = int(os.environ.get("FLIP", 1))
stream = torch.cuda.current_stream(dev).cuda_stream


def split(x, neg=1 << 30):
    m, k = x.shape
    out = torch.empty(m, 3 * k, dtype=torch.int16, device=dev)
    rc = lib.aimnet_debug_split_bf3(x.data_ptr(), k, m, k, out.data_ptr(), 3 * k, neg, stream)
    assert rc == 0, _lib.last_error()
    return out


def unsplit(c3, n):
    """bf3 [M][n/32][3][32] int16 -> fp64 [M][n]"""
    m = c3.shape[0]
    v = c3.view(m, n // 32, 3, 32).to(torch.int32) << 16
    f = v.view(torch.float32).double()
    return f.sum(dim=2).reshape(m, n)


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


tot = {c: 0.0 for c in cfgs}
tot32 = 0.0
for (N, K) in shapes:
    A = torch.randn(M, K, device=dev)
    Bt = torch.randn(N, K, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    z = A.double() @ Bt.double().T
    if EPI == 2:
        ref = torch.nn.functional.gelu(z + bias.double())
    elif EPI == 1:
        ref = z + bias.double()
    else:
        ref = z
    assert (unsplit(split(Bt), K) - Bt.double()).abs().max().item() == 0.0, "split is not exact"
    KNEG = (2 * (K // 32) + 1) // 3 if FLIP else -1  # sign-flipped last third of the accumulation
    B3 = split(Bt, KNEG if FLIP else 1 << 30)
    # exact-fp32 kernel: time and error
    C32 = torch.empty(M, N, device=dev)
    D32 = torch.empty(M, N, device=dev)

    def run32():
        rc = lib.aimnet_debug_gemm(0, EPI if EPI != 3 else 0, A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), C32.data_ptr(),
                                   D32.data_ptr(), N, stream)
        assert rc == 0, _lib.last_error()

    us32 = timeit(run32)
    tot32 += us32
    err32 = (C32.double() - ref).abs().max().item()
    rms32 = (C32.double() - ref).pow(2).mean().sqrt().item()
    line = f"N={N:4d} K={K:4d}: f32 {us32:6.1f}us {2*M*N*K/us32/1e6:5.0f}TF err {err32:.1e} rms {rms32:.1e} |"
    for cfg in cfgs:
        Cm = torch.empty(M, N, device=dev)
        D = torch.empty(M, N, device=dev)

        def run():
            rc = lib.aimnet_debug_gemm_bf3(cfg, EPI, A.data_ptr(), K, B3.data_ptr(), 3 * K, M, N, K, bias.data_ptr(), Cm.data_ptr(),
                                           D.data_ptr(), N, KNEG, stream)
            assert rc == 0, _lib.last_error()

        us = timeit(run)
        tot[cfg] += us
        got = Cm.double()
        err = (got - ref).abs().max().item()
        rms = (got - ref).pow(2).mean().sqrt().item()
        line += f" {cfg}: {us:6.1f}us {2*M*N*K/us/1e6:5.0f}TF err {err:.1e} rms {rms:.1e} |"
    print(line, flush=True)
print("sum over shapes (us): f32", round(tot32, 1), {c: round(v, 1) for c, v in tot.items()})
