#!/usr/bin/env python
"""One MLP sweep in isolation (aimnet_engine_debug_mlp_sweep): the single launch of csrc/gemm_chain.hip against the per-layer launches
of csrc/gemm_h2.hip on the same random input - every output compared bitwise, both timed with HIP events.

Env: M (rows, default 10080), PASSES ("0,1,2"), REPS."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from aimnetcentral_amd import _lib, loader  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402

M = int(os.environ.get("M", 10080))
REPS = int(os.environ.get("REPS", 20))
passes = [int(p) for p in os.environ.get("PASSES", "0,1,2").split(",")]
spec = loader.synthetic_spec(0)
eng = HipEngine(spec, "cuda:0")
lib, dev = eng.lib, eng.device
stream = torch.cuda.current_stream(dev).cuda_stream
vp = C.c_void_p


def pad32(n):
    return (n + 31) // 32 * 32


def split2(x, mode=1):
    m, k = x.shape
    out = torch.zeros(m, 2 * pad32(k), dtype=torch.int16, device=dev)
    assert lib.aimnet_debug_split_h2(x.data_ptr(), k, m, k, out.data_ptr(), 2 * pad32(k), mode, stream) == 0, _lib.last_error()
    return out


def ptrs(ts):
    return (vp * len(ts))(*[t.data_ptr() if t is not None else None for t in ts])


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


dims = spec["arch"]["layer_dims"] if isinstance(spec, dict) and "arch" in spec else None
arch = eng.arch if hasattr(eng, "arch") else None
layer_dims = [[704, 512, 380, 258], [733, 512, 380, 258], [733, 512, 380, 380, 256]]
numbers = torch.tensor([1, 6, 7, 8], dtype=torch.int32, device=dev)[torch.randint(0, 4, (M,), device=dev)]
tot = {0: 0.0, 1: 0.0}
for p in passes:
    d = layer_dims[p]
    nl = len(d) - 1
    kp = [pad32(v) for v in d]
    torch.manual_seed(p)
    # ---- forward
    x = torch.randn(M, d[0], device=dev)
    x2 = split2(x)
    res = {}
    for chain in (0, 1):
        H = [torch.zeros(M, kp[l + 1], device=dev) for l in range(nl)]
        D = [torch.zeros(M, kp[l + 1], device=dev) for l in range(nl)]

        def run():
            rc = lib.aimnet_engine_debug_mlp_sweep(eng._h, p, 0, chain, 0, x2.data_ptr(), M, numbers.data_ptr(), ptrs(H), ptrs(D), None, None, stream)
            assert rc == 0, _lib.last_error()

        us = timeit(run)
        tot[chain] += us
        res[chain] = (H, D, us)
    line = f"pass {p} fwd M={M}: per-layer {res[0][2]:7.1f} us, chain {res[1][2]:7.1f} us |"
    for l in range(nl):
        a, b = res[0][1][l], res[1][1][l]
        if l < nl - 1 or p == 2:
            ne = (a != b)
            line += f" D{l}: {'bitwise' if not ne.any() else f'{int(ne.sum())} differ max {(a - b).abs().max().item():.2e}'};"
    if os.environ.get("AIMNET_CHAIN_DUMP"):
        for l in range(nl - 1):
            a, b = res[0][0][l].view(torch.int16), res[1][0][l].view(torch.int16)
            a, b = a.view(M, -1)[:, : 2 * kp[l + 1]].reshape(M, -1, 2, 32), b.view(M, -1)[:, : 2 * kp[l + 1]].reshape(M, -1, 2, 32)
            for pl, nm in ((0, "hi"), (1, "lo")):
                ne = a[:, :, pl] != b[:, :, pl]
                line += f" H{l}.{nm}: {'bitwise' if not ne.any() else f'{int(ne.sum())} differ (rows {sorted(set(ne.nonzero()[:, 0].tolist()))[:6]}, kb {sorted(set(ne.nonzero()[:, 1].tolist()))[:6]}, cols {sorted(set(ne.nonzero()[:, 2].tolist()))[:8]})'};"
    a, b = res[0][0][nl - 1], res[1][0][nl - 1]
    ne = (a != b)
    line += f" out: {'bitwise' if not ne.any() else f'{int(ne.sum())} differ max {(a - b).abs().max().item():.2e}'}"
    print(line, flush=True)
    if os.environ.get("STAMPS") and hasattr(lib, "aimnet_debug_chain_stamps"):
        import numpy as np
        buf = (C.c_ulonglong * 256)()
        lib.aimnet_debug_chain_stamps.argtypes = [C.c_void_p]
        assert lib.aimnet_debug_chain_stamps(buf) == 0
        t = np.array(buf[:], dtype=np.int64).reshape(4, 64)
        for g, nm in enumerate(("block 0 wave 0", "block 0 wave 4", "last block wave 0", "last block wave 4")):
            n = int((t[g] > 0).sum())
            if n:
                rel = t[g][:n] - t[0][0]
                print(f"   stamps {nm}: first {rel[0]} last {rel[-1]} | deltas:", " ".join(str(int(v)) for v in np.diff(rel)))
    # ---- backward (GELU' of the forward as the chain-rule factors)
    Dv = res[0][1]
    zbar = torch.randn(M, d[nl], device=dev)
    mw = max(kp)
    for flag in ((1, 0) if p == 0 else (0,)):
        resb = {}
        for chain in (0, 1):
            zb = [torch.zeros(M, 2 * mw, dtype=torch.int16, device=dev) for _ in range(2)]
            which = C.c_int(-1)

            def run():
                zb[0].view(-1)[: M * 2 * kp[nl]] = split2(zbar).view(-1)  # (dense rows: row stride 2 * k_out of the last layer)
                rc = lib.aimnet_engine_debug_mlp_sweep(eng._h, p, 1, chain, flag, zb[0].data_ptr(), M, numbers.data_ptr(), None, ptrs(Dv), ptrs(zb),
                                                       C.byref(which), stream)
                assert rc == 0, _lib.last_error()

            def prep():
                zb[0].view(-1)[: M * 2 * kp[nl]] = split2(zbar).view(-1)

            us = timeit(run)
            run()
            torch.cuda.synchronize()
            xbar = zb[which.value].view(-1).view(torch.float32)[: M * kp[0]].view(M, kp[0]).clone()  # (row stride k_in of the first layer)
            us -= timeit(prep)  # (the refill of the input buffer is part of `run`)
            tot[chain] += us
            resb[chain] = (xbar, us)
        a, b = resb[0][0], resb[1][0]
        if flag:
            a, b = a[:, 256:], b[:, 256:]
        ne = (a != b)
        print(f"pass {p} bwd{' (conv columns)' if flag else ''}: per-layer {resb[0][1]:7.1f} us, chain {resb[1][1]:7.1f} us | xbar: "
              f"{'bitwise' if not ne.any() else f'{int(ne.sum())} of {ne.numel()} differ max {(a - b).abs().max().item():.2e}'}", flush=True)
print(f"sum (with both pass-0 backward variants): per-layer {tot[0]:.1f} us, chain {tot[1]:.1f} us")
