#!/usr/bin/env python
"""The layers of one MLP as ONE persistent launch (csrc/gemm_h2_chain.hip) against one gemm_h2 launch per layer (GPU box): time of the
whole chain and BITWISE equality of every output (hidden activations in h2 form, GELU', last layer).

Env: M (rows), CHAINS=fwd1,bwd1,fwd2,bwd2,fwd0,bwd0 (the MLP shapes of the three passes), REPS."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
M = int(os.environ.get("M", 10080))
REPS = int(os.environ.get("REPS", 20))
stream = torch.cuda.current_stream(dev).cuda_stream
pad32 = lambda n: (n + 31) // 32 * 32  # noqa: E731
# (K_in, [(N, epi, out2)]) : forward = GELU hidden layers with split output, last layer bias / GELU fp32; backward = chain rule, last none
CHAINS = {
    "fwd0": (448, [(512, 2, 1), (384, 2, 1), (288, 1, 0)]),
    "fwd1": (736, [(512, 2, 1), (384, 2, 1), (288, 2, 0)]),
    "fwd2": (736, [(512, 2, 1), (384, 2, 1), (384, 2, 1), (256, 2, 1)]),
    "bwd2": (256, [(384, 3, 1), (384, 3, 1), (512, 3, 1), (736, 0, 0)]),
    "bwd1": (288, [(384, 3, 1), (512, 3, 1), (736, 0, 0)]),
    "bwd0": (288, [(384, 3, 1), (512, 3, 1), (448, 0, 0)]),
}


def split2(x, mode):
    m, k = x.shape
    out = torch.empty(m, 2 * pad32(k), dtype=torch.int16, device=dev)
    assert lib.aimnet_debug_split_h2(x.data_ptr(), k, m, k, out.data_ptr(), 2 * pad32(k), mode, stream) == 0, _lib.last_error()
    return out


def timeit(fn, n=REPS):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot_sep, tot_chain = 0.0, 0.0
for name in os.environ.get("CHAINS", "fwd0,fwd1,fwd2,bwd2,bwd1,bwd0").split(","):
    k_in, layers = CHAINS[name]
    torch.manual_seed(len(name) + k_in)
    A0 = split2(torch.randn(M, k_in, device=dev), 1)
    Ws, biases, Ds = [], [], []
    k = k_in
    for (n, epi, out2) in layers:
        Ws.append(split2(torch.randn(n, k, device=dev) * (1.5 / k**0.5), 2))
        biases.append(torch.randn(n, device=dev) * 0.1)
        Ds.append(torch.rand(M, n, device=dev))
        k = n

    def buffers():
        return ([torch.zeros(M, n, device=dev) for (n, _, _) in layers], [torch.zeros(M, 2 * pad32(n), dtype=torch.int16, device=dev) for (n, _, _) in layers],
                [d.clone() for d in Ds])

    Cs, C2s, Dw = buffers()
    Cc, C2c, Dc = buffers()

    def run_sep():
        a, lda, kk = A0, 2 * pad32(k_in), k_in
        for li, (n, epi, out2) in enumerate(layers):
            rc = lib.aimnet_debug_gemm_h2(452, epi, out2, a.data_ptr(), lda, Ws[li].data_ptr(), 2 * kk, M, n, kk, biases[li].data_ptr(), Cs[li].data_ptr(),
                                          C2s[li].data_ptr(), 2 * pad32(n), Dw[li].data_ptr(), n, 1, stream)
            assert rc == 0, _lib.last_error()
            a, lda, kk = C2s[li], 2 * pad32(n), n

    def run_sep_auto():  # what the engine does today: the tile chooser's pick per layer
        a, lda, kk = A0, 2 * pad32(k_in), k_in
        for li, (n, epi, out2) in enumerate(layers):
            rc = lib.aimnet_debug_gemm_h2(0, epi, out2, a.data_ptr(), lda, Ws[li].data_ptr(), 2 * kk, M, n, kk, biases[li].data_ptr(), Cs[li].data_ptr(),
                                          C2s[li].data_ptr(), 2 * pad32(n), Dw[li].data_ptr(), n, 1, stream)
            assert rc == 0, _lib.last_error()
            a, lda, kk = C2s[li], 2 * pad32(n), n

    n_sync = int(lib.aimnet_debug_gemm_h2_chain_sync_words(len(layers), M))
    sync = torch.zeros(n_sync, dtype=torch.int32, device=dev)
    arr = (_lib.DebugChainLayer * len(layers))()
    a_ptr, lda, kk = A0.data_ptr(), 2 * pad32(k_in), k_in
    for li, (n, epi, out2) in enumerate(layers):
        L = arr[li]
        L.A2, L.lda2, L.Bt2, L.ldb, L.N, L.K = a_ptr, lda, Ws[li].data_ptr(), 2 * kk, n, kk
        L.bias, L.C, L.C2, L.ldc2, L.D, L.ldc = biases[li].data_ptr(), Cc[li].data_ptr(), C2c[li].data_ptr(), 2 * pad32(n), Dc[li].data_ptr(), n
        L.epi, L.out2, L.alt = epi, out2, 1
        a_ptr, lda, kk = C2c[li].data_ptr(), 2 * pad32(n), n

    def run_chain():
        rc = lib.aimnet_debug_gemm_h2_chain(arr, len(layers), M, sync.data_ptr(), n_sync, stream)
        assert rc == 0, _lib.last_error()

    run_sep()
    run_chain()
    torch.cuda.synchronize()
    err = int(sync[1].item())
    same = True
    for li, (n, epi, out2) in enumerate(layers):
        ok = torch.equal(C2s[li], C2c[li]) if out2 else torch.equal(Cs[li], Cc[li])
        if epi == 2:
            ok = ok and torch.equal(Dw[li], Dc[li])
        same = same and ok
    us_sep, us_auto, us_chain = timeit(run_sep), timeit(run_sep_auto), timeit(run_chain)
    # repeatability of the chain under its own dynamic schedule
    rep_ok = True
    for _ in range(5):
        run_chain()
        torch.cuda.synchronize()
        for li, (n, epi, out2) in enumerate(layers):
            rep_ok = rep_ok and (torch.equal(C2s[li], C2c[li]) if out2 else torch.equal(Cs[li], Cc[li]))
    tot_sep += us_auto
    tot_chain += us_chain
    print(f"{name}: K_in {k_in} layers {[n for n, _, _ in layers]}: per-layer launches {us_sep:6.1f} us (160x128 tiles) / {us_auto:6.1f} us (chooser) | chain {us_chain:6.1f} us | "
          f"bitwise equal {same} repeatable {rep_ok} err {err}", flush=True)
print(f"sum: per-layer launches (chooser) {tot_sep:.1f} us, chains {tot_chain:.1f} us")
