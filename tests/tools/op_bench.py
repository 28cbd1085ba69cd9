#!/usr/bin/env python
"""Timing of the stand-alone conv_sv_2d_sp ops (the reference's torch.ops.aimnet.* boundary, INTEGRATION.md level 3) at the size
of BASELINE config 3: B = 10 081 rows, M = the widest 5 A neighbour row, A = G = 16."""
import json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from aimnetcentral_amd import engine as E, workloads

dev = torch.device("cuda:0")
c, z, cell = workloads.glucose_supercell((7, 3, 5))
n = len(z)
nb, num, sh, xw, (mx, ovf) = E.neighbor_list(torch.from_numpy(c.astype(np.float32)).to(dev), 5.0, torch.zeros(n, dtype=torch.int32, device=dev),
                                             cell=torch.from_numpy(cell.astype(np.float32)).to(dev), max_nb=112)
M = int(mx)
idx = torch.full((n + 1, M), n, dtype=torch.int32, device=dev)
idx[:n] = nb[:, :M]
a = torch.randn(n + 1, 16, 16, device=dev); a[-1] = 0
g = torch.randn(n + 1, M, 16, 4, device=dev)
go = torch.randn(n + 1, 16, 16, 4, device=dev); go[-1] = 0
pairs = int(num.sum())


def T(f, reps=20):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e6


res = {"B": n + 1, "M": M, "ordered_pairs": pairs,
       "fwd_us": T(lambda: E.conv_sv_2d_sp_fwd(a, idx, g)), "bwd_us": T(lambda: E.conv_sv_2d_sp_bwd(go, a, idx, g)),
       "bwd_bwd_us": T(lambda: E.conv_sv_2d_sp_bwd_bwd(go, a, g, a, idx, g))}
res["fwd_GBps_g_stream"] = (n + 1) * M * 256 / res["fwd_us"] * 1e-3  # the materialised g tensor is the op's dominant stream
print(json.dumps(res))
