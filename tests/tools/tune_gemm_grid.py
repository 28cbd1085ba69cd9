#!/usr/bin/env python
"""Time every GEMM tile configuration on a grid of M for the six MLP layer shapes and both heavy epilogues;
writes gpurun_out/gemm_grid.json (input of the tile-choice model in csrc/gemm.hip)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
Ms = [int(m) for m in os.environ.get("MS", "1024,2048,3200,4096,6400,8192,10080,10518,12800,16384,25600,51200").split(",")]
shapes = [(512, 736), (384, 512), (288, 384), (256, 384), (736, 512), (512, 384), (384, 384), (128, 256), (128, 128), (448, 512)]
cfgs = [152, 142, 132, 122, 153, 143, 223, 213, 222, 233, 351, 331, 381, 371, 361, 341, 321, 412, 411, 410, 409, 5]
cfgs += [1000 + c for c in cfgs if c != 5]
stream = torch.cuda.current_stream(dev).cuda_stream
out = []
for epi in (2, 3):
    for M in Ms:
        for (N, K) in shapes:
            A = torch.randn(M, K, device=dev)
            Bt = torch.randn(N, K, device=dev) * 0.05
            bias = torch.randn(N, device=dev)
            Cm = torch.empty(M, N, device=dev)
            D = torch.rand(M, N, device=dev)
            row = {"epi": epi, "M": M, "N": N, "K": K, "us": {}}
            for cfg in [0] + cfgs:
                def run():
                    rc = lib.aimnet_debug_gemm(cfg, epi, A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), Cm.data_ptr(), D.data_ptr(), N, stream)
                    assert rc == 0, _lib.last_error()
                for _ in range(3): run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): run()
                e1.record(); torch.cuda.synchronize()
                row["us"][str(cfg)] = e0.elapsed_time(e1) / 10 * 1e3
            out.append(row)
    print("epi", epi, "done", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemm_grid.json", "w"))
