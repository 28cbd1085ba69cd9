#!/usr/bin/env python
"""The one-launch MLP sweeps (csrc/gemm_chain.hip, option "gemm_chain") against the per-layer launches: outputs compared BITWISE
(same products, same accumulation order), step time of both on config 3 (and other sizes with CELLS="2,3,4;7,3,5").

  python tests/tools/chain_check.py            # bitwise + timing
Env: CELLS (supercells, ';'-separated), STEPS, COLD=1 (cold weights), NSE=1 (two charge channels)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from aimnetcentral_amd import loader, workloads  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402

steps = int(os.environ.get("STEPS", 40))
VIEWS = ["d0_0", "d0_1", "h0_2", "q0", "d1_0", "d1_1", "h1_2", "d2_0", "d2_1", "d2_2", "d2_3", "e_atom", "zb0", "zb1", "abar", "qbar", "fgrad"]
spec = loader.synthetic_spec(0)
eng = HipEngine(spec, "cuda:0")
dev = eng.device
cells = [tuple(int(v) for v in c.split(",")) for c in os.environ.get("CELLS", "7,3,5;2,3,4").split(";")]


def evaluate(c, z, cell):
    return eng.eval(c, z, torch.zeros(len(z), dtype=torch.int64, device=dev), torch.zeros(1, device=dev), cell=cell, forces=True,
                    stress=True, coulomb="dsf", dsf_rc=15.0)


for rep in cells:
    c, z, cell = workloads.glucose_supercell(rep)
    rng = np.random.default_rng(1)
    c = torch.from_numpy((c + rng.normal(0, 0.02, c.shape)).astype(np.float32)).to(dev)
    z = torch.from_numpy(z).to(dev)
    cell = torch.from_numpy(cell.astype(np.float32)).to(dev)
    out = {}
    for mode in (0, 1):
        eng.set_option("gemm_chain", mode)
        r = evaluate(c, z, cell)
        torch.cuda.synchronize()
        out[mode] = {k: v.clone() for k, v in r.items()}
        for name in VIEWS:
            try:
                out[mode][name] = eng.debug_view(name).clone()
            except KeyError:
                pass
        for _ in range(10):
            evaluate(c, z, cell)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            evaluate(c, z, cell)
        torch.cuda.synchronize()
        out[mode]["ms"] = (time.perf_counter() - t0) / steps * 1e3
    line = f"cells {rep} atoms {len(z)}: per-layer {out[0]['ms']:.4f} ms, chain {out[1]['ms']:.4f} ms |"
    for k in ("energy", "forces", "charges", "stress"):
        a, b = out[0][k], out[1][k]
        same = torch.equal(a, b)
        d = (a.double() - b.double()).abs().max().item()
        fin = bool(torch.isfinite(b).all())
        line += f" {k}: {'BITWISE' if same else f'max diff {d:.3e}'}{'' if fin else ' NONFINITE'};"
    print(line, flush=True)
    for name in VIEWS:
        if name in out[0] and name in out[1]:
            a, b = out[0][name], out[1][name]
            if a.dtype != torch.float32:
                a, b = a.view(torch.int16), b.view(torch.int16)
            ne = (a != b)
            print(f"   view {name}: {'BITWISE' if not ne.any() else f'{int(ne.sum())} of {ne.numel()} differ, max {(a.double() - b.double()).abs().max().item():.3e}, columns {sorted(set((ne.nonzero()[:, -1] if ne.ndim > 1 else ne.nonzero()[:, 0]).tolist()))[:12]}'}")
