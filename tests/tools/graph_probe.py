"""Does a HIP graph of one evaluation beat back-to-back launches?  (taxol: ~70 launches of < 5 us; config 3: 1.6 ms)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aimnetcentral_amd import loader, workloads  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402

eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
t = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a)).to(dt).cuda()  # noqa: E731
g = np.load(os.path.join(ROOT, "tests", "golden", "taxol.npz"))
cases = {"taxol": (t(g["coord"]), t(g["numbers"], torch.int32), torch.zeros(113, dtype=torch.int32, device="cuda"), t([0.0]), {})}
c, z, cell = workloads.glucose_supercell()
cases["pbc10k"] = (t(c), t(z, torch.int32), torch.zeros(len(z), dtype=torch.int32, device="cuda"), t([0.0]),
                   dict(cell=t(cell), coulomb="dsf", stress=True))
for name, (x, zz, mol, q, kw) in cases.items():
    for _ in range(3):
        eng.eval(x, zz, mol, q, forces=True, **kw)
    torch.cuda.synchronize()
    n = 200 if name == "taxol" else 50
    t0 = time.perf_counter()
    for _ in range(n):
        eng.eval(x, zz, mol, q, forces=True, sync=False, **kw)
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / n * 1e3
    try:
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            eng.eval(x, zz, mol, q, forces=True, sync=False, **kw)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(gr):
            out = eng.eval(x, zz, mol, q, forces=True, sync=False, **kw)
        torch.cuda.synchronize()
        gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            gr.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / n * 1e3
        ref = eng.eval(x, zz, mol, q, forces=True, **kw)
        print(f"{name}: back-to-back launches {plain:.4f} ms/step, graph replay {graph:.4f} ms/step, "
              f"max|dF| {float((ref['forces'] - out['forces']).abs().max()):.1e}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{name}: back-to-back launches {plain:.4f} ms/step, graph capture failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
