"""PME reciprocal part of a cell against the same cell replicated (the potential and its gradient at the images must agree to the
mesh accuracy).  GPU box."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from aimnetcentral_amd import loader
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
lib, dev = eng.lib, eng.device
def recip(x, q, cell, acc):
    n = len(x)
    xd = torch.from_numpy(x.astype(np.float32)).to(dev); qd = torch.from_numpy(q.astype(np.float32)).to(dev); cd = torch.from_numpy(cell.astype(np.float32)).to(dev)
    e = torch.zeros(n, dtype=torch.float64, device=dev); qb = torch.zeros(n, device=dev); fg = torch.zeros(n, 3, device=dev); va = torch.zeros(n, 9, device=dev)
    info = (C.c_double * 8)()
    torch.cuda.synchronize()
    rc = lib.aimnet_debug_pme_recip(xd.data_ptr(), qd.data_ptr(), None, cd.data_ptr(), float(q.sum()), n, acc, 1 << 24, e.data_ptr(), qb.data_ptr(), fg.data_ptr(), va.data_ptr(), info, None)
    assert rc == 0
    return qb.cpu().numpy() / 2, fg.cpu().numpy(), list(info)
rng = np.random.default_rng(0)
cell = np.array([[9.964, 0, 0], [0, 12.562, 0], [-0.466, 0, 11.814]])
n = 192
x = rng.random((n, 3)) @ cell
q = rng.normal(0, 0.3, n); q -= q.mean()
for acc in (1e-6, 1e-8):
    p1, g1, i1 = recip(x, q, cell, acc)
    for rep in ((2, 2, 2), (4, 3, 3), (7, 6, 6)):
        xs = np.concatenate([x + a * cell[0] + b * cell[1] + c * cell[2] for a in range(rep[0]) for b in range(rep[1]) for c in range(rep[2])])
        qs = np.tile(q, rep[0] * rep[1] * rep[2])
        p2, g2, i2 = recip(xs, qs, cell * np.array(rep)[:, None], acc)
        print(acc, rep, len(xs), "alpha", round(i1[0], 4), round(i2[0], 4), "mesh", [int(v) for v in i2[2:5]], "dphi", np.abs(p2[:n] - p1).max(), "dgrad*q", np.abs((g2[:n] - g1)).max(), "|grad|", np.abs(g1).max())
