"""Which central stencil / step gives the smallest H @ v error from fp32 analytic forces?  (hvp40 golden: the reference's
double backward.)  Stencil points are evaluated as one flat batch, as AIMNet2Calculator._fd_hvp does."""
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import golden
from aimnetcentral_amd import AIMNet2Calculator, loader

ST = {4: ([1, 2], [2 / 3, -1 / 12]), 6: ([1, 2, 3], [3 / 4, -3 / 20, 1 / 60]), 8: ([1, 2, 3, 4], [4 / 5, -1 / 5, 4 / 105, -1 / 280])}
for name in ("hvp40", "hvp40_rxn"):
    g = golden(name)
    spec = loader.synthetic_spec(0, rxn=True) if name.endswith("rxn") else loader.synthetic_spec(0)
    calc = AIMNet2Calculator(spec, device="cuda:0")
    eng, dev = calc.engine, calc.device
    coord = torch.from_numpy(g["coord"]).to(dev); z = torch.from_numpy(g["numbers"]).to(dev)
    n = coord.shape[0]
    v = torch.from_numpy(g["v4"]).to(dev).reshape(-1, n, 3)
    K = v.shape[0]
    scale = v.norm(dim=-1).amax(dim=-1); u = v / scale.view(K, 1, 1)
    q = torch.full((1,), float(g["charge"]), device=dev)
    ref = g["hv4"].reshape(K, n, 3)
    for order, (ks, ws) in ST.items():
        for h in (2.5e-3, 5e-3, 7e-3, 1e-2, 1.4e-2, 2e-2, 4e-2):
            offs = torch.tensor([s * k for k in ks for s in (1.0, -1.0)], device=dev) * h
            wts = torch.tensor([s * w for w in ws for s in (1.0, -1.0)], device=dev) / h
            m = len(offs)
            x = (coord.view(1, 1, n, 3) + offs.view(1, m, 1, 1) * u.view(K, 1, n, 3)).reshape(K * m * n, 3)
            res = eng.eval(x, z.repeat(K * m), torch.arange(K * m, device=dev, dtype=torch.int32).repeat_interleave(n),
                           q.repeat(K * m), forces=True, coulomb="simple")
            f = res["forces"].view(K, m, n, 3)
            hv = -(f * wts.view(1, m, 1, 1)).sum(1) * scale.view(K, 1, 1)   # H v = -dF/dx . v
            err = np.abs(hv.cpu().numpy() - ref)
            viol = err - (1e-3 + 1e-3 * np.abs(ref))  # the reference's gate: allclose(rtol=1e-3, atol=1e-3), tests/test_hvp.py:75
            print(f"{name} order {order} h={h:7.4f}: max {err.max():.2e}  rms {np.sqrt((err**2).mean()):.2e}  (|Hv|max {np.abs(ref).max():.1f})"
                  f"  worst err-atol-rtol|ref| {viol.max():+.2e}  over the gate {(viol > 0).sum()}/{viol.size}")
