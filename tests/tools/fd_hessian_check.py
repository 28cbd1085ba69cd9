import sys, numpy as np, torch, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from conftest import golden
from aimnetcentral_amd import AIMNet2Calculator, loader
calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
g = golden("hvp40")
data = {"coord": g["coord"], "numbers": g["numbers"], "charge": float(g["charge"])}
for h in (2e-3, 5e-3, 1e-2):
    calc.FD_STEP = h
    out = calc(data, hessian=True); torch.cuda.synchronize()
    t = time.time(); out = calc(data, hessian=True); torch.cuda.synchronize(); dt = time.time() - t
    H = out["hessian"].cpu().numpy().reshape(120, 120)
    hv = calc.hessian_vector_product(data, torch.from_numpy(g["v4"])).cpu().numpy()
    print(f"h={h}: |dH|max={np.abs(H - g['hessian'].reshape(120,120)).max():.2e}  |dHv|max={np.abs(hv - g['hv4']).max():.2e} (|Hv|max {np.abs(g['hv4']).max():.1f})  hessian wall {dt*1e3:.1f} ms")
