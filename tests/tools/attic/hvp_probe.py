import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conftest import golden
from aimnetcentral_amd import AIMNet2Calculator, loader
g = golden("hvp40")
calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
data = {"coord": g["coord"], "numbers": g["numbers"], "charge": float(g["charge"])}
def rep(tag, got, ref):
    err = np.abs(got - ref); print(f"{tag}: max|err| {err.max():.2e} over gate {(err > 1e-3 + 1e-3*np.abs(ref)).sum()}/{err.size}")
for mode in (1, 0, 2):
    calc.engine.set_option("gemm_bf3", mode)
    for order, h in ((6, 0.010), (4, 0.005)):
        calc.FD_ORDER, calc.FD_STEP = order, h
        v1 = torch.from_numpy(g["v1"]); v4 = torch.from_numpy(g["v4"])
        rep(f"mode {mode} order {order} v1 alone      ", calc.hessian_vector_product(data, v1).cpu().numpy(), g["hv1"])
        rep(f"mode {mode} order {order} v1 x4 stacked ", calc.hessian_vector_product(data, torch.stack([v1] * 4)).cpu().numpy()[0], g["hv1"])
        rep(f"mode {mode} order {order} v1 x8 stacked ", calc.hessian_vector_product(data, torch.stack([v1] * 8)).cpu().numpy()[3], g["hv1"])
        rep(f"mode {mode} order {order} v4[0] alone   ", calc.hessian_vector_product(data, v4[0]).cpu().numpy(), g["hv4"][0])
        rep(f"mode {mode} order {order} v4[0] in batch", calc.hessian_vector_product(data, v4).cpu().numpy()[0], g["hv4"][0])
