#!/usr/bin/env python
"""Does the finite-difference H.v / Hessian meet the REFERENCE's gate (tests/test_hvp.py:75: allclose(rtol=1e-3, atol=1e-3))?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conftest import golden
from aimnetcentral_amd import AIMNet2Calculator, loader

for name in ("hvp40",):
    g = golden(name)
    spec = loader.synthetic_spec(0, rxn=True) if name.endswith("rxn") else loader.synthetic_spec(0)
    calc = AIMNet2Calculator(spec, device="cuda:0")
    data = {"coord": g["coord"], "numbers": g["numbers"], "charge": float(g["charge"])}
    for order, h in ((4, 5e-3), (6, 0.010), (6, 0.012), (6, 0.013), (6, 0.014), (6, 0.015), (6, 0.016), (6, 0.018), (8, 0.014), (8, 0.018)):
        calc.FD_ORDER, calc.FD_STEP = order, h
        mode = f"order {order} h {h}"
        hv4 = calc.hessian_vector_product(data, g["v4"]).cpu().numpy()
        hv1 = calc.hessian_vector_product(data, g["v1"]).cpu().numpy()
        H = calc(data, hessian=True)["hessian"].cpu().numpy().reshape(120, 120)
        for tag, got, ref in (("hv1", hv1, g["hv1"]), ("hv4", hv4, g["hv4"]), ("H", H, g["hessian"].reshape(120, 120))):
            err = np.abs(got - ref)
            viol = err - (1e-3 + 1e-3 * np.abs(ref))
            print(f"{name} gemm_bf3={mode} {tag}: max|err| {err.max():.2e}  |ref|max {np.abs(ref).max():.1f}  worst (err - atol - rtol|ref|) {viol.max():+.2e}  "
                  f"elements over the gate: {(viol > 0).sum()} / {viol.size}")
