"""Cost of the analytic Hessian-vector product against force evaluations (<= 2 force evaluations per vector),
and accuracy of both operators against the reference goldens.  python tests/tools/hvp_bench.py > gpurun_out/hvp_bench.txt"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aimnetcentral_amd import AIMNet2Calculator, loader  # noqa: E402
from aimnetcentral_amd import workloads  # noqa: E402


def golden(n):
    return np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"))


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    res = {}
    for name, rxn in (("hvp40", False), ("hvp40_rxn", True)):
        g = golden(name)
        calc = AIMNet2Calculator(loader.synthetic_spec(0, rxn=rxn), device="cuda:0")
        data = {"coord": g["coord"], "numbers": g["numbers"], "charge": 0.0}
        Href = g["hessian"].reshape(120, 120)
        row = {}
        for method in ("analytic", "fd"):
            calc.hvp_method = method
            H = calc(data, hessian=True)["hessian"].cpu().numpy().reshape(120, 120)
            hv1 = calc.hessian_vector_product(data, g["v1"]).cpu().numpy()
            hv4 = calc.hessian_vector_product(data, g["v4"]).cpu().numpy()
            row[method] = {
                "dH_max": float(np.abs(H - Href).max()), "dHv1_max": float(np.abs(hv1 - g["hv1"]).max()),
                "dHv4_max": float(np.abs(hv4 - g["hv4"]).max()),
                "outside_allclose_1e-3": int((~np.isclose(H, Href, rtol=1e-3, atol=1e-3)).sum() + (~np.isclose(hv1, g["hv1"], rtol=1e-3, atol=1e-3)).sum()
                                             + (~np.isclose(hv4, g["hv4"], rtol=1e-3, atol=1e-3)).sum()),
                "hessian_ms": timed(lambda: calc(data, hessian=True), 5),
                "hv1_ms": timed(lambda: calc.hessian_vector_product(data, g["v1"])),
                "hv4_ms": timed(lambda: calc.hessian_vector_product(data, g["v4"])),
            }
        row["force_eval_ms"] = timed(lambda: calc(data, forces=True))
        res[name] = row
    # a large system: one direction on the 10 080-atom crystal of config 3 (engine level, DSF 15 A)
    calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
    c, z, cell = workloads.glucose_supercell()
    n = len(z)
    eng = calc.engine
    t = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a)).to(dt).cuda()  # noqa: E731
    args = (t(c), t(z, torch.int32), torch.zeros(n, dtype=torch.int32, device="cuda"), t([0.0]))
    kw = dict(cell=t(cell), coulomb="dsf", dsf_rc=15.0, dsf_alpha=0.2)
    v = torch.randn(1, n, 3, device="cuda:0")
    res["pbc10k"] = {"n_atoms": n, "hvp_1dir_ms": timed(lambda: eng.hvp(*args, v, **kw), 3),
                     "force_eval_ms": timed(lambda: eng.eval(*args, forces=True, **kw), 5)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
