"""Step time of the three periodic Coulomb methods (dsf / ewald / pme) on glucose supercells.  GPU box.
usage: python tests/tools/pme_bench.py [7,3,5 [14,6,10 ...]]"""
import json
import sys
import time

import numpy as np
import torch

import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from aimnetcentral_amd import loader, workloads  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402


def main():
    reps = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(7, 3, 5), (14, 6, 5)]
    eng = HipEngine(loader.synthetic_spec(0, cold=True), device="cuda:0")
    dev = eng.device
    for rep in reps:
        c, z, cell = workloads.glucose_supercell(rep)
        rng = np.random.default_rng(0)
        c = (c + rng.normal(0, 0.02, c.shape)).astype(np.float32)
        n = len(z)
        args = (torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.zeros(n, dtype=torch.int64, device=dev),
                torch.zeros(1, device=dev))
        rec = {"atoms": n, "rep": rep}
        ref = None
        for method in ("dsf", "ewald", "pme"):
            kw = dict(cell=torch.from_numpy(cell.astype(np.float32)).to(dev), forces=True, stress=True, coulomb=method)
            r = eng.eval(*args, **kw)
            torch.cuda.synchronize()
            steps = 10 if n < 50000 else 4
            t = time.perf_counter()
            for _ in range(steps):
                r = eng.eval(*args, sync=False, **kw)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / steps * 1e3
            rec[method + "_ms"] = round(ms, 3)
            if method != "dsf":
                rec[method + "_status7"] = int(eng.last_status[7])
                cur = {k: r[k].cpu().numpy() for k in ("energy", "forces", "stress")}
                if ref is None:
                    ref = cur
                else:
                    rec["pme_vs_ewald"] = {"dE": float(abs(cur["energy"][0] - ref["energy"][0])),
                                           "dF_max": float(np.abs(cur["forces"] - ref["forces"]).max()),
                                           "F_max": float(np.abs(ref["forces"]).max()),
                                           "dstress_max": float(np.abs(cur["stress"] - ref["stress"]).max())}
        print(json.dumps(rec), flush=True)


main()
