"""Calibration of the PME mesh rule (oracle/pme.py pme_oversampling): mesh against the exact structure-factor sum at the same alpha,
for spline orders and oversampling factors.  CPU only.  usage: python tests/tools/pme_calibrate.py [n_cells]"""
import math
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import pme  # noqa: E402


def system(rep, seed=0):
    rng = np.random.default_rng(seed)
    base = np.array([[4.982, 0.0, 0.0], [0.0, 12.562, 0.0], [-0.233, 0.0, 11.814]])
    cell = base * np.array(rep)[:, None]
    n = 96 * rep[0] * rep[1] * rep[2]
    x = rng.random((n, 3)) @ cell
    q = rng.normal(0.0, 0.3, n)
    q -= q.mean()
    return x, q, cell


def main():
    rep = (2, 1, 1) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split(","))
    x, q, cell = system(rep)
    n = x.shape[0]
    for acc in (1e-4, 1e-6, 1e-8):
        f = math.sqrt(-2.0 * math.log(acc))
        alpha, rc, _ = pme.pme_parameters(n, cell, acc)
        kc = math.sqrt(2.0) * f * alpha
        ex = pme.exact_reciprocal(x, q, cell, alpha, kc * 1.25)
        ex0 = pme.exact_reciprocal(x, q, cell, alpha, kc)
        fr = np.sqrt(((q[:, None] * ex["grad"]) ** 2).sum(1).mean())
        print(f"n={n} acc={acc:g} alpha={alpha:.4f} rc={rc:.2f} kc={kc:.3f} E_rec={ex['e']:.6f} rmsF={fr:.4f}  exact(kc) vs exact(1.25 kc): "
              f"dE={abs(ex0['e'] - ex['e']):.2e} dF={np.sqrt(((q[:, None] * (ex0['grad'] - ex['grad'])) ** 2).sum(1).mean()) / fr:.2e}")
        for p in (4, 6, 8):
            for over in (1.0, 1.5, 2.0, 2.5, 3.0):
                mesh = []
                for a in range(3):
                    nmax = kc * np.linalg.norm(cell[a]) / (2 * math.pi)
                    k = int(math.ceil(over * (2 * nmax + 1)))
                    mesh.append(max(8, k + (k & 1)))
                t = time.time()
                r = pme.pme_reciprocal(x, q, cell, alpha, mesh, p)
                dt = time.time() - t
                dF = np.sqrt(((q[:, None] * (r["grad"] - ex["grad"])) ** 2).sum(1).mean()) / fr
                dphi = np.abs(r["phi"] - ex["phi"]).max()
                print(f"   p={p} over={over:.1f} mesh={mesh} dE/E={abs(r['e'] - ex['e']) / abs(ex['e']):.2e} dF_rms/F_rms={dF:.2e} dphi_max={dphi:.2e} "
                      f"dstrain={np.abs(r['strain'] - ex['strain']).max():.2e} ({dt:.1f}s)")


main()
