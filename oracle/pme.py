"""Smooth particle-mesh Ewald, reciprocal space (numpy, float64).  TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's
cpu_baseline leg and tests/tools; the product path (aimnetcentral_amd/csrc/pme.hip) never touches it.

PARITY UNPINNED against the reference's production kernel: the reference delegates LRCoulomb "pme" to
nvalchemiops.torch.interactions.electrostatics.particle_mesh_ewald (aimnet/modules/lr.py:752-775, parameters from
estimate_pme_parameters, aimnet/calculators/calculator.py:1579-1586); nvalchemiops 0.4.0 is not vendored in the reference tree.
What is restated here is the published method (Essmann, Perera, Berkowitz, Darden, Lee, Pedersen, J. Chem. Phys. 103, 8577
(1995)): cardinal B-spline charge assignment of order p, a discrete Fourier transform of the charge mesh, the Ewald influence
function times the inverse squared moduli of the spline's Fourier coefficients, and analytic differentiation of the splines for
the forces.  It is pinned to the exact structure-factor sum of oracle/aimnet2_oracle.py (ewald_reciprocal, itself pinned to the
reference's in-tree torch Ewald, aimnet/ops.py:196-276): both converge to the same reciprocal-space energy, and
tests/test_oracle_pme.py checks the difference against the requested accuracy.

Conventions (those of csrc/ewald.hip): E_rec / k_e = 1/2 sum_i q_i phi_i, phi = phi_mesh + phi_bg with the neutralising
background phi_bg = -pi Q / (V alpha^2); dE/dq_i = k_e phi_i; dE/dr_i = k_e q_i grad phi_i;
dE/d eps_ab = k_e/2 sum_m theta(m) |Q^(m)|^2 (2 k_a k_b (1/k^2 + 1/(4 alpha^2)) - delta_ab) - delta_ab k_e/2 Q phi_bg ... (row-vector strain).
"""
from __future__ import annotations

import math

import numpy as np

PME_ORDER = 8          # cardinal B-spline order (points per axis an atom touches)
PME_RC_MAX = 10.0      # Angstrom: cap of the real-space cutoff (the Ewald balance would let it grow as N^(1/6))
PME_MIN_MESH = 8       # mesh points per axis at least


def pme_parameters(n_atoms: int, cell: np.ndarray, accuracy: float, rc_max: float = PME_RC_MAX, order: int = PME_ORDER):
    """(alpha, rc, mesh dims) of one system.  The splitting follows the Ewald estimate (oracle ewald_parameters: eta, f) until
    its real-space cutoff reaches rc_max; from there rc stays and alpha = f / (sqrt(2) rc) grows no further, so the real-space
    walk is O(N) and the mesh carries the rest.  Mesh: the reciprocal cutoff kc = sqrt(2) f alpha is resolved with `over`
    points per shortest wavelength, over = oversampling(accuracy, order)."""
    c = np.asarray(cell, dtype=np.float64)
    vol = abs(np.linalg.det(c))
    eta = (vol * vol / max(n_atoms, 1)) ** (1.0 / 6.0) / math.sqrt(2.0 * math.pi)
    f = math.sqrt(-2.0 * math.log(accuracy))
    rc = min(f * eta, rc_max)
    alpha = f / (math.sqrt(2.0) * rc)
    kc = math.sqrt(2.0) * f * alpha
    over = pme_oversampling(accuracy, order)
    mesh = []
    for a in range(3):
        length = np.linalg.norm(c[a])
        nmax = kc * length / (2.0 * math.pi)
        k = int(math.ceil(over * (2.0 * nmax + 1.0)))
        k += k & 1
        mesh.append(max(PME_MIN_MESH, k))
    return alpha, rc, tuple(mesh)


def pme_oversampling(accuracy: float, order: int = PME_ORDER) -> float:
    """Mesh points per reciprocal-cutoff wavelength half: calibrated so that the rms force error of the mesh against the exact
    sum stays below `accuracy` relative to the rms reciprocal force (tests/tools/pme_calibrate.py; order 8: 1.0 at 1e-4, 1.5 at
    1e-6, 2.0 at 1e-8)."""
    return 1.0 + 0.25 * max(0.0, -math.log10(accuracy) - 4.0)


def bspline(u_frac: np.ndarray, p: int):
    """Cardinal B-spline weights M_p(w + p - 1 - j) ... for the p mesh points an atom touches.  u_frac in [0, 1): the fractional
    part of the scaled coordinate; returns (w [n, p], dw [n, p]) with w[:, j] the weight of mesh point floor(u) - (p - 1) + j + ...
    (see pme_reciprocal for the index) and dw the derivative with respect to u."""
    n = u_frac.shape[0]
    w = np.zeros((n, p))
    w[:, 0] = 1.0 - u_frac
    w[:, 1] = u_frac
    for k in range(3, p + 1):          # raise the order to k (Essmann eq. 4.1 recursion), p - 1 last for the derivative
        if k == p:
            dw = np.zeros((n, p))
            dw[:, 0] = -w[:, 0]
            for j in range(1, p):
                dw[:, j] = w[:, j - 1] - w[:, j]
        div = 1.0 / (k - 1)
        w[:, k - 1] = div * u_frac * w[:, k - 2]
        for j in range(1, k - 1):
            w[:, k - 1 - j] = div * ((u_frac + j) * w[:, k - 2 - j] + (k - j - u_frac) * w[:, k - 1 - j])
        w[:, 0] = div * (1.0 - u_frac) * w[:, 0]
    if p == 2:
        dw = np.stack([-np.ones(n), np.ones(n)], axis=1)
    return w, dw


def bspline_moduli(K: int, p: int) -> np.ndarray:
    """|b(m)|^2 of Essmann eq. 4.4 for m = 0 .. K-1: 1 / |sum_j M_p(j + 1) exp(2 pi i m j / K)|^2."""
    w, _ = bspline(np.zeros(1), p)      # M_p at the integers: w[0, j] = M_p(p - 1 - j) for j = 0 .. p-1 (w[0, p-1] = M_p(0) = 0)
    mp = w[0, ::-1][1:]                 # M_p(1) .. M_p(p - 1)
    m = np.arange(K)
    s = np.zeros(K, dtype=np.complex128)
    for j, v in enumerate(mp):
        s += v * np.exp(2j * math.pi * m * j / K)
    d = (s * s.conj()).real
    # zeros of the sum (even order at m = K/2): take the neighbours' mean, as every SPME code does
    for i in np.nonzero(d < 1e-10)[0]:
        d[i] = 0.5 * (d[(i - 1) % K] + d[(i + 1) % K])
    return 1.0 / d


def pme_reciprocal(x: np.ndarray, q: np.ndarray, cell: np.ndarray, alpha: float, mesh, order: int = PME_ORDER, dft: str = "fft"):
    """Reciprocal space + neutralising background on the mesh.  Returns dict(e = E_rec / k_e, phi [n], grad [n, 3] = grad phi_i,
    strain [3, 3] = dE/d eps / k_e).  dft = "direct" forms the three axis transforms as dense matrix products (what pme.hip does)."""
    x = np.asarray(x, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    c = np.asarray(cell, dtype=np.float64)
    n = x.shape[0]
    p = order
    K = np.asarray(mesh, dtype=np.int64)
    inv = np.linalg.inv(c)              # row-vector convention: frac = x @ inv
    vol = abs(np.linalg.det(c))
    frac = x @ inv
    frac -= np.floor(frac)
    u = frac * K                        # scaled fractional coordinates
    fl = np.floor(u).astype(np.int64)
    w, dw, idx = [], [], []
    for a in range(3):
        wa, dwa = bspline(u[:, a] - fl[:, a], p)
        w.append(wa)
        dw.append(dwa)
        # weight j belongs to mesh point floor(u) - (p - 1) + j + ... : with the recursion above w[:, j] = M_p(u - k) for
        # k = floor(u) - (p - 1) + j, j = 0 .. p-1
        idx.append((fl[:, a, None] - (p - 1) + np.arange(p)[None, :]) % K[a])
    Q = np.zeros(tuple(K))
    for i in range(n):
        Q[np.ix_(idx[0][i], idx[1][i], idx[2][i])] += q[i] * w[0][i][:, None, None] * w[1][i][None, :, None] * w[2][i][None, None, :]
    if dft == "direct":
        Qh = Q.astype(np.complex128)
        for a in range(3):
            m = np.arange(K[a])
            F = np.exp(-2j * math.pi * np.outer(m, m) / K[a])
            Qh = np.moveaxis(np.tensordot(F, Qh, axes=([1], [a])), 0, a)
    else:
        Qh = np.fft.fftn(Q)
    mm = [np.where(np.arange(K[a]) <= K[a] // 2, np.arange(K[a]), np.arange(K[a]) - K[a]) for a in range(3)]
    m1, m2, m3 = np.meshgrid(*mm, indexing="ij")
    b = 2.0 * math.pi * inv.T           # rows: reciprocal vectors
    kx = m1[..., None] * b[0] + m2[..., None] * b[1] + m3[..., None] * b[2]
    k2 = (kx * kx).sum(-1)
    k2[0, 0, 0] = 1.0
    mod = bspline_moduli(K[0], p)[:, None, None] * bspline_moduli(K[1], p)[None, :, None] * bspline_moduli(K[2], p)[None, None, :]
    theta = 4.0 * math.pi / vol * np.exp(-k2 / (4.0 * alpha * alpha)) / k2 * mod
    theta[0, 0, 0] = 0.0
    s2 = (Qh * Qh.conj()).real
    e_mesh = 0.5 * (theta * s2).sum()
    vfac = 2.0 * (1.0 / k2 + 1.0 / (4.0 * alpha * alpha))
    ts = theta * s2
    strain = 0.5 * (np.einsum("xyz,xyza,xyzb->ab", ts * vfac, kx, kx) - np.eye(3) * ts.sum())
    if dft == "direct":
        ph = theta * Qh
        for a in range(3):
            m = np.arange(K[a])
            F = np.exp(2j * math.pi * np.outer(m, m) / K[a])
            ph = np.moveaxis(np.tensordot(F, ph, axes=([1], [a])), 0, a)
        pot = ph.real
    else:
        pot = np.fft.ifftn(theta * Qh).real * K.prod()
    phi = np.zeros(n)
    g = np.zeros((n, 3))
    for i in range(n):
        sub = pot[np.ix_(idx[0][i], idx[1][i], idx[2][i])]
        phi[i] = np.einsum("xyz,x,y,z->", sub, w[0][i], w[1][i], w[2][i])
        du = np.array([np.einsum("xyz,x,y,z->", sub, dw[0][i], w[1][i], w[2][i]) * K[0],
                       np.einsum("xyz,x,y,z->", sub, w[0][i], dw[1][i], w[2][i]) * K[1],
                       np.einsum("xyz,x,y,z->", sub, w[0][i], w[1][i], dw[2][i]) * K[2]])
        g[i] = inv @ du                 # d frac_a / d x_c = inv[c, a]
    qtot = q.sum()
    phi_bg = -math.pi * qtot / (vol * alpha * alpha)
    e = e_mesh + 0.5 * qtot * phi_bg
    strain = strain - np.eye(3) * 0.5 * qtot * phi_bg
    return {"e": e, "phi": phi + phi_bg, "grad": g, "strain": strain, "mesh_e": e_mesh}


def exact_reciprocal(x, q, cell, alpha, kc):
    """The exact structure-factor sum in the same conventions (numpy twin of oracle ewald_reciprocal, with phi / grad / strain)."""
    x = np.asarray(x, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    c = np.asarray(cell, dtype=np.float64)
    vol = abs(np.linalg.det(c))
    b = 2.0 * math.pi * np.linalg.inv(c).T
    nmax = [int(math.floor(kc * np.linalg.norm(c[a]) / (2.0 * math.pi))) for a in range(3)]
    g = np.stack(np.meshgrid(*[np.arange(-m, m + 1) for m in nmax], indexing="ij"), axis=-1).reshape(-1, 3)
    g = g[(g != 0).any(1)]
    k = g @ b
    k2 = (k * k).sum(-1)
    keep = k2 <= kc * kc
    k, k2 = k[keep], k2[keep]
    n = x.shape[0]
    phi = np.zeros(n)
    grad = np.zeros((n, 3))
    strain = np.zeros((3, 3))
    e = 0.0
    for c0 in range(0, k.shape[0], 512):
        kk, kq = k[c0:c0 + 512], k2[c0:c0 + 512]
        th = x @ kk.T
        cs, sn = np.cos(th), np.sin(th)
        sre, sim = q @ cs, q @ sn
        A = 4.0 * math.pi / vol * np.exp(-kq / (4.0 * alpha * alpha)) / kq
        t = cs * (A * sre) + sn * (A * sim)
        phi += t.sum(1)
        grad += (cs * (A * sim) - sn * (A * sre)) @ kk
        ts = A * (sre * sre + sim * sim)
        e += 0.5 * ts.sum()
        vf = 2.0 * (1.0 / kq + 1.0 / (4.0 * alpha * alpha))
        strain += 0.5 * (np.einsum("k,ka,kb->ab", ts * vf, kk, kk) - np.eye(3) * ts.sum())
    qtot = q.sum()
    phi_bg = -math.pi * qtot / (vol * alpha * alpha)
    return {"e": e + 0.5 * qtot * phi_bg, "phi": phi + phi_bg, "grad": grad, "strain": strain - np.eye(3) * 0.5 * qtot * phi_bg}
