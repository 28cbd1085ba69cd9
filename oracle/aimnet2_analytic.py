"""ORACLE - test infrastructure only.  Hand-derived analytic forward + backward of the AIMNet2
energy/force/virial path, written WITHOUT autograd, in the exact algebraic form the HIP kernels
use (centre-major gather form of every scatter).

It is the specification of aimnetcentral_amd/csrc/*.hip: each block below names the kernel that
implements it.  It is validated against oracle/aimnet2_oracle.py (the autograd restatement of the
reference, itself pinned to reference goldens) in tests/test_oracle_analytic.py, normally in fp64
where both must agree to ~1e-9.

Reference lines this derivation differentiates: models/aimnet2.py:141-187, modules/aev.py:94-110
and :156-189, ops.py:37-145, modules/lr.py:21-62, :311-331, :559-615, calculators/derivatives.py:96-146.

Layout (product layout, no padding row): real atoms 0..N-1, neighbour matrix nb (N, M) int with
`cnt[i]` valid leading entries per row, integer shifts sh (N, M, 3), per-atom cell index.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import Tensor

from .aimnet2_oracle import COULOMB_FACTOR, OracleModel

SQRT_2_OVER_PI_HALF = 1.0 / math.sqrt(2.0 * math.pi)


def gelu(z: Tensor) -> Tensor:
    return 0.5 * z * (1.0 + torch.erf(z / math.sqrt(2.0)))


def gelu_grad(z: Tensor) -> Tensor:
    """exact GELU'(z) = Phi(z) + z phi(z)."""
    return 0.5 * (1.0 + torch.erf(z / math.sqrt(2.0))) + z * torch.exp(-0.5 * z * z) * SQRT_2_OVER_PI_HALF


def _strip(nbmat: np.ndarray, shifts, n: int):
    """(N+1, M) sentinel layout of the reference -> (N, M) + counts."""
    nb = torch.as_tensor(np.asarray(nbmat)[:n]).long()
    valid = nb < n
    sh = None if shifts is None else torch.as_tensor(np.asarray(shifts)[:n])
    return nb.clamp(max=n - 1), valid, sh


def pair_geometry(x: Tensor, nb: Tensor, valid: Tensor, sh, cell_at):
    """kernel: pair_geometry  ->  r, d, u   (r = x_j + s.C - x_i, ops.py:52-65)."""
    r = x[nb] - x.unsqueeze(1)
    if sh is not None:
        r = r + torch.einsum("nmd,ndh->nmh", sh.to(x.dtype), cell_at)
    r = torch.where(valid.unsqueeze(-1), r, torch.ones_like(r))
    d = r.norm(dim=-1)
    return r, d, r / d.unsqueeze(-1)


def radial_basis(model: OracleModel, d: Tensor, valid: Tensor):
    """gs_g(d) = exp(-eta (d-s_g)^2) * fc(d) and its d-derivative (aev.py:98-103, ops.py:82-96)."""
    rc = model.rc
    dc = d.clamp(min=1e-6, max=float(rc))
    fc = 0.5 * (torch.cos(dc * (math.pi / rc)) + 1.0)
    inside = (d > 1e-6) & (d < rc)
    dfc = torch.where(inside, -0.5 * (math.pi / rc) * torch.sin(dc * (math.pi / rc)), torch.zeros_like(d))
    G = torch.exp(-model.eta * (d.unsqueeze(-1) - model.shifts) ** 2)
    dG = -2.0 * model.eta * (d.unsqueeze(-1) - model.shifts) * G
    gs = G * fc.unsqueeze(-1)
    dgs = dG * fc.unsqueeze(-1) + G * dfc.unsqueeze(-1)
    v = valid.unsqueeze(-1).to(d.dtype)
    return gs * v, dgs * v


def evaluate(
    model: OracleModel,
    coord_wrapped,
    numbers,
    charge,
    mol_idx,
    nbmat,
    shifts=None,
    cell=None,
    coulomb: str = "simple",
    nbmat_lr=None,
    shifts_lr=None,
    dsf_rc: float = 15.0,
    dsf_alpha: float = 0.2,
    stress: bool = False,
    mult=None,
) -> dict[str, np.ndarray]:
    dt = model.dtype
    x = torch.as_tensor(np.asarray(coord_wrapped)).to(dt)
    n = x.shape[0]
    Z = torch.as_tensor(np.asarray(numbers)).long()
    mol = torch.as_tensor(np.asarray(mol_idx)).long()
    Q = torch.as_tensor(np.atleast_1d(np.asarray(charge))).to(dt)
    n_mol = Q.shape[0]
    nq = model.nq  # charge channels: 1, or 2 for the open-shell NSE family (aimnet2.py:94-106); everything below carries them
    if nq == 2:
        mt = torch.ones_like(Q) if mult is None else torch.as_tensor(np.atleast_1d(np.asarray(mult))).to(dt)
        Q = torch.stack([0.5 * Q + 0.5 * (mt - 1.0), 0.5 * Q - 0.5 * (mt - 1.0)], dim=-1)
    else:
        Q = Q.unsqueeze(-1)
    A, G = model.A, model.G
    H = model.agh_a.shape[2]
    cell_at = None
    if cell is not None:
        c = torch.as_tensor(np.asarray(cell)).to(dt)
        if c.ndim == 2:
            c = c.unsqueeze(0)
        cell_at = c[mol] if c.shape[0] > 1 else c.expand(n, 3, 3)
    nb, valid, sh = _strip(nbmat, shifts, n)

    def msum(v: Tensor) -> Tensor:  # kernel: mol_reduce (deterministic per-molecule block sum)
        out = torch.zeros((n_mol,) + v.shape[1:], dtype=v.dtype)
        return out.index_add_(0, mol, v)

    # ------------------------------------------------------------------ forward
    r, d, u = pair_geometry(x, nb, valid, sh, cell_at)
    gs, dgs = radial_basis(model, d, valid)  # (N, M, G)
    one_u = torch.cat([torch.ones_like(d).unsqueeze(-1), u], dim=-1)  # (N, M, 4)

    a = model.afv[Z].view(n, A, G)
    q = torch.zeros(n, nq, dtype=dt)
    npass = len(model.mlps)
    saved = []
    for p in range(npass):
        # kernel: conv_fwd  S[i,a,g,c] = sum_m a_j[a,g] gs[i,m,g] (1,u)[c]   (aev.py:177-178)
        S = torch.einsum("nmag,nmg,nmc->nagc", a[nb], gs, one_u)
        V = torch.einsum("agh,nagk->nahk", model.agh_a, S[..., 1:])
        xin = [a.reshape(n, -1), S[..., 0].reshape(n, -1), V.pow(2).sum(-1).reshape(n, -1)]
        Vq = None
        if p > 0:
            Sq = torch.einsum("nmq,nmg,nmc->nqgc", q[nb], gs, one_u)
            Vq = torch.einsum("qgh,nqgk->nqhk", model.agh_q, Sq[..., 1:])
            xin += [q, Sq[..., 0].reshape(n, -1), Vq.pow(2).sum(-1).reshape(n, -1)]
        h = torch.cat(xin, dim=-1)
        zs = []
        layers = model.mlps[p]
        last_linear = p == 0  # aimnet2.py:58-85: only the first MLP ends linear
        for li, (w, b) in enumerate(layers):  # kernel: gemm_bias_gelu
            z = h @ w.T + b
            zs.append(z)
            h = z if (last_linear and li == len(layers) - 1) else gelu(z)
        rec = {"a": a, "q": q, "V": V, "Vq": Vq, "zs": zs, "y": h}
        if p < npass - 1:
            qt, ft, da = h[:, :nq], h[:, nq : 2 * nq], h[:, 2 * nq :]
            qr = q + qt if p > 0 else qt
            f = ft * ft
            F = msum(f) + 1.0e-6  # kernel: nse (ops.py:99-145)
            D = Q - msum(qr)
            q = qr + f / F[mol] * D[mol]
            a = a + da.view(n, A, G)
            rec.update(ft=ft, f=f, F=F, D=D)
        saved.append(rec)
    aim = saved[-1]["y"]
    hz = []
    h = aim
    for li, (w, b) in enumerate(model.head):
        z = h @ w.T + b
        hz.append(z)
        h = z if li == len(model.head) - 1 else gelu(z)
    e_atom = h.squeeze(-1)
    energy = msum(e_atom.double() + model.sae[Z])

    # ------------------------------------------------------------------ Coulomb (fwd + bwd)
    q_ch = q          # per-channel charges of the last NSE step
    q = q_ch.sum(-1)  # the Coulomb terms see alpha + beta; their dE/dq seeds every channel alike
    qbar = torch.zeros(n, dtype=dt)
    xbar = torch.zeros(n, 3, dtype=dt)
    virial = torch.zeros(n_mol, 3, 3, dtype=torch.float64)

    def pair_term(w: Tensor, dw: Tensor, nb_, valid_, r_, d_, u_, sign: float):
        """kernel: coulomb_pairs.  E += sign k sum_im w q_i q_j over a full symmetric list."""
        nonlocal energy, qbar, xbar, virial
        vm = valid_.to(dt)
        qq = q.unsqueeze(1) * q[nb_] * vm
        energy = energy + sign * COULOMB_FACTOR * msum((w * qq).sum(-1, dtype=torch.float64))
        qbar = qbar + sign * 2.0 * COULOMB_FACTOR * (w * q[nb_] * vm).sum(-1)
        rbar = (sign * COULOMB_FACTOR * dw * qq).unsqueeze(-1) * u_  # dE/dr_im of the ordered pair
        xbar = xbar - 2.0 * rbar.sum(1)
        if stress:
            virial = virial + msum(torch.einsum("nma,nmb->nab", r_, rbar).double())

    w_sr, dw_sr, _ = sr_weight(model, d, getattr(model, "sr_envelope", "exp"))
    pair_term(w_sr, dw_sr, nb, valid, r, d, u, -1.0)  # embedded SRCoulomb (lr.py:1020), exp or cosine envelope
    if coulomb != "none":
        nbl, validl, shl = _strip(nbmat_lr, shifts_lr, n)
        rl, dl, ul = pair_geometry(x, nbl, validl, shl, cell_at)
        if coulomb == "simple":
            pair_term(1.0 / dl, -1.0 / (dl * dl), nbl, validl, rl, dl, ul, 1.0)
        else:
            al, Rc = dsf_alpha, dsf_rc
            erfc_rc = math.erfc(al * Rc)
            sv = erfc_rc / Rc
            slope = erfc_rc / Rc**2 + 2.0 * al / math.sqrt(math.pi) * math.exp(-(al**2) * Rc**2) / Rc
            inside = validl & (dl < Rc)
            w = torch.erfc(al * dl) / dl - sv + (dl - Rc) * slope
            dw = -torch.erfc(al * dl) / (dl * dl) - 2.0 * al / math.sqrt(math.pi) * torch.exp(-(al * dl) ** 2) / dl + slope
            pair_term(w, dw, nbl, inside, rl, dl, ul, 1.0)
            cs = -(sv / 2.0 + al / math.sqrt(math.pi))
            energy = energy + 2.0 * COULOMB_FACTOR * msum((cs * q * q).double())
            qbar = qbar + 4.0 * COULOMB_FACTOR * cs * q

    qbar = qbar.unsqueeze(-1).expand(n, nq).clone()
    # ------------------------------------------------------------------ backward: head
    g = model.head[-1][0].expand(n, -1).clone()  # d e / d h2
    for li in range(len(model.head) - 2, -1, -1):  # kernel: gemm_dx_gelugrad
        g = (g * gelu_grad(hz[li])) @ model.head[li][0]
    ybar = g  # d E / d aim
    abar = torch.zeros(n, A, G, dtype=dt)

    for p in range(npass - 1, -1, -1):
        rec = saved[p]
        layers = model.mlps[p]
        last_linear = p == 0
        gcur = ybar
        for li in range(len(layers) - 1, -1, -1):
            if not (last_linear and li == len(layers) - 1):
                gcur = gcur * gelu_grad(rec["zs"][li])
            gcur = gcur @ layers[li][0]
        xb = gcur  # (N, 704|733)
        # kernel: unconcat -> abar, Sbar, (qbar, Sqbar)
        abar = abar + xb[:, : A * G].view(n, A, G)
        Sbar = torch.empty(n, A, G, 4, dtype=dt)
        Sbar[..., 0] = xb[:, A * G : 2 * A * G].view(n, A, G)
        vbar = xb[:, 2 * A * G : 2 * A * G + A * H].view(n, A, H)
        Vbar = 2.0 * rec["V"] * vbar.unsqueeze(-1)
        Sbar[..., 1:] = torch.einsum("agh,nahk->nagk", model.agh_a, Vbar)
        a_p, q_p = rec["a"], rec["q"]
        # kernel: conv_bwd (centre-major; u_ji = -u_ij; same nb/sh as forward)
        Sb_j = Sbar[nb]  # (N, M, A, G, 4)
        Pp = Sb_j[..., 0] - torch.einsum("nmk,nmagk->nmag", u, Sb_j[..., 1:])
        if p > 0:
            abar = abar + torch.einsum("nmg,nmag->nag", gs, Pp)
        P = Sbar[..., 0].unsqueeze(1) + torch.einsum("nmk,nagk->nmag", u, Sbar[..., 1:])
        a_j = a_p[nb]
        dbar_ij = torch.einsum("nmag,nmag,nmg->nm", a_j, P, dgs)
        ubar_ij = torch.einsum("nmg,nmag,nagk->nmk", gs, a_j, Sbar[..., 1:])
        dbar_ji = torch.einsum("nag,nmag,nmg->nm", a_p, Pp, dgs)
        ubar_ji = torch.einsum("nmg,nag,nmagk->nmk", gs, a_p, Sb_j[..., 1:])
        if p > 0:
            c0 = 2 * A * G + A * H
            qbar = qbar + xb[:, c0 : c0 + nq]
            Sqbar = torch.empty(n, nq, G, 4, dtype=dt)
            Sqbar[..., 0] = xb[:, c0 + nq : c0 + nq + nq * G].view(n, nq, G)
            vqbar = xb[:, c0 + nq + nq * G : c0 + nq + nq * G + nq * H].view(n, nq, H)
            Sqbar[..., 1:] = torch.einsum("qgh,nqhk->nqgk", model.agh_q, 2.0 * rec["Vq"] * vqbar.unsqueeze(-1))
            Sq_j = Sqbar[nb]  # (N, M, nq, G, 4)
            Pqp = Sq_j[..., 0] - torch.einsum("nmk,nmqgk->nmqg", u, Sq_j[..., 1:])
            qbar = qbar + torch.einsum("nmg,nmqg->nq", gs, Pqp)
            Pq = Sqbar[..., 0].unsqueeze(1) + torch.einsum("nmk,nqgk->nmqg", u, Sqbar[..., 1:])
            qj = q_p[nb]  # (N, M, nq)
            dbar_ij = dbar_ij + torch.einsum("nmq,nmqg,nmg->nm", qj, Pq, dgs)
            ubar_ij = ubar_ij + torch.einsum("nmq,nmg,nqgk->nmk", qj, gs, Sqbar[..., 1:])
            dbar_ji = dbar_ji + torch.einsum("nq,nmqg,nmg->nm", q_p, Pqp, dgs)
            ubar_ji = ubar_ji + torch.einsum("nq,nmg,nmqgk->nmk", q_p, gs, Sq_j[..., 1:])
        vm = valid.to(dt).unsqueeze(-1)
        dinv = (1.0 / d).unsqueeze(-1)
        rbar_ij = (dbar_ij.unsqueeze(-1) * u + (ubar_ij - (ubar_ij * u).sum(-1, keepdim=True) * u) * dinv) * vm
        uji = -u
        rbar_ji = (dbar_ji.unsqueeze(-1) * uji + (ubar_ji - (ubar_ji * uji).sum(-1, keepdim=True) * uji) * dinv) * vm
        xbar = xbar + (rbar_ji - rbar_ij).sum(1)
        if stress:
            virial = virial + msum(torch.einsum("nma,nmb->nab", r * vm, rbar_ij).double())
        if p == 0:
            break
        # kernel: nse_bwd for pass p-1 (its NSE produced q_p and a_p = a_{p-1} + delta_a)
        prev = saved[p - 1]
        wl = prev["f"] / prev["F"][mol]
        Wbar = msum(qbar * wl)
        qrbar = qbar - Wbar[mol]
        fbar = (prev["D"] / prev["F"])[mol] * qrbar
        ftbar = 2.0 * prev["ft"] * fbar
        ybar = torch.cat([qrbar, ftbar, abar.reshape(n, -1)], dim=-1)
        qbar = qrbar.clone() if p - 1 > 0 else torch.zeros_like(qbar)  # q_raw = q_prev + q~ only for pass >= 1
        # abar carries over unchanged (a_p = a_{p-1} + delta_a)

    res = {
        "energy": energy.numpy().copy(),
        "charges": q.numpy().copy(),
        **({"spin_charges": (q_ch[:, 0] - q_ch[:, 1]).numpy().copy()} if nq == 2 else {}),
        "forces": (-xbar).numpy().copy(),
        "_aim": aim.numpy().copy(),
        "_e_atom": e_atom.numpy().copy(),
    }
    if stress:
        c = torch.as_tensor(np.asarray(cell)).double()
        if c.ndim == 2:
            c = c.unsqueeze(0)
        vol = torch.linalg.det(c).abs().view(-1, 1, 1)
        st = virial / vol
        res["stress"] = (st[0] if np.asarray(cell).ndim == 2 else st).to(dt).numpy().copy()
    return res


# ======================================================================================================================
# Tangent sweep: H v = d/d eps [dE/dx (x + eps v)] by forward-mode differentiation of the forward + backward sweep above,
# hand-derived (no autograd), K directions at once.  It is the specification of aimnetcentral_amd/csrc/hvp.hip: every block
# names the kernel that implements it and keeps that kernel's centre-major gather form.  Reference operator:
# calculators/calculator.py:1753-1989 (hessian_vector_product: vjp of the force graph), derivatives.py:149-192 (dense Hessian),
# the double-backward op kernels/conv_sv_2d_sp_wp.py:167-244.  Validated in tests/test_oracle_analytic.py against the autograd
# Hessian of oracle/aimnet2_oracle.py and against central differences of `evaluate` in fp64.
#
# Notation: X = primal value (as in `evaluate`), tX = its directional derivative, leading axis K.
def gelu_grad2(z: Tensor) -> Tensor:
    """GELU''(z) = phi(z) (2 - z^2)."""
    return torch.exp(-0.5 * z * z) * SQRT_2_OVER_PI_HALF * (2.0 - z * z)


def radial_basis2(model: OracleModel, d: Tensor, valid: Tensor):
    """gs, gs', gs'' (the envelope's second derivative jumps at rc: 0.5 (pi/rc)^2 inside, 0 outside)."""
    rc = model.rc
    w = math.pi / rc
    dc = d.clamp(min=1e-6, max=float(rc))
    inside = ((d > 1e-6) & (d < rc)).to(d.dtype)
    fc = 0.5 * (torch.cos(dc * w) + 1.0)
    dfc = -0.5 * w * torch.sin(dc * w) * inside
    d2fc = -0.5 * w * w * torch.cos(dc * w) * inside
    x = d.unsqueeze(-1) - model.shifts
    G = torch.exp(-model.eta * x * x)
    dG = -2.0 * model.eta * x * G
    d2G = (4.0 * model.eta * model.eta * x * x - 2.0 * model.eta) * G
    f, df, d2f = fc.unsqueeze(-1), dfc.unsqueeze(-1), d2fc.unsqueeze(-1)
    v = valid.unsqueeze(-1).to(d.dtype)
    return G * f * v, (dG * f + G * df) * v, (d2G * f + 2.0 * dG * df + G * d2f) * v


def sr_weight(model: OracleModel, d: Tensor, envelope: str = "exp"):
    """w = fc(d) / d of the embedded short-range Coulomb term with two derivatives (lr.py:21-62, ops.py:82-96)."""
    rc = float(model.sr_rc)
    if envelope == "exp":
        tr = d / rc
        t = tr.clamp(0, 1.0 - 1e-6)
        om = 1.0 - t * t
        fc = torch.exp(-1.0 / om) / 0.36787944117144233
        live = (tr < 1.0 - 1e-6).to(d.dtype)
        s1 = -2.0 * t / om**2
        s2 = -2.0 * (1.0 + 3.0 * t * t) / om**3
        dfc = fc * s1 / rc * live
        d2fc = fc * (s1 * s1 + s2) / (rc * rc) * live
    else:
        w = math.pi / rc
        dc = d.clamp(min=1e-6, max=rc)
        live = ((d > 1e-6) & (d < rc)).to(d.dtype)
        fc = 0.5 * (torch.cos(dc * w) + 1.0)
        dfc = -0.5 * w * torch.sin(dc * w) * live
        d2fc = -0.5 * w * w * torch.cos(dc * w) * live
    inv = 1.0 / d
    return fc * inv, dfc * inv - fc * inv * inv, d2fc * inv - 2.0 * dfc * inv * inv + 2.0 * fc * inv**3


def evaluate_hvp(
    model: OracleModel,
    coord_wrapped,
    numbers,
    charge,
    mol_idx,
    nbmat,
    vectors,
    shifts=None,
    cell=None,
    coulomb: str = "simple",
    nbmat_lr=None,
    shifts_lr=None,
    dsf_rc: float = 15.0,
    dsf_alpha: float = 0.2,
    mult=None,
    sr_coulomb: bool = True,
    sr_envelope: str | None = None,
    return_intermediates: bool = False,
) -> dict[str, np.ndarray]:
    """H v for K directions `vectors` (K, N, 3) (+ the primal forces of the same sweep).  Arguments as `evaluate`."""
    dt = model.dtype
    x = torch.as_tensor(np.asarray(coord_wrapped)).to(dt)
    n = x.shape[0]
    tv = torch.as_tensor(np.asarray(vectors)).to(dt).reshape(-1, n, 3)
    K = tv.shape[0]
    Z = torch.as_tensor(np.asarray(numbers)).long()
    mol = torch.as_tensor(np.asarray(mol_idx)).long()
    Q = torch.as_tensor(np.atleast_1d(np.asarray(charge))).to(dt)
    n_mol = Q.shape[0]
    nq = model.nq
    if nq == 2:
        mt = torch.ones_like(Q) if mult is None else torch.as_tensor(np.atleast_1d(np.asarray(mult))).to(dt)
        Q = torch.stack([0.5 * Q + 0.5 * (mt - 1.0), 0.5 * Q - 0.5 * (mt - 1.0)], dim=-1)
    else:
        Q = Q.unsqueeze(-1)
    A, G = model.A, model.G
    H = model.agh_a.shape[2]
    cell_at = None
    if cell is not None:
        c = torch.as_tensor(np.asarray(cell)).to(dt)
        if c.ndim == 2:
            c = c.unsqueeze(0)
        cell_at = c[mol] if c.shape[0] > 1 else c.expand(n, 3, 3)
    nb, valid, sh = _strip(nbmat, shifts, n)
    dbg: dict[str, Tensor] = {}

    def msum(v: Tensor) -> Tensor:
        out = torch.zeros((n_mol,) + v.shape[1:], dtype=v.dtype)
        return out.index_add_(0, mol, v)

    def msumK(v: Tensor) -> Tensor:  # (K, N, ...) -> (K, n_mol, ...)
        out = torch.zeros((v.shape[0], n_mol) + v.shape[2:], dtype=v.dtype)
        return out.index_add_(1, mol, v)

    def pair_tangent(nb_, valid_, d_, u_):
        """kernel: every pair kernel starts with this: t_r = v_j - v_i (the cell is fixed), t_d = u . t_r, t_u = (t_r - u t_d) / d."""
        vm = valid_.to(dt)
        t_r = (tv[:, nb_] - tv.unsqueeze(2)) * vm.unsqueeze(-1)
        t_d = (u_ * t_r).sum(-1)
        t_u = (t_r - u_ * t_d.unsqueeze(-1)) / d_.unsqueeze(-1)
        return t_d, t_u

    # ------------------------------------------------------------------ forward + its tangent
    r, d, u = pair_geometry(x, nb, valid, sh, cell_at)
    t_d, t_u = pair_tangent(nb, valid, d, u)
    gs, dgs, d2gs = radial_basis2(model, d, valid)
    t_gs = dgs * t_d.unsqueeze(-1)    # (K, N, M, G)
    t_dgs = d2gs * t_d.unsqueeze(-1)
    ou = torch.cat([torch.ones_like(d).unsqueeze(-1), u], dim=-1)
    t_ou = torch.cat([torch.zeros_like(t_d).unsqueeze(-1), t_u], dim=-1)

    a = model.afv[Z].view(n, A, G)
    t_a = torch.zeros(K, n, A, G, dtype=dt)
    q = torch.zeros(n, nq, dtype=dt)
    t_q = torch.zeros(K, n, nq, dtype=dt)
    npass = len(model.mlps)
    saved = []

    def mlp_fwd(h, t_h, layers, last_linear):
        """kernels: gemm (primal rows, bias) + gemm (K N tangent rows, no bias) + hvp_act: t_h = GELU'(z) t_z."""
        zs, t_zs = [], []
        for li, (w, b) in enumerate(layers):
            z = h @ w.T + b
            t_z = t_h @ w.T
            zs.append(z)
            t_zs.append(t_z)
            if last_linear and li == len(layers) - 1:
                h, t_h = z, t_z
            else:
                h, t_h = gelu(z), gelu_grad(z) * t_z
        return h, t_h, zs, t_zs

    for p in range(npass):
        # kernel: hvp_conv_fwd   t_S = sum_m t_a_j gs (1,u) + a_j t_gs (1,u) + a_j gs (0,t_u)
        aj = a[nb]
        S = torch.einsum("nmag,nmg,nmc->nagc", aj, gs, ou)
        t_S = (torch.einsum("knmag,nmg,nmc->knagc", t_a[:, nb], gs, ou) + torch.einsum("nmag,knmg,nmc->knagc", aj, t_gs, ou)
               + torch.einsum("nmag,nmg,knmc->knagc", aj, gs, t_ou))
        V = torch.einsum("agh,nagk->nahk", model.agh_a, S[..., 1:])
        t_V = torch.einsum("agh,knagc->knahc", model.agh_a, t_S[..., 1:])
        xin = [a.reshape(n, -1), S[..., 0].reshape(n, -1), V.pow(2).sum(-1).reshape(n, -1)]
        t_xin = [t_a.reshape(K, n, -1), t_S[..., 0].reshape(K, n, -1), (2.0 * V * t_V).sum(-1).reshape(K, n, -1)]
        Vq = t_Vq = None
        if p > 0:
            qj = q[nb]
            Sq = torch.einsum("nmq,nmg,nmc->nqgc", qj, gs, ou)
            t_Sq = (torch.einsum("knmq,nmg,nmc->knqgc", t_q[:, nb], gs, ou) + torch.einsum("nmq,knmg,nmc->knqgc", qj, t_gs, ou)
                    + torch.einsum("nmq,nmg,knmc->knqgc", qj, gs, t_ou))
            Vq = torch.einsum("qgh,nqgk->nqhk", model.agh_q, Sq[..., 1:])
            t_Vq = torch.einsum("qgh,knqgc->knqhc", model.agh_q, t_Sq[..., 1:])
            xin += [q, Sq[..., 0].reshape(n, -1), Vq.pow(2).sum(-1).reshape(n, -1)]
            t_xin += [t_q, t_Sq[..., 0].reshape(K, n, -1), (2.0 * Vq * t_Vq).sum(-1).reshape(K, n, -1)]
        h, t_h, zs, t_zs = mlp_fwd(torch.cat(xin, dim=-1), torch.cat(t_xin, dim=-1), model.mlps[p], p == 0)
        rec = {"a": a, "t_a": t_a, "q": q, "t_q": t_q, "V": V, "t_V": t_V, "Vq": Vq, "t_Vq": t_Vq, "zs": zs, "t_zs": t_zs}
        dbg[f"t_x{p}"] = torch.cat(t_xin, dim=-1)
        dbg[f"t_y{p}"] = t_h
        if p < npass - 1:
            # kernel: hvp_nse_fwd
            qt, ft, da = h[:, :nq], h[:, nq : 2 * nq], h[:, 2 * nq :]
            t_qt, t_ft, t_da = t_h[..., :nq], t_h[..., nq : 2 * nq], t_h[..., 2 * nq :]
            qr = q + qt if p > 0 else qt
            t_qr = t_q + t_qt if p > 0 else t_qt
            f = ft * ft
            t_f = 2.0 * ft * t_ft
            F = msum(f) + 1.0e-6
            t_F = msumK(t_f)
            D = Q - msum(qr)
            t_D = -msumK(t_qr)
            Fi, Di = F[mol], D[mol]
            q = qr + f / Fi * Di
            t_q = t_qr + t_f / Fi * Di - f * t_F[:, mol] / (Fi * Fi) * Di + f / Fi * t_D[:, mol]
            a = a + da.view(n, A, G)
            t_a = t_a + t_da.reshape(K, n, A, G)
            rec.update(ft=ft, t_ft=t_ft, f=f, t_f=t_f, F=F, t_F=t_F, D=D, t_D=t_D)
            dbg[f"t_q{p}"] = t_q
        saved.append(rec)
    _, _, hz, t_hz = mlp_fwd(h, t_h, model.head, True)

    # ------------------------------------------------------------------ Coulomb: adjoint seeds and their tangents
    qs, t_qs = q.sum(-1), t_q.sum(-1)
    qbar = torch.zeros(n, dtype=dt)
    t_qbar = torch.zeros(K, n, dtype=dt)
    xbar = torch.zeros(n, 3, dtype=dt)
    t_xbar = torch.zeros(K, n, 3, dtype=dt)

    def pair_term(w, dw, d2w, nb_, valid_, d_, u_, t_d_, t_u_, sign: float):
        """kernel: hvp_coulomb_pairs (full symmetric list, ordered pairs)."""
        nonlocal qbar, t_qbar, xbar, t_xbar
        vm = valid_.to(dt)
        qj, t_qj = qs[nb_], t_qs[:, nb_]
        k2 = sign * 2.0 * COULOMB_FACTOR
        qbar = qbar + k2 * (w * qj * vm).sum(-1)
        t_qbar = t_qbar + k2 * ((dw * t_d_ * qj + w * t_qj) * vm).sum(-1)
        qq = qs.unsqueeze(1) * qj * vm
        t_qq = (t_qs.unsqueeze(2) * qj + qs.unsqueeze(1) * t_qj) * vm
        xbar = xbar - k2 * ((dw * qq).unsqueeze(-1) * u_).sum(1)
        t_xbar = t_xbar - k2 * ((d2w * t_d_ * qq + dw * t_qq).unsqueeze(-1) * u_ + (dw * qq).unsqueeze(-1) * t_u_).sum(2)

    if sr_coulomb:
        w, dw, d2w = sr_weight(model, d, sr_envelope or getattr(model, "sr_envelope", "exp"))
        pair_term(w, dw, d2w, nb, valid, d, u, t_d, t_u, -1.0)
    if coulomb != "none":
        nbl, validl, shl = _strip(nbmat_lr, shifts_lr, n)
        rl, dl, ul = pair_geometry(x, nbl, validl, shl, cell_at)
        t_dl, t_ul = pair_tangent(nbl, validl, dl, ul)
        if coulomb == "simple":
            pair_term(1.0 / dl, -1.0 / (dl * dl), 2.0 / dl**3, nbl, validl, dl, ul, t_dl, t_ul, 1.0)
        else:
            al, Rc = dsf_alpha, dsf_rc
            cpi = 2.0 * al / math.sqrt(math.pi)
            erfc_rc = math.erfc(al * Rc)
            slope = erfc_rc / Rc**2 + cpi * math.exp(-(al**2) * Rc**2) / Rc
            inside = validl & (dl < Rc)
            ec, ex = torch.erfc(al * dl), torch.exp(-(al * dl) ** 2)
            w = ec / dl - erfc_rc / Rc + (dl - Rc) * slope
            dw = -ec / (dl * dl) - cpi * ex / dl + slope
            d2w = 2.0 * ec / dl**3 + 2.0 * cpi * ex / (dl * dl) + 2.0 * al * al * cpi * ex
            pair_term(w, dw, d2w, nbl, inside, dl, ul, t_dl, t_ul, 1.0)
            cs = -(erfc_rc / Rc / 2.0 + al / math.sqrt(math.pi))
            qbar = qbar + 4.0 * COULOMB_FACTOR * cs * qs
            t_qbar = t_qbar + 4.0 * COULOMB_FACTOR * cs * t_qs
    qbar = qbar.unsqueeze(-1).expand(n, nq).clone()
    t_qbar = t_qbar.unsqueeze(-1).expand(K, n, nq).clone()
    dbg["t_qbar_seed"] = t_qbar.clone()
    dbg["t_xbar_seed"] = t_xbar.clone()

    # ------------------------------------------------------------------ backward + its tangent
    def mlp_bwd(g, t_g, layers, zs, t_zs, last_linear):
        """kernels: hvp_act_bwd  t_t = t_g GELU'(z) + g GELU''(z) t_z, t = g GELU'(z);  gemm (t . W) for both."""
        for li in range(len(layers) - 1, -1, -1):
            if not (last_linear and li == len(layers) - 1):
                t_g = t_g * gelu_grad(zs[li]) + g * gelu_grad2(zs[li]) * t_zs[li]
                g = g * gelu_grad(zs[li])
            g, t_g = g @ layers[li][0], t_g @ layers[li][0]
        return g, t_g

    nh = len(model.head)
    g = model.head[-1][0].expand(n, -1).clone()
    ybar, t_ybar = mlp_bwd(g, torch.zeros(K, *g.shape, dtype=dt), model.head[: nh - 1], hz, t_hz, False)
    abar = torch.zeros(n, A, G, dtype=dt)
    t_abar = torch.zeros(K, n, A, G, dtype=dt)
    dinv = (1.0 / d).unsqueeze(-1)
    vm = valid.to(dt).unsqueeze(-1)

    def rbar_of(dbar, ubar, t_dbar, t_ubar, uu, t_uu):
        """dE/dr of an ordered pair from (dE/dd, dE/du) and its tangent: rbar = dbar u + (ubar - (ubar.u) u) / d."""
        pu = (ubar * uu).sum(-1, keepdim=True)
        t_pu = (t_ubar * uu).sum(-1, keepdim=True) + (ubar.unsqueeze(0) * t_uu).sum(-1, keepdim=True)
        perp = ubar - pu * uu
        t_perp = t_ubar - t_pu * uu - pu * t_uu
        rb = dbar.unsqueeze(-1) * uu + perp * dinv
        t_rb = (t_dbar.unsqueeze(-1) * uu + dbar.unsqueeze(-1) * t_uu + t_perp * dinv
                - perp * dinv * dinv * t_d.unsqueeze(-1))
        return rb * vm, t_rb * vm

    for p in range(npass - 1, -1, -1):
        rec = saved[p]
        xb, t_xb = mlp_bwd(ybar, t_ybar, model.mlps[p], rec["zs"], rec["t_zs"], p == 0)
        dbg[f"t_xb{p}"] = t_xb
        # kernel: hvp_unconcat
        AG, AH = A * G, A * H
        abar = abar + xb[:, :AG].view(n, A, G)
        t_abar = t_abar + t_xb[..., :AG].reshape(K, n, A, G)
        vbar, t_vbar = xb[:, 2 * AG : 2 * AG + AH].view(n, A, H), t_xb[..., 2 * AG : 2 * AG + AH].reshape(K, n, A, H)
        Sbar = torch.empty(n, A, G, 4, dtype=dt)
        t_Sbar = torch.empty(K, n, A, G, 4, dtype=dt)
        Sbar[..., 0] = xb[:, AG : 2 * AG].view(n, A, G)
        t_Sbar[..., 0] = t_xb[..., AG : 2 * AG].reshape(K, n, A, G)
        Sbar[..., 1:] = torch.einsum("agh,nahk->nagk", model.agh_a, 2.0 * rec["V"] * vbar.unsqueeze(-1))
        t_Sbar[..., 1:] = torch.einsum("agh,knahc->knagc", model.agh_a,
                                       2.0 * (rec["t_V"] * vbar.unsqueeze(-1) + rec["V"] * t_vbar.unsqueeze(-1)))
        a_p, t_a_p, q_p, t_q_p = rec["a"], rec["t_a"], rec["q"], rec["t_q"]
        # kernel: hvp_conv_bwd (centre-major, both halves of every ordered pair)
        Sb_j, t_Sb_j = Sbar[nb], t_Sbar[:, nb]
        Pp = Sb_j[..., 0] - torch.einsum("nmk,nmagk->nmag", u, Sb_j[..., 1:])
        t_Pp = (t_Sb_j[..., 0] - torch.einsum("knmc,nmagc->knmag", t_u, Sb_j[..., 1:])
                - torch.einsum("nmc,knmagc->knmag", u, t_Sb_j[..., 1:]))
        if p > 0:
            abar = abar + torch.einsum("nmg,nmag->nag", gs, Pp)
            t_abar = t_abar + torch.einsum("knmg,nmag->knag", t_gs, Pp) + torch.einsum("nmg,knmag->knag", gs, t_Pp)
        P = Sbar[..., 0].unsqueeze(1) + torch.einsum("nmk,nagk->nmag", u, Sbar[..., 1:])
        t_P = (t_Sbar[..., 0].unsqueeze(2) + torch.einsum("knmc,nagc->knmag", t_u, Sbar[..., 1:])
               + torch.einsum("nmc,knagc->knmag", u, t_Sbar[..., 1:]))
        a_j, t_a_j = a_p[nb], t_a_p[:, nb]
        Sv, t_Sv = Sbar[..., 1:], t_Sbar[..., 1:]
        Svj, t_Svj = Sb_j[..., 1:], t_Sb_j[..., 1:]
        dbar_ij = torch.einsum("nmag,nmag,nmg->nm", a_j, P, dgs)
        t_dbar_ij = (torch.einsum("knmag,nmag,nmg->knm", t_a_j, P, dgs) + torch.einsum("nmag,knmag,nmg->knm", a_j, t_P, dgs)
                     + torch.einsum("nmag,nmag,knmg->knm", a_j, P, t_dgs))
        ubar_ij = torch.einsum("nmg,nmag,nagk->nmk", gs, a_j, Sv)
        t_ubar_ij = (torch.einsum("knmg,nmag,nagc->knmc", t_gs, a_j, Sv) + torch.einsum("nmg,knmag,nagc->knmc", gs, t_a_j, Sv)
                     + torch.einsum("nmg,nmag,knagc->knmc", gs, a_j, t_Sv))
        dbar_ji = torch.einsum("nag,nmag,nmg->nm", a_p, Pp, dgs)
        t_dbar_ji = (torch.einsum("knag,nmag,nmg->knm", t_a_p, Pp, dgs) + torch.einsum("nag,knmag,nmg->knm", a_p, t_Pp, dgs)
                     + torch.einsum("nag,nmag,knmg->knm", a_p, Pp, t_dgs))
        ubar_ji = torch.einsum("nmg,nag,nmagk->nmk", gs, a_p, Svj)
        t_ubar_ji = (torch.einsum("knmg,nag,nmagc->knmc", t_gs, a_p, Svj) + torch.einsum("nmg,knag,nmagc->knmc", gs, t_a_p, Svj)
                     + torch.einsum("nmg,nag,knmagc->knmc", gs, a_p, t_Svj))
        if p > 0:
            c0 = 2 * AG + AH
            qbar = qbar + xb[:, c0 : c0 + nq]
            t_qbar = t_qbar + t_xb[..., c0 : c0 + nq]
            Sqbar = torch.empty(n, nq, G, 4, dtype=dt)
            t_Sqbar = torch.empty(K, n, nq, G, 4, dtype=dt)
            Sqbar[..., 0] = xb[:, c0 + nq : c0 + nq + nq * G].view(n, nq, G)
            t_Sqbar[..., 0] = t_xb[..., c0 + nq : c0 + nq + nq * G].reshape(K, n, nq, G)
            vqbar = xb[:, c0 + nq + nq * G : c0 + nq + nq * G + nq * H].view(n, nq, H)
            t_vqbar = t_xb[..., c0 + nq + nq * G : c0 + nq + nq * G + nq * H].reshape(K, n, nq, H)
            Sqbar[..., 1:] = torch.einsum("qgh,nqhk->nqgk", model.agh_q, 2.0 * rec["Vq"] * vqbar.unsqueeze(-1))
            t_Sqbar[..., 1:] = torch.einsum("qgh,knqhc->knqgc", model.agh_q,
                                            2.0 * (rec["t_Vq"] * vqbar.unsqueeze(-1) + rec["Vq"] * t_vqbar.unsqueeze(-1)))
            Sq_j, t_Sq_j = Sqbar[nb], t_Sqbar[:, nb]
            Pqp = Sq_j[..., 0] - torch.einsum("nmk,nmqgk->nmqg", u, Sq_j[..., 1:])
            t_Pqp = (t_Sq_j[..., 0] - torch.einsum("knmc,nmqgc->knmqg", t_u, Sq_j[..., 1:])
                     - torch.einsum("nmc,knmqgc->knmqg", u, t_Sq_j[..., 1:]))
            qbar = qbar + torch.einsum("nmg,nmqg->nq", gs, Pqp)
            t_qbar = t_qbar + torch.einsum("knmg,nmqg->knq", t_gs, Pqp) + torch.einsum("nmg,knmqg->knq", gs, t_Pqp)
            Pq = Sqbar[..., 0].unsqueeze(1) + torch.einsum("nmk,nqgk->nmqg", u, Sqbar[..., 1:])
            t_Pq = (t_Sqbar[..., 0].unsqueeze(2) + torch.einsum("knmc,nqgc->knmqg", t_u, Sqbar[..., 1:])
                    + torch.einsum("nmc,knqgc->knmqg", u, t_Sqbar[..., 1:]))
            qj, t_qj = q_p[nb], t_q_p[:, nb]
            Sqv, t_Sqv, Sqvj, t_Sqvj = Sqbar[..., 1:], t_Sqbar[..., 1:], Sq_j[..., 1:], t_Sq_j[..., 1:]
            dbar_ij = dbar_ij + torch.einsum("nmq,nmqg,nmg->nm", qj, Pq, dgs)
            t_dbar_ij = t_dbar_ij + (torch.einsum("knmq,nmqg,nmg->knm", t_qj, Pq, dgs) + torch.einsum("nmq,knmqg,nmg->knm", qj, t_Pq, dgs)
                                     + torch.einsum("nmq,nmqg,knmg->knm", qj, Pq, t_dgs))
            ubar_ij = ubar_ij + torch.einsum("nmq,nmg,nqgk->nmk", qj, gs, Sqv)
            t_ubar_ij = t_ubar_ij + (torch.einsum("knmq,nmg,nqgc->knmc", t_qj, gs, Sqv) + torch.einsum("nmq,knmg,nqgc->knmc", qj, t_gs, Sqv)
                                     + torch.einsum("nmq,nmg,knqgc->knmc", qj, gs, t_Sqv))
            dbar_ji = dbar_ji + torch.einsum("nq,nmqg,nmg->nm", q_p, Pqp, dgs)
            t_dbar_ji = t_dbar_ji + (torch.einsum("knq,nmqg,nmg->knm", t_q_p, Pqp, dgs) + torch.einsum("nq,knmqg,nmg->knm", q_p, t_Pqp, dgs)
                                     + torch.einsum("nq,nmqg,knmg->knm", q_p, Pqp, t_dgs))
            ubar_ji = ubar_ji + torch.einsum("nq,nmg,nmqgk->nmk", q_p, gs, Sqvj)
            t_ubar_ji = t_ubar_ji + (torch.einsum("knq,nmg,nmqgc->knmc", t_q_p, gs, Sqvj) + torch.einsum("nq,knmg,nmqgc->knmc", q_p, t_gs, Sqvj)
                                     + torch.einsum("nq,nmg,knmqgc->knmc", q_p, gs, t_Sqvj))
        rb_ij, t_rb_ij = rbar_of(dbar_ij, ubar_ij, t_dbar_ij, t_ubar_ij, u, t_u)
        rb_ji, t_rb_ji = rbar_of(dbar_ji, ubar_ji, t_dbar_ji, t_ubar_ji, -u, -t_u)
        xbar = xbar + (rb_ji - rb_ij).sum(1)
        t_xbar = t_xbar + (t_rb_ji - t_rb_ij).sum(2)
        if p == 0:
            break
        # kernel: hvp_nse_bwd (adjoint of pass p-1's charge update and its tangent)
        prev = saved[p - 1]
        Fi, Di = prev["F"][mol], prev["D"][mol]
        t_Fi, t_Di = prev["t_F"][:, mol], prev["t_D"][:, mol]
        wl = prev["f"] / Fi
        t_wl = prev["t_f"] / Fi - prev["f"] * t_Fi / (Fi * Fi)
        Wbar = msum(qbar * wl)
        t_Wbar = msumK(t_qbar * wl + qbar * t_wl)
        qrbar = qbar - Wbar[mol]
        t_qrbar = t_qbar - t_Wbar[:, mol]
        fbar = Di / Fi * qrbar
        t_fbar = (t_Di / Fi - Di * t_Fi / (Fi * Fi)) * qrbar + Di / Fi * t_qrbar
        ftbar = 2.0 * prev["ft"] * fbar
        t_ftbar = 2.0 * (prev["t_ft"] * fbar + prev["ft"] * t_fbar)
        ybar = torch.cat([qrbar, ftbar, abar.reshape(n, -1)], dim=-1)
        t_ybar = torch.cat([t_qrbar, t_ftbar, t_abar.reshape(K, n, -1)], dim=-1)
        dbg[f"t_ybar{p - 1}"] = t_ybar
        if p - 1 > 0:
            qbar, t_qbar = qrbar.clone(), t_qrbar.clone()
        else:
            qbar, t_qbar = torch.zeros_like(qbar), torch.zeros_like(t_qbar)

    res = {"hv": t_xbar.numpy().copy(), "forces": (-xbar).numpy().copy()}
    if return_intermediates:
        res.update({"_" + k: v.numpy().copy() for k, v in dbg.items()})
    return res
