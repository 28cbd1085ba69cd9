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

    rc_sr = float(model.sr_rc)
    t = (d / rc_sr).clamp(0, 1.0 - 1e-6)
    fce = torch.exp(-1.0 / (1.0 - t * t)) / 0.36787944117144233
    dfce = torch.where(d / rc_sr < 1.0 - 1e-6, fce * (-2.0 * t / (1.0 - t * t) ** 2) / rc_sr, torch.zeros_like(d))
    pair_term(fce / d, dfce / d - fce / (d * d), nb, valid, r, d, u, -1.0)  # embedded SRCoulomb (lr.py:1020)
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
