"""ORACLE - test infrastructure only.  CPU restatement of the reference AIMNet2 energy/force path.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this file;
the product (aimnetcentral_amd/) never does and fails loudly when its HIP library is missing.

Parity status: PINNED - tests/test_oracle_golden.py checks this file against the golden
vectors in tests/golden/*.npz, which were produced by the unmodified reference
(tests/golden/make_golden.py) in the build container.

It restates, op for op, the PyTorch eager sequence of the reference (all paths relative to
/root/reference/aimnet):
  * flat "mode 1" layout with a trailing padding atom and sentinel neighbour index N
    (calculators/calculator.py:1521-1710 make_nbmat/pad_input, nbops.py:61-133 calc_masks)
  * ops.calc_distances ops.py:37-66, cosine_cutoff :82-85, exp_cutoff :88-90, exp_expand :93-96
  * AEVSV._calc_aev modules/aev.py:94-110, ConvSV.forward aev.py:156-189
  * AIMNet2.forward models/aimnet2.py:141-187 (_update_q :122-139), ops.nse ops.py:99-145
  * Output/AtomicShift/AtomicSum modules/core.py:71-132, SRCoulomb lr.py:986-1032 (_calc_coulomb_sr :21-62)
  * LRCoulomb.coul_simple lr.py:311-331, _coul_dsf_torch lr.py:559-615
  * LRCoulomb "ewald" (lr.py:617-720, calculator.py:1560-1603): the production arithmetic and parameter estimate live in
    nvalchemi-toolkit-ops 0.4.0 (ewald_summation / estimate_ewald_parameters), which is not in the reference tree - against THAT the
    parity is unpinned.  What the tree does hold is its own pure-PyTorch Ewald, `aimnet.ops.coulomb_matrix_ewald` (ops.py:196-276,
    "kept for ... regression cross-checks"), with the same parameter formula (eta = (V^2 / N)^(1/6) / sqrt(2 pi),
    rc = sqrt(-2 ln eps) eta, kc = sqrt(-2 ln eps) / eta, alpha = 1 / (sqrt(2) eta); also calculator.py:660-667): the restatement
    here is PINNED to golden matrices of that function (tests/golden/ewald_matrix.npz, energies and potentials to its fp32 rounding)
    and, beyond fp32, to published Madelung constants, independence of the splitting and finite differences
    (tests/test_oracle_ewald.py).
  * derivatives: calculators/derivatives.py:47-146 (autograd forces; row-vector strain stress)
Forces/stress come from torch.autograd exactly as in the reference; oracle/aimnet2_analytic.py
holds the hand-derived backward that the HIP kernels implement and is itself checked against
this file.
"""
from __future__ import annotations

import math
from typing import Any

import numpy as np
import torch
from torch import Tensor

HARTREE = 27.211386024367243
BOHR = 0.5291772105638411
COULOMB_FACTOR = 0.5 * HARTREE * BOHR


# --------------------------------------------------------------------------------------------
# neighbour list (brute force; the reference delegates to nvalchemiops, un-vendored, 0.4.0)
# --------------------------------------------------------------------------------------------
def wrap_into_cell(coord: np.ndarray, cell: np.ndarray, mol_idx: np.ndarray, pbc: np.ndarray) -> np.ndarray:
    """neighbors.py:331-381 move_coord_to_cell: fractional % 1 on periodic axes (fp32 like the reference)."""
    c = torch.as_tensor(coord, dtype=torch.float32)
    cells = torch.as_tensor(cell, dtype=torch.float32)
    if cells.ndim == 2:
        cells = cells.unsqueeze(0)
    pb = torch.as_tensor(pbc, dtype=torch.bool)
    if pb.ndim == 1:
        pb = pb.unsqueeze(0).expand(cells.shape[0], -1)
    m = torch.as_tensor(mol_idx, dtype=torch.long)
    if cells.shape[0] == 1:
        inv = torch.linalg.inv(cells[0])
        f = c @ inv
        f = torch.where(pb[0], f % 1, f)
        return (f @ cells[0]).numpy()
    inv = torch.linalg.inv(cells)
    f = torch.bmm(c.unsqueeze(1), inv[m]).squeeze(1)
    f = torch.where(pb[m], f % 1, f)
    return torch.bmm(f.unsqueeze(1), cells[m]).squeeze(1).numpy()


def neighbor_list(coord, cutoff: float, mol_idx, cell=None, pbc=None):
    """Full neighbour matrix: returns (nbmat (N+1, M) int64 with sentinel N incl. padding row,
    shifts (N+1, M, 3) float32 or None).  Rows real-first; same-molecule pairs only.
    Brute force over images in fp64, vectorised per image; row order = (image, j) ascending."""
    pos = np.asarray(coord, dtype=np.float64)
    n = pos.shape[0]
    mol = np.asarray(mol_idx, dtype=np.int64)
    ii_all, jj_all, sh_all = [], [], []
    starts = np.concatenate([[0], np.nonzero(np.diff(mol))[0] + 1, [n]])
    for a, b in zip(starts[:-1], starts[1:]):  # molecules are contiguous (sorted mol_idx)
        p = pos[a:b]
        if cell is None:
            if math.isinf(cutoff):
                ok = np.ones((b - a, b - a), dtype=bool)
            else:
                ok = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1) < cutoff * cutoff
            np.fill_diagonal(ok, False)
            i, j = np.nonzero(ok)
            ii_all.append(i + a)
            jj_all.append(j + a)
            sh_all.append(np.zeros((i.size, 3), dtype=np.int64))
            continue
        cells = np.asarray(cell, dtype=np.float64)
        c = cells if cells.ndim == 2 else cells[mol[a]]
        pba = np.ones((1, 3), dtype=bool) if pbc is None else np.asarray(pbc, dtype=bool).reshape(-1, 3)
        pb = pba[mol[a]] if pba.shape[0] > 1 else pba[0]  # per-system flags (normalize_pbc, neighbors.py:309-321)
        vol = abs(np.linalg.det(c))
        nimg = []
        for k in range(3):
            h = vol / np.linalg.norm(np.cross(c[(k + 1) % 3], c[(k + 2) % 3]))
            nimg.append(int(math.ceil(cutoff / h)) if pb[k] else 0)
        for sx in range(-nimg[0], nimg[0] + 1):
            for sy in range(-nimg[1], nimg[1] + 1):
                for sz in range(-nimg[2], nimg[2] + 1):
                    off = sx * c[0] + sy * c[1] + sz * c[2]
                    ok = ((p[None, :, :] + off - p[:, None, :]) ** 2).sum(-1) < cutoff * cutoff
                    if sx == 0 and sy == 0 and sz == 0:
                        np.fill_diagonal(ok, False)
                    i, j = np.nonzero(ok)
                    ii_all.append(i + a)
                    jj_all.append(j + a)
                    sh_all.append(np.broadcast_to(np.array([sx, sy, sz]), (i.size, 3)))
    ii = np.concatenate(ii_all) if ii_all else np.zeros(0, dtype=np.int64)
    jj = np.concatenate(jj_all) if jj_all else np.zeros(0, dtype=np.int64)
    sh = np.concatenate(sh_all) if sh_all else np.zeros((0, 3), dtype=np.int64)
    order = np.argsort(ii, kind="stable")
    ii, jj, sh = ii[order], jj[order], sh[order]
    counts = np.bincount(ii, minlength=n)
    m = max(1, int(counts.max()) if counts.size else 1)
    first = np.concatenate([[0], np.cumsum(counts)[:-1]])
    slot = np.arange(ii.size) - first[ii]
    nbmat = np.full((n + 1, m), n, dtype=np.int64)
    shifts = np.zeros((n + 1, m, 3), dtype=np.float32)
    nbmat[ii, slot] = jj
    shifts[ii, slot] = sh
    return nbmat, (shifts if cell is not None else None)


def neighbor_list_fast(coord, cutoff: float, mol_idx, cell=None, pbc=None):
    """Same contract and the same pair SETS as `neighbor_list` (checked in tests/test_oracle_golden.py), built with a k-d tree over
    the explicit periodic images instead of N x N distance matrices per image: what bench.py's cpu_baseline times, so that the
    CPU figure is not dominated by a quadratic list builder the reference does not have (its CPU lists come from
    nvalchemiops' cell list, neighbors.py:106-125).  Row order: ascending (j, image) per row."""
    from scipy.spatial import cKDTree

    pos = np.asarray(coord, dtype=np.float64)
    n = pos.shape[0]
    mol = np.asarray(mol_idx, dtype=np.int64)
    if math.isinf(cutoff):
        return neighbor_list(coord, cutoff, mol_idx, cell, pbc)
    starts = np.concatenate([[0], np.nonzero(np.diff(mol))[0] + 1, [n]])
    ii_all, jj_all, sh_all = [], [], []
    for a, b in zip(starts[:-1], starts[1:]):
        p = pos[a:b]
        if cell is None:
            img, img_j, img_s = p, np.arange(b - a), np.zeros((b - a, 3), dtype=np.int64)
        else:
            cells = np.asarray(cell, dtype=np.float64)
            c = cells if cells.ndim == 2 else cells[mol[a]]
            pba = np.ones((1, 3), dtype=bool) if pbc is None else np.asarray(pbc, dtype=bool).reshape(-1, 3)
            pb = pba[mol[a]] if pba.shape[0] > 1 else pba[0]
            vol = abs(np.linalg.det(c))
            nimg = []
            for k in range(3):
                h = vol / np.linalg.norm(np.cross(c[(k + 1) % 3], c[(k + 2) % 3]))
                nimg.append(int(math.ceil(cutoff / h)) if pb[k] else 0)
            sh = np.array([(sx, sy, sz) for sx in range(-nimg[0], nimg[0] + 1) for sy in range(-nimg[1], nimg[1] + 1)
                           for sz in range(-nimg[2], nimg[2] + 1)], dtype=np.int64)
            img = (p[None, :, :] + (sh.astype(np.float64) @ c)[:, None, :]).reshape(-1, 3)
            img_j = np.tile(np.arange(b - a), len(sh))
            img_s = np.repeat(sh, b - a, axis=0)
            lo, hi = p.min(0) - cutoff, p.max(0) + cutoff  # images farther than the cutoff from every atom cannot pair
            keep = ((img >= lo) & (img <= hi)).all(1)
            img, img_j, img_s = img[keep], img_j[keep], img_s[keep]
        tree = cKDTree(img)
        hits = tree.query_ball_point(p, cutoff, return_sorted=True)
        for i, h in enumerate(hits):
            h = np.asarray(h, dtype=np.int64)
            d2 = ((img[h] - p[i]) ** 2).sum(-1)
            ok = d2 < cutoff * cutoff  # strict, like the brute-force builder
            ok &= ~((img_j[h] == i) & (img_s[h] == 0).all(1))
            h = h[ok]
            ii_all.append(np.full(h.size, i + a, dtype=np.int64))
            jj_all.append(img_j[h] + a)
            sh_all.append(img_s[h])
    ii = np.concatenate(ii_all) if ii_all else np.zeros(0, dtype=np.int64)
    jj = np.concatenate(jj_all) if jj_all else np.zeros(0, dtype=np.int64)
    sh = np.concatenate(sh_all) if sh_all else np.zeros((0, 3), dtype=np.int64)
    counts = np.bincount(ii, minlength=n)
    m = max(1, int(counts.max()) if counts.size else 1)
    first = np.concatenate([[0], np.cumsum(counts)[:-1]])
    slot = np.arange(ii.size) - first[ii]
    nbmat = np.full((n + 1, m), n, dtype=np.int64)
    shifts = np.zeros((n + 1, m, 3), dtype=np.float32)
    nbmat[ii, slot] = jj
    shifts[ii, slot] = sh
    return nbmat, (shifts if cell is not None else None)


# --------------------------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------------------------
class OracleModel:
    """Weights of one AIMNet2 core model (state-dict keys as in SURVEY.md 2.1)."""

    def __init__(self, state_dict: dict[str, Any], dtype: torch.dtype = torch.float32):
        self.dtype = dtype
        sd = {k: torch.as_tensor(np.asarray(v)) for k, v in state_dict.items()}
        f = lambda k: sd[k].to(dtype)  # noqa: E731
        self.rc = f("aev.rc_s")
        self.eta = f("aev.eta_s")
        self.shifts = f("aev.shifts_s")
        self.afv = f("afv.weight")
        self.agh_a = f("conv_a.agh")
        self.agh_q = f("conv_q.agh")
        self.mlps: list[list[tuple[Tensor, Tensor]]] = []
        p = 0
        while f"mlps.{p}.0.weight" in sd:
            layers, li = [], 0
            while f"mlps.{p}.{2 * li}.weight" in sd:
                layers.append((f(f"mlps.{p}.{2 * li}.weight"), f(f"mlps.{p}.{2 * li}.bias")))
                li += 1
            self.mlps.append(layers)
            p += 1
        self.head = []
        li = 0
        while f"outputs.energy_mlp.mlp.{2 * li}.weight" in sd:
            self.head.append((f(f"outputs.energy_mlp.mlp.{2 * li}.weight"), f(f"outputs.energy_mlp.mlp.{2 * li}.bias")))
            li += 1
        self.sae = sd["outputs.atomic_shift.shifts.weight"].to(torch.float64).squeeze(-1)
        self.sr_rc = f("outputs.srcoulomb.rc") if "outputs.srcoulomb.rc" in sd else torch.tensor(4.6, dtype=dtype)
        self.sr_envelope = "exp"  # SRCoulomb(envelope=...) of the model YAML (lr.py:986-1032): "exp" mollifier or "cosine"
        self.A = self.agh_a.shape[0]
        self.G = self.agh_a.shape[1]
        self.nq = int(self.agh_q.shape[0])  # num_charge_channels (aimnet2.py:21): 2 = open-shell NSE


def _gelu(x: Tensor) -> Tensor:
    return torch.nn.functional.gelu(x)  # exact erf form (core.py:27, torch.nn.GELU default)


def _mlp(x: Tensor, layers, last_linear: bool) -> Tensor:
    for i, (w, b) in enumerate(layers):
        x = torch.nn.functional.linear(x, w, b)
        if not (last_linear and i == len(layers) - 1):
            x = _gelu(x)
    return x


def _distances(coord_p: Tensor, nbmat: Tensor, shifts, cell, mol_idx_p: Tensor):
    """ops.calc_distances for mode 1 (ops.py:37-66): masked pairs get r=(1,1,1)."""
    n = coord_p.shape[0] - 1
    coord_j = coord_p.index_select(0, nbmat.flatten()).unflatten(0, nbmat.shape)
    if shifts is not None:
        if cell.ndim == 2:
            off = shifts @ cell
        else:
            off = torch.einsum("nmd,ndh->nmh", shifts, cell[mol_idx_p])
        coord_j = coord_j + off
    r = coord_j - coord_p.unsqueeze(1)
    mask = nbmat == n
    r = r.masked_fill(mask.unsqueeze(-1), 1.0)
    d = torch.linalg.vector_norm(r, ord=2, dim=-1)
    return d, r, mask


def _cosine_cutoff(d: Tensor, rc: Tensor) -> Tensor:
    return 0.5 * (torch.cos(d.clamp(min=torch.full_like(rc, 1e-6), max=rc) * (math.pi / rc)) + 1.0)


def _exp_cutoff(d: Tensor, rc: Tensor) -> Tensor:
    return torch.exp(-1.0 / (1.0 - (d / rc).clamp(0, 1.0 - 1e-6).pow(2))) / 0.36787944117144233


def _mol_sum(x: Tensor, mol_idx_p: Tensor, n_mol: int) -> Tensor:
    if x.ndim == 1:
        res = torch.zeros(n_mol, dtype=x.dtype)
        return res.scatter_add_(0, mol_idx_p, x)
    res = torch.zeros(n_mol, x.shape[1], dtype=x.dtype)
    return res.scatter_add_(0, mol_idx_p.unsqueeze(-1).expand(-1, x.shape[1]), x)


def _conv(a: Tensor, g_sv: Tensor, nbmat: Tensor, agh: Tensor, d2: bool) -> Tensor:
    a_j = a.index_select(0, nbmat.flatten()).unflatten(0, nbmat.shape)
    if d2:
        avf_sv = torch.einsum("...mag,...mgd->...agd", a_j, g_sv)
    else:
        avf_sv = torch.einsum("...ma,...mgd->...agd", a_j, g_sv)
    avf_s, avf_v = avf_sv.split([1, 3], dim=-1)
    avf_v = torch.einsum("agh,...agd->...ahd", agh, avf_v).pow(2).sum(-1)
    return torch.cat([avf_s.squeeze(-1).flatten(-2, -1), avf_v.flatten(-2, -1)], dim=-1)


BOHR_INV = 1.0 / 0.5291772105638411
HALF_HARTREE = 0.5 * 27.211386024367243


def dftd3_energy(d_ij: Tensor, mask: Tensor, numbers_p: Tensor, nbmat: Tensor, mol_p: Tensor, n_mol: int, par: dict) -> Tensor:
    """DFT-D3(BJ) two-body energy per molecule in eV on a full neighbour matrix (distances in Angstrom):
    restatement of DFTD3._compute_energy_torch (lr.py:1626-1660) with its helpers _calc_torch_coord_num (:1595),
    _calc_torch_c6ij (:1605) and _s5_switch_torch (:1580).  `par`: s6, s8, a1, a2, cutoff, smoothing_fraction and the
    reference tables c6ab / cn_ref [Z,Z,5,5], rcov / r4r2 [Z] (aimnet/dftd3_data.pt)."""
    dt = d_ij.dtype
    c6ab, cn_ref = torch.as_tensor(par["c6ab"]).to(dt), torch.as_tensor(par["cn_ref"]).to(dt)
    rcov, r4r2 = torch.as_tensor(par["rcov"]).to(dt), torch.as_tensor(par["r4r2"]).to(dt)
    d = d_ij.clamp_min(1.0e-12) * BOHR_INV
    zi = numbers_p.unsqueeze(1).expand_as(nbmat)
    zj = numbers_p[nbmat]
    # coordination numbers
    cn_ij = torch.sigmoid(16.0 * ((rcov[zi] + rcov[zj]) / d.clamp_min(1.0e-12) - 1.0)).masked_fill(mask, 0.0)
    cn = cn_ij.sum(-1)
    # C6 interpolation over the 5 x 5 reference systems
    cn_i = cn.view(-1, 1, 1, 1)
    cn_j = cn[nbmat].unsqueeze(-1).unsqueeze(-1)
    c6ref = c6ab[zi, zj]
    cnref_i = cn_ref[zi, zj]
    cnref_j = cn_ref[zj, zi].transpose(-1, -2)
    valid = c6ref != 0
    exp_arg = -4.0 * ((cn_i - cnref_i).pow(2) + (cn_j - cnref_j).pow(2))
    max_exp = exp_arg.masked_fill(~valid, -torch.inf).amax(dim=(-1, -2), keepdim=True)
    finite_max = torch.isfinite(max_exp)
    shifted = torch.where(finite_max, exp_arg - max_exp, torch.zeros_like(exp_arg))
    weights = torch.where(valid & finite_max & (shifted >= -12.0), shifted.exp(), torch.zeros_like(shifted))
    wsum = weights.sum(dim=(-1, -2))
    c6sum = (c6ref * weights).sum(dim=(-1, -2))
    c6ij = torch.where(wsum > 1.0e-12, c6sum / wsum.clamp_min(1.0e-12), torch.zeros_like(wsum))
    # BJ damping + S5 switch
    r4r2_ij = 3.0 * r4r2[zi] * r4r2[zj]
    r0 = par["a1"] * r4r2_ij.sqrt() + par["a2"]
    d2 = d.pow(2)
    d6, d8 = d2.pow(3), d2.pow(4)
    damping = par.get("s6", 1.0) / (d6 + r0.pow(6)) + par["s8"] * r4r2_ij / (d8 + r0.pow(8))
    r_off = float(par.get("cutoff", 15.0)) * BOHR_INV
    r_on = r_off * (1.0 - float(par.get("smoothing_fraction", 0.2)))
    if r_off > r_on:
        t = ((d - r_on) / (r_off - r_on)).clamp(0.0, 1.0)
        sw = 1.0 - (10.0 * t**3 - 15.0 * t**4 + 6.0 * t**5)
        sw = torch.where(d <= r_on, torch.ones_like(sw), sw)
    else:
        sw = torch.ones_like(d)
    e_ij = (-c6ij * damping * sw).masked_fill(mask, 0.0)
    return HALF_HARTREE * _mol_sum(e_ij.sum(-1), mol_p, n_mol)


# --------------------------------------------------------------------------------------------
# Ewald summation (periodic point charges in a neutralising background)
# --------------------------------------------------------------------------------------------
def ewald_parameters(n_atoms: int, volume: float, accuracy: float) -> tuple[float, float, float]:
    """(alpha, real-space cutoff, reciprocal-space cutoff) of one system for a target accuracy: the choice that balances the two
    sums (both O(N^1.5)); what estimate_ewald_parameters of nvalchemiops documents (called at calculator.py:1572-1578)."""
    eta = (volume * volume / max(n_atoms, 1)) ** (1.0 / 6.0) / math.sqrt(2.0 * math.pi)
    f = math.sqrt(-2.0 * math.log(accuracy))
    return 1.0 / (math.sqrt(2.0) * eta), f * eta, f / eta


def ewald_kvectors(cell: np.ndarray, kc: float) -> np.ndarray:
    """Integer triplets n of the half space (n1 > 0, or n1 = 0 and n2 > 0, or n1 = n2 = 0 and n3 > 0) with |2 pi n C^-T| <= kc."""
    c = np.asarray(cell, dtype=np.float64)
    binv = 2.0 * math.pi * np.linalg.inv(c).T  # rows: reciprocal vectors
    nmax = [int(math.floor(kc * np.linalg.norm(c[a]) / (2.0 * math.pi))) for a in range(3)]
    g = np.stack(np.meshgrid(np.arange(0, nmax[0] + 1), np.arange(-nmax[1], nmax[1] + 1), np.arange(-nmax[2], nmax[2] + 1), indexing="ij"),
                 axis=-1).reshape(-1, 3)
    half = (g[:, 0] > 0) | ((g[:, 0] == 0) & (g[:, 1] > 0)) | ((g[:, 0] == 0) & (g[:, 1] == 0) & (g[:, 2] > 0))
    g = g[half]
    k = g @ binv
    return g[(k * k).sum(-1) <= kc * kc]


def ewald_reciprocal(x: Tensor, q: Tensor, cell: Tensor, n_half: np.ndarray, alpha: float, chunk: int = 256) -> Tensor:
    """E_rec / k_e = (2 pi / V) sum_{k != 0} exp(-k^2 / 4 alpha^2) / k^2 |S(k)|^2 (written over the half space) minus the
    neutralising-background term pi Q^2 / (2 V alpha^2); x [n,3], q [n], cell [3,3] (strained cell: k follows the strain)."""
    vol = torch.linalg.det(cell).abs()
    binv = 2.0 * math.pi * torch.linalg.inv(cell).T
    e = x.new_zeros(())
    nh = torch.as_tensor(n_half, dtype=x.dtype)
    for c0 in range(0, nh.shape[0], chunk):
        k = nh[c0 : c0 + chunk] @ binv
        k2 = (k * k).sum(-1)
        th = x @ k.T
        sre = (q.unsqueeze(-1) * torch.cos(th)).sum(0)
        sim = (q.unsqueeze(-1) * torch.sin(th)).sum(0)
        e = e + (torch.exp(-k2 / (4.0 * alpha * alpha)) / k2 * (sre * sre + sim * sim)).sum()
    return 4.0 * math.pi / vol * e - math.pi * q.sum() ** 2 / (2.0 * vol * alpha * alpha)


def evaluate(
    model: OracleModel,
    coord,
    numbers,
    charge,
    mol_idx=None,
    cell=None,
    pbc=None,
    coulomb: str = "simple",
    dsf_rc: float = 15.0,
    dsf_alpha: float = 0.2,
    ewald_accuracy: float = 1e-6,
    forces: bool = True,
    stress: bool = False,
    hessian: bool = False,
    dftd3: dict | None = None,
    mult=None,
    return_intermediates: bool = False,
    nbmat=None,
    shifts=None,
    nbmat_lr=None,
    shifts_lr=None,
) -> dict[str, np.ndarray]:
    """One AIMNet2Calculator.eval on a flat (N,3) system (calculator.py:879-947) with external
    Coulomb `coulomb` in {"simple","dsf","ewald","none"} and sr_embedded SRCoulomb subtraction."""
    dt = model.dtype
    coord_np = np.asarray(coord, dtype=np.float32)
    n = coord_np.shape[0]
    numbers = np.asarray(numbers, dtype=np.int64)
    mol = np.zeros(n, dtype=np.int64) if mol_idx is None else np.asarray(mol_idx, dtype=np.int64)
    charge_t = torch.as_tensor(np.atleast_1d(np.asarray(charge, dtype=np.float32))).to(dt)
    n_mol = charge_t.shape[0]
    cell_t = None
    if cell is not None:
        pbc_np = np.ones(3, dtype=bool) if pbc is None else np.asarray(pbc, dtype=bool)
        coord_np = wrap_into_cell(coord_np, np.asarray(cell, dtype=np.float32), mol, pbc_np)
        cell_t = torch.as_tensor(np.asarray(cell, dtype=np.float32)).to(dt)
    else:
        pbc_np = None
    if nbmat is None:
        nbmat, shifts = neighbor_list(coord_np, float(model.rc), mol, cell, pbc_np)
    if coulomb == "simple" and cell is None:
        if nbmat_lr is None:
            nbmat_lr, shifts_lr = neighbor_list(coord_np, math.inf, mol)
    elif coulomb == "dsf" or (coulomb == "simple" and cell is not None):
        coulomb = "dsf"
        if nbmat_lr is None:
            nbmat_lr, shifts_lr = neighbor_list(coord_np, dsf_rc, mol, cell, pbc_np)
    ew = None
    if coulomb == "ewald":  # per-system parameters from the UNSTRAINED cell (constants of the evaluation)
        if cell is None or not pbc_np.all():
            raise ValueError("Ewald summation needs a cell that is periodic along all three axes (lr.py:655-657)")
        cells_np = np.asarray(cell, dtype=np.float64).reshape(-1, 3, 3)
        ew = []
        for m_ in range(n_mol):
            c_ = cells_np[m_ if cells_np.shape[0] > 1 else 0]
            al_, rc_, kc_ = ewald_parameters(int((mol == m_).sum()), abs(np.linalg.det(c_)), ewald_accuracy)
            ew.append((al_, rc_, kc_, ewald_kvectors(c_, kc_)))
        if nbmat_lr is None:
            nbmat_lr, shifts_lr = neighbor_list(coord_np, max(e_[1] for e_ in ew), mol, cell, pbc_np)
    nb = torch.as_tensor(nbmat)
    sh = None if shifts is None else torch.as_tensor(shifts).to(dt)

    # pad_input: padding atom row (coord 0, Z 0, mol_idx = last) calculator.py:1704-1710
    coord_p = torch.cat([torch.as_tensor(coord_np).to(dt), torch.zeros(1, 3, dtype=dt)], dim=0)
    numbers_p = torch.cat([torch.as_tensor(numbers), torch.zeros(1, dtype=torch.long)])
    mol_p = torch.cat([torch.as_tensor(mol), torch.as_tensor(mol[-1:])])

    coord_p.requires_grad_(forces or stress or hessian)
    x = coord_p
    cell_x = cell_t
    scaling = None
    if stress:
        assert cell_t is not None
        if cell_t.ndim == 2:
            scaling = torch.eye(3, dtype=dt, requires_grad=True)
            x = coord_p @ scaling
            cell_x = cell_t @ scaling
        else:
            scaling = torch.eye(3, dtype=dt).unsqueeze(0).repeat(cell_t.shape[0], 1, 1).requires_grad_(True)
            x = (coord_p.unsqueeze(1) @ scaling[mol_p]).squeeze(1)
            cell_x = cell_t @ scaling

    inter: dict[str, Tensor] = {}
    a = model.afv[numbers_p].unflatten(-1, (model.A, model.G))
    d_ij, r_ij, mask = _distances(x, nb, sh, cell_x, mol_p)
    fc = _cosine_cutoff(d_ij, model.rc).masked_fill(mask, 0.0)
    gs = torch.exp(-model.eta * (d_ij.unsqueeze(-1) - model.shifts) ** 2) * fc.unsqueeze(-1)
    u = r_ij / d_ij.unsqueeze(-1)
    g_sv = torch.cat([gs.unsqueeze(-1), gs.unsqueeze(-1) * u.unsqueeze(-2)], dim=-1)
    inter["d_ij"] = d_ij

    q = None
    nq = model.nq
    if nq == 2:  # _preprocess_spin_polarized_charge, aimnet2.py:94-100
        mult_t = torch.ones_like(charge_t) if mult is None else torch.as_tensor(np.atleast_1d(np.asarray(mult, dtype=np.float32))).to(dt)
        half_spin, half_q = 0.5 * (mult_t - 1.0), 0.5 * charge_t
        Q = torch.stack([half_q + half_spin, half_q - half_spin], dim=-1)
    else:
        Q = charge_t.unsqueeze(-1)
    npass = len(model.mlps)
    for ip, layers in enumerate(model.mlps):
        _in = torch.cat([a.flatten(-2, -1), _conv(a, g_sv, nb, model.agh_a, True)], dim=-1)
        if ip > 0:
            _in = torch.cat([_in, q, _conv(q, g_sv, nb, model.agh_q, False)], dim=-1)
        inter[f"mlp{ip}_in"] = _in
        out = _mlp(_in, layers, last_linear=(ip == 0))
        out = torch.cat([out[:-1], torch.zeros_like(out[:1])], dim=0)  # mask_i_ on padding row
        inter[f"mlp{ip}_out"] = out
        if ip < npass - 1:
            _q, _f, da = out.split([nq, nq, out.shape[-1] - 2 * nq], dim=-1)
            qr = q + _q if ip > 0 else _q
            f = _f.pow(2)
            F = _mol_sum(f, mol_p, n_mol) + 1.0e-6
            Qu = _mol_sum(qr, mol_p, n_mol)
            dQ = Q - Qu
            q = qr + f / F[mol_p] * dQ[mol_p]
            a = a + da.view_as(a)
            inter[f"q{ip}"] = q.squeeze(-1)
        else:
            aim = out
    spin = q[:, 0] - q[:, 1] if nq == 2 else None  # _postprocess_spin_polarized_charge, aimnet2.py:102-106
    charges = q.sum(-1)
    inter["aim"] = aim

    e_at = _mlp(aim, model.head, last_linear=True).squeeze(-1)
    e_at = torch.cat([e_at[:-1], torch.zeros_like(e_at[:1])])
    inter["e_atom"] = e_at
    e_at64 = e_at + model.sae[numbers_p]  # fp32 + fp64 -> fp64 (core.py:95)
    energy = _mol_sum(e_at64, mol_p, n_mol)

    def pair_sum(e_ij: Tensor) -> Tensor:
        return COULOMB_FACTOR * _mol_sum(e_ij.sum(-1, dtype=torch.float64), mol_p, n_mol)

    # embedded SRCoulomb: energy -= E_sr (lr.py:1020-1032)
    q_i, q_j = charges.unsqueeze(1), charges[nb]
    fc_sr = _exp_cutoff(d_ij, model.sr_rc) if model.sr_envelope == "exp" else _cosine_cutoff(d_ij, model.sr_rc)  # lr.py:54-57
    e_sr = (fc_sr * q_i * q_j / d_ij).masked_fill(mask, 0.0)
    energy = energy.double() - pair_sum(e_sr)

    if coulomb != "none":
        nbl = torch.as_tensor(nbmat_lr)
        shl = None if shifts_lr is None else torch.as_tensor(shifts_lr).to(dt)
        d_lr, _, mask_lr = _distances(x, nbl, shl, cell_x, mol_p)
        q_jl = charges[nbl]
        if coulomb == "simple":
            e_lr = (q_i * q_jl / d_lr).masked_fill(mask_lr, 0.0)
            energy = energy + pair_sum(e_lr)
        elif coulomb == "ewald":
            # real space: erfc(alpha d) / d inside the system's cutoff; self term -alpha / sqrt(pi) q_i^2; reciprocal space + background
            al_at = d_lr.new_tensor([e_[0] for e_ in ew])[mol_p].unsqueeze(-1)
            rc_at = d_lr.new_tensor([e_[1] for e_ in ew])[mol_p].unsqueeze(-1)
            e_lr = (q_i * q_jl * torch.erfc(al_at * d_lr) / d_lr * (d_lr < rc_at).to(dt)).masked_fill(mask_lr, 0.0)
            energy = energy + pair_sum(e_lr)
            q_self = torch.cat([charges[:-1], torch.zeros_like(charges[:1])])
            energy = energy + 2.0 * COULOMB_FACTOR * _mol_sum((-al_at.squeeze(-1) / math.sqrt(math.pi) * q_self.pow(2)).double(), mol_p, n_mol)
            e_rec = []
            for m_ in range(n_mol):
                idx = torch.as_tensor(np.nonzero(mol == m_)[0])
                c_ = cell_x if cell_x.ndim == 2 else cell_x[m_]
                e_rec.append(ewald_reciprocal(x[idx], charges[idx], c_, ew[m_][3], ew[m_][0]))
            energy = energy + 2.0 * COULOMB_FACTOR * torch.stack(e_rec).double()
        else:
            al = d_lr.new_tensor(dsf_alpha)
            rc = d_lr.new_tensor(dsf_rc)
            erfc_rc = torch.erfc(al * rc)
            shift_val = erfc_rc / rc
            slope = erfc_rc / rc.pow(2) + d_lr.new_tensor(2.0 * dsf_alpha / math.sqrt(math.pi)) * torch.exp(-(al**2) * rc**2) / rc
            e_pair = torch.erfc(al * d_lr) / d_lr - shift_val + (d_lr - rc) * slope
            e_lr = (q_i * q_jl * e_pair * (d_lr < rc).to(dt)).masked_fill(mask_lr, 0.0)
            energy = energy + pair_sum(e_lr)
            self_coeff = -(shift_val / 2.0 + d_lr.new_tensor(dsf_alpha / math.sqrt(math.pi)))
            q_self = torch.cat([charges[:-1], torch.zeros_like(charges[:1])])
            energy = energy + 2.0 * COULOMB_FACTOR * _mol_sum((self_coeff * q_self.pow(2)).double(), mol_p, n_mol)

    if dftd3 is not None:  # external DFT-D3 on its own list of the D3 cutoff (calculator.py:999-1032)
        rc3 = float(dftd3.get("cutoff", 15.0))
        nb3, sh3 = neighbor_list(coord_np, rc3, mol, cell, pbc_np)
        nb3_t = torch.as_tensor(nb3)
        sh3_t = None if sh3 is None else torch.as_tensor(sh3).to(dt)
        d3_d, _, d3_mask = _distances(x, nb3_t, sh3_t, cell_x, mol_p)
        e_d3 = dftd3_energy(d3_d, d3_mask, numbers_p, nb3_t, mol_p, n_mol, dftd3)
        inter["e_dftd3"] = e_d3
        energy = energy + e_d3.double()

    res: dict[str, np.ndarray] = {"energy": energy.detach().numpy().copy(), "charges": charges[:-1].detach().numpy().copy()}
    if spin is not None:
        res["spin_charges"] = spin[:-1].detach().numpy().copy()
    if hessian:
        # dense (N,3,N,3) Hessian by double backward, row by row (calculate_hessian, derivatives.py:149-192)
        (g,) = torch.autograd.grad(energy.sum(), coord_p, create_graph=True)
        rows = []
        for k in range(3 * n):
            (r,) = torch.autograd.grad(g[k // 3, k % 3], coord_p, retain_graph=True)
            rows.append(r[:-1])
        res["hessian"] = torch.stack(rows).reshape(n, 3, n, 3).detach().numpy().copy()
        if forces:
            res["forces"] = (-g[:-1]).detach().numpy().copy()
    elif forces or stress:
        wrt = [coord_p] + ([scaling] if stress else [])
        grads = torch.autograd.grad(energy.sum(), wrt)
        if forces:
            res["forces"] = (-grads[0][:-1]).detach().numpy().copy()
        if stress:
            dedc = grads[1]
            cell0 = cell_t.detach()
            vol = torch.linalg.det(cell0).abs()
            if cell0.ndim == 3:
                vol = vol.unsqueeze(-1).unsqueeze(-1)
            res["stress"] = (dedc / vol).detach().numpy().copy()
    if return_intermediates:
        res["nbmat"] = np.asarray(nbmat)
        if shifts is not None:
            res["shifts"] = np.asarray(shifts)
        res["coord_wrapped"] = coord_np
        for k, v in inter.items():
            res["_" + k] = v.detach().numpy().copy()
    return res
