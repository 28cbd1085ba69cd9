"""Stand-alone C-ABI operators against the reference operator contracts."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from aimnetcentral_amd import engine as E
from aimnetcentral_amd import workloads
from oracle import aimnet2_oracle as O

pytestmark = pytest.mark.gpu


def reference_conv_sv_2d_sp_einsum(a, idx, g):
    """Masked einsum semantics the reference pins its Warp kernel to (tests/test_conv_sv_2d_sp.py:73-101)."""
    B, M = idx.shape
    valid = (idx < B - 1).unsqueeze(-1).unsqueeze(-1)
    a_sel = a.index_select(0, idx.clamp(0, B - 1).flatten()).unflatten(0, (B, M))
    out = torch.einsum("bmag,bmgd->bagd", a_sel, g * valid)
    out[-1] = 0
    return out


def make_idx(B, M, n_real, gen):
    idx = torch.full((B, M), B - 1, dtype=torch.int64)
    for b in range(B):
        n = int(torch.randint(0, min(n_real, M) + 1, (1,), generator=gen))
        idx[b, :n] = torch.randint(0, B - 1, (n,), generator=gen)
    return idx


@pytest.mark.parametrize("B,A,G,M", [(8, 16, 12, 10), (113, 16, 16, 62), (33, 4, 8, 5), (2, 16, 16, 1)])
def test_conv_sv_fwd_bwd_match_einsum_reference(B, A, G, M):
    gen = torch.Generator().manual_seed(B * 1000 + M)
    a = torch.randn(B, A, G, generator=gen)
    g = torch.randn(B, M, G, 4, generator=gen)
    idx = make_idx(B, M, M, gen)
    dev = torch.device("cuda:0")
    out = E.conv_sv_2d_sp_fwd(a.to(dev), idx.to(dev), g.to(dev)).cpu()
    ref = reference_conv_sv_2d_sp_einsum(a, idx, g)
    torch.testing.assert_close(out, ref, atol=1e-5, rtol=1e-4)  # test_conv_sv_2d_sp.py:158
    assert (out[-1] == 0).all()
    # backward against autograd of the einsum reference (test_conv_sv_2d_sp.py:186-192)
    a_r, g_r = a.clone().requires_grad_(True), g.clone().requires_grad_(True)
    go = torch.randn(B, A, G, 4, generator=gen)
    go[-1] = 0
    reference_conv_sv_2d_sp_einsum(a_r, idx, g_r).backward(go)
    ga, gg = E.conv_sv_2d_sp_bwd(go.to(dev), a.to(dev), idx.to(dev), g.to(dev))
    torch.testing.assert_close(ga.cpu(), a_r.grad, atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(gg.cpu(), g_r.grad, atol=1e-4, rtol=1e-3)


@pytest.mark.parametrize("B,A,G,M", [(8, 16, 12, 10), (57, 16, 16, 31), (33, 4, 8, 5)])
def test_conv_sv_double_backward_matches_autograd(B, A, G, M):
    """conv_sv_2d_sp_bwd_bwd against torch's double backward of the einsum reference (the reference pins its Warp
    kernels the same way, tests/test_conv_sv_2d_sp.py:197-253): cotangents v_a, v_g of (grad_a, grad_g)."""
    gen = torch.Generator().manual_seed(B * 77 + M)
    a = torch.randn(B, A, G, generator=gen, dtype=torch.float64, requires_grad=True)
    g = torch.randn(B, M, G, 4, generator=gen, dtype=torch.float64, requires_grad=True)
    idx = make_idx(B, M, M, gen)
    go = torch.randn(B, A, G, 4, generator=gen, dtype=torch.float64)
    go[-1] = 0
    go.requires_grad_(True)
    va = torch.randn(B, A, G, generator=gen, dtype=torch.float64)
    vg = torch.randn(B, M, G, 4, generator=gen, dtype=torch.float64)
    out = reference_conv_sv_2d_sp_einsum(a, idx, g)
    ga, gg = torch.autograd.grad(out, (a, g), go, create_graph=True)
    want = torch.autograd.grad((ga * va).sum() + (gg * vg).sum(), (go, a, g))
    dev = torch.device("cuda:0")
    got = E.conv_sv_2d_sp_bwd_bwd(go.detach().float().to(dev), va.float().to(dev), vg.float().to(dev), a.detach().float().to(dev),
                                  idx.to(dev), g.detach().float().to(dev))
    for name, x, w in zip(("grad_grad_output", "grad_a_double", "grad_g_double"), got, want):
        w = w.clone()
        if name == "grad_grad_output":
            w[-1] = 0  # the op leaves the padding row zero
        torch.testing.assert_close(x.cpu().double(), w, atol=2e-4, rtol=1e-3, msg=name)


def test_conv_sv_padding_counts():
    """rows with 0, partial and full neighbour counts (test_conv_sv_2d_sp.py:294-318)."""
    B, A, G, M = 6, 16, 16, 4
    gen = torch.Generator().manual_seed(7)
    a, g = torch.randn(B, A, G, generator=gen), torch.randn(B, M, G, 4, generator=gen)
    idx = torch.full((B, M), B - 1, dtype=torch.int64)
    idx[1, :1] = 0
    idx[2, :2] = torch.tensor([3, 1])
    idx[3, :4] = torch.tensor([0, 1, 2, 4])
    dev = torch.device("cuda:0")
    out = E.conv_sv_2d_sp_fwd(a.to(dev), idx.to(dev), g.to(dev)).cpu()
    assert (out[0] == 0).all() and (out[4] == 0).all() and (out[5] == 0).all()
    torch.testing.assert_close(out, reference_conv_sv_2d_sp_einsum(a, idx, g), atol=1e-5, rtol=1e-4)


def rows_as_sets(nbmat, num, shifts=None):
    out = []
    nb = nbmat.cpu().numpy()
    cnt = num.cpu().numpy()
    sh = shifts.cpu().numpy() if shifts is not None else None
    for i in range(nb.shape[0]):
        if sh is None:
            out.append(sorted((int(j), 0, 0, 0) for j in nb[i, : cnt[i]]))
        else:
            out.append(sorted((int(j), int(s[0]), int(s[1]), int(s[2])) for j, s in zip(nb[i, : cnt[i]], sh[i, : cnt[i]])))
    return out


def oracle_sets(nbmat, shifts, n):
    out = []
    for i in range(n):
        v = nbmat[i] < n
        if shifts is None:
            out.append(sorted((int(j), 0, 0, 0) for j in nbmat[i][v]))
        else:
            out.append(sorted((int(j), int(s[0]), int(s[1]), int(s[2])) for j, s in zip(nbmat[i][v], shifts[i][v])))
    return out


def test_neighbor_list_nonperiodic_batch_as_sets():
    """neighbour rows compared as SETS per atom, as the reference does (test_calculator_gpu.py:181-184)."""
    c, z, mol, q = workloads.random_batch(7, 5, 70, seed=9)
    dev = torch.device("cuda:0")
    nb, num, sh, xw, (mx, ovf) = E.neighbor_list(torch.from_numpy(c).to(dev), 5.0, torch.from_numpy(mol).to(dev), max_nb=96)
    ref, _ = O.neighbor_list(c, 5.0, mol)
    assert ovf == 0 and sh is None
    assert rows_as_sets(nb, num) == oracle_sets(ref, None, len(z))
    assert mx == (ref[:-1] < len(z)).sum(1).max()
    # packed real-first + fill_value padding (conv_sv_2d_sp_wp.py:630-636)
    nbn, cnt = nb.cpu().numpy(), num.cpu().numpy()
    for i in range(len(z)):
        assert (nbn[i, cnt[i]:] == len(z)).all() and (nbn[i, : cnt[i]] < len(z)).all()
    # symmetry of the full list
    pairs = {(i, j) for i, r in enumerate(rows_as_sets(nb, num)) for (j, *_s) in r}
    assert all((j, i) in pairs for (i, j) in pairs)


@pytest.mark.parametrize("cutoff,pbc", [(5.0, (True, True, True)), (9.0, (True, True, True)), (6.0, (True, False, True))])
def test_neighbor_list_periodic_small_cell_as_sets(cutoff, pbc):
    """96-atom monoclinic cell, a = 4.98 A < cutoff: several images of the same atom and self images."""
    c, z, cell = workloads.glucose_cell()
    rng = np.random.default_rng(1)
    c = c + rng.integers(-2, 3, size=(96, 3)) @ cell  # atoms outside the box: exercises wrapping
    mol = np.zeros(96, dtype=np.int64)
    dev = torch.device("cuda:0")
    nb, num, sh, xw, (mx, ovf) = E.neighbor_list(torch.from_numpy(c.astype(np.float32)).to(dev), cutoff,
                                                 torch.from_numpy(mol).to(dev), cell=torch.from_numpy(cell.astype(np.float32)).to(dev),
                                                 pbc=pbc, max_nb=640)
    assert ovf == 0
    pb = np.array(pbc)
    xw_ref = O.wrap_into_cell(c.astype(np.float32), cell.astype(np.float32), mol, pb)
    assert np.abs(xw.cpu().numpy() - xw_ref).max() < 5e-5
    ref, rsh = O.neighbor_list(xw.cpu().numpy(), cutoff, mol, cell, pb)
    got, want = rows_as_sets(nb, num, sh), oracle_sets(ref, rsh, 96)
    # pairs within 1e-4 A of the cutoff may legitimately differ between fp32 (GPU) and fp64 (oracle)
    diff = sum(len(set(a) ^ set(b)) for a, b in zip(got, want))
    assert diff <= 2, f"{diff} differing pairs"
    # exact symmetry (i,j,s) <-> (j,i,-s)
    allp = {(i, j, sx, sy, sz) for i, r in enumerate(got) for (j, sx, sy, sz) in r}
    assert all((j, i, -sx, -sy, -sz) in allp for (i, j, sx, sy, sz) in allp)


def test_neighbor_list_overflow_reported():
    c, z, mol, q = workloads.random_batch(1, 60, 60, seed=2)
    dev = torch.device("cuda:0")
    nb, num, sh, xw, (mx, ovf) = E.neighbor_list(torch.from_numpy(c).to(dev), 5.0, max_nb=8)
    assert ovf == 1 and mx > 8 and int(num.max()) == 8


def test_neighbor_list_batched_cells():
    c, z, cell = workloads.glucose_cell()
    cell2 = cell * np.array([[1.03], [0.97], [1.0]])
    c2 = (c @ np.linalg.inv(cell)) @ cell2
    cc = np.concatenate([c, c2]).astype(np.float32)
    mol = np.repeat(np.arange(2), 96)
    cells = np.stack([cell, cell2]).astype(np.float32)
    dev = torch.device("cuda:0")
    nb, num, sh, xw, (mx, ovf) = E.neighbor_list(torch.from_numpy(cc).to(dev), 5.0, torch.from_numpy(mol).to(dev),
                                                 cell=torch.from_numpy(cells).to(dev), max_nb=128)
    ref, rsh = O.neighbor_list(xw.cpu().numpy(), 5.0, mol, cells, np.ones(3, bool))
    got, want = rows_as_sets(nb, num, sh), oracle_sets(ref, rsh, 192)
    assert sum(len(set(a) ^ set(b)) for a, b in zip(got, want)) <= 2
    assert all(j < 96 for r in got[:96] for (j, *_x) in r) and all(j >= 96 for r in got[96:] for (j, *_x) in r)


# ---- MLP GEMM (csrc/gemm.hip): every tile configuration against an fp64 product ---------------------------
_PANEL = [152, 142, 132, 122, 153, 143, 223, 213, 222, 233, 412, 411, 410, 409, 381, 371, 361, 351, 341, 331, 321]
GEMM_CFGS = [0, 5, 7, *_PANEL, *[1000 + c for c in _PANEL]]  # 1000 + id: the 32-deep-stage variant of a panel tile


@pytest.mark.parametrize("cfg", GEMM_CFGS)
@pytest.mark.parametrize("M,N,K", [(1000, 288, 384), (333, 736, 512), (47, 128, 32)])
def test_gemm_tiles_match_fp64(cfg, M, N, K):
    """C = epilogue(A . Bt^T): bias+GELU (with its derivative D), plain, and the backward's multiply-by-D form.
    Tolerance: fp32 accumulation over K <= 512 terms of O(1) products -> |err| <= 5e-5 (observed ~1e-5)."""
    from aimnetcentral_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(1234 + cfg + M)
    A = torch.randn(M, K, generator=gen).to(dev)
    Bt = (torch.randn(N, K, generator=gen) * 0.1).to(dev)
    bias = torch.randn(N, generator=gen).to(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    z = A.double() @ Bt.double().T
    for epi in (0, 1, 2, 3):
        Cm = torch.full((M, N), float("nan"), device=dev)
        D = torch.rand(M, N, generator=gen).to(dev) if epi == 3 else torch.full((M, N), float("nan"), device=dev)
        D0 = D.clone()
        rc = lib.aimnet_debug_gemm(cfg, epi, A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), Cm.data_ptr(),
                                   D.data_ptr(), N, stream)
        assert rc == 0, _lib.last_error()
        torch.cuda.synchronize()
        if epi == 0:
            want = z
        elif epi == 1:
            want = z + bias.double()
        elif epi == 2:
            zb = z + bias.double()
            cdf = 0.5 * (1 + torch.erf(zb / 2**0.5))
            want = zb * cdf
            want_d = cdf + zb * torch.exp(-0.5 * zb * zb) / (2 * torch.pi) ** 0.5
            assert (D.double() - want_d).abs().max().item() < 5e-5
        else:
            want = z * D0.double()
        assert (Cm.double() - want).abs().max().item() < 5e-5, (cfg, epi)


# ---- bf16x3-split MLP GEMM (csrc/gemm_bf3.hip): every tile against an fp64 product, same tolerance as the exact-fp32 kernels ----
BF3_CFGS = [0, 452, 442, 432, 422, 223, 224, 234, 851, 861, 871, 891, 1452, 1442, 1432, 1422, 1223, 1224, 1234, 1851]


def _split_bf3(lib, x, stream, neg_from_block=1 << 30):
    m, k = x.shape
    out = torch.empty(m, 3 * k, dtype=torch.int16, device=x.device)
    assert lib.aimnet_debug_split_bf3(x.data_ptr(), k, m, k, out.data_ptr(), 3 * k, neg_from_block, stream) == 0
    return out


def test_bf3_split_is_exact_and_round_to_nearest():
    """fp32 == plane0 + plane1 + plane2 bit for bit, plane0 = round-to-nearest-even bf16 (what torch's own cast does)."""
    from aimnetcentral_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev).cuda_stream
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(257, 96, generator=gen) * torch.logspace(-6, 4, 96)).to(dev)
    x[0, :4] = torch.tensor([0.0, -0.0, 1.0, -3.5e-30])
    s3 = _split_bf3(lib, x, stream)
    planes = (s3.view(257, 3, 3, 32).to(torch.int32) << 16).view(torch.float32)
    assert torch.equal(planes.double().sum(dim=2).reshape(257, 96), x.double())
    assert torch.equal(planes[:, :, 0, :].reshape(257, 96), x.to(torch.bfloat16).float())


@pytest.mark.parametrize("cfg", BF3_CFGS)
@pytest.mark.parametrize("M,N,K", [(1000, 288, 384), (333, 736, 512), (47, 128, 32), (170, 128, 96)])
def test_gemm_bf3_tiles_match_fp64(cfg, M, N, K):
    """The exact-fp32 test above on the bf16x3-split kernels: activations fp32, weights pre-split, six bf16 MFMA products per
    tile, fp32 accumulation.  Same 5e-5 gate (observed ~1e-5, a little below the fp32 MFMA chain)."""
    from aimnetcentral_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(4321 + cfg + M)
    A = torch.randn(M, K, generator=gen).to(dev)
    Bt = (torch.randn(N, K, generator=gen) * 0.1).to(dev)
    bias = torch.randn(N, generator=gen).to(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    z = A.double() @ Bt.double().T
    for epi in (0, 1, 2, 3):
        # the sign-flipped second accumulation phase (weights stored negated from k-block KNEG on) at every split point class
        KNEG = (-1, (2 * (K // 32) + 1) // 3, 0, 1)[epi] if K > 32 else (-1, 0, 0, -1)[epi]
        B3 = _split_bf3(lib, Bt, stream, KNEG if KNEG >= 0 else 1 << 30)
        Cm = torch.full((M, N), float("nan"), device=dev)
        D = torch.rand(M, N, generator=gen).to(dev) if epi == 3 else torch.full((M, N), float("nan"), device=dev)
        D0 = D.clone()
        rc = lib.aimnet_debug_gemm_bf3(cfg, epi, A.data_ptr(), K, B3.data_ptr(), 3 * K, M, N, K, bias.data_ptr(), Cm.data_ptr(),
                                       D.data_ptr(), N, KNEG, stream)
        assert rc == 0, _lib.last_error()
        torch.cuda.synchronize()
        if epi == 0:
            want = z
        elif epi == 1:
            want = z + bias.double()
        elif epi == 2:
            zb = z + bias.double()
            cdf = 0.5 * (1 + torch.erf(zb / 2**0.5))
            want = zb * cdf
            want_d = cdf + zb * torch.exp(-0.5 * zb * zb) / (2 * torch.pi) ** 0.5
            assert (D.double() - want_d).abs().max().item() < 5e-5
        else:
            want = z * D0.double()
        assert (Cm.double() - want).abs().max().item() < 5e-5, (cfg, epi)


def test_neighbor_list_large_nonperiodic_uses_bounding_box_cells():
    """>= 1500 atoms per non-periodic molecule switches the builder from the O(n^2) scan to a cell list over the
    molecule's bounding box (nlist.hip bbox_setup_kernel): same neighbour SETS as a brute-force numpy search."""
    c, z, cell = workloads.glucose_supercell((3, 2, 3))   # 1728 atoms, coordinates both sides of the origin after the shift
    c = (c - c.mean(0)).astype(np.float32)
    dev = torch.device("cuda:0")
    nb, num, _, _, status = E.neighbor_list(torch.from_numpy(c).to(dev), 5.0, max_nb=128)
    nb, num = nb.cpu().numpy(), num.cpu().numpy()
    assert int(status[1]) == 0
    d = np.linalg.norm(c[:, None, :] - c[None, :, :], axis=-1)
    np.fill_diagonal(d, 99.0)
    for i in range(0, len(c), 37):
        want = set(np.nonzero(d[i] < 5.0)[0].tolist())
        assert set(nb[i, : num[i]].tolist()) == want
    assert num.max() < 128 and (num == (d < 5.0).sum(1)).all()


@pytest.mark.parametrize("cfg", [5, 7, 351])
def test_gelu_epilogue_accuracy(cfg):
    """The fused GELU / GELU' epilogue (common.h erf_fast: branch-free polynomial erf) against the fp64 formula of
    torch.nn.GELU on exact pre-activations z in [-9, 9] (K = 32 product with a single non-zero term):
    |h - gelu(z)| <= 2e-7 (1 + |z|), |d - gelu'(z)| <= 4e-7."""
    from aimnetcentral_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda:0")
    M, N, K = 8192, 16, 32
    z = torch.linspace(-9.0, 9.0, M, dtype=torch.float32)
    z[::97] = torch.tensor([0.0, -0.0, 0.921875 * 2**0.5, -0.921875 * 2**0.5, 1e-8, -1e-8, 5.7, -5.7])[torch.arange(len(z[::97])) % 8]
    A = torch.zeros(M, K)
    A[:, 0] = z
    Bt = torch.zeros(N, K)
    Bt[:, 0] = 1.0
    A, Bt, bias = A.to(dev), Bt.to(dev), torch.zeros(N, device=dev)
    Cm, D = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    rc = lib.aimnet_debug_gemm(cfg, 2, A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), Cm.data_ptr(), D.data_ptr(), N,
                               torch.cuda.current_stream(dev).cuda_stream)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    zd = z.double()
    cdf = 0.5 * (1 + torch.erf(zd / 2**0.5))
    h, d = zd * cdf, cdf + zd * torch.exp(-0.5 * zd * zd) / (2 * torch.pi) ** 0.5
    eh = (Cm[:, 3].cpu().double() - h).abs() / (1 + zd.abs())
    ed = (D[:, 5].cpu().double() - d).abs()
    assert eh.max().item() <= 2e-7, (eh.max().item(), z[eh.argmax()].item())
    assert ed.max().item() <= 4e-7, (ed.max().item(), z[ed.argmax()].item())


def test_mfma_4x4x1_16b_lane_layout():
    """The conv kernels of csrc/conv_mfma.hip assume v_mfma_f32_4x4x1_16B_f32 computes, for the 16 blocks b = lane >> 2,
    D_b[r][col] = A_b[r] * B_b[col] with A at lane 4b + r, B at lane 4b + col and D in VGPR r, lane 4b + col."""
    import ctypes as C

    from aimnetcentral_amd import _lib

    lib = _lib.load()
    out = torch.zeros(64, 4, 64, device="cuda:0")
    _lib.check(lib.aimnet_debug_mfma4_probe(C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "probe")
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    exp = np.zeros((64, 4, 64), dtype=np.float32)
    for lb in range(64):  # B = one-hot(lb): block lb >> 2, col lb & 3;  A[l] = l + 1
        b, col = lb >> 2, lb & 3
        for r in range(4):
            exp[lb, r, 4 * b + col] = 4 * b + r + 1
    assert np.array_equal(got, exp)


def _symmetric_op_inputs(seed=5, frames=12):
    """The op's inputs as the model calls it (aev.py:163-165): a real, symmetric neighbour matrix of a molecule batch with
    the padding row appended, sentinel = B - 1, rows packed real-first, g = the AEV basis (any values do)."""
    c, z, mol, q = workloads.random_batch(frames, 20, 45, seed=seed)
    dev = torch.device("cuda:0")
    nb, num, sh, xw, (mx, ovf) = E.neighbor_list(torch.from_numpy(c).to(dev), 5.0, torch.from_numpy(mol).to(dev), max_nb=64)
    assert ovf == 0
    n = len(z)
    M = int(mx)
    idx = torch.full((n + 1, M), n, dtype=torch.int32, device=dev)
    idx[:n] = nb[:, :M]
    gen = torch.Generator().manual_seed(seed)
    a = torch.randn(n + 1, 16, 16, generator=gen)
    a[-1] = 0
    g = torch.randn(n + 1, M, 16, 4, generator=gen)
    go = torch.randn(n + 1, 16, 16, 4, generator=gen)
    go[-1] = 0
    return a, idx.cpu().long(), g, go


def test_conv_sv_ops_on_a_symmetric_list_are_deterministic_and_exact():
    """On the lists the model produces (symmetric, no repeated neighbour) grad_a is formed by a GATHER with plain stores - the
    atomic remainder pass finds nothing to add - so two calls are bitwise identical (the Warp kernel's atomic scatter is not,
    conv_sv_2d_sp_wp.py:115-136), and everything equals the einsum reference."""
    a, idx, g, go = _symmetric_op_inputs()
    dev = torch.device("cuda:0")
    ad, idd, gd, god = a.to(dev), idx.to(dev), g.to(dev), go.to(dev)
    out = E.conv_sv_2d_sp_fwd(ad, idd, gd).cpu()
    torch.testing.assert_close(out, reference_conv_sv_2d_sp_einsum(a, idx, g), atol=1e-5, rtol=1e-4)
    a_r, g_r = a.clone().requires_grad_(True), g.clone().requires_grad_(True)
    reference_conv_sv_2d_sp_einsum(a_r, idx, g_r).backward(go)
    ga1, gg1 = E.conv_sv_2d_sp_bwd(god, ad, idd, gd)
    ga2, gg2 = E.conv_sv_2d_sp_bwd(god, ad, idd, gd)
    assert torch.equal(ga1, ga2) and torch.equal(gg1, gg2)
    torch.testing.assert_close(ga1.cpu(), a_r.grad, atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(gg1.cpu(), g_r.grad, atol=1e-4, rtol=1e-3)
