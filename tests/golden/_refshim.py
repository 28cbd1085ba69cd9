"""Import shim that lets the UNMODIFIED reference package (/root/reference/aimnet) run on
CPU in the build container.  Used ONLY by tests/golden/make_golden.py to produce the committed
golden vectors; nothing in the product, the `-m gpu` tests, smoke() or bench.py imports it
(/root/reference does not exist on the GPU box).

Three shims:
  1. typing.Self / typing.NotRequired back-ports (reference needs python >= 3.11).
  2. a stub `warp` module: the reference registers its Warp kernels at import time but never
     executes them on CPU (aimnet/modules/aev.py:163-178 falls through to einsum).
  3. a stub `nvalchemiops` package with a brute-force `neighbor_list` that honours the
     reference call contract (aimnet/calculators/neighbors.py:106-125): full (both-direction)
     lists, rows packed real-first, fill_value padding, NeighborOverflowError on overflow,
     integer PBC shifts.
"""
from __future__ import annotations

import itertools
import math
import os
import sys
import types
import typing

REFERENCE_ROOT = "/root/reference"


def _stub_warp() -> types.ModuleType:
    wp = types.ModuleType("warp")

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

        def __getitem__(self, item):
            return self

    def kernel(fn=None, **kwargs):
        if fn is None:
            return lambda f: f
        return fn

    wp.kernel = kernel
    for name in ("array", "array1d", "array2d", "array3d", "array4d"):
        setattr(wp, name, lambda *a, **k: _Any())
    for name in ("vec4f", "vec3f", "float32", "int32", "float64", "int64"):
        setattr(wp, name, _Any())
    wp.init = lambda: None
    wp.get_cuda_device_count = lambda: 0
    wp.is_cuda_available = lambda: False
    wp.config = types.SimpleNamespace(quiet=True)
    wp.tid = lambda: 0
    wp.dot = lambda a, b: 0
    wp.atomic_add = lambda *a: None
    wp.launch = lambda *a, **k: None
    wp.from_torch = lambda *a, **k: None
    wp.device_from_torch = lambda *a, **k: None
    wp.stream_from_torch = lambda *a, **k: None
    return wp


class NeighborOverflowError(RuntimeError):
    pass


def _brute_force_neighbor_list(
    positions,
    cutoff,
    cell=None,
    pbc=None,
    batch_idx=None,
    max_neighbors=None,
    half_fill=False,
    fill_value=None,
    method=None,
    **_ignored,
):
    """O(N^2 * images) neighbour matrix in float64 on CPU.  Pair order inside a row is
    ascending (shift, j) - the reference's tests treat rows as sets."""
    import torch

    assert not half_fill
    pos = positions.detach().double().cpu()
    n = pos.shape[0]
    if fill_value is None:
        fill_value = n
    if batch_idx is None:
        bidx = torch.zeros(n, dtype=torch.long)
    else:
        bidx = batch_idx.detach().long().cpu()
    same = bidx[:, None] == bidx[None, :]
    rows: list[list[tuple[int, int, int, int]]] = [[] for _ in range(n)]
    if cell is None:
        d2 = ((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1)
        ok = (d2 < cutoff * cutoff) & same
        ok.fill_diagonal_(False)
        for i in range(n):
            rows[i] = [(int(j), 0, 0, 0) for j in torch.nonzero(ok[i]).flatten().tolist()]
    else:
        cells = cell.detach().double().cpu()
        if cells.ndim == 2:
            cells = cells.unsqueeze(0)
        pbcs = torch.as_tensor(pbc).bool().cpu() if pbc is not None else torch.ones(cells.shape[0], 3, dtype=torch.bool)
        if pbcs.ndim == 1:
            pbcs = pbcs.unsqueeze(0).expand(cells.shape[0], -1)
        for s in range(cells.shape[0]):
            idx = torch.nonzero(bidx == s).flatten()
            if idx.numel() == 0:
                continue
            c = cells[s]
            vol = abs(torch.linalg.det(c).item())
            nimg = []
            for k in range(3):
                a1, a2 = c[(k + 1) % 3], c[(k + 2) % 3]
                h = vol / torch.linalg.norm(torch.linalg.cross(a1, a2)).item()
                nimg.append(int(math.ceil(cutoff / h)) if bool(pbcs[s, k]) else 0)
            p = pos[idx]
            for sx, sy, sz in itertools.product(*[range(-m, m + 1) for m in nimg]):
                off = sx * c[0] + sy * c[1] + sz * c[2]
                d2 = ((p[None, :, :] + off - p[:, None, :]) ** 2).sum(-1)  # [i, j]
                ok = d2 < cutoff * cutoff
                if sx == 0 and sy == 0 and sz == 0:
                    ok.fill_diagonal_(False)
                ii, jj = torch.nonzero(ok, as_tuple=True)
                for i_loc, j_loc in zip(ii.tolist(), jj.tolist()):
                    rows[int(idx[i_loc])].append((int(idx[j_loc]), sx, sy, sz))
    counts = [len(r) for r in rows]
    if max_neighbors is None:
        max_neighbors = max(counts + [1])
    if max(counts + [0]) > max_neighbors:
        raise NeighborOverflowError(f"max_neighbors={max_neighbors} < {max(counts)}")
    dev = positions.device
    nbmat = torch.full((n, max_neighbors), int(fill_value), dtype=torch.int32)
    shifts = torch.zeros((n, max_neighbors, 3), dtype=torch.int32)
    for i, r in enumerate(rows):
        for m, (j, sx, sy, sz) in enumerate(r):
            nbmat[i, m] = j
            shifts[i, m, 0] = sx
            shifts[i, m, 1] = sy
            shifts[i, m, 2] = sz
    num = torch.tensor(counts, dtype=torch.int32)
    if cell is None:
        return nbmat.to(dev), num.to(dev)
    return nbmat.to(dev), num.to(dev), shifts.to(dev)


def _unavailable(name):
    def fn(*a, **k):
        raise RuntimeError(f"nvalchemiops.{name} is not available in the oracle shim")

    return fn


def install() -> None:
    """Install the shims and put the reference on sys.path (idempotent)."""
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    import typing_extensions

    if not hasattr(typing, "Self"):
        typing.Self = typing_extensions.Self
    if not hasattr(typing, "NotRequired"):
        typing.NotRequired = typing_extensions.NotRequired
    if "warp" not in sys.modules:
        sys.modules["warp"] = _stub_warp()
    if "nvalchemiops" not in sys.modules:
        def mod(name):
            m = types.ModuleType(name)
            sys.modules[name] = m
            return m

        root = mod("nvalchemiops")
        nb = mod("nvalchemiops.neighbors")
        nb.NeighborOverflowError = NeighborOverflowError
        t = mod("nvalchemiops.torch")
        tn = mod("nvalchemiops.torch.neighbors")
        tn.neighbor_list = _brute_force_neighbor_list
        ti = mod("nvalchemiops.torch.interactions")
        td = mod("nvalchemiops.torch.interactions.dispersion")
        td.dftd3 = _unavailable("dftd3")
        te = mod("nvalchemiops.torch.interactions.electrostatics")
        for name in (
            "dsf_coulomb",
            "ewald_summation",
            "particle_mesh_ewald",
            "estimate_ewald_parameters",
            "estimate_pme_parameters",
        ):
            setattr(te, name, _unavailable(name))
        root.neighbors, root.torch = nb, t
        t.neighbors, t.interactions = tn, ti
        ti.dispersion, ti.electrostatics = td, te
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
