"""Spatial domain decomposition of ONE periodic system over the GPUs of a node.

The reference has no counterpart: it evaluates a system on one device and points users with several GPUs at independent
processes (docs/tutorials/performance.md:275-285).  What the model's structure implies (aimnet/models/aimnet2.py:135-206:
three message-passing passes of radius rc, the NSE charge normalisation aimnet/ops.py:99-145 per molecule, the external
Coulomb term aimnet/modules/lr.py:559-615 over 15 A):

* a rank OWNS the atoms of one slab of the cell (fractional coordinate along the widest axis) and holds HALO copies - periodic
  images resolved here, on the host - of every atom within `halo` = max(3 rc, Coulomb cutoff) of the slab.  Owned + halo atoms
  go to the engine as ONE NON-PERIODIC cluster (`slab_partition`);
* the energy of an owned atom is exact inside that cluster except for two things that are global: the NSE sums (two scalars per
  molecule, charge channel and pass: all-reduced through the engine's exchange function, include/aimnet_hip.h
  aimnet_engine_set_dd) and the final charges of halo copies further than one cutoff from the slab (taken from their owners in
  front of the Coulomb block, same function);
* the backward sweep differentiates E_rank = sum of the owned atoms' energies with respect to EVERY local position (the adjoint
  sums of the NSE steps are all-reduced, mirror image of the forward); the direct Coulomb force and dE/dq are formed at the owned
  centre for both directions of a pair, so no charge adjoint crosses ranks; at the end the partial forces of halo copies are
  added onto their owners (one reverse halo exchange) and the rank energies (and, for the stress, virial shares) are summed;
* external DFT-D3: the reference weights and dE/dcn of halo copies come from their owners (a coordination number needs the
  atom's own 15 A neighbourhood) - two more exchanges of the same owner-to-halo kind.

The exchanges here are the simple form - global-size arrays all-reduced over the group (40 KB of charges, 120 KB of forces at
10^4 atoms): correct for any number of ranks on `nccl` (= RCCL) and `gloo`; the scalable form (neighbour-to-neighbour halo
messages) changes `_Exchange` only.  No performance claim is attached to this module: at 10^4 atoms one MI355X
evaluates the whole system in 1.2 ms; decomposition is for systems beyond one GPU's memory or time budget.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Any

import numpy as np

from . import _lib
from .engine import NONFINITE_FLAG, _round16, describe_input_flags

DD_SUM, DD_CHARGES, DD_ROWS = 0, 1, 2  # AIMNET_DD_* of include/aimnet_hip.h
EXCHANGE_FN = _lib.DD_EXCHANGE_FN


@dataclass
class SlabDomain:
    """One rank's cluster: the first `n_owned` local atoms are owned (wrapped into the cell), the rest are halo copies."""

    rank: int
    world: int
    axis: int
    n_owned: int
    gid: np.ndarray  # [n_loc] global index of every local atom
    coord: np.ndarray  # [n_loc, 3] Cartesian positions with the periodic images resolved (float64)
    shift: np.ndarray  # [n_loc, 3] integer lattice shift of every local atom relative to its wrapped position

    @property
    def n_local(self) -> int:
        return int(self.gid.shape[0])

    @property
    def owned_mask(self) -> np.ndarray:
        m = np.zeros(self.n_local, dtype=np.float32)
        m[: self.n_owned] = 1.0
        return m


def perpendicular_widths(cell: np.ndarray) -> np.ndarray:
    """Distance between the two faces of the cell (row vectors) that bound fractional coordinate d, for d = 0, 1, 2."""
    cell = np.asarray(cell, dtype=np.float64)
    inv = np.linalg.inv(cell)  # columns = reciprocal vectors (coord = frac @ cell)
    return 1.0 / np.linalg.norm(inv, axis=0)


def wrapped_fractional(coord: np.ndarray, cell: np.ndarray) -> np.ndarray:
    f = np.asarray(coord, dtype=np.float64) @ np.linalg.inv(np.asarray(cell, dtype=np.float64))
    f = f - np.floor(f)
    f[f >= 1.0] = 0.0  # (x - floor(x) rounds to 1.0 for tiny negative x)
    return f


def brick_grid(cell: np.ndarray, world: int, halo: float = 15.0) -> tuple[int, int, int]:
    """Ranks along the three cell axes (g0 g1 g2 = world) that evaluate the fewest local atoms per owned atom (`halo_fraction`)."""
    w = perpendicular_widths(cell)
    best, best_cost = (1, 1, world), None
    for g0 in range(1, world + 1):
        if world % g0:
            continue
        for g1 in range(1, world // g0 + 1):
            if (world // g0) % g1:
                continue
            g = (g0, g1, world // (g0 * g1))
            cost = float(np.prod([(w[d] / g[d] + 2.0 * halo) / (w[d] / g[d]) for d in range(3)]))
            if best_cost is None or cost < best_cost - 1e-12:
                best, best_cost = g, cost
    return best


def _grid_of(cell, world: int, grid, axis):
    if grid is not None:
        g = tuple(int(x) for x in grid)
        if len(g) != 3 or min(g) < 1 or g[0] * g[1] * g[2] != world:
            raise ValueError(f"domain decomposition: grid {grid} does not multiply to the {world} ranks of the group")
        return g
    if axis is None:
        axis = int(np.argmax(perpendicular_widths(cell)))
    g = [1, 1, 1]
    g[axis] = world
    return tuple(g)


def _check_every_rank_owns(counts, g) -> None:
    """Raised on EVERY rank alike (each rank sees the owner of every atom): a rank that stopped alone would leave the others waiting
    in the first collective of the evaluation."""
    empty = [int(r) for r in np.nonzero(np.asarray(counts) == 0)[0]]
    if empty:
        raise ValueError(f"domain decomposition: rank(s) {empty} of the {g[0]} x {g[1]} x {g[2]} rank grid own no atom - use fewer "
                         "ranks or another grid for this system")


def owners(coord: np.ndarray, cell: np.ndarray, world: int, axis: int | None = None, grid=None) -> tuple[np.ndarray, tuple[int, int, int]]:
    """Owner rank of every atom and the rank grid.  Default: `world` equal-width slabs of the fractional coordinate along `axis`
    (the widest axis); `grid = (g0, g1, g2)`: bricks, rank = (i0 g1 + i1) g2 + i2."""
    g = _grid_of(cell, world, grid, axis)
    f = wrapped_fractional(coord, cell)
    idx = [np.minimum((f[:, d] * g[d]).astype(np.int64), g[d] - 1) for d in range(3)]
    return (idx[0] * g[1] + idx[1]) * g[2] + idx[2], g


def slab_partition(coord: np.ndarray, cell: np.ndarray, world: int, rank: int, halo: float, axis: int | None = None,
                   grid=None) -> SlabDomain:
    """Owned atoms of slab (or brick, `grid`) `rank` plus every periodic image of every atom inside the region's box padded by `halo`
    along all three cell axes (a superset of the points within `halo` of the region: a point at Euclidean distance <= h from it is
    at most h / w_d outside it in fractional coordinate d, w_d the perpendicular width)."""
    cell = np.asarray(cell, dtype=np.float64)
    if world < 1 or not (0 <= rank < world):
        raise ValueError("slab_partition: need 0 <= rank < world")
    own, g = owners(coord, cell, world, axis, grid)
    _check_every_rank_owns(np.bincount(own, minlength=world), g)
    f = wrapped_fractional(coord, cell)
    w = perpendicular_widths(cell)
    ri = (rank // (g[1] * g[2]), (rank // g[2]) % g[1], rank % g[2])
    lo = np.array([ri[d] / g[d] for d in range(3)])
    hi = np.array([(ri[d] + 1) / g[d] for d in range(3)])
    axis = int(np.argmax(g))
    pad = halo / w
    n_lo = np.floor(lo - pad).astype(int)
    n_hi = np.floor(hi + pad).astype(int)
    idx_owned = np.nonzero(own == rank)[0]
    gids, shifts = [idx_owned], [np.zeros((idx_owned.shape[0], 3), dtype=np.int64)]
    for nx in range(n_lo[0], n_hi[0] + 1):
        for ny in range(n_lo[1], n_hi[1] + 1):
            for nz in range(n_lo[2], n_hi[2] + 1):
                n = np.array([nx, ny, nz])
                gg = f + n
                inside = np.all((gg >= lo - pad) & (gg < hi + pad), axis=1)
                if nx == 0 and ny == 0 and nz == 0:
                    inside &= own != rank  # (the owned atoms themselves)
                k = np.nonzero(inside)[0]
                if k.size:
                    gids.append(k)
                    shifts.append(np.broadcast_to(n, (k.size, 3)).astype(np.int64))
    gid = np.concatenate(gids)
    shift = np.concatenate(shifts)
    pos = (f[gid] + shift) @ cell
    return SlabDomain(rank=rank, world=world, axis=axis, n_owned=int(idx_owned.shape[0]), gid=gid, coord=pos, shift=shift)


def slab_partition_device(coord, cell, world: int, rank: int, halo: float, axis: int | None = None, grid=None):
    """`slab_partition` in torch on the device that holds `coord` (float64 arithmetic, same region, same image set): for callers whose
    positions live on the GPU - a 10^6-atom system costs the numpy form a second of host time per call.  Returns a `SlabDomain` whose
    `gid` / `coord` / `shift` are torch tensors on that device (owned atoms first, in global order; the halo order is the shift loop's)."""
    import torch

    cel = np.asarray(cell.detach().cpu() if hasattr(cell, "detach") else cell, dtype=np.float64)
    if world < 1 or not (0 <= rank < world):
        raise ValueError("slab_partition: need 0 <= rank < world")
    g = _grid_of(cel, world, grid, axis)
    w = perpendicular_widths(cel)
    dev = coord.device
    cell_t = torch.as_tensor(cel, dtype=torch.float64, device=dev)
    f = coord.to(torch.float64) @ torch.linalg.inv(cell_t)
    f = f - torch.floor(f)
    f = torch.where(f >= 1.0, torch.zeros_like(f), f)
    g_t = torch.tensor(g, dtype=torch.float64, device=dev)
    idx = torch.minimum((f * g_t).to(torch.int64), (g_t - 1).to(torch.int64))
    own = (idx[:, 0] * g[1] + idx[:, 1]) * g[2] + idx[:, 2]
    _check_every_rank_owns(torch.bincount(own, minlength=world).cpu().numpy(), g)
    ri = (rank // (g[1] * g[2]), (rank // g[2]) % g[1], rank % g[2])
    lo = np.array([ri[d] / g[d] for d in range(3)])
    hi = np.array([(ri[d] + 1) / g[d] for d in range(3)])
    pad = halo / w
    n_lo, n_hi = np.floor(lo - pad).astype(int), np.floor(hi + pad).astype(int)
    lo_t = torch.as_tensor(lo - pad, dtype=torch.float64, device=dev)
    hi_t = torch.as_tensor(hi + pad, dtype=torch.float64, device=dev)
    mine = own == rank
    idx_owned = torch.nonzero(mine).reshape(-1)
    gids, shifts = [idx_owned], [torch.zeros((idx_owned.shape[0], 3), dtype=torch.int64, device=dev)]
    for nx in range(n_lo[0], n_hi[0] + 1):
        for ny in range(n_lo[1], n_hi[1] + 1):
            for nz in range(n_lo[2], n_hi[2] + 1):
                n = torch.tensor([nx, ny, nz], dtype=torch.float64, device=dev)
                gg = f + n
                inside = ((gg >= lo_t) & (gg < hi_t)).all(dim=1)
                if nx == 0 and ny == 0 and nz == 0:
                    inside = inside & ~mine
                k = torch.nonzero(inside).reshape(-1)
                gids.append(k)
                shifts.append(n.to(torch.int64).expand(k.shape[0], 3))
    gid = torch.cat(gids)
    shift = torch.cat(shifts)
    pos = (f[gid] + shift.to(torch.float64)) @ cell_t
    return SlabDomain(rank=rank, world=world, axis=int(np.argmax(g)), n_owned=int(idx_owned.shape[0]), gid=gid, coord=pos, shift=shift)


class _Exchange:
    """The engine's exchange function for one evaluation: all-reduce of the NSE sums, owner values for the halo charges."""

    def __init__(self, dde: "DomainDecomposedEngine", dom: SlabDomain, gid_t, n_global: int):
        self.dde, self.dom, self.gid, self.n_global = dde, dom, gid_t, n_global
        self.error: BaseException | None = None
        self.calls = {DD_SUM: 0, DD_CHARGES: 0, DD_ROWS: 0}
        self.cb = EXCHANGE_FN(self._call)

    def _view(self, ptr: int, n: int):
        import torch

        ws = self.dde.engine._ws
        off = ptr - ws.data_ptr()
        if off < 0 or off + 4 * n > ws.numel():
            raise RuntimeError("domain decomposition: the engine handed out a pointer outside its workspace")
        return ws[off : off + 4 * n].view(torch.float32)

    def _call(self, _ctx, what, ptr, n, _stream) -> int:
        import torch

        try:
            t = self._view(int(ptr), int(n))
            self.calls[int(what)] += 1
            if what == DD_SUM:
                self.dde.all_reduce_(t)
            elif what == DD_ROWS:  # per-atom records (DFT-D3 weights, dE/dcn): halo rows take their owners' rows
                rows = t.view(self.dom.n_local, -1)
                glob = torch.zeros((self.n_global, rows.shape[1]), dtype=torch.float32, device=t.device)
                glob[self.gid[: self.dom.n_owned]] = rows[: self.dom.n_owned]
                self.dde.all_reduce_(glob)  # exactly one non-zero contribution per row: exact
                rows[self.dom.n_owned :] = glob[self.gid[self.dom.n_owned :]]
            else:
                nq = self.dde.engine.nq
                planes = t.view(nq, self.dom.n_local)
                glob = torch.zeros((nq, self.n_global), dtype=torch.float32, device=t.device)
                glob[:, self.gid[: self.dom.n_owned]] = planes[:, : self.dom.n_owned]
                self.dde.all_reduce_(glob)  # every entry has exactly one non-zero contribution: exact
                planes[:, self.dom.n_owned :] = glob[:, self.gid[self.dom.n_owned :]]
            return 0
        except BaseException as exc:  # noqa: BLE001 - must not propagate through the C frame
            self.error = exc
            return 1


class DomainDecomposedEngine:
    """`HipEngine.eval` for one periodic system cut into slabs over the ranks of a torch.distributed group.

    Every rank calls `eval` with the SAME full system (coord [N,3], numbers [N], cell [3,3]); every rank gets the same result:
    energy (float64 scalar tensor), forces [N,3], charges [N] in the caller's atom order."""

    def __init__(self, engine, group=None):
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("DomainDecomposedEngine needs an initialised torch.distributed process group")
        self.engine, self.group = engine, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.last_domain: SlabDomain | None = None
        self.last_calls: dict[int, int] | None = None

    def all_reduce_(self, t, op=None):
        """In-place sum (or `op`) over the group, ordered on torch's current stream.  `nccl` (= RCCL) reduces device memory
        directly; `gloo` (CPU tests, two ranks sharing one GPU) goes through the host."""
        import torch.distributed as dist

        op = dist.ReduceOp.SUM if op is None else op
        if self.backend == "nccl" or not t.is_cuda:
            dist.all_reduce(t, op=op, group=self.group)
        else:
            h = t.cpu()  # synchronises the current stream: everything the engine has enqueued so far is done
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
        return t

    def eval(self, coord, numbers, cell, charge: float = 0.0, forces: bool = True, stress: bool = False, coulomb: str = "dsf",
             dsf_rc: float = 15.0, dsf_alpha: float = 0.2, dftd3: dict[str, float] | None = None, halo: float | None = None,
             axis: int | None = None, grid=None) -> dict[str, Any]:
        import torch

        eng = self.engine
        if coulomb not in ("none", "dsf"):
            raise ValueError("DomainDecomposedEngine: coulomb must be 'none' or 'dsf' (Ewald / PME reciprocal sums and the all-pairs "
                             "'simple' form are not decomposed)")
        cel = np.asarray(cell.detach().cpu() if hasattr(cell, "detach") else cell, dtype=np.float64)
        dev = eng.device
        n_global = int(coord.shape[0])
        if halo is None:
            halo = max(3.0 * float(eng.spec.rc), float(dsf_rc) if coulomb == "dsf" else 0.0,
                       float(dftd3.get("cutoff", 15.0)) if dftd3 is not None else 0.0) + 1e-3
        on_device = hasattr(coord, "is_cuda") and coord.is_cuda
        if on_device:  # positions already on the GPU (an MD driver's state): partition there, nothing crosses to the host
            dom = slab_partition_device(coord.to(dev), cel, self.world, self.rank, float(halo), axis, grid)
            gid = dom.gid
            xc = dom.coord
            z_loc = torch.as_tensor(numbers, device=dev).to(torch.int32)[gid]
        else:
            xyz = np.asarray(coord.detach().cpu() if hasattr(coord, "detach") else coord, dtype=np.float64)
            z = np.asarray(numbers.detach().cpu() if hasattr(numbers, "detach") else numbers).astype(np.int64)
            dom = slab_partition(xyz, cel, self.world, self.rank, float(halo), axis, grid)
            gid = torch.as_tensor(dom.gid, device=dev)
            xc = torch.as_tensor(dom.coord, device=dev)
            z_loc = torch.as_tensor(z[dom.gid], dtype=torch.int32, device=dev)
        self.last_domain = dom
        # (the cluster is centred on its owned atoms before it is rounded to fp32: the engine's results are translation invariant,
        # and an image position of magnitude 50 A carries twice the rounding of one at 25 A)
        x_loc = (xc - xc[: dom.n_owned].mean(dim=0)).to(torch.float32)
        mol = torch.zeros(dom.n_local, dtype=torch.int32, device=dev)
        if eng.nq == 2:
            q_in = torch.as_tensor(np.asarray(charge, dtype=np.float32).reshape(1, 2), device=dev)
        else:
            q_in = torch.tensor([float(charge)], dtype=torch.float32, device=dev)
        owned = torch.zeros(dom.n_local, dtype=torch.float32, device=dev)
        owned[: dom.n_owned] = 1.0
        ex = _Exchange(self, dom, gid, n_global)
        try:
            while True:
                _lib.check(eng.lib.aimnet_engine_set_dd(eng._h, owned.data_ptr(), ex.cb, None), "aimnet_engine_set_dd")
                try:
                    # (stress: the engine returns the rank's share of dE/d(strain), undivided - the cluster has no cell)
                    res = eng.eval(x_loc, z_loc, mol, q_in, cell=None, forces=forces, stress=stress, coulomb=coulomb, dsf_rc=dsf_rc,
                                   dsf_alpha=dsf_alpha, dftd3=dftd3, sync=False)
                except _lib.HipLibraryError:
                    if ex.error is not None:
                        raise ex.error
                    raise
                # the row capacities must grow on EVERY rank together (a rank that repeated its evaluation alone would leave the
                # others waiting in a collective): status words max-reduced over the group, then the engine's own growth rule
                import torch.distributed as dist

                st = self.all_reduce_(res["status"].clone(), op=dist.ReduceOp.MAX).cpu().numpy()
                flags = int(st[6]) & ~NONFINITE_FLAG
                if flags:
                    raise ValueError("DomainDecomposedEngine: invalid input: " + describe_input_flags(flags, 1))
                retry = False
                if st[2]:
                    eng.max_nb = _round16(int(max(eng.max_nb * 1.5, st[0])))
                    retry = True
                if st[3]:
                    eng._max_nb_lr[float(dsf_rc)] = _round16(int(max(eng._max_nb_lr[float(dsf_rc)] * 1.5, st[1])))
                    retry = True
                if st[5]:  # DFT-D3 matrix (it may live in the long-range buffers: grow both, as HipEngine.eval does)
                    d3_rc = float(dftd3.get("cutoff", 15.0))
                    eng._max_nb_lr[d3_rc] = _round16(int(max(eng._lr_capacity(d3_rc) * 1.5, st[4], st[1])))
                    retry = True
                if not retry:
                    break
        finally:
            _lib.check(eng.lib.aimnet_engine_set_dd(eng._h, None, EXCHANGE_FN(), None), "aimnet_engine_set_dd")
        self.last_calls = dict(ex.calls)
        # one all-reduce for the outputs: [energy, forces of owned AND halo copies added onto their global rows, owned charges];
        # float64 so that the sum over a row's copies does not depend on the order the device adds them in
        nf = 3 * n_global if forces else 0
        buf = torch.zeros(1 + nf + n_global + (9 if stress else 0), dtype=torch.float64, device=dev)
        buf[0] = res["energy"][0]
        if forces:
            buf[1 : 1 + nf].view(n_global, 3).index_add_(0, gid, res["forces"].double())
        buf[1 + nf : 1 + nf + n_global].index_add_(0, gid[: dom.n_owned], res["charges"][: dom.n_owned].double())
        if stress:
            buf[1 + nf + n_global :] = res["stress"].reshape(9).double()
        self.all_reduce_(buf)
        if not bool(torch.isfinite(buf).all()):
            raise FloatingPointError("DomainDecomposedEngine: non-finite energy / forces (for an fp16-range overflow of the GEMM "
                                     "operands set engine.set_option('gemm_h2', 0) on every rank and repeat)")
        out = {"energy": buf[0].clone(), "charges": buf[1 + nf : 1 + nf + n_global].float()}
        if stress:  # sum of the ranks' virial shares over the cell volume (model.hip, stress_partial_kernel: sum / |det cell|)
            out["stress"] = (buf[1 + nf + n_global :] / abs(float(np.linalg.det(cel)))).view(3, 3).float()
        if forces:
            out["forces"] = buf[1 : 1 + nf].view(n_global, 3).float()
        return out


def halo_fraction(cell: np.ndarray, world: int, halo: float, grid=None) -> float:
    """Local atoms per owned atom for a homogeneous system (the cost model of DESIGN.md 6: form (a) pays this factor)."""
    w = perpendicular_widths(cell)
    g = _grid_of(cell, world, grid, None)
    return float(np.prod([(w[d] / g[d] + 2.0 * halo) / (w[d] / g[d]) for d in range(3)]))


__all__ = ["DomainDecomposedEngine", "SlabDomain", "slab_partition", "slab_partition_device", "owners", "brick_grid", "perpendicular_widths", "wrapped_fractional",
           "halo_fraction"]
