"""Host logic of the spatial domain decomposition (aimnetcentral_amd/dd.py): the slab partitioner against brute-force periodic
images, and the engine's exchange function (all-reduce of the NSE sums, owner values for halo charges) over gloo with two CPU
ranks and a stand-in for the engine's workspace.  No GPU."""
from __future__ import annotations

import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from aimnetcentral_amd import dd, workloads


def _cell_and_atoms(seed: int, n: int = 60):
    rng = np.random.default_rng(seed)
    cell = np.array([[9.0, 0.0, 0.0], [2.5, 11.0, 0.0], [-1.5, 3.0, 14.0]]) + rng.normal(0.0, 0.2, (3, 3))
    frac = rng.uniform(-1.5, 2.5, (n, 3))  # atoms up to two cells outside the box
    return frac @ cell, cell


def _periodic_neighbour_distances(x, cell, i, h):
    """Sorted distances < h from atom i to every periodic image of every atom (excluding itself at shift 0)."""
    reach = int(np.ceil(h / dd.perpendicular_widths(cell).min())) + 1
    x = dd.wrapped_fractional(x, cell) @ cell  # (the input atoms sit up to two cells outside the box)
    out = []
    rng_ = range(-reach, reach + 1)
    for a in rng_:
        for b in rng_:
            for c in rng_:
                d = np.linalg.norm(x + np.array([a, b, c]) @ cell - x[i], axis=1)
                if a == 0 and b == 0 and c == 0:
                    d = np.delete(d, i)
                out.append(d[d < h])
    return np.sort(np.concatenate(out))


def test_partition_covers_every_owned_neighbourhood():
    h = 6.0
    for seed, world, grid in ((0, 2, None), (1, 3, None), (2, 1, None), (3, 4, (2, 1, 2)), (4, 8, (2, 2, 2))):
        x, cell = _cell_and_atoms(seed)
        own, g = dd.owners(x, cell, world, grid=grid)
        assert g[0] * g[1] * g[2] == world and (grid is None or g == grid)
        seen = np.zeros(len(x), dtype=int)
        for rank in range(world):
            dom = dd.slab_partition(x, cell, world, rank, h, grid=grid)
            assert dom.axis == int(np.argmax(g)) and dom.n_owned == int((own == rank).sum())
            assert np.array_equal(np.sort(dom.gid[: dom.n_owned]), np.nonzero(own == rank)[0])
            assert np.all(dom.shift[: dom.n_owned] == 0) and dom.owned_mask.sum() == dom.n_owned
            seen[dom.gid[: dom.n_owned]] += 1
            # positions are periodic images of the input atoms
            f_in = x @ np.linalg.inv(cell)
            f_loc = dom.coord @ np.linalg.inv(cell)
            delta = f_loc - f_in[dom.gid]
            assert np.abs(delta - np.round(delta)).max() < 1e-9
            # no (atom, image) twice
            key = np.concatenate([dom.gid[:, None], np.round(delta).astype(int)], axis=1)
            assert len(np.unique(key, axis=0)) == dom.n_local
            # every owned atom sees inside the cluster exactly what it sees in the periodic system
            for k in range(0, dom.n_owned, 3):
                d = np.linalg.norm(dom.coord - dom.coord[k], axis=1)
                d = np.sort(np.delete(d, k))
                ref = _periodic_neighbour_distances(x, cell, int(dom.gid[k]), h)
                got = d[d < h]
                assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-9
        assert np.all(seen == 1)  # every atom is owned by exactly one rank


def test_torch_partitioner_equals_the_numpy_one():
    """`slab_partition_device` (what a caller with positions on the GPU gets) on CPU tensors: the same atoms, images, order and
    positions as the numpy partitioner."""
    for seed, world, grid in ((5, 2, None), (6, 3, None), (7, 4, (2, 1, 2)), (8, 8, (2, 2, 2))):
        x, cell = _cell_and_atoms(seed, 80)
        for rank in range(world):
            a = dd.slab_partition(x, cell, world, rank, 6.0, grid=grid)
            b = dd.slab_partition_device(torch.as_tensor(x), cell, world, rank, 6.0, grid=grid)
            assert a.n_owned == b.n_owned and a.n_local == b.n_local and a.axis == b.axis
            assert np.array_equal(a.gid, b.gid.numpy()) and np.array_equal(a.shift, b.shift.numpy())
            assert np.abs(a.coord - b.coord.numpy()).max() < 1e-12


def test_an_empty_rank_is_refused_on_every_rank():
    """All atoms in one half of the cell: the rank that would own nothing is named by EVERY rank's partition call (a rank that
    stopped alone would leave the others in a collective)."""
    import pytest

    cell = np.diag([10.0, 10.0, 20.0])
    x = np.random.default_rng(0).uniform(0.0, 1.0, (30, 3)) * np.array([10.0, 10.0, 9.0])
    for rank in (0, 1):
        with pytest.raises(ValueError, match=r"rank\(s\) \[1\]"):
            dd.slab_partition(x, cell, 2, rank, 5.0)
        with pytest.raises(ValueError, match=r"rank\(s\) \[1\]"):
            dd.slab_partition_device(torch.as_tensor(x), cell, 2, rank, 5.0)


def test_widths_and_halo_fraction():
    _, _, cell = workloads.glucose_supercell((7, 3, 5))
    w = dd.perpendicular_widths(cell)
    assert abs(w[1] - 37.6872) < 1e-3 and int(np.argmax(w)) == 2 and w[0] < np.linalg.norm(cell[0]) + 1e-9
    assert dd.halo_fraction(cell, 2, 15.0) > dd.halo_fraction(cell * 10.0, 2, 15.0) > 1.0
    cube = np.eye(3) * 200.0
    assert dd.brick_grid(cube, 8) == (2, 2, 2) and dd.halo_fraction(cube, 8, 15.0, (2, 2, 2)) < dd.halo_fraction(cube, 8, 15.0)
    assert dd.brick_grid(cell, 2) == (1, 1, 2)  # two ranks: slabs along the widest axis
    f = dd.wrapped_fractional(np.array([[-1e-18, 0.0, 0.0]]), np.eye(3))
    assert (f >= 0).all() and (f < 1).all()


class _StubEngine:
    nq = 1
    _ws = None


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, cell = _cell_and_atoms(3, 40)
        dom = dd.slab_partition(x, cell, world, rank, 5.0)
        eng = _StubEngine()
        eng._ws = torch.zeros(64 + 4 * dom.n_local + 64, dtype=torch.uint8)  # stand-in for the engine's workspace
        dde = dd.DomainDecomposedEngine(eng)
        ex = dd._Exchange(dde, dom, torch.as_tensor(dom.gid), len(x))
        base = eng._ws.data_ptr()
        # AIMNET_DD_SUM: two floats summed over the ranks, in place
        sums = eng._ws[16:24].view(torch.float32)
        sums[:] = torch.tensor([1.0 + rank, 10.0 * (rank + 1)])
        assert ex.cb(None, dd.DD_SUM, base + 16, 2, None) == 0
        # AIMNET_DD_CHARGES: the charge of atom g is 0.5 + g on its owner, garbage on halo copies
        q = eng._ws[64 : 64 + 4 * dom.n_local].view(torch.float32)
        q[: dom.n_owned] = torch.as_tensor(0.5 + dom.gid[: dom.n_owned], dtype=torch.float32)
        q[dom.n_owned :] = -777.0
        assert ex.cb(None, dd.DD_CHARGES, base + 64, dom.n_local, None) == 0
        # a pointer outside the workspace is refused (returns non-zero, the engine aborts the evaluation)
        bad = ex.cb(None, dd.DD_SUM, base - 4096, 2, None)
        ret[rank] = (sums.clone().numpy(), q.clone().numpy(), dom.gid.copy(), bad, dict(ex.calls))
    finally:
        dist.destroy_process_group()


def test_exchange_function_over_gloo_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [mp.get_context("spawn").Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    for r in range(world):
        sums, q, gid, bad, calls = ret[r]
        assert np.array_equal(sums, np.array([3.0, 30.0], dtype=np.float32))
        assert np.array_equal(q, (0.5 + gid).astype(np.float32))  # halo copies carry their owners' values
        assert bad != 0 and calls[dd.DD_SUM] == 1 and calls[dd.DD_CHARGES] == 1
