"""world_size-2 gloo test of the batch-sharding helpers (the N>1 path of bench.py); CPU only."""
from __future__ import annotations

import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from aimnetcentral_amd import dist as adist
from aimnetcentral_amd import workloads


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, sizes, ret):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ranges = adist.shard_frames(sizes, world)
        lo, hi = ranges[rank]
        frames = [b - a for a, b in ranges]
        e_local = torch.arange(lo, hi, dtype=torch.float64) * 1.5  # energy of frame f is 1.5 f
        e_all = adist.all_gather_energies(e_local, frames)
        atoms = [int(np.sum(sizes[a:b])) for a, b in ranges]
        start = int(np.sum(sizes[:lo]))
        f_local = (torch.arange(atoms[rank], dtype=torch.float32) + start).unsqueeze(-1).expand(-1, 3).contiguous()
        f_all = adist.all_gather_atoms(f_local, atoms)
        ret[rank] = (e_all.numpy(), f_all.numpy())
    finally:
        dist.destroy_process_group()


def test_shard_and_allgather_world2():
    sizes = np.array([5, 9, 3, 7, 8, 2, 6], dtype=np.int64)
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [mp.get_context("spawn").Process(target=_worker, args=(r, world, port, sizes, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(world):
        e_all, f_all = ret[r]
        assert np.array_equal(e_all, np.arange(len(sizes)) * 1.5)
        assert np.array_equal(f_all[:, 0], np.arange(sizes.sum(), dtype=np.float32))


def test_shard_frames_properties():
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        sizes = rng.integers(20, 61, size=256)
        ranges = adist.shard_frames(sizes, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == 256
        assert all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
        loads = [sizes[a:b].sum() for a, b in ranges]
        assert max(loads) - min(loads) <= 2 * sizes.max()
    assert adist.shard_frames([10, 10], 4) == [(0, 0), (0, 1), (1, 1), (1, 2)] or len(adist.shard_frames([10, 10], 4)) == 4


def test_local_batch_rebases_mol_idx():
    c, z, mol, q = workloads.random_batch(6, 5, 9, seed=3)
    cl, zl, ml, ql = adist.local_batch(c, z, mol, q, 2, 5)
    assert ml.min() == 0 and ml.max() == 2 and len(ql) == 3
    assert np.array_equal(cl, c[(mol >= 2) & (mol < 5)])
