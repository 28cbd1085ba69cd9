"""Batch sharding of independent frames over the GPUs of one node.

The reference has no multi-GPU inference ("run independent processes",
docs/tutorials/performance.md:275-285).  Molecules/frames are independent - mol_sum and nse are
per molecule (nbops.py:309, ops.py:99) and neighbour lists are per batch_idx (neighbors.py:111) -
so the path shards with NO data-path collective: one process per GPU evaluates a contiguous range
of frames; the only exchange is an all-gather of the per-frame fp64 energies (8 B per frame) over
RCCL/xGMI, optionally of forces.  Works with backend "nccl" (= RCCL) on GPUs and "gloo" on CPU.
"""
from __future__ import annotations

import numpy as np


def shard_frames(atoms_per_frame, world_size: int) -> list[tuple[int, int]]:
    """Contiguous frame ranges [lo, hi) per rank, balanced by atom count (greedy prefix split)."""
    sizes = np.asarray(atoms_per_frame, dtype=np.int64)
    n = len(sizes)
    csum = np.concatenate([[0], np.cumsum(sizes)])
    total = int(csum[-1])
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        k = int(np.searchsorted(csum, target, side="left"))
        k = min(max(k, bounds[-1]), n)
        # pick the closer of k-1 / k
        if k > bounds[-1] and abs(csum[k - 1] - target) <= abs(csum[min(k, n)] - target):
            k -= 1
        bounds.append(max(k, bounds[-1]))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def local_batch(coord, numbers, mol_idx, charge, lo: int, hi: int):
    """Slice a flat batch to frames [lo, hi) and re-base mol_idx to start at 0."""
    mol_idx = np.asarray(mol_idx)
    sel = (mol_idx >= lo) & (mol_idx < hi)
    return np.asarray(coord)[sel], np.asarray(numbers)[sel], mol_idx[sel] - lo, np.asarray(charge)[lo:hi]


_GATHER_BUFFERS: dict = {}  # (device, dtype, world, nmax) -> (send, recv): a per-step collective must not time allocations


def _gather_buffers(device, dtype, world: int, nmax: int):
    import torch

    key = (str(device), dtype, world, nmax)
    if key not in _GATHER_BUFFERS:
        _GATHER_BUFFERS[key] = (torch.zeros(nmax, dtype=dtype, device=device), torch.empty(world * nmax, dtype=dtype, device=device))
    return _GATHER_BUFFERS[key]


def all_gather_energies(energy_local, frames_per_rank: list[int], group=None, reuse_buffer: bool = False):
    """All-gather per-frame energies (fp64) from every rank: returns a tensor [sum(frames_per_rank)]
    in global frame order.  Ragged counts are padded to the maximum (one fixed-size collective).  The padded send buffer
    and the receive buffer are allocated once per (device, world, size) and reused.  The result is a fresh tensor (a copy of a
    few bytes per frame) unless `reuse_buffer=True`, which hands out the cached receive buffer itself - valid only until the
    next call (a per-step loop that consumes the energies at once)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    nmax = max(frames_per_rank)
    buf, out = _gather_buffers(energy_local.device, torch.float64, world, nmax)
    buf[: energy_local.shape[0]] = energy_local
    dist.all_gather_into_tensor(out, buf, group=group)
    if all(f == nmax for f in frames_per_rank):
        return out if reuse_buffer else out.clone()
    v = out.view(world, nmax)
    return torch.cat([v[r, : frames_per_rank[r]] for r in range(world)])


def all_gather_atoms(x_local, atoms_per_rank: list[int], group=None):
    """All-gather a per-atom tensor [n_local, ...] (forces, charges) in global atom order."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    nmax = max(atoms_per_rank)
    tail = tuple(x_local.shape[1:])
    buf = torch.zeros((nmax,) + tail, dtype=x_local.dtype, device=x_local.device)
    buf[: x_local.shape[0]] = x_local
    out = torch.empty((world * nmax,) + tail, dtype=x_local.dtype, device=x_local.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    out = out.view((world, nmax) + tail)
    return torch.cat([out[r, : atoms_per_rank[r]] for r in range(world)])
