"""Synthetic / crystallographic inputs for the BASELINE.json configs.

Pure NumPy, deterministic (PCG64), no file access: the GPU box has neither the reference
tree nor ASE, so the allose/glucose cell of examples/2019828.cif (reference, lines 49-55 cell,
13687-13691 symmetry operations, 13704-13727 fractional sites) is restated here as data.
"""
from __future__ import annotations

import math

import numpy as np

# (Z, fx, fy, fz) of the 24 asymmetric-unit sites, 2019828.cif:13704-13727.
_CIF_SITES = [
    (8, 0.1621, 0.79703, 0.67894), (8, 0.1676, 0.85269, 0.91826), (8, -0.0213, 1.05723, 0.88005),
    (8, 0.3426, 1.21319, 0.80258), (8, 0.2673, 0.96785, 0.63757), (8, 0.6115, 1.14807, 0.56175),
    (6, 0.1454, 0.90015, 0.71937), (6, 0.2848, 0.91736, 0.83377), (6, 0.2552, 1.03366, 0.86734),
    (6, 0.3680, 1.10357, 0.77418), (6, 0.2263, 1.07910, 0.66196), (6, 0.3259, 1.14427, 0.56431),
    (1, 0.3191, 0.7774, 0.6812), (1, -0.0442, 0.9196, 0.7254), (1, 0.2377, 0.7936, 0.9180),
    (1, 0.4758, 0.8998, 0.8280), (1, -0.0440, 1.0847, 0.9421), (1, 0.3529, 1.0470, 0.9387),
    (1, 0.5592, 1.0872, 0.7668), (1, 0.1849, 1.2268, 0.8151), (1, 0.0336, 1.0920, 0.6699),
    (1, 0.6640, 1.1146, 0.5069), (1, 0.2577, 1.1140, 0.4939), (1, 0.2565, 1.2162, 0.5702),
]
_CIF_A, _CIF_B, _CIF_C, _CIF_BETA = 4.98211, 12.5624, 11.8156, 91.1262


def glucose_cell() -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """96-atom P2_1/c unit cell: (coord (96,3) f64, numbers (96,) i64, cell (3,3) f64, row vectors)."""
    beta = math.radians(_CIF_BETA)
    cell = np.array(
        [[_CIF_A, 0.0, 0.0], [0.0, _CIF_B, 0.0], [_CIF_C * math.cos(beta), 0.0, _CIF_C * math.sin(beta)]]
    )
    z = np.array([s[0] for s in _CIF_SITES], dtype=np.int64)
    f = np.array([s[1:] for s in _CIF_SITES], dtype=np.float64)
    ops = [
        lambda p: p,
        lambda p: np.stack([-p[:, 0], 0.5 + p[:, 1], 0.5 - p[:, 2]], axis=1),
        lambda p: -p,
        lambda p: np.stack([p[:, 0], 0.5 - p[:, 1], 0.5 + p[:, 2]], axis=1),
    ]
    frac = np.concatenate([op(f) for op in ops], axis=0) % 1.0
    numbers = np.concatenate([z] * 4)
    return frac @ cell, numbers, cell


def glucose_supercell(reps=(7, 3, 5)) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Config 3: (7,3,5) -> 10 080 atoms, cell 34.87 x 37.69 x 59.08 A."""
    coord, numbers, cell = glucose_cell()
    nx, ny, nz = reps
    out = []
    for ix in range(nx):
        for iy in range(ny):
            for iz in range(nz):
                out.append(coord + ix * cell[0] + iy * cell[1] + iz * cell[2])
    sc = cell * np.array([[nx], [ny], [nz]], dtype=np.float64)
    return np.concatenate(out, axis=0), np.tile(numbers, nx * ny * nz), sc


def random_organic(n: int, rng: np.random.Generator, min_dist: float = 0.9) -> tuple[np.ndarray, np.ndarray]:
    """A connected blob of n atoms (H 50 % / C 30 % / N 10 % / O 10 %): each new atom is
    placed 1.0-1.5 A from a random earlier atom, rejected if closer than `min_dist` to any
    atom or outside a sphere of radius 1.2 n^(1/3) + 1 A."""
    numbers = rng.choice(np.array([1, 6, 7, 8]), size=n, p=[0.5, 0.3, 0.1, 0.1]).astype(np.int64)
    numbers[0] = 6
    rmax = 1.2 * n ** (1.0 / 3.0) + 1.0
    pos = np.zeros((n, 3))
    k = 1
    while k < n:
        parent = pos[rng.integers(0, k)]
        v = rng.standard_normal(3)
        v /= np.linalg.norm(v)
        cand = parent + v * rng.uniform(1.0, 1.5)
        if np.linalg.norm(cand) > rmax:
            continue
        if np.min(np.linalg.norm(pos[:k] - cand, axis=1)) < min_dist:
            continue
        pos[k] = cand
        k += 1
    return pos, numbers


def random_batch(n_mol: int, size_lo: int, size_hi: int, seed: int):
    """Configs 2 / 5: flat batch -> coord (Ntot,3) f32, numbers (Ntot,) i64, mol_idx (Ntot,) i64,
    charge (n_mol,) f32 (all neutral)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    coords, nums, mol = [], [], []
    for m in range(n_mol):
        n = int(rng.integers(size_lo, size_hi + 1))
        c, z = random_organic(n, rng)
        coords.append(c)
        nums.append(z)
        mol.append(np.full(n, m, dtype=np.int64))
    return (
        np.concatenate(coords).astype(np.float32),
        np.concatenate(nums),
        np.concatenate(mol),
        np.zeros(n_mol, dtype=np.float32),
    )


def pad_batch(coord, numbers, mol_idx, n_mol: int):
    """Flat batch -> zero-padded dense (B, Nmax, 3) / (B, Nmax) layout (mode-0 style input)."""
    sizes = np.bincount(mol_idx, minlength=n_mol)
    nmax = int(sizes.max())
    c = np.zeros((n_mol, nmax, 3), dtype=np.float32)
    z = np.zeros((n_mol, nmax), dtype=np.int64)
    start = 0
    for m, s in enumerate(sizes):
        c[m, :s] = coord[start : start + s]
        z[m, :s] = numbers[start : start + s]
        start += s
    return c, z
