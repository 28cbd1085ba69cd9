"""Verlet-skin reuse of the neighbour matrices across MD steps.

The reference's counterpart is `StaticInputCache` (aimnet/calculators/neighbors.py:150-250): it keeps `nbmat` / `shifts` /
`nbmat_lr` / `nbmat_dftd3` for a caller whose coordinate tensor is literally the same object - static geometry only.  This
module keeps them while the atoms MOVE: the matrices are built with cutoff + skin (the engine's own builder,
aimnet_neighbor_list), stay valid until some atom has moved more than skin / 2 from where it stood at build time, and go to the
engine as caller-supplied matrices (aimnet_inputs.nbmat ...: no list is built, calculator.py:1069-1071 of the reference) - the
kernels cut every pair at the true cutoff themselves (cosine cutoff, `d < dsf_rc`, the D3 switch), so the extra skin pairs
contribute exactly nothing.

Frames: the builder wraps the atoms into the cell and its shifts refer to the wrapped positions.  On a reuse step the current
coordinates are translated by the same per-atom lattice vectors as at build time (an atom may leave the cell by up to skin / 2 -
the engine takes caller-supplied coordinates as given), so a trajectory has to be continuous: a caller that re-wraps an atom
moves it by a cell vector, which reads as a displacement > skin / 2 and triggers a rebuild (correct, merely no reuse).

What it buys (profiles/r6_verlet.md): the engine's fused cell-grid builder costs 0.05 ms of a 1.22 ms step on the 10 080-atom
crystal, and periodic DSF needs no 15 A list at all (it walks the grid), so on the headline configuration reuse LOSES - the
matrices have to be imported and the 15 A rows read.  It is for callers who already hold lists, for very long cutoffs on small
systems, and it is what SURVEY 8f names; the default MD path stays the per-step rebuild (aimnet2ase.py, md_throughput).
"""
from __future__ import annotations

from typing import Any

from .engine import HipEngine, NeighborOverflowError, _round16, neighbor_list


class VerletSkinLists:
    """`HipEngine.eval` with neighbour matrices that live for as long as no atom has moved more than skin / 2.

    sync=True (default): one displacement check per step decides on the host (one scalar D2H, beside the status read the
    synchronous path pays anyway).  sync=False: nothing is read back - the matrices are rebuilt every `rebuild_every` steps and the
    per-step "an atom left the skin" flags are verified by `check_deferred()` together with the engine's deferred status words;
    a raised flag invalidates the steps since the last check (NeighborOverflowError, as for a row overflow)."""

    def __init__(self, engine: HipEngine, skin: float = 0.5, rebuild_every: int = 20):
        if not skin > 0.0:
            raise ValueError("VerletSkinLists: skin must be positive")
        self.engine, self.skin, self.rebuild_every = engine, float(skin), int(rebuild_every)
        self._key = None
        self._lists: dict[str, Any] = {}
        self._x_ref = self._offset = self._cell_ref = None
        self._age = 0
        self._pending_flags: list = []
        self.builds = self.reuses = 0
        self._cap: dict[float, int] = {}

    def invalidate(self) -> None:
        self._key = None

    def _build_one(self, coord, cutoff: float, mol_idx, cell, pbc):
        import math

        cap = self._cap.get(cutoff) or _round16(int(0.2 * 4.0 / 3.0 * math.pi * cutoff**3))
        while True:
            nbmat, _num, shifts, xw, (longest, overflow) = neighbor_list(coord, cutoff, mol_idx, cell, pbc, max_nb=cap)
            if not overflow:
                break
            cap = _round16(int(max(cap * 1.5, longest)))  # AdaptiveNeighborList growth rule, neighbors.py:127-130
        self._cap[cutoff] = cap
        width = max(16, _round16(longest))
        return nbmat[:, :width].contiguous(), (shifts[:, :width].contiguous() if shifts is not None else None), xw

    def _build(self, coord, mol_idx, cell, pbc, need_lr: float | None, need_d3: float | None) -> None:
        rc = float(self.engine.spec.rc)
        nb, sh, xw = self._build_one(coord, rc + self.skin, mol_idx, cell, pbc)
        self._lists = {"nbmat": nb, "shifts": sh}
        if need_lr is not None:
            self._lists["nbmat_lr"], self._lists["shifts_lr"], _ = self._build_one(coord, need_lr + self.skin, mol_idx, cell, pbc)
        if need_d3 is not None:
            self._lists["nbmat_d3"], self._lists["shifts_d3"], _ = self._build_one(coord, need_d3 + self.skin, mol_idx, cell, pbc)
        self._x_ref = coord.clone()
        self._offset = coord - xw  # per-atom lattice translation of the build-time wrap (zero without a cell)
        self._cell_ref = None if cell is None else cell.clone()
        self._age = 0
        self.builds += 1

    def eval(self, coord, numbers, mol_idx, charge, cell=None, pbc=(True, True, True), coulomb: str = "simple", dsf_rc: float = 15.0,
             dftd3: dict[str, float] | None = None, sync: bool = True, **kw) -> dict[str, Any]:
        import torch

        eng = self.engine
        dev = eng.device
        coord = coord.to(device=dev, dtype=torch.float32).contiguous()
        mol_idx = mol_idx.to(device=dev, dtype=torch.int32).contiguous()
        if cell is not None:
            cell = cell.to(device=dev, dtype=torch.float32).contiguous()
            if cell.ndim != 2:
                raise NotImplementedError("VerletSkinLists: one cell per call (the list builder's per-system cells are not exposed)")
        if coulomb in ("ewald", "pme"):
            raise ValueError("VerletSkinLists: Ewald / PME walk the engine's own cell grid and take no caller-supplied matrices")
        d3_rc = float(dftd3.get("cutoff", 15.0)) if dftd3 is not None else None
        # with caller-supplied matrices DSF runs over a matrix (no grid walk); 'simple' sums all pairs of a molecule and needs none
        need_lr = float(dsf_rc) if coulomb == "dsf" else None
        # one cutoff for DSF and D3: ONE matrix serves both terms (include/aimnet_hip.h: pass it as nbmat_lr only)
        need_d3 = d3_rc if (d3_rc is not None and d3_rc != need_lr) else None
        key = (tuple(coord.shape), numbers.data_ptr(), None if cell is None else tuple(cell.shape), tuple(bool(b) for b in pbc),
               need_lr, need_d3)
        rebuild = key != self._key
        flag = None
        if not rebuild:
            moved = (coord - self._x_ref).square().sum(dim=1).max() > (0.5 * self.skin) ** 2
            if cell is not None:
                moved = moved | (cell != self._cell_ref).any()  # a changed cell (NPT) moves every image
            if sync:
                rebuild = bool(moved)  # the step's one extra scalar read
            else:
                flag = moved
                rebuild = self._age + 1 >= self.rebuild_every
        if rebuild:
            self._build(coord, mol_idx, cell, pbc, need_lr, need_d3)
            self._key = key
            flag = None
        else:
            self.reuses += 1
            self._age += 1
        if flag is not None:
            self._pending_flags.append(flag)
        x = coord - self._offset
        res = eng.eval(x, numbers, mol_idx, charge, cell=cell, pbc=pbc, coulomb=coulomb, dsf_rc=dsf_rc, dftd3=dftd3, sync=sync,
                       **self._lists, **kw)
        return res

    def check_deferred(self) -> None:
        """The deferred mode's check: an atom that left the skin on a reuse step invalidates the steps since the last check."""
        import torch

        flags, self._pending_flags = self._pending_flags, []
        self.engine.check_deferred()
        if flags and bool(torch.stack(flags).any()):
            self.invalidate()
            raise NeighborOverflowError("VerletSkinLists: an atom moved more than skin / 2 on a step that reused the neighbour matrices; "
                                        "the evaluations since the last check are invalid - repeat them (the matrices will be rebuilt), "
                                        "or use a larger skin / a smaller rebuild_every")
