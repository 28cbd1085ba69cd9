"""HipEngine: thin Python owner of one `aimnet_engine` handle (include/aimnet_hip.h).

PyTorch-ROCm tensors are used purely as device buffers (`data_ptr()`), the HIP stream is torch's
current stream; there is no autograd and no torch compute on the hot path.  One engine per
device; not re-entrant (same contract as the reference calculator, calculator.py:283-328).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Any

import numpy as np

from . import _lib

_DTYPES = {4: "float32", 8: "float64"}
_INT_VIEWS = {"nb_idx", "nb_shift", "nb_cnt", "lr_idx", "lr_shift", "lr_cnt"}


def _round16(n: int) -> int:
    return ((int(n) + 15) // 16) * 16


@dataclass
class ModelSpec:
    """Architecture + host weights of one AIMNet2 core model (built by loader.py / synth.py)."""

    nfeature: int
    nshifts: int
    ncomb_v: int
    mlp_dims: list[list[int]]          # per pass: [n_in, hidden..., n_out]
    last_linear: list[bool]
    head_dims: list[int]
    rc: float
    eta: float
    shifts: list[float]
    sr_coulomb: bool
    sr_envelope: str
    sr_rc: float
    weights: dict[str, np.ndarray] = field(repr=False, default_factory=dict)  # state-dict keys
    metadata: dict[str, Any] = field(default_factory=dict)
    num_charge_channels: int = 1       # 2 = open-shell NSE family (aimnet2.py:21-28)


class ActivationRangeError(RuntimeError):
    """A DEFERRED evaluation produced non-finite energies while the fp16x2-split GEMM operands were in use: an MLP activation left
    fp16's range (|x| >= 65504, csrc/gemm_h2_common.h).  The engine has switched to the bf16x3 operands; repeat the evaluations
    since the last check (the synchronous path does this by itself)."""


class NeighborOverflowError(RuntimeError):
    """A neighbour row overflowed in a DEFERRED evaluation (the synchronous path retries by itself; nvalchemiops raises the
    exception of this name, neighbors.py:127-130)."""


NONFINITE_FLAG = 32  # status[6] bit 5 (csrc/kernels.h STATUS_NONFINITE): a non-finite energy or force was written - not an input flag


def describe_input_flags(flags: int, n_mol: int | None = None) -> str:
    """Text of the input sanity flags the engine raises in status[6] (it clamps for memory safety; the results are meaningless)."""
    what = []
    if flags & 1:
        what.append("atomic numbers outside [0, 63] (the embedding has 64 rows, core.py:49)")
    if flags & 2:
        what.append("mol_idx entries outside [0, %s) (n_mol is taken from the charge array)" % ("n_mol" if n_mol is None else n_mol))
    if flags & 4:
        what.append("mol_idx is not sorted (the atoms of a molecule must be contiguous)")
    if flags & 8:
        what.append("a caller-supplied neighbour matrix holds a lattice shift outside +-127 or an unshifted self pair")
    if flags & 16:
        what.append("a caller-supplied neighbour matrix is not a full symmetric matrix (an entry i -> j without exactly one mirror j -> i, "
                    "or a duplicated entry)")
    if flags & ~31:
        what.append(f"unknown flag bits {flags & ~31:#x}")
    return " and ".join(what) if what else "no flags"


class HipEngine:
    def __init__(self, spec: ModelSpec, device: str | int = 0):
        import torch

        self.spec = spec
        self.lib = _lib.load()  # raises HipLibraryError when the .so is missing: no fallback
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("HipEngine needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if self.device.type != "cuda":
            raise _lib.HipLibraryError(f"HipEngine device must be a GPU, got {self.device}")
        self.dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._keep: list[np.ndarray] = []
        arch = _lib.Arch()
        arch.nfeature, arch.nshifts, arch.ncomb_v = spec.nfeature, spec.nshifts, spec.ncomb_v
        arch.n_pass = len(spec.mlp_dims)
        for p, dims in enumerate(spec.mlp_dims):
            arch.n_layers[p] = len(dims) - 1
            for k, d in enumerate(dims):
                arch.layer_dims[p][k] = d
            arch.last_linear[p] = 1 if spec.last_linear[p] else 0
        arch.head_n_layers = len(spec.head_dims) - 1
        for k, d in enumerate(spec.head_dims):
            arch.head_dims[k] = d
        arch.rc, arch.eta = float(spec.rc), float(spec.eta)
        for g, s in enumerate(spec.shifts):
            arch.shifts[g] = float(s)
        arch.sr_coulomb = 1 if spec.sr_coulomb else 0
        arch.sr_envelope = 0 if spec.sr_envelope == "exp" else 1
        arch.sr_rc = float(spec.sr_rc)
        arch.n_charge_channels = self.nq = int(spec.num_charge_channels)

        w = _lib.Weights()
        sd = spec.weights

        def ptr(key: str, dtype=np.float32) -> int:
            a = np.ascontiguousarray(np.asarray(sd[key]), dtype=dtype)
            self._keep.append(a)
            return a.ctypes.data

        w.afv, w.agh_a, w.agh_q = ptr("afv.weight"), ptr("conv_a.agh"), ptr("conv_q.agh")
        for p, dims in enumerate(spec.mlp_dims):
            for layer in range(len(dims) - 1):
                w.mlp_w[p][layer] = ptr(f"mlps.{p}.{2 * layer}.weight")
                w.mlp_b[p][layer] = ptr(f"mlps.{p}.{2 * layer}.bias")
        for layer in range(len(spec.head_dims) - 1):
            w.head_w[layer] = ptr(f"outputs.energy_mlp.mlp.{2 * layer}.weight")
            w.head_b[layer] = ptr(f"outputs.energy_mlp.mlp.{2 * layer}.bias")
        w.sae = ptr("outputs.atomic_shift.shifts.weight", np.float64)
        handle = C.c_void_p()
        _lib.check(self.lib.aimnet_engine_create(C.byref(arch), C.byref(w), self.dev_index, C.byref(handle)), "aimnet_engine_create")
        self._h = handle
        self._keep.clear()  # weights are on the device now
        self._ws = None
        self._ws_stream = None  # stream of the last launch (the workspace is tied to it)
        self.pending_status: list = []  # status words of evaluations run with sync=False and defer=True (check_deferred)
        self._pending_energy: list = []  # their energies, while the h2 operand form is on (the fp16-range sentinel, check_deferred)
        # AdaptiveNeighborList policy (neighbors.py:49-63): density 0.2 -> 112 @ 5 A, 2832 @ 15 A
        self.max_nb = _round16(int(0.2 * 4.0 / 3.0 * math.pi * spec.rc**3))
        self._max_nb_lr: dict[float, int] = {}
        self.last_status: np.ndarray | None = None
        self.has_dftd3 = False
        self._ewald_max_k = 8192  # capacity of the Ewald k arrays (entries; grows to what status[7] reports)
        self._pme_max_mesh = 8192  # capacity of one system's PME mesh (points; every system of a batch gets a slice: grows to what status[7] reports)
        self._status7_pme = False  # what status[7] of the pending deferred evaluations counts (mesh points or k entries)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self.lib.aimnet_engine_destroy(h)
            except Exception:
                pass
            self._h = None

    # ------------------------------------------------------------------------------------------
    def set_option(self, name: str, value: int) -> None:
        """Engine switch for A/B and parity runs (include/aimnet_hip.h, aimnet_engine_set_option): "gemm_bf3", "gemm_presplit", "gemm_h2", "head_fused", "prep_fused", "energy_rides", "status_rides", "setup_rides", "status_owned", "sums_whole", "conv_mfma",
        "conv_xe", "split_max", "p0_moments", "spatial_order", "overlap_coulomb"."""
        _lib.check(self.lib.aimnet_engine_set_option(self._h, name.encode(), int(value)), "aimnet_engine_set_option")
        self._ws = None  # the workspace layout depends on the switches

    def get_option(self, name: str) -> int:
        """Current value of an engine switch (aimnet_engine_get_option)."""
        v = C.c_int(0)
        _lib.check(self.lib.aimnet_engine_get_option(self._h, name.encode(), C.byref(v)), "aimnet_engine_get_option")
        return int(v.value)

    def set_dftd3_tables(self, tables: dict[str, Any]) -> None:
        """Upload the DFT-D3 reference tables (what DFTD3.__init__ reads from aimnet/dftd3_data.pt, lr.py:1405-1423):
        `c6ab`, `cn_ref` [Z,Z,5,5] and `rcov`, `r4r2` [Z] (see loader.load_dftd3_tables)."""
        arrs = {k: np.ascontiguousarray(np.asarray(tables[k]), dtype=np.float32) for k in ("c6ab", "cn_ref", "rcov", "r4r2")}
        nz = arrs["rcov"].shape[0]
        if arrs["c6ab"].shape != (nz, nz, 5, 5) or arrs["cn_ref"].shape != (nz, nz, 5, 5) or arrs["r4r2"].shape != (nz,):
            raise ValueError("DFT-D3 tables must be c6ab/cn_ref [Z,Z,5,5] and rcov/r4r2 [Z]")
        t = _lib.DftD3Tables()
        t.n_z = nz
        t.c6ab, t.cn_ref, t.rcov, t.r4r2 = (arrs[k].ctypes.data for k in ("c6ab", "cn_ref", "rcov", "r4r2"))
        _lib.check(self.lib.aimnet_engine_set_dftd3(self._h, C.byref(t)), "aimnet_engine_set_dftd3")
        self.has_dftd3 = True

    @staticmethod
    def _shrunk(capacity: int, actual_max: int, target_utilization: float = 0.75) -> int:
        if actual_max < (2.0 / 3.0) * target_utilization * capacity:
            return max(16, _round16(int(actual_max / target_utilization)))
        return capacity

    def _grow_pme_mesh(self, need: int) -> None:
        if need >= 2**31 - 1:
            raise ValueError("HipEngine: a cell vector of this system needs more than 512 PME mesh points (csrc/pme.hip PME_MAX_AXIS); "
                             "use coulomb='dsf' or a looser ewald_accuracy")
        self._pme_max_mesh = need * 9 // 8 + 64

    def check_deferred(self) -> None:
        """Synchronise once and verify the status words of every evaluation run with `sync=False, defer=True` since the last
        check: a device-resident MD driver calls this every K steps instead of paying one host read per step.  On a neighbour
        overflow the row capacity is grown (x1.5, as the synchronous path does) and NeighborOverflowError is raised - the
        evaluations since the last check are invalid and have to be repeated."""
        import torch

        if not self.pending_status:
            return
        st = torch.stack(self.pending_status).cpu().numpy()
        self.pending_status.clear()
        self.last_status = st[-1]
        energies, self._pending_energy = self._pending_energy, []
        if (st[:, 6] & ~NONFINITE_FLAG).any():  # (first: invalid input also produces non-finite outputs)
            raise ValueError("HipEngine: invalid input in a deferred evaluation: " +
                             describe_input_flags(int(np.bitwise_or.reduce(st[:, 6])) & ~NONFINITE_FLAG))
        rows_ok = not (st[:, 2].any() or st[:, 3].any() or st[:, 5].any())
        # status[6] bit 5: the kernels that write the energies / forces saw a non-finite value (covers the adjoint sweep too, which
        # the energies alone would not)
        nonfinite = bool((st[:, 6] & NONFINITE_FLAG).any()) or not all(bool(torch.isfinite(e).all()) for e in energies)
        if rows_ok and nonfinite and self.get_option("gemm_h2"):
            self.set_option("gemm_h2", 0)
            raise ActivationRangeError("non-finite energies or forces in one of the last %d deferred evaluations with fp16x2-split GEMM "
                                       "operands: an activation may have left fp16's range; the engine now uses the bf16x3 operands - "
                                       "repeat them (if they stay non-finite the input itself is the cause)" % len(st))
        grown = False
        if st[:, 2].any():
            self.max_nb = _round16(int(max(self.max_nb * 1.5, st[:, 0].max())))
            grown = True
        if st[:, 3].any() or st[:, 5].any():
            for rc in list(self._max_nb_lr):
                self._max_nb_lr[rc] = _round16(int(max(self._max_nb_lr[rc] * 1.5, st[:, 1].max(), st[:, 4].max())))
            grown = True
        if self._status7_pme:
            if st[:, 7].max() > self._pme_max_mesh:  # PME mesh of the largest system
                self._grow_pme_mesh(int(st[:, 7].max()))
                grown = True
        elif st[:, 7].max() > self._ewald_max_k:  # Ewald k arrays (status[7] is 0 for the other methods)
            self._ewald_max_k = (int(st[:, 7].max()) * 5 // 4 + 7) // 8 * 8
            grown = True
        if grown:
            raise NeighborOverflowError("neighbour-row overflow in one of the last %d deferred evaluations: their results are invalid; "
                                        "the row capacity has been grown - repeat them" % len(st))

    def _lr_capacity(self, rc: float) -> int:
        if rc not in self._max_nb_lr:
            self._max_nb_lr[rc] = _round16(int(0.2 * 4.0 / 3.0 * math.pi * rc**3))
        return self._max_nb_lr[rc]

    def eval(
        self,
        coord,
        numbers,
        mol_idx,
        charge,
        cell=None,
        pbc=(True, True, True),
        forces: bool = False,
        stress: bool = False,
        coulomb: str = "simple",
        dsf_rc: float = 15.0,
        dsf_alpha: float = 0.2,
        ewald_accuracy: float = 1e-6,
        sync: bool = True,
        dftd3: dict[str, float] | None = None,
        host_out: bool = False,
        defer: bool = False,
        nbmat=None,
        shifts=None,
        nbmat_lr=None,
        shifts_lr=None,
        nbmat_d3=None,
        shifts_d3=None,
    ) -> dict[str, Any]:
        """One evaluation on device tensors (coord f32 [N,3], numbers/mol_idx i32 [N], charge f32
        [n_mol] - for a 2-channel NSE model [n_mol, 2] = the alpha / beta charges of aimnet2.py:94-100 -,
        cell f32 [3,3]|[n_mol,3,3]).  Returns device tensors (NSE: plus spin_charges); retries with x1.5 row
        capacity on neighbour overflow (neighbors.py:127-130).  `dftd3` = {s8, a1, a2[, s6, cutoff, smoothing_fraction]}
        adds the external DFT-D3(BJ) term (needs set_dftd3_tables).  All outputs and the status words live in ONE device
        buffer; `host_out=True` returns CPU tensors taken from the single D2H copy that fetches the status anyway (the
        ASE adapter's path: no further .cpu() round trips).
        `nbmat` (+ `shifts`, `nbmat_lr` / `shifts_lr`, `nbmat_d3` / `shifts_d3`): caller-supplied FULL neighbour matrices, int
        [N, width] with entries outside [0, N) as padding, integer shifts [N, width, 3] - the engine then builds no list and takes the
        coordinates as given (include/aimnet_hip.h, aimnet_inputs.nbmat)."""
        import torch

        dev = self.device
        coord = coord.to(device=dev, dtype=torch.float32).contiguous()
        numbers = numbers.to(device=dev, dtype=torch.int32).contiguous()
        mol_idx = mol_idx.to(device=dev, dtype=torch.int32).contiguous()
        charge = charge.to(device=dev, dtype=torch.float32)
        if self.nq == 2:
            if charge.ndim != 2 or charge.shape[1] != 2:
                raise ValueError("HipEngine.eval: a 2-channel (NSE) model needs charge of shape [n_mol, 2] (alpha, beta)")
            n_mol = charge.shape[0]
            charge = charge.t().contiguous()  # channel-major planes, include/aimnet_hip.h aimnet_inputs.charge
        else:
            if charge.ndim != 1:
                raise ValueError("HipEngine.eval: charge must have shape [n_mol]")
            charge = charge.contiguous()
            n_mol = charge.shape[0]
        n = coord.shape[0]
        if n == 0 or n_mol == 0:
            raise ValueError("HipEngine.eval: empty input (no atoms or no molecules)")
        n_cell = 0
        if cell is not None:
            cell = cell.to(device=dev, dtype=torch.float32).contiguous()
            n_cell = 1 if cell.ndim == 2 else cell.shape[0]
        # pbc: three flags, or per-system flags [n_cell, 3] (tensor / array): normalize_pbc, neighbors.py:309-321
        pbc_sys = None
        if cell is not None and not isinstance(pbc, (tuple, list)) and getattr(pbc, "ndim", 1) == 2:
            pbc_sys = torch.as_tensor(pbc).to(device=dev, dtype=torch.int32).contiguous()
            if tuple(pbc_sys.shape) != (n_cell, 3):
                raise ValueError(f"pbc must have shape (3,) or ({n_cell}, 3), got {tuple(pbc_sys.shape)}")
        elif not isinstance(pbc, (tuple, list)):
            pbc = tuple(bool(x) for x in torch.as_tensor(pbc).reshape(-1).tolist())
        # "ewald": the exact structure-factor sum (csrc/ewald.hip); "pme": the same splitting with the reciprocal sum on a mesh
        # (csrc/pme.hip)
        method = {"none": _lib.COULOMB_NONE, "simple": _lib.COULOMB_SIMPLE, "dsf": _lib.COULOMB_DSF, "ewald": _lib.COULOMB_EWALD,
                  "pme": _lib.COULOMB_PME}[coulomb]
        is_pme = method == _lib.COULOMB_PME
        if method in (_lib.COULOMB_EWALD, _lib.COULOMB_PME):
            if cell is None:
                raise ValueError(f"HipEngine.eval: coulomb={coulomb!r} needs a periodic cell (lr.py:655-657)")
            if nbmat is not None:
                raise ValueError("HipEngine.eval: Ewald summation does not take caller-supplied neighbour matrices")
            if not (0.0 < float(ewald_accuracy) < 1.0):
                raise ValueError("HipEngine.eval: ewald_accuracy must lie in (0, 1)")
        ext = {}
        for key, mat, sh in (("nbmat", nbmat, shifts), ("nbmat_lr", nbmat_lr, shifts_lr), ("nbmat_d3", nbmat_d3, shifts_d3)):
            if mat is None:
                if sh is not None:
                    raise ValueError(f"HipEngine.eval: shifts given without {key}")
                continue
            if nbmat is None:
                raise ValueError("HipEngine.eval: nbmat_lr / nbmat_d3 are only read together with nbmat")
            mat = mat.to(device=dev, dtype=torch.int32).contiguous()
            if mat.ndim != 2 or mat.shape[0] != n or mat.shape[1] < 1:
                raise ValueError(f"HipEngine.eval: {key} must have shape [{n}, width], got {tuple(mat.shape)}")
            if sh is not None:
                sh = sh.to(device=dev, dtype=torch.int32).contiguous()
                if tuple(sh.shape) != (n, mat.shape[1], 3):
                    raise ValueError(f"HipEngine.eval: the shifts of {key} must have shape {(n, mat.shape[1], 3)}, got {tuple(sh.shape)}")
            elif cell is not None:
                raise ValueError(f"HipEngine.eval: periodic input needs the shifts of {key}")
            ext[key] = (mat, sh)
        # one allocation for status + every output (16-byte aligned sections); status is zeroed by the engine
        sections = [("status", torch.int32, (8,)), ("energy", torch.float64, (n_mol,)), ("charges", torch.float32, (n,))]
        if self.nq == 2:
            sections.append(("spin_charges", torch.float32, (n,)))
        if forces:
            sections.append(("forces", torch.float32, (n, 3)))
        if stress:
            sections.append(("stress", torch.float32, (max(n_cell, 1), 3, 3)))
        esize = {torch.int32: 4, torch.float32: 4, torch.float64: 8}
        offs, nbytes, total = [], [], 0
        for _, dt, shape in sections:
            nb = math.prod(shape) * esize[dt]
            offs.append(total)
            nbytes.append(nb)
            total += (nb + 15) // 16 * 16
        outbuf = torch.empty(total, dtype=torch.uint8, device=dev)

        def views(buf):
            return {name: buf[o : o + nb].view(dt).view(shape) for (name, dt, shape), o, nb in zip(sections, offs, nbytes)}

        dv = views(outbuf)
        energy, charges, status = dv["energy"], dv["charges"], dv["status"]
        spin, f_out, s_out = dv.get("spin_charges"), dv.get("forces"), dv.get("stress")
        stream = torch.cuda.current_stream(dev).cuda_stream
        host = None
        h2_on, h2_retry = bool(self.get_option("gemm_h2")) and bool(self.get_option("gemm_presplit")), False
        pme_shrunk = False
        while True:
            opt = _lib.EvalOptions()
            opt.flags = (_lib.FORCES if forces else 0) | (_lib.STRESS if stress else 0)
            opt.coulomb = method
            opt.dsf_rc, opt.dsf_alpha = float(dsf_rc), float(dsf_alpha)
            opt.max_nb = self.max_nb
            # periodic DSF walks the cell grid (no list); only non-periodic DSF materialises a long-range list
            opt.max_nb_lr = self._lr_capacity(float(dsf_rc)) if (method == _lib.COULOMB_DSF and cell is None) else 0
            # ... and neither do non-periodic systems large enough for the bounding-box cell grid (engine.hip `np_walk`: the same rule)
            if (opt.max_nb_lr and n >= 1500 * n_mol and nbmat is None and self.get_option("dsf_np_walk") and
                    not (dftd3 is not None and float(dftd3.get("cutoff", 15.0)) == float(dsf_rc))):
                opt.max_nb_lr = 0
            if method in (_lib.COULOMB_EWALD, _lib.COULOMB_PME):
                opt.ewald_accuracy = float(ewald_accuracy)
                opt.ewald_max_k = self._ewald_max_k
                opt.pme_max_mesh = self._pme_max_mesh
                # the mesh workspace is n_mol slices of the LARGEST system's mesh this engine has seen (40 B per point): after one big
                # cell, a batch of many small ones would ask for n_mol x that.  Shrink to the starting capacity once (status[7] grows
                # it back to what THIS batch needs); if the batch itself is beyond the limit, say so instead of running out of memory.
                # (the PME kernels index the systems through grid.y: <= 65 535)
                if method == _lib.COULOMB_PME and (n_mol > 65535 or n_mol * self._pme_max_mesh * 40 > 64 << 30):
                    if n_mol <= 65535 and not pme_shrunk and self._pme_max_mesh > 8192:
                        pme_shrunk = True
                        self._pme_max_mesh = opt.pme_max_mesh = 8192
                    else:
                        raise ValueError(f"HipEngine.eval(coulomb='pme'): {n_mol} systems x a mesh of {self._pme_max_mesh} points for the "
                                         "largest of them exceeds the PME workspace limit (64 GiB, 65 535 systems); split the batch, or use "
                                         "coulomb='ewald' / 'dsf'")
            if ext:  # caller-supplied matrices: the row capacities are their widths (nothing can overflow)
                opt.max_nb = _round16(ext["nbmat"][0].shape[1])
                opt.max_nb_lr = _round16(ext["nbmat_lr"][0].shape[1]) if "nbmat_lr" in ext else 0
            if dftd3 is not None:
                if not self.has_dftd3:
                    raise RuntimeError("dftd3 requested but no DFT-D3 tables were uploaded (HipEngine.set_dftd3_tables)")
                d3_rc = float(dftd3.get("cutoff", 15.0))
                opt.dftd3 = 1
                opt.d3_s6, opt.d3_s8 = float(dftd3.get("s6", 1.0)), float(dftd3["s8"])
                opt.d3_a1, opt.d3_a2 = float(dftd3["a1"]), float(dftd3["a2"])
                opt.d3_cutoff = d3_rc
                opt.d3_smoothing_on = d3_rc * (1.0 - float(dftd3.get("smoothing_fraction", 0.2)))
                opt.max_nb_d3 = self._lr_capacity(d3_rc)
                if ext:
                    src = ext.get("nbmat_d3") or ext.get("nbmat_lr")
                    opt.max_nb_d3 = _round16(src[0].shape[1]) if src is not None else 16
            need = int(self.lib.aimnet_engine_workspace_bytes(self._h, n, n_mol, n_cell, C.byref(opt)))
            if self._ws is None or self._ws.numel() < need:
                if self._ws is not None and self._ws_stream is not None:
                    # earlier (possibly still running, sync=False) evaluations used the old workspace on that stream: keep the
                    # caching allocator from handing its memory to another stream before they finish
                    self._ws.record_stream(self._ws_stream)
                self._ws = None
                self._ws = torch.empty(int(need * 1.1) + 4096, dtype=torch.uint8, device=dev)
            cur_stream = torch.cuda.current_stream(dev)
            if self._ws_stream is not None and self._ws_stream != cur_stream:
                cur_stream.wait_stream(self._ws_stream)  # one workspace: evaluations on different streams are serialised
            self._ws_stream = cur_stream
            inp = _lib.Inputs()
            inp.n_atoms, inp.n_mol = n, n_mol
            inp.coord, inp.numbers, inp.mol_idx, inp.charge = coord.data_ptr(), numbers.data_ptr(), mol_idx.data_ptr(), charge.data_ptr()
            inp.cell = cell.data_ptr() if cell is not None else None
            inp.n_cell = n_cell
            if pbc_sys is not None:
                inp.pbc_sys = pbc_sys.data_ptr()
                for k in range(3):
                    inp.pbc[k] = 1
            else:
                inp.pbc_sys = None
                for k in range(3):
                    inp.pbc[k] = 1 if bool(pbc[k]) else 0
            for key, (mat, sh) in ext.items():
                setattr(inp, key, mat.data_ptr())
                setattr(inp, "shifts" + key[5:], sh.data_ptr() if sh is not None else None)
                setattr(inp, key + "_width", int(mat.shape[1]))
            out = _lib.Outputs()
            out.energy, out.charges = energy.data_ptr(), charges.data_ptr()
            out.forces = f_out.data_ptr() if f_out is not None else None
            out.stress = s_out.data_ptr() if s_out is not None else None
            out.status = status.data_ptr()
            out.spin_charges = spin.data_ptr() if spin is not None else None
            with torch.cuda.device(dev):
                rc = self.lib.aimnet_engine_eval(self._h, C.byref(inp), C.byref(opt), C.byref(out), self._ws.data_ptr(),
                                                 self._ws.numel(), stream)
            _lib.check(rc, "aimnet_engine_eval")
            if not sync:
                break
            # the one D2H sync of a step (the reference has one per list, neighbors.py:133)
            if host_out:
                # ONE D2H copy brings the status words and every output back.  (A persistent pinned staging buffer was tried and
                # dropped: torch's pinned host memory is uncached for the CPU on this platform - reading 160 KB of results out of
                # it took 1.4 ms, and the evaluation as a whole 15 ms instead of 2.2, tests/tools/ase_prof2.py.)
                host = outbuf.cpu()
                st = host[:32].view(torch.int32).numpy()
                finite = (all(bool(torch.isfinite(v).all()) for k, v in views(host).items() if k != "status") and
                          not (int(st[6]) & NONFINITE_FLAG)) if (h2_on or h2_retry) else True
            else:
                # the status words and the molecule energies (adjacent sections) in one copy: the energies are the fp16-range
                # sentinel of the h2 GEMM operands (below)
                head = outbuf[: offs[1] + nbytes[1]].cpu()
                st = head[:32].view(torch.int32).numpy()
                # (status[6] bit 5 = a non-finite energy or FORCE seen on the device: an overflow in the adjoint sweep leaves the energies finite)
                finite = (bool(torch.isfinite(head[offs[1]:].view(torch.float64)).all()) and
                          not (int(st[6]) & NONFINITE_FLAG)) if (h2_on or h2_retry) else True
            self.last_status = st
            if int(st[6]) & ~NONFINITE_FLAG:  # input sanity flags raised by the engine (it clamps for memory safety, the results are meaningless)
                raise ValueError("HipEngine.eval: invalid input: " + describe_input_flags(int(st[6]) & ~NONFINITE_FLAG, n_mol))
            retry = False
            rows_overflowed = bool(st[2] or st[3] or st[5] or (method == _lib.COULOMB_EWALD and st[7] > opt.ewald_max_k) or
                                   (is_pme and st[7] > opt.pme_max_mesh))
            if not finite and h2_on and not rows_overflowed:
                # fp16x2-split GEMM operands (csrc/gemm_h2_common.h) hold |x| < 65504: an activation beyond that turns into inf / NaN
                # and surfaces in the outputs.  Repeat the call with the bf16x3 operands (fp32's range); if THAT is finite the
                # engine stays on them.
                self.set_option("gemm_h2", 0)
                h2_on, h2_retry = False, True
                continue
            if h2_retry and not rows_overflowed:
                h2_retry = False
                if finite:
                    import warnings
                    warnings.warn("HipEngine: an MLP activation of this model exceeded fp16's range (|x| >= 65504); the engine has "
                                  "switched from the fp16x2-split to the bf16x3-split GEMM operands (set_option('gemm_h2', 0)) for "
                                  "this and all later evaluations", RuntimeWarning, stacklevel=2)
                else:  # non-finite with either operand form: not a range problem of the h2 form
                    self.set_option("gemm_h2", 1)
            if ext:  # caller-supplied rows cannot overflow, and say nothing about the capacities of the engine's own lists
                break
            if st[2]:
                self.max_nb = _round16(int(max(self.max_nb * 1.5, st[0])))
                retry = True
            if st[3]:
                self._max_nb_lr[float(dsf_rc)] = _round16(int(max(opt.max_nb_lr * 1.5, st[1])))
                retry = True
            if method == _lib.COULOMB_EWALD and st[7] > opt.ewald_max_k:  # the k boxes did not fit: their size is now known
                self._ewald_max_k = (int(st[7]) * 5 // 4 + 7) // 8 * 8
                retry = True
            if is_pme and st[7] > opt.pme_max_mesh:  # the mesh did not fit: its size is now known
                self._grow_pme_mesh(int(st[7]))
                retry = True
            if st[5]:  # D3 list (it may be stored in the LR buffers: grow both capacities)
                d3_rc = float(dftd3.get("cutoff", 15.0))
                self._max_nb_lr[d3_rc] = _round16(int(max(max(opt.max_nb_d3, opt.max_nb_lr) * 1.5, st[4])))
                retry = True
            if not retry:
                # AdaptiveNeighborList shrink rule (neighbors.py:135-139): a row capacity used to less than 2/3 of the 75 % target
                # shrinks to actual / 0.75 (multiples of 16, at least 16) - next call's workspace and list traffic follow
                self.max_nb = self._shrunk(self.max_nb, int(st[0]))
                if opt.max_nb_lr > 0:
                    self._max_nb_lr[float(dsf_rc)] = self._shrunk(self._max_nb_lr[float(dsf_rc)], int(st[1]))
                if dftd3 is not None and opt.max_nb_d3 > 0 and st[4] > 0:
                    d3_rc = float(dftd3.get("cutoff", 15.0))
                    self._max_nb_lr[d3_rc] = self._shrunk(self._max_nb_lr[d3_rc], int(st[4]))
                break
        if host is not None:
            hv = views(host)
            energy, charges, spin, f_out, s_out = hv["energy"], hv["charges"], hv.get("spin_charges"), hv.get("forces"), hv.get("stress")
        res: dict[str, Any] = {"energy": energy, "charges": charges}
        if not sync:
            # no host round trip: the caller owns the overflow check (status[2], [3], [5] must be 0, include/aimnet_hip.h)
            # and reads it whenever it next synchronises - e.g. a device-resident MD loop once per block of steps
            res["status"] = status
            if defer:
                self.pending_status.append(status)  # verified in one go by check_deferred()
                self._status7_pme = is_pme
                if h2_on:
                    self._pending_energy.append(energy)
        if spin is not None:
            res["spin_charges"] = spin
        if forces:
            res["forces"] = f_out
        if stress:
            res["stress"] = s_out[0] if (cell is not None and cell.ndim == 2) else s_out
        return res

    # ------------------------------------------------------------------------------------------
    HVP_BYTES_BUDGET = 6 << 30  # workspace of one tangent sweep; more directions than fit are processed in several sweeps
    HVP_MAX_DIRECTIONS = 65535  # the tangent kernels put the direction index in grid.y (HIP limit 65535; 4 K with DFT-D3: 16383)
    # one direction costs ~3/4 us per atom (7.5 ms on 10 080 atoms, profiles/r3_hvp.md): calls that would run for more than
    # this many seconds are announced (or refused, HVP_ON_LONG) (a dense Hessian of a 10 k-atom crystal is 30 240 directions = minutes)
    HVP_MAX_SECONDS = 120.0
    HVP_ON_LONG = "warn"  # "warn": say so and run (the reference runs such requests too, slowly); "raise": refuse with a ValueError

    def hvp(self, coord, numbers, mol_idx, charge, vectors, cell=None, pbc=(True, True, True), coulomb: str = "simple",
            dsf_rc: float = 15.0, dsf_alpha: float = 0.2, want_forces: bool = False,
            dftd3: dict[str, float] | None = None) -> dict[str, Any]:
        """Analytic Hessian-vector products (csrc/hvp.hip, aimnet_engine_hvp): vectors [K, N, 3] -> hv [K, N, 3] on the device
        (+ the forces of the same sweep with `want_forces`).  Inputs as `eval`; with `dftd3` the dispersion block is added as a
        central difference of the D3 gradient inside the same call.  The K directions are processed in as many sweeps as the
        workspace budget asks for."""
        import torch

        dev = self.device
        coord = coord.to(device=dev, dtype=torch.float32).contiguous()
        numbers = numbers.to(device=dev, dtype=torch.int32).contiguous()
        mol_idx = mol_idx.to(device=dev, dtype=torch.int32).contiguous()
        charge = charge.to(device=dev, dtype=torch.float32)
        if self.nq == 2:
            if charge.ndim != 2 or charge.shape[1] != 2:
                raise ValueError("HipEngine.hvp: a 2-channel (NSE) model needs charge of shape [n_mol, 2] (alpha, beta)")
            n_mol = charge.shape[0]
            charge = charge.t().contiguous()
        else:
            charge = charge.reshape(-1).contiguous()
            n_mol = charge.shape[0]
        n = coord.shape[0]
        K = int(vectors.numel()) // max(1, 3 * n)
        if n == 0 or n_mol == 0 or K == 0:
            raise ValueError("HipEngine.hvp: empty input")
        est = 0.75e-6 * n * K if n > 2000 else 0.0  # (small systems are launch-bound: ~0.04 force evaluations per direction)
        if est > self.HVP_MAX_SECONDS and self.HVP_ON_LONG == "raise":
            raise ValueError(f"HipEngine.hvp: {K} directions on {n} atoms would take ~{est:.0f} s (HVP_ON_LONG = 'raise')")
        if est > self.HVP_MAX_SECONDS:  # the reference computes such requests (slowly): warn, do not refuse
            import warnings
            warnings.warn(f"HipEngine.hvp: {K} directions on {n} atoms will take ~{est:.0f} s (~0.75 us per atom and direction on one "
                          f"MI355X, more with DFT-D3); a Krylov / Davidson solver needs tens of directions, not 3N "
                          f"(threshold: HipEngine.HVP_MAX_SECONDS = {self.HVP_MAX_SECONDS:.0f} s)", RuntimeWarning, stacklevel=2)
        vectors = vectors.to(device=dev, dtype=torch.float32).reshape(-1, n, 3).contiguous()
        n_cell = 0
        if cell is not None:
            cell = cell.to(device=dev, dtype=torch.float32).contiguous()
            n_cell = 1 if cell.ndim == 2 else cell.shape[0]
        if not isinstance(pbc, (tuple, list)):
            pbc = tuple(bool(x) for x in torch.as_tensor(pbc).reshape(-1).tolist())[:3]
        if coulomb in ("ewald", "pme"):
            raise ValueError("HipEngine.hvp: the analytic tangent sweep covers the pair-wise Coulomb methods only; with Ewald summation use "
                             "differences of the forces (AIMNet2Calculator does: hvp_method 'fd' is taken automatically)")
        method = {"none": _lib.COULOMB_NONE, "simple": _lib.COULOMB_SIMPLE, "dsf": _lib.COULOMB_DSF}[coulomb]
        hv = torch.empty_like(vectors)
        f_out = torch.empty(n, 3, dtype=torch.float32, device=dev) if want_forces else None
        status = torch.empty(8, dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        inp = _lib.Inputs()
        inp.n_atoms, inp.n_mol = n, n_mol
        inp.coord, inp.numbers, inp.mol_idx, inp.charge = coord.data_ptr(), numbers.data_ptr(), mol_idx.data_ptr(), charge.data_ptr()
        inp.cell = cell.data_ptr() if cell is not None else None
        inp.n_cell = n_cell
        inp.pbc_sys = None
        for k in range(3):
            inp.pbc[k] = 1 if bool(pbc[k]) else 0
        k0 = 0
        while k0 < K:
            opt = _lib.EvalOptions()
            opt.coulomb = method
            opt.dsf_rc, opt.dsf_alpha = float(dsf_rc), float(dsf_alpha)
            opt.max_nb = self.max_nb
            opt.max_nb_lr = self._lr_capacity(float(dsf_rc)) if method == _lib.COULOMB_DSF else 0
            if dftd3 is not None:
                if not self.has_dftd3:
                    raise RuntimeError("dftd3 requested but no DFT-D3 tables were uploaded (HipEngine.set_dftd3_tables)")
                d3_rc = float(dftd3.get("cutoff", 15.0))
                opt.dftd3 = 1
                opt.d3_s6, opt.d3_s8 = float(dftd3.get("s6", 1.0)), float(dftd3["s8"])
                opt.d3_a1, opt.d3_a2 = float(dftd3["a1"]), float(dftd3["a2"])
                opt.d3_cutoff = d3_rc
                opt.d3_smoothing_on = d3_rc * (1.0 - float(dftd3.get("smoothing_fraction", 0.2)))
                opt.max_nb_d3 = self._lr_capacity(d3_rc)
            per_dir = (int(self.lib.aimnet_engine_hvp_workspace_bytes(self._h, n, n_mol, 2, C.byref(opt)))
                       - int(self.lib.aimnet_engine_hvp_workspace_bytes(self._h, n, n_mol, 1, C.byref(opt))))
            kmax = self.HVP_MAX_DIRECTIONS // (4 if dftd3 is not None else 1)  # grid.y limit of the tangent kernels
            kc = max(1, min(K - k0, kmax, self.HVP_BYTES_BUDGET // max(1, per_dir)))
            need = int(self.lib.aimnet_engine_hvp_workspace_bytes(self._h, n, n_mol, kc, C.byref(opt)))
            ws = torch.empty(need + 4096, dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                rc = self.lib.aimnet_engine_hvp(self._h, C.byref(inp), C.byref(opt), vectors[k0 : k0 + kc].data_ptr(), kc,
                                                hv[k0 : k0 + kc].data_ptr(), f_out.data_ptr() if f_out is not None else None,
                                                status.data_ptr(), ws.data_ptr(), ws.numel(), stream)
            _lib.check(rc, "aimnet_engine_hvp")
            st = status.cpu().numpy()
            del ws
            self.last_status = st
            if st[6]:
                raise ValueError("HipEngine.hvp: invalid input: " + describe_input_flags(int(st[6]), n_mol))
            if st[2] or st[3] or st[5]:  # neighbour-row overflow: grow and repeat this sweep
                if st[2]:
                    self.max_nb = _round16(int(max(self.max_nb * 1.5, st[0])))
                if st[3]:
                    self._max_nb_lr[float(dsf_rc)] = _round16(int(max(opt.max_nb_lr * 1.5, st[1])))
                if st[5]:
                    self._max_nb_lr[float(dftd3.get("cutoff", 15.0))] = _round16(int(max(opt.max_nb_d3 * 1.5, st[4])))
                continue
            k0 += kc
        res: dict[str, Any] = {"hv": hv}
        if f_out is not None:
            res["forces"] = f_out
        return res

    def set_profiling(self, level: int, every: int = 1) -> None:
        """0 off, 1 GEMM-vs-rest, 2 per kernel family (HIP events on the eval stream); `every` = record them on every
        n-th evaluation only (the events cost ~3 % of a 2 ms step; read_profile()["evals"] says how many were covered)."""
        _lib.check(self.lib.aimnet_engine_set_profile_sampling(self._h, int(every)), "aimnet_engine_set_profile_sampling")
        _lib.check(self.lib.aimnet_engine_set_profiling(self._h, int(level)), "aimnet_engine_set_profiling")

    def read_profile(self, reset: bool = True) -> dict[str, float]:
        """Milliseconds per kernel family accumulated since the last reset."""
        nf = len(_lib.PROF_FAMILIES)
        buf = (C.c_double * (nf + 1))()
        _lib.check(self.lib.aimnet_engine_profile_read(self._h, buf, nf + 1, 1 if reset else 0), "aimnet_engine_profile_read")
        res = {k: float(buf[i]) for i, k in enumerate(_lib.PROF_FAMILIES)}
        res["evals"] = float(buf[nf])
        return res

    def gemm_flops_per_atom(self, backward: bool = True) -> float:
        """Algorithmic (unpadded) FLOPs per atom of all GEMM launches of one eval: 2 MACs forward,
        plus the input-gradient GEMMs of the backward; the k->1 head layer is a dot kernel."""
        macs = 0
        for dims in self.spec.mlp_dims:
            macs += sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
        hd = self.spec.head_dims
        macs += sum(hd[i] * hd[i + 1] for i in range(len(hd) - 2))
        total = 2.0 * macs * (2 if backward else 1)
        # pass 0, first layer: the 256 embedding columns are folded into a per-element bias table at create time, the GEMM
        # runs over the 448 conv columns (engine.hip, emb_bias0) - not counted, it is not executed
        total -= 2.0 * 256 * self.spec.mlp_dims[0][1]
        if backward:
            # the pass-0 input gradient with respect to the (constant) embedding block is never formed:
            # the last backward GEMM of pass 0 has N = 448, not 704 (engine.hip, "only the conv columns")
            d0 = self.spec.mlp_dims[0]
            total -= 2.0 * 256 * d0[1]
        return total

    def debug_view(self, name: str):
        """Intermediate of the last eval as a torch tensor view into the workspace (tests only)."""
        import torch

        off, n_elem = C.c_size_t(), C.c_size_t()
        esz, stride = C.c_int32(), C.c_int32()
        rc = self.lib.aimnet_engine_debug_view(self._h, name.encode(), C.byref(off), C.byref(n_elem), C.byref(esz), C.byref(stride))
        if rc != 0:
            raise KeyError(name)
        raw = self._ws[off.value : off.value + n_elem.value * esz.value]
        if name in _INT_VIEWS:
            t = raw.view(torch.int32)
        elif esz.value == 16:
            t = raw.view(torch.float32)
        else:
            t = raw.view(getattr(torch, _DTYPES[esz.value]))
        if stride.value > 1:
            per = stride.value * (4 if esz.value == 16 else 1)
            t = t.view(-1, per)
        return t


# ---- stand-alone op wrappers (parity tests of the reference operator boundaries) ------------------
def neighbor_list(coord, cutoff: float, mol_idx=None, cell=None, pbc=(True, True, True), max_nb: int = 128, fill_value=None):
    """nvalchemiops-contract neighbour list on the GPU: returns (nbmat i32 [N,max_nb], num_nb i32 [N],
    shifts i32 [N,max_nb,3] | None, coord_wrapped f32 [N,3], status (max_count, overflow))."""
    import torch

    lib = _lib.load()
    dev = coord.device
    coord = coord.to(torch.float32).contiguous()
    n = coord.shape[0]
    if mol_idx is None:
        mol_idx = torch.zeros(n, dtype=torch.int32, device=dev)
    mol_idx = mol_idx.to(device=dev, dtype=torch.int32).contiguous()
    n_mol = int(mol_idx.max().item()) + 1
    fill = n if fill_value is None else int(fill_value)
    n_cell = 0
    if cell is not None:
        cell = cell.to(device=dev, dtype=torch.float32).contiguous()
        n_cell = 1 if cell.ndim == 2 else cell.shape[0]
    nbmat = torch.empty(n, max_nb, dtype=torch.int32, device=dev)
    shifts = torch.empty(n, max_nb, 3, dtype=torch.int32, device=dev) if cell is not None else None
    num = torch.empty(n, dtype=torch.int32, device=dev)
    status = torch.zeros(2, dtype=torch.int32, device=dev)
    xw = torch.empty(n, 3, dtype=torch.float32, device=dev)
    nbytes = int(lib.aimnet_neighbor_list_workspace_bytes(n, n_mol, max_nb))
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    p3 = (C.c_int32 * 3)(*[1 if bool(b) else 0 for b in pbc])
    with torch.cuda.device(dev):
        rc = lib.aimnet_neighbor_list(coord.data_ptr(), mol_idx.data_ptr(), n, n_mol, cell.data_ptr() if cell is not None else None,
                                      n_cell, C.byref(p3), float(cutoff), int(max_nb), fill, nbmat.data_ptr(),
                                      shifts.data_ptr() if shifts is not None else None, num.data_ptr(), status.data_ptr(),
                                      xw.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "aimnet_neighbor_list")
    st = status.cpu().numpy()
    return nbmat, num, shifts, xw, (int(st[0]), int(st[1]))


def conv_sv_2d_sp_fwd(a, idx, g):
    """torch.ops.aimnet.conv_sv_2d_sp_fwd contract (conv_sv_2d_sp_wp.py:252-275)."""
    import torch

    lib = _lib.load()
    B, A, G = a.shape
    M = idx.shape[1]
    a = a.to(torch.float32).contiguous()
    idx = idx.to(torch.int32).contiguous()
    g = g.to(torch.float32).contiguous()
    out = torch.empty(B, A, G, 4, dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = lib.aimnet_conv_sv_2d_sp_fwd(a.data_ptr(), idx.data_ptr(), g.data_ptr(), out.data_ptr(), B, A, G, M,
                                          torch.cuda.current_stream(a.device).cuda_stream)
    _lib.check(rc, "aimnet_conv_sv_2d_sp_fwd")
    return out


def conv_sv_2d_sp_bwd(grad_out, a, idx, g):
    """torch.ops.aimnet.conv_sv_2d_sp_bwd contract (conv_sv_2d_sp_wp.py:285-340): (grad_a, grad_g)."""
    import torch

    lib = _lib.load()
    B, A, G = a.shape
    M = idx.shape[1]
    grad_out = grad_out.to(torch.float32).contiguous()
    a = a.to(torch.float32).contiguous()
    idx = idx.to(torch.int32).contiguous()
    g = g.to(torch.float32).contiguous()
    grad_a = torch.empty_like(a)
    grad_g = torch.empty_like(g)
    with torch.cuda.device(a.device):
        rc = lib.aimnet_conv_sv_2d_sp_bwd(grad_out.data_ptr(), a.data_ptr(), idx.data_ptr(), g.data_ptr(), grad_a.data_ptr(),
                                          grad_g.data_ptr(), B, A, G, M, torch.cuda.current_stream(a.device).cuda_stream)
    _lib.check(rc, "aimnet_conv_sv_2d_sp_bwd")
    return grad_a, grad_g


def conv_sv_2d_sp_bwd_bwd(grad_out, grad2_a, grad2_g, a, idx, g):
    """torch.ops.aimnet.conv_sv_2d_sp_bwd_bwd contract (conv_sv_2d_sp_wp.py:342-446):
    (grad_grad_output [B,A,G,4], grad_a_double [B,A,G], grad_g_double [B,M,G,4])."""
    import torch

    lib = _lib.load()
    B, A, G = a.shape
    M = idx.shape[1]
    f = lambda t: t.to(torch.float32).contiguous()  # noqa: E731
    grad_out, grad2_a, grad2_g, a, g = f(grad_out), f(grad2_a), f(grad2_g), f(a), f(g)
    idx = idx.to(torch.int32).contiguous()
    ggo = torch.empty(B, A, G, 4, dtype=torch.float32, device=a.device)
    ga2 = torch.empty_like(a)
    gg2 = torch.empty_like(g)
    with torch.cuda.device(a.device):
        rc = lib.aimnet_conv_sv_2d_sp_bwd_bwd(grad_out.data_ptr(), grad2_a.data_ptr(), grad2_g.data_ptr(), a.data_ptr(), idx.data_ptr(),
                                              g.data_ptr(), ggo.data_ptr(), ga2.data_ptr(), gg2.data_ptr(), B, A, G, M,
                                              torch.cuda.current_stream(a.device).cuda_stream)
    _lib.check(rc, "aimnet_conv_sv_2d_sp_bwd_bwd")
    return ggo, ga2, gg2
