"""AIMNet2TorchSim - TorchSim model adapter over the native AIMNet2Calculator.

Observable contract of aimnet/calculators/aimnet2torchsim.py:41-175 of the reference: a
TorchSim `SimState` is a flat multi-system batch - `positions (N,3)`, `atomic_numbers (N)`, `system_idx (N)`,
`row_vector_cell (S,3,3)`, `pbc`, optional per-system `charge` - which is exactly the layout the HIP engine
evaluates, so the adapter only renames fields:  coord <- positions, mol_idx <- system_idx, cell <-
row_vector_cell (when any axis is periodic and the cell is non-zero).  Outputs are the calculator's, detached,
plus `partial_charges` as an alias of `charges`; `implemented_properties` follows the compute_forces /
compute_stress switches.  float32 on the calculator's device whatever the state's dtype (AIMNet2 is fp32).

`torch_sim` is optional: with it the class derives from `torch_sim.models.interface.ModelInterface`; without it
(this build image) the adapter is a plain callable that works with any object exposing the attributes above, which
is what the host tests use.  (The reference refuses to construct without torch_sim; accepting duck-typed states is
a superset, not a behavioural change for TorchSim users.)
"""
from __future__ import annotations

from typing import Any

try:  # pragma: no cover - optional dependency
    from torch_sim.models.interface import ModelInterface as _Base

    HAVE_TORCHSIM = True
except ImportError:
    HAVE_TORCHSIM = False

    class _Base:  # minimal stand-in: callable with the two switches TorchSim integrators flip
        def __init__(self) -> None:
            pass

        def __call__(self, state, **kw):
            return self.forward(state, **kw)


class AIMNet2TorchSim(_Base):
    def __init__(self, base_calc, *, compute_forces: bool = True, compute_stress: bool = False, validate_species: bool = True,
                 status_check_every: int = 1):
        """`status_check_every = K > 1` (not in the reference): device-resident stepping.  forward() only ENQUEUES the
        evaluation - no host read, no synchronisation - and the neighbour-overflow status of the last K evaluations is
        verified in one go every K-th call (`engine.NeighborOverflowError` then says that those K results are invalid and
        the row capacity has been grown).  K = 1 is the reference's behaviour: one status read per call."""
        super().__init__()
        import torch

        self._check_every = max(1, int(status_check_every))
        self._since_check = 0
        self._cell_key = None  # identity of the last cell tensor that was checked for being non-zero (one sync per new cell)

        self._base_calc = base_calc
        self._device = torch.device(base_calc.device)
        self._dtype = torch.float32
        self._validate_species = bool(validate_species)
        self._memory_scales_with = "n_atoms_x_density"
        self._compute_forces = bool(compute_forces)
        self._compute_stress = bool(compute_stress)
        self._refresh_properties()

    # ---- the switches ---------------------------------------------------------------------------------------
    @property
    def base_calc(self):
        return self._base_calc

    @property
    def metadata(self):
        return self._base_calc.metadata

    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        return self._dtype

    @property
    def compute_forces(self) -> bool:
        return self._compute_forces

    @compute_forces.setter
    def compute_forces(self, value: bool) -> None:
        self._compute_forces = bool(value)
        self._refresh_properties()

    @property
    def compute_stress(self) -> bool:
        return self._compute_stress

    @compute_stress.setter
    def compute_stress(self, value: bool) -> None:
        self._compute_stress = bool(value)
        self._refresh_properties()

    def _refresh_properties(self) -> None:
        props = ["energy"] + (["forces"] if self._compute_forces else []) + (["stress"] if self._compute_stress else [])
        props += ["charges", "partial_charges"]
        if self._base_calc.is_nse:
            props.append("spin_charges")
        self.implemented_properties = props

    # ---- evaluation -----------------------------------------------------------------------------------------
    def forward(self, state, **kwargs: Any) -> dict[str, Any]:
        import torch

        if hasattr(state, "to") and (getattr(state, "device", self._device) != self._device or getattr(state, "dtype", self._dtype) != self._dtype):
            state = state.to(self._device, self._dtype)
        defer = self._check_every > 1
        out = self._base_calc.eval(self.state_to_data(state), forces=self._compute_forces, stress=self._compute_stress,
                                   validate_species=self._validate_species, **({"defer_status": True} if defer else {}))
        if defer:
            self._since_check += 1
            if self._since_check >= self._check_every:
                self._since_check = 0
                self._base_calc.check_status()
        if "charges" in out:
            out["partial_charges"] = out["charges"]
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}

    def state_to_data(self, state) -> dict[str, Any]:
        """SimState -> calculator input dict (flat layout; see the module docstring)."""
        import torch

        n_sys = int(state.n_systems)
        data = {
            "coord": torch.as_tensor(state.positions, dtype=torch.float32).clone(),
            # static per-atom integers: converted once per source tensor (identity + version), not once per MD step
            "numbers": self._converted("numbers", torch.as_tensor(state.atomic_numbers), torch.int64),
            "mol_idx": self._converted("mol_idx", torch.as_tensor(state.system_idx), torch.int64),
            "charge": self._per_system(state, ("charge",), 0.0, n_sys),
        }
        if self._base_calc.is_nse:
            data["mult"] = self._per_system(state, ("mult", "spin"), 1.0, n_sys)
        pbc = torch.as_tensor(state.pbc, dtype=torch.bool)
        if pbc.numel() == 1:  # TorchSim allows a single flag for all three axes
            pbc = pbc.reshape(1).expand(3).clone()
        cell = torch.as_tensor(state.row_vector_cell, dtype=torch.float32)
        nonzero = False
        if bool(pbc.any()):
            # "(cell != 0).any()" is a device read: done once per cell tensor (identity + version), not once per MD step
            key = (cell.data_ptr(), getattr(cell, "_version", None), tuple(cell.shape))
            if self._cell_key is None or self._cell_key[0] != key or self._cell_key[2] is not cell:
                self._cell_key = (key, bool((cell != 0).any()), cell)  # (holds the tensor: see _converted)
            nonzero = self._cell_key[1]
        if bool(pbc.any()) and nonzero:
            data["cell"] = cell.contiguous()
            data["pbc"] = pbc
        elif self._compute_stress:
            raise ValueError("AIMNet2 stress calculation requires a periodic TorchSim state with a non-zero cell.")
        return data

    def _converted(self, name: str, src, dtype):
        """`src.to(dtype)`, cached per source tensor: the atomic numbers and system indices of an MD run never change, and a dtype
        conversion is a kernel launch (plus an allocation) per step otherwise."""
        if src.dtype == dtype:
            return src
        cache = self.__dict__.setdefault("_conv_cache", {})
        key = (src.data_ptr(), getattr(src, "_version", None), tuple(src.shape), src.dtype, src.device)
        hit = cache.get(name)
        # the entry HOLDS the source tensor: its storage cannot be recycled for another state's numbers while the entry lives, so
        # an equal (address, version) key can only come from the very same tensor (a freed tensor's address is reused at once by
        # the caching allocator, with _version 0 again)
        if hit is None or hit[0] != key or hit[1] is not src:
            hit = (key, src, src.to(dtype))
            cache[name] = hit
        return hit[2]

    @staticmethod
    def _per_system(state, names: tuple[str, ...], default: float, n_sys: int):
        import torch

        value = next((getattr(state, nm) for nm in names if getattr(state, nm, None) is not None), None)
        if value is None:
            return torch.full((n_sys,), float(default), dtype=torch.float32)
        t = torch.as_tensor(value, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            return t.expand(n_sys).clone()
        if t.numel() != n_sys:
            raise ValueError(f"TorchSim system extra '{'/'.join(names)}' must be scalar or have one value per system "
                             f"({n_sys}); got {t.numel()} values.")
        return t
