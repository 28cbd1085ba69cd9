"""AIMNet2ASE - ASE calculator adapter over the native AIMNet2Calculator.

Observable contract of aimnet/calculators/aimnet2ase.py:35-274 of the reference:
implemented_properties, species check in set_atoms, atoms.info["charge"] precedence and cache
invalidation, non-periodic systems passed as a 3-D batch of one, periodic systems flat with
cell + pbc, results {energy (float), charges, dipole_moment (sum q r on the host), forces?, stress?}.
Pure host code; needs `ase` at run time (absent in the build image -> a minimal stand-in base class
with the same `calculate/reset/check_state` protocol is used so that the adapter stays testable
with any Atoms-like object exposing numbers, positions, cell, pbc, info).
"""
from __future__ import annotations

from typing import ClassVar

import numpy as np

try:  # pragma: no cover - ase is an optional extra
    from ase.calculators.calculator import Calculator, PropertyNotImplementedError, all_changes
    HAVE_ASE = True
except ImportError:
    HAVE_ASE = False
    all_changes = ["positions", "numbers", "cell", "pbc", "initial_charges", "initial_magmoms", "info"]

    class PropertyNotImplementedError(NotImplementedError):
        pass

    class Calculator:
        """Smallest subset of ase.calculators.calculator.Calculator the adapter relies on."""

        def __init__(self, *a, **k):
            self.results = {}
            self.atoms = None

        def reset(self):
            self.results = {}

        def check_state(self, atoms, tol=1e-15):
            old = self.atoms
            if old is None:
                return list(all_changes)
            changes = []
            if len(old.numbers) != len(atoms.numbers) or (np.asarray(old.numbers) != np.asarray(atoms.numbers)).any():
                changes.append("numbers")
            elif np.abs(np.asarray(old.positions) - np.asarray(atoms.positions)).max(initial=0.0) > tol:
                changes.append("positions")
            return changes

        def calculate(self, atoms=None, properties=None, system_changes=None):
            if atoms is not None:
                self.atoms = _copy_atoms(atoms)


def _copy_atoms(atoms):
    return atoms.copy() if hasattr(atoms, "copy") else atoms


def _cell_array(atoms):
    cell = getattr(atoms, "cell", None)
    if cell is None:
        return None
    return np.asarray(cell.array if hasattr(cell, "array") else cell, dtype=np.float64)


class AIMNet2ASE(Calculator):
    implemented_properties: ClassVar[list[str]] = ["energy", "forces", "free_energy", "charges", "stress", "dipole_moment"]

    def __init__(self, base_calc="aimnet2", charge=0, mult=1, validate_species: bool = True):
        from .calculator import AIMNet2Calculator

        super().__init__()
        if isinstance(base_calc, str):
            base_calc = AIMNet2Calculator(base_calc)
        self.base_calc = base_calc
        self.validate_species = validate_species
        if self.base_calc.is_nse:  # aimnet2ase.py:62-63
            self.__dict__["implemented_properties"] = [*self.__class__.implemented_properties, "spin_charges"]
        self.reset()
        self.charge = charge
        self.mult = mult
        meta = base_calc.metadata
        species = meta.get("implemented_species") if meta is not None else None
        self.implemented_species = np.array(species, dtype=np.int64) if species else None

    def reset(self):
        # ASE calls reset() on every change of the positions: only the results go; the device copies of numbers /
        # charge / mult below are validated by content, so an MD step uploads nothing but the coordinates
        super().reset()

    _numbers = None  # (host copy, device int32 tensor)
    _qm = None       # ((charge, mult), device charge [1], device mult [1])

    def _device_inputs(self, numbers):
        import torch

        dev = self.base_calc.device
        if self._numbers is None or self._numbers[0].shape != numbers.shape or not np.array_equal(self._numbers[0], numbers):
            self._numbers = (numbers.copy(), torch.as_tensor(numbers.astype(np.int32), device=dev))
        key = (float(self.charge), float(self.mult))
        if self._qm is None or self._qm[0] != key:
            self._qm = (key, torch.as_tensor(np.array([key[0]], np.float32), device=dev),
                        torch.as_tensor(np.array([key[1]], np.float32), device=dev))
        return self._numbers[1], self._qm[1], self._qm[2]

    def set_atoms(self, atoms):
        if self.implemented_species is not None and not np.isin(atoms.numbers, self.implemented_species).all():
            raise ValueError("Some species are not implemented in the AIMNet2Calculator")
        self.reset()
        self.atoms = atoms

    def check_state(self, atoms, tol=1e-15):
        state = super().check_state(atoms, tol=tol)
        if (not state) and getattr(self, "atoms", None) is not None:
            old_info, new_info = getattr(self.atoms, "info", {}), getattr(atoms, "info", {})
            if old_info.get("charge") != new_info.get("charge"):
                state.append("info")
            elif self.base_calc.is_nse:  # spin / multiplicity matter for NSE models only (aimnet2ase.py:100-105)
                if old_info.get("spin", old_info.get("mult")) != new_info.get("spin", new_info.get("mult")):
                    state.append("info")
        return state

    def set_charge(self, charge):
        self.charge = charge

    def set_mult(self, mult):
        self.mult = mult

    def _charge_from_info(self, atoms):
        info = getattr(atoms, "info", {})
        charge = info.get("charge")
        if charge is not None and charge != self.charge:
            self.charge = charge
        if self.base_calc.is_nse:  # "mult" (AIMNet2 style) or "spin" (MACE style), both 2S+1 (aimnet2ase.py:136-142)
            mult = info.get("mult", info.get("spin"))
            if mult is not None and mult != self.mult:
                self.mult = mult

    def get_dipole_moment(self, atoms=None):
        atoms = self.atoms if atoms is None else atoms
        return np.sum(np.asarray(self.results["charges"])[:, None] * np.asarray(atoms.positions), axis=0)

    def get_spin_charges(self, atoms=None):
        if "spin_charges" not in self.results:
            raise PropertyNotImplementedError("spin_charges is not available. Use an NSE model (e.g. 'aimnet2nse').")
        return self.results["spin_charges"]

    def get_hessian(self, atoms=None):
        atoms = getattr(self, "atoms", None) if atoms is None else atoms
        if atoms is None:
            raise PropertyNotImplementedError("get_hessian() requires an attached Atoms object or an explicit argument.")
        if np.asarray(atoms.pbc).any():
            raise PropertyNotImplementedError("Hessian for periodic systems is not supported by AIMNet2ASE.get_hessian().")
        self._charge_from_info(atoms)
        res = self.base_calc({"coord": np.asarray(atoms.positions, dtype=np.float32), "numbers": np.asarray(atoms.numbers),
                              "charge": float(self.charge), "mult": float(self.mult)}, forces=True, hessian=True,
                             validate_species=self.validate_species)
        H = res["hessian"].detach()
        n = H.shape[0]
        return H.reshape(n * 3, n * 3).cpu().numpy()

    def calculate(self, atoms=None, properties=None, system_changes=all_changes):
        if properties is None:
            properties = ["energy"]
        super().calculate(atoms, properties, system_changes)
        self._charge_from_info(self.atoms)
        pbc = np.asarray(self.atoms.pbc, dtype=bool)
        cell = _cell_array(self.atoms) if pbc.any() else None
        coord = np.ascontiguousarray(self.atoms.positions, dtype=np.float32)
        # The reference hands non-periodic systems over as a 3-D batch of one (aimnet2ase.py:248-251) because its dense
        # mode wants that; the native engine is flat either way, so one flat system is passed.  Per step the only uploads
        # are the coordinates (and the cell), and every output returns with the engine's single status copy (host_out).
        numbers_t, charge_t, mult_t = self._device_inputs(np.asarray(self.atoms.numbers))
        data = {"coord": coord, "numbers": numbers_t, "charge": charge_t}
        if self.base_calc.is_nse:
            data["mult"] = mult_t
        elif float(self.mult) != 1.0 and not getattr(self.base_calc, "_mult_ignored_checked", True):
            data["mult"] = np.float32(self.mult)  # lets the closed-shell calculator warn once that it is ignored
        if cell is not None:
            data["cell"] = cell.astype(np.float32)
            data["pbc"] = pbc
        out = self.base_calc.eval(data, forces="forces" in properties, stress="stress" in properties,
                                  validate_species=self.validate_species, host_out=True)
        res = {k: v.detach().cpu().numpy() for k, v in out.items()}
        self.results["energy"] = float(np.asarray(res["energy"]).reshape(-1)[0])
        self.results["free_energy"] = self.results["energy"]
        self.results["charges"] = res["charges"]
        self.results["dipole_moment"] = self.get_dipole_moment(self.atoms)
        if "spin_charges" in res:
            self.results["spin_charges"] = res["spin_charges"]
        if "forces" in properties:
            self.results["forces"] = res["forces"]
        if "stress" in properties:
            self.results["stress"] = res["stress"]
