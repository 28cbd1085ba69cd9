"""Deterministic synthetic AIMNet2 weights in the reference's v2 artifact format.

No pretrained weights exist offline (registry downloads need network,
aimnet/calculators/model_registry.yaml:41-45 of the reference), so parity and benchmarks run
on random-init weights of the real architecture.  The generator is NumPy PCG64 -> fp32 so the
GPU box regenerates bit-identical tensors from the seed alone; `state_dict_digest` lets a test
prove that.  The artifact layout follows docs/model_format.md:204-222 and the export path
aimnet/train/export_model.py:133-260 (state-dict key names, fp64 atomic shifts, NaN rows for
unsupported species, `srcoulomb` appended last, metadata keys).
"""
from __future__ import annotations

import hashlib
import math
from typing import Any

import numpy as np

# Species of the shipped `aimnet2` family (H B C N O F Si P S Cl As Se Br I).
AIMNET2_SPECIES = [1, 5, 6, 7, 8, 9, 14, 15, 16, 17, 33, 34, 35, 53]

# Core architecture of aimnet/models/aimnet2.yaml after strip_lr_modules_from_yaml
# (aimnet/models/utils.py:379-580): `lrcoulomb` removed, `srcoulomb` appended last.  Key order
# is significant (outputs run in YAML order, aimnet2.py:184-185), hence a literal string.
AIMNET2_CORE_YAML = """\
class: aimnet.models.AIMNet2
kwargs:
  nfeature: 16
  d2features: true
  ncomb_v: 12
  hidden:
  - - 512
    - 380
  - - 512
    - 380
  - - 512
    - 380
    - 380
  aim_size: 256
  aev:
    rc_s: 5.0
    nshifts_s: 16
  outputs:
    energy_mlp:
      class: aimnet.modules.Output
      kwargs:
        n_in: 256
        n_out: 1
        key_in: aim
        key_out: energy
        mlp:
          activation_fn: torch.nn.GELU
          last_linear: true
          hidden:
          - 128
          - 128
    atomic_shift:
      class: aimnet.modules.AtomicShift
      kwargs:
        key_in: energy
        key_out: energy
    atomic_sum:
      class: aimnet.modules.AtomicSum
      kwargs:
        key_in: energy
        key_out: energy
    srcoulomb:
      class: aimnet.modules.SRCoulomb
      kwargs:
        rc: 4.6
        key_in: charges
        key_out: energy
        envelope: exp
"""

# Rough wB97M-D3 self-atomic energies (eV) so |E| has realistic magnitude and the fp64
# accumulation path (core.py:71-97, utils.py:369-376) is exercised.
_SAE_EV = {
    1: -16.30, 5: -675.4, 6: -1035.6, 7: -1488.2, 8: -2045.1, 9: -2715.8, 14: -7876.0,
    15: -9288.5, 16: -10832.9, 17: -12520.5, 33: -60835.0, 34: -65348.0, 35: -70045.3, 53: -8102.5,
}


def _mlp_sizes(n_in: int, hidden: list[int], n_out: int) -> list[tuple[int, int]]:
    sizes = [n_in, *hidden, n_out]
    return [(sizes[i + 1], sizes[i]) for i in range(len(sizes) - 1)]


_RXN_EXTRA_OUTPUTS = """    dipole:
      class: aimnet.modules.Dipole
      kwargs:
        key_in: charges
        key_out: dipole
    quadrupole:
      class: aimnet.modules.Quadrupole
      kwargs:
        key_in: charges
        key_out: quadrupole
"""


def rxn_yaml() -> str:
    """The core YAML of the `aimnet2_rxn` architecture (aimnet/models/aimnet2_rxn.yaml after strip_lr_modules_from_yaml,
    utils.py:379): the aimnet2 network with an explicit `num_charge_channels: 1` and the Dipole / Quadrupole output modules
    in front of the SRCoulomb block (BASELINE config 4 runs on it)."""
    y = AIMNET2_CORE_YAML.replace("kwargs:\n  nfeature: 16\n", "kwargs:\n  nfeature: 16\n  num_charge_channels: 1\n", 1)
    assert "    srcoulomb:\n" in y
    return y.replace("    srcoulomb:\n", _RXN_EXTRA_OUTPUTS + "    srcoulomb:\n", 1)


def core_yaml(num_charge_channels: int = 1, rxn: bool = False) -> str:
    """AIMNET2_CORE_YAML, with `num_charge_channels: 2` for the open-shell NSE family (aimnet2.py:21,94-106)."""
    if rxn:
        return rxn_yaml()
    if num_charge_channels == 1:
        return AIMNET2_CORE_YAML
    return AIMNET2_CORE_YAML.replace("kwargs:\n  nfeature: 16\n", f"kwargs:\n  num_charge_channels: {int(num_charge_channels)}\n  nfeature: 16\n", 1)


# "Cold" variant of a seed (parity at the reference's LITERAL gates): the same random tensors with the hidden MLP layers
# scaled to a gain of 1.1 (hot: 1.4), the (q~, f~, delta_a) layers to 0.3 (0.5) and the energy head's last layer to 0.8 (1.2): random
# organics with 0.9 A contacts then have max|F| ~ 5 eV/A (hot: 50 - 800), taxol 3 eV/A, |q| <= 0.4 e - the force scale of real
# molecules near equilibrium, where |dE| < 1e-5 eV and allclose(rtol 1e-4, atol 1e-5) on every force component
# (tests/test_calculator_gpu.py:137,445,464 of the reference) can be asked of fp32 arithmetic without fp64 anchoring.
_COLD_GAINS = {"hidden": 1.1 / 1.4, "charge_out": 0.3 / 0.5, "head_last": 0.8 / 1.2}


def synthetic_state_dict(seed: int = 0, species: list[int] | None = None, num_charge_channels: int = 1,
                         cold: bool = False) -> dict[str, np.ndarray]:
    """All 37 tensors of the aimnet2 core state dict as NumPy arrays (fp32; SAE fp64).  num_charge_channels = 2 gives the
    shapes of an NSE model (conv_q.agh (2,G,H), MLP inputs 704 / 762, outputs 260; aimnet2.py:53-85).  cold: see _COLD_GAINS."""
    if cold:
        sd = synthetic_state_dict(seed, species, num_charge_channels)
        for p in range(3):
            nl = 3 if p < 2 else 4
            for li in range(nl):
                k = f"mlps.{p}.{2 * li}.weight"
                g = _COLD_GAINS["charge_out"] if (li == nl - 1 and p < 2) else _COLD_GAINS["hidden"]
                sd[k] = (sd[k] * np.float32(g)).astype(np.float32)
        k = "outputs.energy_mlp.mlp.4.weight"
        sd[k] = (sd[k] * np.float32(_COLD_GAINS["head_last"])).astype(np.float32)
        return sd
    species = list(AIMNET2_SPECIES if species is None else species)
    rng = np.random.Generator(np.random.PCG64(seed))

    def normal(shape, scale):
        return (rng.standard_normal(shape) * scale).astype(np.float32)

    A, G, H = 16, 16, 12
    sd: dict[str, np.ndarray] = {}
    rc, rmin = 5.0, 0.8
    eta = (1.0 / ((rc - rmin) / G)) ** 2  # aev.py:71-72
    shifts = np.linspace(rmin, rc, G + 1, dtype=np.float32)[:G]  # aev.py:78
    for mod in ("_s", "_v"):
        sd["aev.rc" + mod] = np.asarray(rc, dtype=np.float32)
        sd["aev.eta" + mod] = np.asarray(eta, dtype=np.float32)
        sd["aev.shifts" + mod] = shifts.copy()

    afv = np.full((64, A * G), np.nan, dtype=np.float32)
    afv[0] = 0.0
    for z in species:
        # one 16-vector per element, broadcast over the 16 shifts (aimnet2.py:46-51) plus jitter
        base = normal((A, 1), 0.6)
        afv[z] = (base + normal((A, G), 0.15)).reshape(-1)
    sd["afv.weight"] = afv
    sd["conv_a.agh"] = normal((A, G, H), 0.35)
    nq = int(num_charge_channels)
    sd["conv_q.agh"] = normal((nq, G, H), 0.35)

    n_conv_a, n_conv_q = A * G + A * H, nq * (G + H)
    n_in0 = n_conv_a + A * G  # 704
    n_in1 = n_in0 + n_conv_q + nq  # 733 (762 for 2 channels)
    mlps = [
        _mlp_sizes(n_in0, [512, 380], A * G + 2 * nq),
        _mlp_sizes(n_in1, [512, 380], A * G + 2 * nq),
        _mlp_sizes(n_in1, [512, 380, 380], 256),
    ]
    for p, layers in enumerate(mlps):
        for li, (fo, fi) in enumerate(layers):
            std = math.sqrt(2.0 / (fi + fo))  # xavier_normal_, core.py:18,40
            last = li == len(layers) - 1
            # pre-activations O(1): GELU in its nonlinear range; the (q~, f~, delta_a) layers smaller - half as large again
            # for two charge channels, where the spin channel otherwise drives |q| > 3 e and forces > 100 eV/A
            gain = (0.5 if nq == 1 else 0.25) if (last and p < 2) else 1.4
            sd[f"mlps.{p}.{2 * li}.weight"] = normal((fo, fi), std * gain)
            sd[f"mlps.{p}.{2 * li}.bias"] = normal((fo,), 0.05)
    for li, (fo, fi) in enumerate(_mlp_sizes(256, [128, 128], 1)):
        std = math.sqrt(2.0 / (fi + fo))
        sd[f"outputs.energy_mlp.mlp.{2 * li}.weight"] = normal((fo, fi), std * 1.2)
        sd[f"outputs.energy_mlp.mlp.{2 * li}.bias"] = normal((fo,), 0.05)

    sae = np.zeros((64, 1), dtype=np.float64)
    for z in species:
        sae[z, 0] = _SAE_EV.get(z, -100.0 * z) + float(rng.standard_normal()) * 0.01
    sd["outputs.atomic_shift.shifts.weight"] = sae
    sd["outputs.srcoulomb.rc"] = np.asarray(4.6, dtype=np.float32)
    return sd


def state_dict_digest(sd: dict[str, Any]) -> str:
    """sha256 over (key, dtype, shape, bytes) in key order; NaN rows hash by bit pattern."""
    h = hashlib.sha256()
    for k in sorted(sd):
        v = np.ascontiguousarray(np.asarray(sd[k]))
        h.update(k.encode())
        h.update(str(v.dtype).encode())
        h.update(str(v.shape).encode())
        h.update(v.tobytes())
    return h.hexdigest()


def synthetic_artifact(seed: int = 0, species: list[int] | None = None, num_charge_channels: int = 1,
                       rxn: bool = False, sr_envelope: str = "exp", sr_rc: float = 4.6, cold: bool = False) -> dict[str, Any]:
    """A v2 artifact dict (torch tensors) as `torch.save` would hold it.  sr_envelope / sr_rc: the SRCoulomb block's
    `envelope` ("exp" mollifier or "cosine", lr.py:986-1032) and radius - same weights, another short-range subtraction."""
    import torch

    species = list(AIMNET2_SPECIES if species is None else species)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synthetic_state_dict(seed, species, num_charge_channels, cold).items()}
    if rxn:
        # Dipole / Quadrupole register a `mass` buffer (119 atomic masses, core.py:163) that the reference loader insists on;
        # it is only read with center_coord=True, which the rxn YAML does not set - the synthetic artifact carries zeros
        for name in ("dipole", "quadrupole"):
            sd[f"outputs.{name}.mass"] = torch.zeros(119, dtype=torch.float32)
    yml = core_yaml(num_charge_channels, rxn)
    if sr_envelope != "exp" or float(sr_rc) != 4.6:
        old = "        rc: 4.6\n        key_in: charges\n        key_out: energy\n        envelope: exp\n"
        assert old in yml
        yml = yml.replace(old, f"        rc: {float(sr_rc)}\n        key_in: charges\n        key_out: energy\n        envelope: {sr_envelope}\n")
        sd["outputs.srcoulomb.rc"] = torch.tensor(float(sr_rc), dtype=torch.float32)
    return {
        "format_version": 2,
        "model_yaml": yml,
        "state_dict": sd,
        "cutoff": 5.0,
        "needs_coulomb": True,
        "needs_dispersion": False,
        "coulomb_mode": "sr_embedded",
        "coulomb_sr_rc": float(sr_rc),
        "coulomb_sr_envelope": sr_envelope,
        "d3_params": None,
        "has_embedded_lr": True,
        "implemented_species": species,
    }
