"""Format-compatible loader for AIMNet2 v2 model artifacts (`.pt`).

Behavioural restatement of the reference loading path - aimnet/models/base.py:116-211
(load_model / _load_v2_model), aimnet/models/artifact_validation.py (envelope, YAML walk,
metadata rules) and aimnet/models/utils.py:300-376 (state-dict key policy, fp64 atomic shifts)
- but instead of instantiating nn.Modules from the YAML it maps the allow-listed class names to
the components of the native engine and emits a `ModelSpec` (host weights + architecture).

Accepted: `torch.save`d dict {format_version?: 2, model_yaml, state_dict, cutoff, needs_coulomb,
needs_dispersion, coulomb_mode, coulomb_sr_rc, coulomb_sr_envelope, d3_params, has_embedded_lr,
implemented_species, ...} exactly as docs/model_format.md:204-222 describes.  Rejected with the
reference's error classes: non-dict payloads, empty/invalid YAML, alias cycles, import paths
outside the allow-list, the keys fn/trainer/evaluator/ptfile, malformed metadata.  Classes the
native engine does not implement (D3TS, DispParam) raise NotImplementedError - the native loader
can only honour what it implements.  Open-shell NSE models (num_charge_channels = 2,
aimnet2.py:21-28) load into a 2-channel ModelSpec and run on the engine.
"""
from __future__ import annotations

import math
import warnings
from collections.abc import Collection, Mapping
from numbers import Real
from typing import Any

import numpy as np
import yaml

from .engine import ModelSpec

DEFAULT_CLASS_PATHS = frozenset({
    "aimnet.models.AIMNet2",
    "aimnet.models.aimnet2.AIMNet2",
    "aimnet.modules.AtomicShift",
    "aimnet.modules.AtomicSum",
    "aimnet.modules.Dipole",
    "aimnet.modules.Output",
    "aimnet.modules.Quadrupole",
    "aimnet.modules.SRCoulomb",
    "aimnet.modules.D3TS",
    "aimnet.modules.lr.D3TS",
    "aimnet.modules.lr.DispParam",
})
DEFAULT_ACTIVATION_PATHS = frozenset({"torch.nn.GELU"})
DEFAULT_INITIALIZER_PATHS = frozenset({"torch.nn.init.xavier_normal_"})
_IMPORT_KEYS = {"class": "class", "activation_fn": "activation", "weight_init_fn": "initializer"}
_FORBIDDEN_IMPORT_KEYS = frozenset({"fn", "trainer", "evaluator"})
_FORBIDDEN_CTOR_KEYS = frozenset({"ptfile"})
_MODEL_CLASSES = {"aimnet.models.AIMNet2", "aimnet.models.aimnet2.AIMNet2"}
# state-dict keys the calculator re-creates externally / buffers that may legitimately be absent
_EXPECTED_UNEXPECTED_PREFIXES = ("outputs.lrcoulomb.", "outputs.dftd3.", "outputs.d3bj.", "outputs.dipole.mass", "outputs.quadrupole.mass")
_EXPECTED_MISSING_PREFIXES = ("outputs.srcoulomb.",)


def _match(path: str, pattern: str) -> bool:
    if pattern.endswith(".*"):
        return path.startswith(pattern[:-1]) and len(path) > len(pattern) - 1
    return path == pattern


class ImportPolicy:
    """Which dotted paths an artifact's YAML may name (artifact_validation.py:129-153,208-240)."""

    def __init__(self, paths: Collection[str] | None = None, mode: str = "extend"):
        if mode not in ("extend", "replace", "unsafe"):
            raise ValueError(f"model_import_mode must be 'extend', 'replace' or 'unsafe', got {mode!r}.")
        if isinstance(paths, str):
            raise TypeError("model_import_paths must be a collection of strings, not a string.")
        extra = frozenset(paths or ())
        for p in extra:
            if not isinstance(p, str) or not p or any(not part.isidentifier() for part in p.removesuffix(".*").split(".")):
                raise ValueError(f"Invalid import path pattern: {p!r}.")
        self.unsafe = mode == "unsafe"
        if self.unsafe and extra:
            raise ValueError("model_import_mode='unsafe' cannot be combined with model_import_paths.")
        if mode == "replace":
            if not extra:
                raise ValueError("model_import_mode='replace' requires a nonempty model_import_paths.")
            self.paths = {"class": extra, "activation": extra, "initializer": extra}
        else:
            self.paths = {
                "class": DEFAULT_CLASS_PATHS | extra,
                "activation": DEFAULT_ACTIVATION_PATHS | extra,
                "initializer": DEFAULT_INITIALIZER_PATHS | extra,
            }

    def require_allowed(self, path: str, role: str) -> None:
        if self.unsafe:
            return
        if not any(_match(path, pat) for pat in self.paths[role]):
            raise ValueError(f"Untrusted import path for {role!r}: {path!r}.")


def parse_model_yaml(model_yaml: Any, policy: ImportPolicy) -> dict[str, Any]:
    """safe_load + walk: alias cycles, forbidden keys, import allow-list (artifact_validation.py:242-285)."""
    if not isinstance(model_yaml, str) or not model_yaml.strip():
        raise ValueError("model_yaml must be a nonempty string.")
    try:
        config = yaml.safe_load(model_yaml)
    except yaml.YAMLError as exc:
        raise ValueError(f"Invalid model_yaml: {exc}") from exc
    if not isinstance(config, dict):
        raise ValueError("model_yaml root must be a mapping.")
    active: set[int] = set()
    seen: set[int] = set()

    def walk(node: Any) -> None:
        if not isinstance(node, (dict, list)):
            return
        nid = id(node)
        if nid in active:
            raise ValueError("model_yaml contains a recursive alias cycle.")
        if nid in seen:
            return
        active.add(nid)
        seen.add(nid)
        if isinstance(node, dict):
            for key, child in node.items():
                if key in _FORBIDDEN_CTOR_KEYS:
                    raise ValueError(f"Key {key!r} is forbidden in model artifacts.")
                if key in _FORBIDDEN_IMPORT_KEYS:
                    raise ValueError(f"Import key {key!r} is forbidden in model artifacts.")
                if key in _IMPORT_KEYS:
                    if not isinstance(child, str):
                        raise ValueError(f"Import key {key!r} must contain a string path.")
                    policy.require_allowed(child, _IMPORT_KEYS[key])
                walk(child)
        else:
            for child in node:
                walk(child)
        active.remove(nid)

    walk(config)
    return config


def _finite_pos(v: Any) -> bool:
    return not isinstance(v, bool) and isinstance(v, Real) and math.isfinite(float(v)) and v > 0


def validate_metadata(md: Mapping[str, Any], *, require_cutoff: bool = True, structural: bool = True) -> None:
    """Scalar metadata rules of artifact_validation.py:394-500 (structural level)."""
    if require_cutoff and "cutoff" not in md:
        raise ValueError("model metadata requires a 'cutoff' field.")
    if "cutoff" in md and not _finite_pos(md["cutoff"]):
        raise ValueError("model metadata field 'cutoff' must be a finite positive real number.")
    if "format_version" in md and (type(md["format_version"]) is not int or md["format_version"] not in (1, 2)):
        raise ValueError("model metadata field 'format_version' must be integer 1 or 2.")
    for key in ("needs_coulomb", "needs_dispersion", "has_embedded_lr", "has_embedded_d3ts"):
        if key in md and type(md[key]) is not bool:
            raise ValueError(f"model metadata field {key!r} must be a bool.")
    if md.get("supports_charged_systems") is not None and type(md["supports_charged_systems"]) is not bool:
        raise ValueError("model metadata field 'supports_charged_systems' must be a bool or null.")
    if "coulomb_mode" in md and md["coulomb_mode"] not in ("none", "sr_embedded", "full_embedded"):
        raise ValueError("model metadata field 'coulomb_mode' has an unsupported value.")
    if md.get("coulomb_sr_rc") is not None and not _finite_pos(md["coulomb_sr_rc"]):
        raise ValueError("model metadata field 'coulomb_sr_rc' must be a finite positive real number.")
    if md.get("coulomb_sr_envelope") is not None and md["coulomb_sr_envelope"] not in ("exp", "cosine"):
        raise ValueError("model metadata field 'coulomb_sr_envelope' has an unsupported value.")
    d3 = md.get("d3_params")
    if d3 is not None:
        if not isinstance(d3, Mapping):
            raise ValueError("model metadata field 'd3_params' must be a mapping or null.")
        for key in ("s6", "s8", "a1", "a2"):
            if key in d3 and (isinstance(d3[key], bool) or not isinstance(d3[key], Real) or not math.isfinite(float(d3[key]))):
                raise ValueError(f"d3_params[{key!r}] must be a finite real number.")
    if "implemented_species" in md:
        sp = md["implemented_species"]
        if not isinstance(sp, list) or any(type(v) is not int or v <= 0 for v in sp):
            raise ValueError("model metadata field 'implemented_species' must be a list of positive integers.")
    if md.get("family") is not None and not isinstance(md["family"], str):
        raise ValueError("model metadata field 'family' must be a string or null.")
    if structural:
        mode = md.get("coulomb_mode", "none")
        emb = md.get("has_embedded_lr", False)
        if mode == "sr_embedded" and (md.get("coulomb_sr_rc") is None or md.get("coulomb_sr_envelope") is None):
            raise ValueError("sr_embedded Coulomb metadata requires cutoff and envelope fields.")
        if mode == "sr_embedded" and not emb:
            raise ValueError("sr_embedded Coulomb metadata requires embedded LR metadata.")
        if mode == "sr_embedded" and md.get("cutoff") is not None and md["coulomb_sr_rc"] > md["cutoff"]:
            raise ValueError("coulomb_sr_rc cannot exceed model cutoff.")
        if mode == "full_embedded" and not emb:
            raise ValueError("full_embedded Coulomb metadata requires embedded LR metadata.")
        if md.get("has_embedded_d3ts", False) and not emb:
            raise ValueError("embedded D3TS metadata requires embedded LR metadata.")


def validate_runtime_metadata(md: Mapping[str, Any], *, needs_coulomb: bool, needs_dispersion: bool) -> None:
    """artifact_validation.py:503-533."""
    eff = dict(md)
    eff["needs_coulomb"], eff["needs_dispersion"] = needs_coulomb, needs_dispersion
    if "format_version" in md:
        legacy = type(eff.get("format_version")) is int and eff["format_version"] == 1
        validate_metadata(eff, require_cutoff=not legacy, structural=not legacy)
    if needs_coulomb and eff.get("coulomb_mode") == "full_embedded":
        raise ValueError("full_embedded Coulomb metadata cannot request external Coulomb.")
    if needs_dispersion:
        d3 = eff.get("d3_params")
        if not isinstance(d3, Mapping):
            raise ValueError("needs_dispersion metadata requires d3_params.")
        missing = {"s8", "a1", "a2"} - set(d3)
        if missing:
            raise ValueError(f"needs_dispersion metadata is missing d3_params: {sorted(missing)}.")
        if eff.get("has_embedded_d3ts", False):
            raise ValueError("needs_dispersion cannot be combined with embedded D3TS.")


def _has_d3ts(config: Mapping[str, Any]) -> bool:
    outputs = (config.get("kwargs") or {}).get("outputs") or {}
    vals = outputs.values() if isinstance(outputs, Mapping) else outputs
    return any(isinstance(v, Mapping) and "D3TS" in str(v.get("class", "")) for v in vals)


def validate_artifact(data: Any, policy: ImportPolicy) -> tuple[dict[str, Any], Mapping[str, Any]]:
    """Envelope rules of artifact_validation.py:330-369; returns (parsed YAML, state_dict)."""
    import torch

    if not isinstance(data, dict):
        raise ValueError(f"v2 artifact must be a dictionary, got {type(data).__name__}.")
    model_yaml = data.get("model_yaml")
    if not isinstance(model_yaml, str) or not model_yaml.strip():
        raise ValueError("v2 artifact field 'model_yaml' must be a nonempty string.")
    try:
        config = parse_model_yaml(model_yaml, policy)
    except ValueError as exc:
        raise ValueError(f"Invalid v2 artifact field 'model_yaml': {exc}") from exc
    validate_metadata(data, require_cutoff=True, structural=True)
    if _has_d3ts(config) != bool(data.get("has_embedded_d3ts", False)):
        raise ValueError("model metadata field 'has_embedded_d3ts' disagrees with D3TS presence in model_yaml.")
    sd = data.get("state_dict")
    if not isinstance(sd, Mapping):
        raise ValueError("v2 artifact field 'state_dict' must be a mapping.")
    for k, v in sd.items():
        if not isinstance(k, str):
            raise ValueError("v2 artifact state_dict keys must be strings.")
        if not isinstance(v, torch.Tensor):
            raise ValueError(f"v2 artifact state_dict value for {k!r} must be a tensor.")
    fv = data.get("format_version", 2)
    if type(fv) is not int or fv != 2:
        raise ValueError("v2 artifact field 'format_version' must be integer 2.")
    return config, sd


METADATA_KEYS = ("format_version", "cutoff", "needs_coulomb", "needs_dispersion", "coulomb_mode", "coulomb_sr_rc",
                 "coulomb_sr_envelope", "d3_params", "has_embedded_lr", "implemented_species", "family",
                 "supports_charged_systems", "has_embedded_d3ts")


def metadata_from_artifact(data: Mapping[str, Any]) -> dict[str, Any]:
    """The ModelMetadata dict of base.py:174-188 (same defaults)."""
    return {
        "format_version": data.get("format_version", 2),
        "cutoff": data["cutoff"],
        "needs_coulomb": data.get("needs_coulomb", False),
        "needs_dispersion": data.get("needs_dispersion", False),
        "coulomb_mode": data.get("coulomb_mode", "none"),
        "coulomb_sr_rc": data.get("coulomb_sr_rc"),
        "coulomb_sr_envelope": data.get("coulomb_sr_envelope"),
        "d3_params": data.get("d3_params"),
        "has_embedded_lr": data.get("has_embedded_lr", False),
        "implemented_species": data.get("implemented_species", []),
        "family": data.get("family"),
        "supports_charged_systems": data.get("supports_charged_systems"),
        "has_embedded_d3ts": data.get("has_embedded_d3ts", False),
    }


def spec_from_config(config: Mapping[str, Any], state_dict: Mapping[str, Any], metadata: dict[str, Any], *,
                     source: str = "<artifact>", unexpected: str = "warn") -> ModelSpec:
    """YAML -> native architecture (the role of build_module config.py:154-202 + AIMNet2.__init__
    aimnet2.py:12-106), then state-dict check (utils.py:300-366) and fp64 shifts (utils.py:369-376)."""
    cls = config.get("class")
    if cls not in _MODEL_CLASSES:
        raise NotImplementedError(f"native engine implements {sorted(_MODEL_CLASSES)} only, artifact root class is {cls!r}")
    kw = dict(config.get("kwargs") or {})
    nq = int(kw.get("num_charge_channels", 1))
    if nq not in (1, 2):
        raise ValueError("num_charge_channels must be 1 (closed shell) or 2 (NSE for open-shell).")  # aimnet2.py:26-27
    if not kw.get("d2features", False):
        raise NotImplementedError("native engine implements d2features=True models only")
    aev = dict(kw.get("aev") or {})
    if aev.get("rc_v") is not None:
        raise NotImplementedError("dual-basis AEV (rc_v) is not implemented by the native engine")
    A, H = int(kw["nfeature"]), int(kw["ncomb_v"])
    G = int(aev.get("nshifts_s", 16))
    hidden = [list(h) for h in kw["hidden"]]
    aim_size = int(kw["aim_size"])
    n_conv_a, n_conv_q = A * G + A * H, nq * (G + H)  # ConvSV.output_size, aev.py:150-154
    n0 = n_conv_a + A * G
    n1 = n0 + n_conv_q + nq  # aimnet2.py:69
    mlp_dims, last_linear = [], []
    for p, h in enumerate(hidden):
        n_in = n0 if p == 0 else n1
        n_out = aim_size if p == len(hidden) - 1 else A * G + 2 * nq
        mlp_dims.append([n_in, *[int(x) for x in h if int(x) > 0], n_out])
        last_linear.append(p == 0)  # aimnet2.py:58-85
    outputs = kw.get("outputs") or {}
    if not isinstance(outputs, Mapping):
        raise NotImplementedError("native engine expects `outputs` as a mapping")
    head_dims: list[int] | None = None
    sr_coulomb, sr_rc, sr_env = False, 4.6, "exp"
    seen = []
    for name, cfg in outputs.items():
        c = cfg.get("class") if isinstance(cfg, Mapping) else None
        okw = dict((cfg.get("kwargs") or {})) if isinstance(cfg, Mapping) else {}
        seen.append(c)
        if c == "aimnet.modules.Output":
            if okw.get("key_in") != "aim" or okw.get("key_out") != "energy" or int(okw.get("n_out", 1)) != 1:
                raise NotImplementedError(f"output head {name!r}: only the aim -> energy head is implemented")
            mlp = dict(okw.get("mlp") or {})
            if not mlp.get("last_linear", True):
                raise NotImplementedError("energy head must end with a linear layer")
            head_dims = [int(okw["n_in"]), *[int(x) for x in mlp.get("hidden", []) if int(x) > 0], 1]
        elif c in ("aimnet.modules.AtomicShift", "aimnet.modules.AtomicSum"):
            if okw.get("key_in", "energy") != "energy" or okw.get("key_out", "energy") != "energy":
                raise NotImplementedError(f"{c} on keys other than energy is not implemented")
        elif c == "aimnet.modules.SRCoulomb":
            sr_coulomb = True
            sr_rc = float(okw.get("rc", 4.6))
            sr_env = okw.get("envelope", "exp")
            if sr_env not in ("exp", "cosine"):
                raise ValueError(f"Unknown envelope {sr_env}, must be 'exp' or 'cosine'")
        elif c in ("aimnet.modules.Dipole", "aimnet.modules.Quadrupole"):
            continue  # never returned by the calculator (keys_out, calculator.py:143)
        else:
            raise NotImplementedError(f"output module {c!r} ({name!r}) is not implemented by the native engine")
    if head_dims is None:
        raise ValueError("model_yaml has no energy Output head")
    want = [c for c in seen if c in ("aimnet.modules.Output", "aimnet.modules.AtomicShift", "aimnet.modules.AtomicSum")]
    if want[:3] != ["aimnet.modules.Output", "aimnet.modules.AtomicShift", "aimnet.modules.AtomicSum"]:
        raise NotImplementedError("native engine expects outputs in the order energy_mlp -> atomic_shift -> atomic_sum")

    # ---- state dict -> host weights --------------------------------------------------------
    def arr(key: str, shape: tuple[int, ...], dtype=np.float32) -> np.ndarray:
        if key not in state_dict:
            raise RuntimeError(f"Missing key(s) in state_dict loading {source}: {key!r}")
        t = state_dict[key]
        a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
        if tuple(a.shape) != tuple(shape):
            raise RuntimeError(f"size mismatch for {key}: artifact {tuple(a.shape)} vs model {tuple(shape)} ({source})")
        return np.ascontiguousarray(a, dtype=dtype)

    w: dict[str, np.ndarray] = {}
    used: set[str] = set()

    def take(key: str, shape, dtype=np.float32):
        w[key] = arr(key, shape, dtype)
        used.add(key)

    take("afv.weight", (64, A * G))
    take("conv_a.agh", (A, G, H))
    take("conv_q.agh", (nq, G, H))
    take("aev.rc_s", ())
    take("aev.eta_s", ())
    take("aev.shifts_s", (G,))
    for p, dims in enumerate(mlp_dims):
        for layer in range(len(dims) - 1):
            take(f"mlps.{p}.{2 * layer}.weight", (dims[layer + 1], dims[layer]))
            take(f"mlps.{p}.{2 * layer}.bias", (dims[layer + 1],))
    for layer in range(len(head_dims) - 1):
        take(f"outputs.energy_mlp.mlp.{2 * layer}.weight", (head_dims[layer + 1], head_dims[layer]))
        take(f"outputs.energy_mlp.mlp.{2 * layer}.bias", (head_dims[layer + 1],))
    take("outputs.atomic_shift.shifts.weight", (64, 1), np.float64)  # fp64 before load (base.py:85)
    optional = {"aev.rc_v", "aev.eta_v", "aev.shifts_v", "outputs.srcoulomb.rc"}
    extra = [k for k in state_dict if k not in used and k not in optional
             and not any(k.startswith(p) for p in _EXPECTED_UNEXPECTED_PREFIXES)]
    if extra:
        msg = f"Unexpected key(s) in state_dict loading {source}: {sorted(extra)}"
        if unexpected == "error":
            raise RuntimeError(msg)
        warnings.warn(msg, stacklevel=2)
    if "outputs.srcoulomb.rc" in state_dict and sr_coulomb:
        sr_rc = float(np.asarray(state_dict["outputs.srcoulomb.rc"].detach().cpu().numpy() if hasattr(state_dict["outputs.srcoulomb.rc"], "detach") else state_dict["outputs.srcoulomb.rc"]))
    spec = ModelSpec(
        nfeature=A, nshifts=G, ncomb_v=H, mlp_dims=mlp_dims, last_linear=last_linear, head_dims=head_dims,
        rc=float(w["aev.rc_s"].reshape(-1)[0]), eta=float(w["aev.eta_s"].reshape(-1)[0]),
        shifts=[float(s) for s in w["aev.shifts_s"]],
        sr_coulomb=sr_coulomb, sr_envelope=sr_env, sr_rc=sr_rc, weights=w, metadata=dict(metadata), num_charge_channels=nq,
    )
    return spec


def spec_from_artifact(data: Mapping[str, Any], *, policy: ImportPolicy | None = None, source: str = "<artifact>",
                       unexpected: str = "warn") -> ModelSpec:
    policy = policy or ImportPolicy()
    config, sd = validate_artifact(data, policy)
    return spec_from_config(config, sd, metadata_from_artifact(data), source=source, unexpected=unexpected)


def load_model(path: str, *, model_import_paths: Collection[str] | None = None, model_import_mode: str = "extend") -> tuple[ModelSpec, dict[str, Any]]:
    """load_model of aimnet/models/base.py:116-162: `.jpt` routes to TorchScript (not available
    natively), everything else is read exactly once with torch.load(weights_only=True)."""
    import torch

    import os

    if os.path.isdir(path):  # the reference's Hugging Face layout on disk: config.json + ensemble_N.safetensors
        return load_hf_dir(path, 0, model_import_paths=model_import_paths, model_import_mode=model_import_mode)
    policy = ImportPolicy(model_import_paths, model_import_mode)
    if str(path).lower().endswith(".jpt"):
        raise NotImplementedError("legacy TorchScript (.jpt) models cannot run on the native HIP engine; convert them to the v2 format")
    data = torch.load(path, map_location="cpu", weights_only=True)
    spec = spec_from_artifact(data, policy=policy, source=str(path))
    return spec, spec.metadata


def _srcoulomb_pairs(obj: Any) -> set[tuple[float, str]]:
    """Complete (rc, envelope) pairs of every SRCoulomb block of a parsed model YAML (hf_hub.py:95-118)."""
    pairs: set[tuple[float, str]] = set()
    if isinstance(obj, Mapping):
        cls = obj.get("class")
        if isinstance(cls, str) and cls.endswith("SRCoulomb"):
            kw = obj.get("kwargs", {})
            if isinstance(kw, Mapping) and kw.get("rc") is not None and kw.get("envelope") is not None:
                if not isinstance(kw["envelope"], str):
                    raise ValueError("SRCoulomb model_yaml field 'coulomb_sr_envelope' must be a supported string.")
                validate_metadata({"coulomb_sr_rc": kw["rc"], "coulomb_sr_envelope": kw["envelope"]}, require_cutoff=False, structural=False)
                pairs.add((float(kw["rc"]), kw["envelope"]))
        for v in obj.values():
            pairs |= _srcoulomb_pairs(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            pairs |= _srcoulomb_pairs(v)
    return pairs


def load_hf_dir(path: str, ensemble_member: int = 0, *, model_import_paths: Collection[str] | None = None,
                model_import_mode: str = "extend") -> tuple[ModelSpec, dict[str, Any]]:
    """A LOCAL directory in the reference's Hugging Face layout - `config.json` (metadata + `model_yaml`) next to
    `ensemble_<N>.safetensors` - as load_from_hf_repo reads it (aimnet/calculators/hf_hub.py:275-398): same order of checks
    (config root, member_names range, YAML import policy, SRCoulomb pair discovery / conflict rules of :121-147, metadata defaults of
    :174-192, structural validation, then the weights).  Repository ids (network download) and the family-level configs without
    `model_yaml` (registry fallback, :337-352) are out of scope here: there is no network on the GPU box."""
    import json
    import os

    if type(ensemble_member) is not int or ensemble_member < 0:
        raise ValueError("ensemble_member must be a non-boolean integer greater than or equal to zero.")
    if not os.path.isdir(path):
        raise NotImplementedError(f"{path!r} is not a local directory: Hugging Face repository ids need network access (out of scope)")
    policy = ImportPolicy(model_import_paths, model_import_mode)
    cfg_path = os.path.join(path, "config.json")
    if not os.path.exists(cfg_path):
        raise FileNotFoundError(f"config.json not found in {path}")
    with open(cfg_path) as f:
        config = json.load(f)
    if not isinstance(config, Mapping):
        raise TypeError("config.json root must be a mapping.")
    config = dict(config)
    validate_metadata(config, require_cutoff=False, structural=False)
    if "member_names" in config:
        names = config["member_names"]
        if not isinstance(names, list) or not names or any(not isinstance(n, str) for n in names):
            raise ValueError("config.json field 'member_names' must be a nonempty list of strings.")
        if ensemble_member >= len(names):
            raise ValueError(f"ensemble_member {ensemble_member} is out of range for config.json 'member_names' with {len(names)} entries.")
    model_yaml = config.get("model_yaml")
    if model_yaml is None:
        raise NotImplementedError("family-level config.json without 'model_yaml' needs the model registry (network): out of scope")
    model_config = parse_model_yaml(model_yaml, policy)
    pairs = _srcoulomb_pairs(model_config)
    if len(pairs) > 1:
        raise ValueError(f"ambiguous SRCoulomb definitions contain distinct parameter pairs: {sorted(pairs)!r}.")
    rc, env = next(iter(pairs), (None, None))
    found = rc is not None and env is not None
    if found and config.get("coulomb_sr_rc") is not None and float(config["coulomb_sr_rc"]) != rc:
        raise ValueError("config.json field 'coulomb_sr_rc' conflicts with the SRCoulomb value discovered in model_yaml.")
    if found and config.get("coulomb_sr_envelope") is not None and config["coulomb_sr_envelope"] != env:
        raise ValueError("config.json field 'coulomb_sr_envelope' conflicts with the SRCoulomb value discovered in model_yaml.")
    if (config.get("coulomb_mode", "none") == "sr_embedded" and not found
            and (config.get("coulomb_sr_rc") is None or config.get("coulomb_sr_envelope") is None)):
        raise ValueError("sr_embedded metadata with an omitted Coulomb field requires exactly one distinct complete "
                         "SRCoulomb parameter pair in model_yaml.")
    if config.get("coulomb_sr_rc") is None and found:
        config["coulomb_sr_rc"] = rc
    if config.get("coulomb_sr_envelope") is None and found:
        config["coulomb_sr_envelope"] = env
    validate_metadata(config, require_cutoff=True, structural=False)
    fv = config.get("format_version", 2)
    if type(fv) is not int or fv != 2:
        raise ValueError("HF model metadata field 'format_version' must be integer 2.")
    metadata = metadata_from_artifact(config)
    validate_metadata(metadata, require_cutoff=True, structural=True)
    if _has_d3ts(model_config) != bool(metadata.get("has_embedded_d3ts", False)):
        raise ValueError("model metadata field 'has_embedded_d3ts' disagrees with D3TS presence in model_yaml.")
    st_name = f"ensemble_{ensemble_member}.safetensors"
    st_path = os.path.join(path, st_name)
    if not os.path.exists(st_path):
        raise FileNotFoundError(f"{st_name} not found in {path}")
    try:
        from safetensors.torch import load_file
    except ImportError as exc:  # pragma: no cover
        raise ImportError("Hugging Face format support requires the `safetensors` package") from exc
    sd = load_file(st_path, device="cpu")
    spec = spec_from_config(model_config, sd, metadata, source=st_path, unexpected="warn")
    return spec, spec.metadata


def load_dftd3_tables(source: Any = None) -> dict[str, np.ndarray]:
    """DFT-D3 reference tables as {c6ab, cn_ref: f32[Z,Z,5,5], rcov, r4r2: f32[Z]} - the data the reference reads from
    aimnet/dftd3_data.pt in DFTD3.__init__ (lr.py:1405-1423; packed legacy layout c6ab[...,3] handled as in
    DFTD3._load_from_state_dict, lr.py:1441-1468).  `source`: a mapping, a path to .npz / .pt, or None = the file named
    by $AIMNET_DFTD3_DATA, else `dftd3_data.pt` of an installed `aimnet` package.  The tables are data of the D3 method;
    this package does not ship a copy."""
    import importlib.util
    import os

    if source is None:
        source = os.environ.get("AIMNET_DFTD3_DATA")
    if source is None:
        spec = importlib.util.find_spec("aimnet")
        if spec is not None and spec.submodule_search_locations:
            cand = os.path.join(list(spec.submodule_search_locations)[0], "dftd3_data.pt")
            if os.path.exists(cand):
                source = cand
    if source is None:
        raise FileNotFoundError("DFT-D3 reference tables not found: pass dftd3_data=..., set AIMNET_DFTD3_DATA to a "
                                "dftd3_data.pt / .npz file, or install the `aimnet` package that ships it")
    if isinstance(source, (str, os.PathLike)):
        path = os.fspath(source)
        if path.endswith(".npz"):
            with np.load(path) as z:
                raw = {k: z[k] for k in z.files}
        else:
            import torch

            raw = {k: (v.numpy() if hasattr(v, "numpy") else np.asarray(v)) for k, v in torch.load(path, map_location="cpu", weights_only=True).items()}
    else:
        raw = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in dict(source).items()}
    c6ab = np.asarray(raw["c6ab"], dtype=np.float32)
    if c6ab.ndim == 5:
        cn_ref = c6ab[..., 1]
        c6ab = c6ab[..., 0]
    else:
        cn_ref = np.asarray(raw["cn_ref"], dtype=np.float32)
    out = {"c6ab": np.ascontiguousarray(c6ab), "cn_ref": np.ascontiguousarray(cn_ref, dtype=np.float32),
           "rcov": np.ascontiguousarray(raw["rcov"], dtype=np.float32), "r4r2": np.ascontiguousarray(raw["r4r2"], dtype=np.float32)}
    nz = out["rcov"].shape[0]
    if out["c6ab"].shape != (nz, nz, 5, 5) or out["cn_ref"].shape != (nz, nz, 5, 5) or out["r4r2"].shape != (nz,):
        raise ValueError("malformed DFT-D3 tables: expected c6ab/cn_ref [Z,Z,5,5] and rcov/r4r2 [Z]")
    return out


def synthetic_spec(seed: int = 0, num_charge_channels: int = 1, rxn: bool = False, sr_envelope: str = "exp",
                   sr_rc: float = 4.6, cold: bool = False) -> ModelSpec:
    """ModelSpec of the deterministic synthetic aimnet2 artifact (aimnetcentral_amd/synth.py); num_charge_channels = 2
    gives the open-shell NSE shape, rxn = True the `aimnet2_rxn` YAML (Dipole / Quadrupole output modules), sr_envelope / sr_rc
    the SRCoulomb block's envelope ("exp" or "cosine") and radius, cold the low-force variant of the seed (synth._COLD_GAINS)."""
    from . import synth

    return spec_from_artifact(synth.synthetic_artifact(seed, num_charge_channels=num_charge_channels, rxn=rxn,
                                                       sr_envelope=sr_envelope, sr_rc=sr_rc, cold=cold),
                              source=f"<synthetic seed {seed}{' cold' if cold else ''}>")
