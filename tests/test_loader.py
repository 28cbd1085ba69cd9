"""Accept / reject behaviour of the v2 artifact loader (mirrors the reference's
tests/test_model_artifact_security.py and tests/test_model.py:276-316 expectations)."""
from __future__ import annotations

import copy

import numpy as np
import pytest
import torch

from aimnetcentral_amd import loader, synth


@pytest.fixture()
def art():
    return synth.synthetic_artifact(0)


def test_roundtrip_through_torch_save(tmp_path, art):
    p = tmp_path / "m.pt"
    torch.save(art, p)
    spec, meta = loader.load_model(str(p))
    assert spec.mlp_dims == [[704, 512, 380, 258], [733, 512, 380, 258], [733, 512, 380, 380, 256]]
    assert spec.last_linear == [True, False, False]
    assert spec.head_dims == [256, 128, 128, 1]
    assert spec.sr_coulomb and spec.sr_envelope == "exp" and abs(spec.sr_rc - 4.6) < 1e-6
    assert meta["coulomb_mode"] == "sr_embedded" and meta["needs_coulomb"] is True and meta["cutoff"] == 5.0
    assert spec.weights["outputs.atomic_shift.shifts.weight"].dtype == np.float64
    assert abs(spec.eta - (16 / 4.2) ** 2) < 1e-4 and len(spec.shifts) == 16
    assert np.isnan(spec.weights["afv.weight"][2]).all() and (spec.weights["afv.weight"][0] == 0).all()


def test_synthetic_weights_are_deterministic():
    a = synth.state_dict_digest(synth.synthetic_state_dict(0))
    b = synth.state_dict_digest(synth.synthetic_state_dict(0))
    c = synth.state_dict_digest(synth.synthetic_state_dict(1))
    assert a == b and a != c


@pytest.mark.parametrize("mutate,err", [
    (lambda a: a.update(model_yaml=""), ValueError),
    (lambda a: a.update(model_yaml="- a\n- b\n"), ValueError),
    (lambda a: a.update(model_yaml="x: &a [*a]\n"), ValueError),
    (lambda a: a.update(format_version=3), ValueError),
    (lambda a: a.update(format_version=2.0), ValueError),
    (lambda a: a.pop("cutoff"), ValueError),
    (lambda a: a.update(cutoff=float("nan")), ValueError),
    (lambda a: a.update(needs_coulomb=1), ValueError),
    (lambda a: a.update(coulomb_mode="weird"), ValueError),
    (lambda a: a.update(coulomb_sr_rc=None), ValueError),
    (lambda a: a.update(coulomb_sr_rc=6.0), ValueError),
    (lambda a: a.update(has_embedded_lr=False), ValueError),
    (lambda a: a.update(implemented_species=[1, -6]), ValueError),
    (lambda a: a.update(state_dict=[1, 2]), ValueError),
    (lambda a: a["state_dict"].update(bad=3), ValueError),
    (lambda a: a.update(has_embedded_d3ts=True), ValueError),
])
def test_rejects_malformed_artifacts(art, mutate, err):
    mutate(art)
    with pytest.raises(err):
        loader.spec_from_artifact(art)


def test_rejects_non_dict_payload():
    with pytest.raises(ValueError):
        loader.spec_from_artifact([1, 2, 3])


@pytest.mark.parametrize("needle,repl", [
    ("class: aimnet.modules.Output", "class: os.system"),
    ("activation_fn: torch.nn.GELU", "activation_fn: torch.nn.ReLU"),
    ("key_in: aim", "key_in: aim\n        fn: os.system"),
    ("rc: 4.6", "rc: 4.6\n        ptfile: /etc/passwd"),
])
def test_rejects_untrusted_yaml(art, needle, repl):
    assert needle in art["model_yaml"]
    art["model_yaml"] = art["model_yaml"].replace(needle, repl, 1)
    with pytest.raises(ValueError):
        loader.spec_from_artifact(art)


def test_import_policy_modes(art):
    with pytest.raises(ValueError):
        loader.ImportPolicy(["x.y"], "unsafe")
    with pytest.raises(ValueError):
        loader.ImportPolicy(None, "replace")
    with pytest.raises(ValueError):
        loader.ImportPolicy(None, "bogus")
    with pytest.raises(TypeError):
        loader.ImportPolicy("aimnet.models.AIMNet2", "extend")
    pol = loader.ImportPolicy(["my_pkg.*"], "extend")
    pol.require_allowed("my_pkg.layers.Foo", "class")
    with pytest.raises(ValueError):
        pol.require_allowed("other.Foo", "class")
    # an allow-listed but natively unimplemented class is a NotImplementedError, not a silent skip
    art["model_yaml"] = art["model_yaml"].replace("class: aimnet.modules.SRCoulomb", "class: aimnet.modules.lr.DispParam")
    with pytest.raises(NotImplementedError):
        loader.spec_from_artifact(art)


def test_missing_and_unexpected_state_dict_keys(art):
    a = copy.deepcopy(art)
    del a["state_dict"]["mlps.1.2.weight"]
    with pytest.raises(RuntimeError, match="Missing key"):
        loader.spec_from_artifact(a)
    a = copy.deepcopy(art)
    a["state_dict"]["mlps.0.0.weight"] = torch.zeros(3, 3)
    with pytest.raises(RuntimeError, match="size mismatch"):
        loader.spec_from_artifact(a)
    a = copy.deepcopy(art)
    a["state_dict"]["surprise.weight"] = torch.zeros(1)
    with pytest.warns(UserWarning, match="Unexpected key"):
        loader.spec_from_artifact(a)
    with pytest.raises(RuntimeError):
        loader.spec_from_artifact(a, unexpected="error")
    a = copy.deepcopy(art)  # keys the calculator re-creates externally are expected leftovers
    a["state_dict"]["outputs.lrcoulomb.rc"] = torch.tensor(4.6)
    loader.spec_from_artifact(a)


def test_jpt_is_loud_and_nse_shapes(tmp_path, art):
    with pytest.raises(NotImplementedError):
        loader.load_model(str(tmp_path / "legacy.JPT"))
    # a 1-channel state dict under a 2-channel YAML is a size mismatch, as torch's load_state_dict reports it
    art["model_yaml"] = art["model_yaml"].replace("  aim_size: 256", "  aim_size: 256\n  num_charge_channels: 2")
    with pytest.raises(RuntimeError, match="size mismatch for conv_q.agh"):
        loader.spec_from_artifact(art)
    art["model_yaml"] = art["model_yaml"].replace("num_charge_channels: 2", "num_charge_channels: 3")
    with pytest.raises(ValueError, match="num_charge_channels must be 1"):
        loader.spec_from_artifact(art)
    # the open-shell NSE family (aimnet2.py:21-28,53-85): MLP rows [a | conv_a | q(2) | conv_q(2 x 28)] = 762, outputs 256 + 4
    spec = loader.synthetic_spec(0, num_charge_channels=2)
    assert spec.num_charge_channels == 2
    assert [d[0] for d in spec.mlp_dims] == [704, 762, 762] and [d[-1] for d in spec.mlp_dims] == [260, 260, 256]
    assert spec.weights["conv_q.agh"].shape == (2, 16, 12)


def test_runtime_metadata_rules():
    md = loader.metadata_from_artifact(synth.synthetic_artifact(0))
    loader.validate_runtime_metadata(md, needs_coulomb=True, needs_dispersion=False)
    with pytest.raises(ValueError, match="d3_params"):
        loader.validate_runtime_metadata(md, needs_coulomb=True, needs_dispersion=True)
    md2 = dict(md, coulomb_mode="full_embedded")
    with pytest.raises(ValueError, match="full_embedded"):
        loader.validate_runtime_metadata(md2, needs_coulomb=True, needs_dispersion=False)


# ---- the reference's Hugging Face layout on disk: config.json + ensemble_N.safetensors (hf_hub.py:275-398) --------------
def _write_hf_dir(tmp_path, art, drop=(), extra=None, member=0):
    import json

    from safetensors.torch import save_file

    d = tmp_path / "hf_model"
    d.mkdir()
    cfg = {k: v for k, v in art.items() if k != "state_dict" and k not in drop}
    cfg.update(extra or {})
    (d / "config.json").write_text(json.dumps(cfg))
    save_file({k: v.contiguous() for k, v in art["state_dict"].items()}, str(d / f"ensemble_{member}.safetensors"))
    return d


def test_hf_directory_loads_like_the_pt_artifact(tmp_path, art):
    d = _write_hf_dir(tmp_path, art)
    spec, meta = loader.load_hf_dir(str(d))
    ref = loader.spec_from_artifact(art)
    assert spec.mlp_dims == ref.mlp_dims and spec.head_dims == ref.head_dims and spec.last_linear == ref.last_linear
    assert spec.sr_envelope == ref.sr_envelope and spec.sr_rc == ref.sr_rc and meta == ref.metadata
    assert set(spec.weights) == set(ref.weights)
    for k in ref.weights:
        assert spec.weights[k].dtype == ref.weights[k].dtype and np.array_equal(spec.weights[k], ref.weights[k], equal_nan=True), k
    spec2, _ = loader.load_model(str(d))  # a directory routes through load_model as well
    assert np.array_equal(spec2.weights["afv.weight"], ref.weights["afv.weight"], equal_nan=True)


def test_hf_directory_fills_sr_coulomb_fields_from_the_yaml(tmp_path, art):
    d = _write_hf_dir(tmp_path, art, drop=("coulomb_sr_rc", "coulomb_sr_envelope"))
    spec, meta = loader.load_hf_dir(str(d))
    assert meta["coulomb_sr_rc"] == 4.6 and meta["coulomb_sr_envelope"] == "exp" and spec.sr_coulomb


@pytest.mark.parametrize("extra,member,exc,msg", [
    ({"coulomb_sr_rc": 4.0}, 0, ValueError, "conflicts with the SRCoulomb value"),
    ({"coulomb_sr_envelope": "cosine"}, 0, ValueError, "conflicts with the SRCoulomb value"),
    ({"member_names": ["a"]}, 1, ValueError, "out of range"),
    ({"member_names": []}, 0, ValueError, "nonempty list of strings"),
    ({"format_version": 1}, 0, ValueError, "must be integer 2"),
    ({"model_yaml": None}, 0, NotImplementedError, "registry"),
    ({}, 3, FileNotFoundError, "ensemble_3.safetensors"),
])
def test_hf_directory_rejections(tmp_path, art, extra, member, exc, msg):
    d = _write_hf_dir(tmp_path, art, extra=extra)
    with pytest.raises(exc, match=msg):
        loader.load_hf_dir(str(d), member)


def test_hf_repo_ids_need_network(tmp_path):
    with pytest.raises(NotImplementedError, match="network"):
        loader.load_hf_dir("isayevlab/aimnet2-wb97m-d3")
    with pytest.raises(FileNotFoundError, match="config.json"):
        loader.load_hf_dir(str(tmp_path))
