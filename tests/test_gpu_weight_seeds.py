"""The bf16x3-split GEMMs on OTHER weight seeds than the one every other parity test uses.

The second accumulation phase of the split kernels runs sign-flipped to cancel the one-signed truncation bias of the bf16 MFMA
accumulation, and the flip point (0.56 K) was chosen on end-to-end energies of weight seed 0.  Here the fixtures the oracle can
regenerate on the GPU box run on weight seeds 1-4 (seed 0 is everywhere else), in GEMM mode 1 (default: split kernels above 256
rows - with pre-split activations and the fused head where the batch is large enough) and mode 2 (split kernels for every batch
size), at the UNCHANGED gates of tests/test_gpu_parity.py / test_gpu_configs.py (the reference's own: tests/test_calculator_gpu.py
:445,464): a small ragged charged batch (batch5 shape), a 2 304-atom periodic cell with DSF + stress, and one rank's shard of
config 5 (128 x 50 atoms) with its rms-over-the-batch distance from the fp64 oracle against the fp32 oracle's own (<= 1.5x).

AIMNET_SEED_TABLE=<file>: the per-seed numbers are also written there as JSON lines (profiles/r4_weight_seeds.jsonl)."""
from __future__ import annotations

import json
import os

import numpy as np
import pytest
import torch

from conftest import CHARGE_ATOL, STRESS_ATOL, assert_forces_close, energy_tol
from aimnetcentral_amd import workloads
from oracle import aimnet2_oracle as O
from test_gpu_configs import check_batch

pytestmark = pytest.mark.gpu

SEEDS = (1, 2, 3, 4)


def _record(**kw):
    path = os.environ.get("AIMNET_SEED_TABLE")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(kw) + "\n")


@pytest.fixture(scope="module", params=SEEDS, ids=[f"seed{s}" for s in SEEDS])
def seeded(request):
    """Engine + fp32 / fp64 oracle of one weight seed, and the oracle's answers on the three inputs (computed once per seed)."""
    from aimnetcentral_amd import loader, synth
    from aimnetcentral_amd.engine import HipEngine

    seed = request.param
    sd = synth.synthetic_state_dict(seed)
    o32, o64 = O.OracleModel(sd, torch.float32), O.OracleModel(sd, torch.float64)
    eng = HipEngine(loader.synthetic_spec(seed), "cuda:0")
    inputs = {}
    # (a) ragged charged batch of the batch5 shape
    c, z, mol, q = workloads.random_batch(5, 11, 40, seed=100 + seed)
    q = np.array([0.0, 1.0, -1.0, 0.0, 2.0], dtype=np.float32)
    inputs["batch5"] = dict(c=c, z=z, mol=mol, q=q, cell=None, kw=dict(coulomb="simple"),
                            ref=O.evaluate(o32, c, z, q, mol, coulomb="simple"),
                            e64=O.evaluate(o64, c, z, q, mol, coulomb="simple", forces=False)["energy"])
    # (b) 2 304 atoms of the config-3 crystal, DSF (9 A keeps the CPU oracle at seconds), forces + stress
    c, z, cell = workloads.glucose_supercell((2, 3, 4))
    rng = np.random.default_rng(7 + seed)
    c = (c + rng.normal(0.0, 0.02, c.shape)).astype(np.float32)
    cell = cell.astype(np.float32)
    mol = np.zeros(len(z), dtype=np.int64)
    q = np.zeros(1, dtype=np.float32)
    kw = dict(coulomb="dsf", stress=True, dsf_rc=9.0, dsf_alpha=0.2)
    pbc = np.ones(3, dtype=bool)
    c = O.wrap_into_cell(c, cell, mol, pbc)  # both sides start from the same wrapped coordinates; lists by the k-d tree builder
    nb, sh = O.neighbor_list_fast(c, 5.0, mol, cell.astype(np.float64), pbc)
    nbl, shl = O.neighbor_list_fast(c, 9.0, mol, cell.astype(np.float64), pbc)
    lists = dict(nbmat=nb, shifts=sh, nbmat_lr=nbl, shifts_lr=shl)
    inputs["pbc2304"] = dict(c=c, z=z, mol=mol, q=q, cell=cell, kw=kw,
                             ref=O.evaluate(o32, coord=c, numbers=z, charge=q, mol_idx=mol, cell=cell, **kw, **lists),
                             e64=O.evaluate(o64, coord=c, numbers=z, charge=q, mol_idx=mol, cell=cell, forces=False, coulomb="dsf",
                                            dsf_rc=9.0, dsf_alpha=0.2, **lists)["energy"])
    # (c) one rank's shard of config 5
    from aimnetcentral_amd import dist as adist

    c, z, mol, q = workloads.random_batch(1024, 50, 50, seed=5)
    a, b = adist.shard_frames(np.bincount(mol, minlength=1024), 8)[0]
    c, z, mol, q = adist.local_batch(c, z, mol, q, a, b)
    inputs["cfg5"] = dict(c=c, z=z, mol=np.asarray(mol), q=q, cell=None, kw=dict(coulomb="simple"),
                          ref=O.evaluate(o32, c, z, q, mol, coulomb="simple", return_intermediates=True),
                          e64=O.evaluate(o64, c, z, q, mol, coulomb="simple", forces=False, return_intermediates=True))
    return seed, eng, inputs


def _eval(eng, inp):
    dev = eng.device
    cell = torch.from_numpy(inp["cell"]).to(dev) if inp["cell"] is not None else None
    r = eng.eval(torch.from_numpy(inp["c"]).to(dev), torch.from_numpy(np.asarray(inp["z"])).to(dev), torch.from_numpy(np.asarray(inp["mol"])).to(dev),
                 torch.from_numpy(inp["q"]).to(dev), cell=cell, forces=True, **inp["kw"])
    return {k: v.cpu().numpy() for k, v in r.items()}


@pytest.mark.parametrize("mode", [1, 2], ids=["bf3_default", "bf3_every_size"])
def test_fixtures_on_other_weight_seeds(seeded, mode):
    seed, eng, inputs = seeded
    eng.set_option("gemm_bf3", mode)
    try:
        # (a) small ragged batch: energy gate widened by the fp32 oracle's own distance from fp64 (as test_gpu_parity.compare does)
        inp = inputs["batch5"]
        r, ref = _eval(eng, inp), inp["ref"]
        sizes = np.bincount(inp["mol"])
        slack = np.abs(ref["energy"] - inp["e64"])
        err = np.abs(r["energy"] - ref["energy"])
        assert (err <= energy_tol(sizes) + slack).all(), f"seed {seed} batch5 energy {err.max():.3e}"
        assert_forces_close(r["forces"], ref["forces"], f"seed {seed} batch5")
        assert np.abs(r["charges"] - ref["charges"]).max() <= CHARGE_ATOL
        _record(seed=seed, mode=mode, fixture="batch5", dE=float(err.max()), gate=float(energy_tol(sizes)), slack_fp64=float(slack.max()),
                dF=float(np.abs(r["forces"] - ref["forces"]).max()))
        # (b) 2 304-atom periodic cell: the plain gate max(1e-5, 5e-7 n) + the fp32 oracle's own distance from fp64
        inp = inputs["pbc2304"]
        r, ref = _eval(eng, inp), inp["ref"]
        n = len(inp["z"])
        slack = abs(float(ref["energy"][0] - inp["e64"][0]))
        err = abs(float(r["energy"][0] - ref["energy"][0]))
        assert err <= energy_tol(n) + slack, f"seed {seed} pbc2304 energy {err:.3e}"
        assert_forces_close(r["forces"], ref["forces"], f"seed {seed} pbc2304")
        assert np.abs(r["charges"] - ref["charges"]).max() <= CHARGE_ATOL
        assert np.abs(r["stress"] - ref["stress"]).max() <= STRESS_ATOL
        _record(seed=seed, mode=mode, fixture="pbc2304", dE=err, gate=float(energy_tol(n)), slack_fp64=slack,
                dE_vs_fp64=abs(float(r["energy"][0] - inp["e64"][0])), dF=float(np.abs(r["forces"] - ref["forces"]).max()),
                dstress=float(np.abs(r["stress"] - ref["stress"]).max()))
        # (c) config-5 shard: rms distance from fp64 <= 1.5 x the fp32 oracle's, per-molecule gates (test_gpu_configs.check_batch)
        inp = inputs["cfg5"]
        r = _eval(eng, inp)
        check_batch(r["energy"], r["forces"], r["charges"], inp["ref"], inp["e64"], inp["mol"], f"seed {seed} config 5 shard")
        e64 = inp["e64"]["energy"]
        rms = float(np.sqrt(np.mean((r["energy"] - e64) ** 2)))
        rms_ref = float(np.sqrt(np.mean((inp["ref"]["energy"] - e64) ** 2)))
        _record(seed=seed, mode=mode, fixture="cfg5_shard", rms_hip_vs_fp64=rms, rms_fp32_oracle_vs_fp64=rms_ref, rms_ratio=rms / rms_ref)
    finally:
        eng.set_option("gemm_bf3", 1)
