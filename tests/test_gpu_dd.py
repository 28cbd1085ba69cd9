"""Spatial domain decomposition of one periodic system (aimnetcentral_amd/dd.py, include/aimnet_hip.h aimnet_engine_set_dd;
SURVEY.md 8f next-4): ranks that share the one GPU of the box over gloo evaluate their slabs (owned + 15 A halo as a
non-periodic cluster, NSE sums and halo charges through the engine's exchange function, reverse halo exchange of the partial
forces) and together reproduce the single-rank periodic evaluation - and the unmodified reference's golden - at the
reference's literal gates (tests/test_calculator_gpu.py:137,445,464 of the reference)."""
from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(case: str, world: int, port: int, tmp_path, backend: str = "gloo", grid: str = "", **env) -> dict:
    out = os.path.join(str(tmp_path), f"dd_{case}_{world}.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dd_worker.py"), case, out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, DD_BACKEND=backend, DD_GRID=grid, **env))
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    with open(out) as fh:
        return json.load(fh)


def _gates(rec: dict, key: str, n: int, dq: float = 1e-4, strict: bool = True):
    """strict: the reference's literal gates.  Otherwise (two fp32 evaluations of one surface whose pair vectors were rounded
    differently - image positions of magnitude 30 A in the cluster, in-cell positions plus a shift vector in the periodic engine)
    at most one force component in a thousand may sit outside allclose(1e-4, 1e-5), and none by more than a factor 2."""
    c = rec[key]
    assert c["dE"] <= max(1e-5, 5e-7 * n), (key, c)
    if strict:
        assert c["dF_violations"] == 0, (key, c)
    else:
        assert c["dF_violations"] <= 3e-3 * n and c["dF_worst_ratio"] <= 2.0, (key, c)
    assert c["dq_max"] <= dq, (key, c)
    assert c["ds_max"] <= 1e-5, (key, c)  # stress: the ranks' virial shares summed, over the cell volume


def test_two_slabs_reproduce_the_reference_golden_2304(tmp_path):
    """The 2 304-atom jittered crystal of tests/golden/coldw_big.npz (outputs of the unmodified reference): two ranks, E + F + q +
    stress."""
    rec = _run("golden2304", 2, 29541, tmp_path)
    assert rec["ranks_agree"] and rec["repeat_bitwise"] and rec["owned_total"] == rec["n_atoms"] == 2304
    assert rec["n_local"] > rec["n_owned"] > 0
    # per evaluation: NSE forward sums of passes 0, 1 + their adjoint sums = 4 all-reduces, one charge exchange
    assert rec["exchange_calls"] == {"0": 4, "1": 1, "2": 0}
    _gates(rec, "vs_single_rank", 2304)
    _gates(rec, "vs_reference_golden", 2304)


@pytest.mark.parametrize("world,port", [(2, 29542), (3, 29543)])
def test_near_cubic_cell_with_atoms_outside_the_box(world, port, tmp_path):
    rec = _run("cube1536", world, port, tmp_path)
    assert rec["ranks_agree"] and rec["owned_total"] == rec["n_atoms"] == 1536
    _gates(rec, "vs_single_rank", 1536, strict=False)


def test_without_coulomb(tmp_path):
    rec = _run("cube1536_nocoul", 2, 29544, tmp_path)
    _gates(rec, "vs_single_rank", 1536, strict=False)


def test_two_charge_channels_charged_cell(tmp_path):
    """Open-shell NSE model (two charge channels, hot weights), total charge +1: one forward all-reduce per channel and pass, one
    (both channels) per pass for the adjoint sums; two charge planes in the one charge exchange."""
    rec = _run("cube1536_nse", 2, 29545, tmp_path)
    assert rec["exchange_calls"] == {"0": 6, "1": 1, "2": 0}
    c = rec["vs_single_rank"]
    # hot synthetic weights: two fp32 evaluations of the same surface in different summation orders (noise floor)
    assert c["dE"] <= 5e-3 and c["dF_max"] <= 2e-4 * max(1.0, c["F_max"]) and c["dq_max"] <= 1e-4 and c["ds_max"] <= 1e-4, c


@pytest.mark.parametrize("case,port", [("cube1536_d3", 29546), ("cube1536_d3rc12", 29547)])
def test_dftd3_weights_and_dEdcn_come_from_the_owners(case, port, tmp_path):
    """External DFT-D3(BJ) (the shipped models run with it): a halo copy's coordination number needs ITS 15 A neighbourhood, so
    the per-atom reference weights and dE/dcn of halo rows are taken from their owners (AIMNET_DD_ROWS, two more exchanges) - with
    one matrix shared with DSF (cutoff 15 A) and with a D3 matrix of its own (12 A, coordination numbers formed by the list build)."""
    rec = _run(case, 2, port, tmp_path)
    assert rec["ranks_agree"] and rec["repeat_bitwise"] and rec["exchange_calls"] == {"0": 4, "1": 1, "2": 2}
    _gates(rec, "vs_single_rank", 1536, strict=False)


def test_one_rank_over_rccl_exercises_the_device_side_exchanges(tmp_path):
    """RCCL refuses two ranks on one device, so the `nccl` branch of the exchanges (all-reduce on views of the engine's workspace,
    ordered on the engine's stream, no host copy) runs here with ONE rank: a single slab whose halo holds the cell's own images."""
    rec = _run("cube1536", 1, 29548, tmp_path, backend="nccl")
    assert rec["backend"] == "nccl" and rec["world"] == 1 and rec["owned_total"] == 1536 and rec["n_local"] > 1536
    assert rec["exchange_calls"] == {"0": 4, "1": 1, "2": 0} and rec["repeat_bitwise"]
    _gates(rec, "vs_single_rank", 1536, strict=False)


def test_four_ranks_as_bricks(tmp_path):
    """2 x 1 x 2 bricks instead of slabs (the form that keeps the halo fraction down on near-cubic cells): four ranks on the one
    GPU, the reference's 2 304-atom golden at the literal gates."""
    rec = _run("golden2304", 4, 29549, tmp_path, grid="2,1,2")
    assert rec["ranks_agree"] and rec["owned_total"] == 2304 and rec["grid"] == [2, 1, 2]
    _gates(rec, "vs_single_rank", 2304)
    _gates(rec, "vs_reference_golden", 2304)


def test_calculator_surface(tmp_path):
    """`AIMNet2Calculator.set_domain_decomposition(True)`: the reference's dict-in / dict-out call on every rank returns the same
    shapes and (at the gates) the same numbers as the undecomposed calculator; a batch is refused; switching it off restores the
    single-rank path bit for bit."""
    rec = _run("calculator", 2, 29550, tmp_path)
    assert rec["shapes"] == rec["ref_shapes"] and rec["shapes"]["energy"] == [1] and rec["shapes"]["stress"] == [3, 3]
    assert rec["dE"] <= max(1e-5, 5e-7 * 1536) and rec["dF_violations"] <= 4 and rec["dF_worst_ratio"] <= 2.0, rec
    assert rec["dq_max"] <= 1e-4 and rec["ds_max"] <= 1e-5, rec
    assert rec["batch_refused"] and "domain decomposition" in rec["batch_refused"] and rec["off_again_bitwise"] and rec["host_out_on_cpu"]


def test_row_overflow_on_one_rank_is_repeated_by_all(tmp_path):
    """Rank 0 starts with neighbour rows that are too short (16 short-range, 64 long-range entries): the status words are max-reduced
    over the group, every rank grows its capacities and repeats the evaluation together - no rank is left in a collective."""
    rec = _run("cube1536", 2, 29551, tmp_path, DD_FORCE_OVERFLOW="1")
    assert rec["max_nb_after"] > 16 and rec["ranks_agree"]
    _gates(rec, "vs_single_rank", 1536, strict=False)
