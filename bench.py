#!/usr/bin/env python
"""bench.py - atoms*steps/s of AIMNet2 energy+forces(+stress) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload pbc10k|md1024|batch256|taxol]

A "step" is one full evaluation of the hot path (neighbour lists rebuilt, 3 message passing
passes, energy head, Coulomb, analytic force/virial backward) on inputs already resident in HBM.

Default workload (BASELINE.json configs[2], the configuration the north-star target is quoted
on): the 2019828.cif allose crystal, (7,3,5) supercell = 10 080 atoms, periodic, DSF Coulomb
(Rc 15 A, alpha 0.2), energy + forces + stress, synthetic weights (seed 0) of the real aimnet2
architecture.  With --gpus N every rank evaluates its own independent frame (batch sharding,
weak scaling) and the per-frame energies are all-gathered over RCCL each step; --workload md1024
is BASELINE configs[4] (1024 frames x 50 atoms, 128 frames per GPU).

The K timed steps are enqueued back to back; their neighbour-overflow status words stay on the
device and are all verified after the closing barrier (warm-up steps run with the per-step check).

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the eval stream
around the dominant kernel family (the MLP GEMMs) on every 4th timed step: algorithmic fp32 GEMM
FLOPs per step / GEMM milliseconds per step.  The GEMMs run with bf16x3-split operands on the bf16
matrix pipe (six bf16 products per fp32 product, fp32 accumulation: csrc/gemm_bf3a.hip / gemm_bf3.hip), so the roof
of that kernel is 2.5 PFLOP/s / 6 = 416.7 TFLOP/s of fp32 work (`peak`); the fraction of the 157.3
TFLOP/s fp32-matrix peak is reported beside it.  `exact_f32` is the same measurement with the
exact-fp32 MFMA kernels (engine option gemm_bf3 = 0), taken after the timed region.  `parity`
compares the engine with the CPU oracle on the samples the CPU baseline is timed on (untimed).
`cpu_baseline` times the oracle (torch-CPU eager restatement of the reference op sequence, kind
"port") on the host cores on a bounded sample of the same workload, rank 0 at N=1 only.
`hessian_config4` (N = 1): BASELINE configs[3], the 40-atom dense Hessian by the analytic tangent sweep against the reference's
golden at the reference's own gate, with its cost in force evaluations.  With
--gpus N > 1 the line also carries `scaling_md1024`: BASELINE configs[4] (128 frames x 50 atoms per
GPU, per-frame energies all-gathered over RCCL every step) timed in the same run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_BF16_DENSE_TFLOPS = 2500.0  # MI355X_MICROARCH.md, dense bf16 MFMA
PEAK_BF16X3_TFLOPS = PEAK_BF16_DENSE_TFLOPS / 6.0  # six bf16 products per fp32 product
PEAK_F16X2_TFLOPS = PEAK_BF16_DENSE_TFLOPS / 3.0   # three fp16 products per fp32 product (fp16 MFMA = the bf16 rate)


def build_workload(name: str, rank: int, world: int):
    """Returns dict(coord, numbers, mol_idx, charge, cell|None, coulomb, stress, frames, label)."""
    from aimnetcentral_amd import dist as adist
    from aimnetcentral_amd import workloads

    if name == "pbc10k":
        c, z, cell = workloads.glucose_supercell((7, 3, 5))
        rng = np.random.Generator(np.random.PCG64(1000 + rank))
        c = c + rng.standard_normal(c.shape) * 0.02  # independent thermal-like jitter per frame
        return dict(coord=c.astype(np.float32), numbers=z, mol_idx=np.zeros(len(z), dtype=np.int64),
                    charge=np.zeros(1, dtype=np.float32), cell=cell.astype(np.float32), coulomb="dsf", stress=True,
                    frames=1, label="aimnet2 periodic: 2019828.cif (7,3,5) supercell 10080 atoms, DSF 15A, E+F+stress")
    if name in ("md1024", "batch256"):
        if name == "md1024":
            n_frames, lo, hi, seed = 128 * world, 50, 50, 5
        else:
            n_frames, lo, hi, seed = 256 * world, 20, 60, 2
        c, z, mol, q = workloads.random_batch(n_frames, lo, hi, seed)
        sizes = np.bincount(mol, minlength=n_frames)
        a, b = adist.shard_frames(sizes, world)[rank]
        c, z, mol, q = adist.local_batch(c, z, mol, q, a, b)
        label = ("aimnet2 MD: 1024 frames x 50 atoms, 128 frames/GPU" if name == "md1024"
                 else "aimnet2 batched: 256 organics of 20-60 atoms per GPU")
        return dict(coord=c, numbers=z, mol_idx=mol, charge=q, cell=None, coulomb="simple", stress=False,
                    frames=b - a, label=label + ", simple Coulomb, E+F")
    if name == "taxol":
        g = np.load(os.path.join(ROOT, "tests", "golden", "taxol.npz"))
        return dict(coord=g["coord"], numbers=g["numbers"], mol_idx=np.zeros(len(g["numbers"]), dtype=np.int64),
                    charge=np.zeros(1, dtype=np.float32), cell=None, coulomb="simple", stress=False, frames=1,
                    label="aimnet2 on taxol (113 atoms), simple Coulomb, E+F")
    raise SystemExit(f"unknown workload {name}")


def parity_samples():
    """The two bounded samples (the CPU baseline is timed on one of them): inputs + a closure that evaluates the oracle."""
    import torch

    from aimnetcentral_amd import synth, workloads
    from oracle import aimnet2_oracle as O

    sd = synth.synthetic_state_dict(0)
    om = O.OracleModel(sd, torch.float32)
    om64 = O.OracleModel(sd, torch.float64)
    c, z, cell = workloads.glucose_supercell((2, 3, 4))  # 2 304 atoms of the config-3 crystal
    mol = np.zeros(len(z), dtype=np.int64)
    pbc = np.ones(3, dtype=bool)
    c32, cell32 = c.astype(np.float32), cell.astype(np.float32)

    def step_pbc():
        xw = O.wrap_into_cell(c32, cell32, mol, pbc)
        nb, sh = O.neighbor_list_fast(xw, 5.0, mol, cell, pbc)
        nbl, shl = O.neighbor_list_fast(xw, 15.0, mol, cell, pbc)
        return O.evaluate(om, coord=xw, numbers=z, charge=np.zeros(1, np.float32), mol_idx=mol, cell=cell32, coulomb="dsf",
                          stress=True, nbmat=nb, shifts=sh, nbmat_lr=nbl, shifts_lr=shl, return_intermediates=True)

    def e64_pbc():  # fp64 oracle, energy only: the anchor of this sample's energy gate (see parity_gate)
        xw = O.wrap_into_cell(c32, cell32, mol, pbc)
        nb, sh = O.neighbor_list_fast(xw, 5.0, mol, cell, pbc)
        nbl, shl = O.neighbor_list_fast(xw, 15.0, mol, cell, pbc)
        return O.evaluate(om64, coord=xw, numbers=z, charge=np.zeros(1, np.float32), mol_idx=mol, cell=cell32, coulomb="dsf", forces=False,
                          nbmat=nb, shifts=sh, nbmat_lr=nbl, shifts_lr=shl, return_intermediates=True)

    cm, zm, molm, qm = workloads.random_batch(48, 50, 50, 5)

    def step_md():
        nb, _ = O.neighbor_list_fast(cm, 5.0, molm)
        nbl, _ = O.neighbor_list(cm, float("inf"), molm)
        return O.evaluate(om, coord=cm, numbers=zm, charge=qm, mol_idx=molm, coulomb="simple", nbmat=nb, nbmat_lr=nbl,
                          return_intermediates=True)

    def e64_md():  # fp64 oracle on the hot random frames: the anchor of their energy gate (see parity_gate)
        nb, _ = O.neighbor_list_fast(cm, 5.0, molm)
        nbl, _ = O.neighbor_list(cm, float("inf"), molm)
        return O.evaluate(om64, coord=cm, numbers=zm, charge=qm, mol_idx=molm, coulomb="simple", nbmat=nb, nbmat_lr=nbl, forces=False,
                          return_intermediates=True)

    gt = np.load(os.path.join(ROOT, "tests", "golden", "taxol.npz"))  # BASELINE configs[0]: the reference's own CPU-runnable case
    ct, zt = gt["coord"].astype(np.float32), gt["numbers"]
    molt, qt = np.zeros(len(zt), dtype=np.int64), np.zeros(1, np.float32)

    def step_taxol():
        nb, _ = O.neighbor_list_fast(ct, 5.0, molt)
        nbl, _ = O.neighbor_list(ct, float("inf"), molt)
        return O.evaluate(om, coord=ct, numbers=zt, charge=qt, mol_idx=molt, coulomb="simple", nbmat=nb, nbmat_lr=nbl, return_intermediates=True)

    def e64_taxol():
        nb, _ = O.neighbor_list_fast(ct, 5.0, molt)
        nbl, _ = O.neighbor_list(ct, float("inf"), molt)
        return O.evaluate(om64, coord=ct, numbers=zt, charge=qt, mol_idx=molt, coulomb="simple", nbmat=nb, nbmat_lr=nbl, forces=False,
                          return_intermediates=True)

    return {
        "taxol113": dict(coord=ct, numbers=zt, mol_idx=molt, charge=qt, cell=None, coulomb="simple", stress=False, step=step_taxol,
                         e64=e64_taxol, label="taxol.xyz (113 atoms), simple Coulomb, E+F"),
        "pbc2304": dict(coord=c32, numbers=z, mol_idx=mol, charge=np.zeros(1, np.float32), cell=cell32, coulomb="dsf", stress=True,
                        step=step_pbc, e64=e64_pbc, label="2019828.cif (2,3,4) supercell, 2304 atoms, DSF 15A, E+F+stress"),
        "md48x50": dict(coord=cm, numbers=zm, mol_idx=molm, charge=qm, cell=None, coulomb="simple", stress=False, step=step_md,
                        e64=e64_md, label="48 frames x 50 atoms (2400 atoms), simple Coulomb, E+F"),
    }


def parity_gate(eng, samples, oracle_out):
    """Engine vs oracle on the parity samples, at the reference's own gates (tests/conftest.py): |dE| <= max(1e-5, 5e-7 n) eV,
    |dF| <= 1e-5 + 1e-4 max|F| eV/A, |dq| <= 1e-4 e, |dstress| <= 1e-5 eV/A^3.  Untimed.
    The random md frames are hot (contacts of 0.9 A, |F| up to 800 eV/A): there the fp32 ORACLE itself sits up to 1e-3 eV from the
    fp64 energy and its per-atom errors are one-signed within a molecule, so their energies are held against the fp64 oracle exactly
    as tests/test_gpu_configs.check_batch holds config 5: per molecule |E_hip - E_64| <= gate + |E_32 - E_64| + 2 sum_atoms
    |e_32 - e_64| (`dE_gate_slack_fp64` = the largest widening), and in the rms over the frames the engine's distance from fp64 may
    exceed the fp32 oracle's by at most 1.5x (`rms_ratio`).  Forces, charges and stress are at the plain gates everywhere."""
    import torch

    dev = eng.device
    res = {}
    ok_all = True
    for name, smp in samples.items():
        ref = oracle_out[name]
        cell = torch.from_numpy(smp["cell"]).to(dev) if smp["cell"] is not None else None
        r = eng.eval(torch.from_numpy(smp["coord"]).to(dev), torch.from_numpy(np.asarray(smp["numbers"])).to(dev),
                     torch.from_numpy(np.asarray(smp["mol_idx"])).to(dev), torch.from_numpy(smp["charge"]).to(dev), cell=cell,
                     forces=True, stress=smp["stress"], coulomb=smp["coulomb"])
        r = {k: v.cpu().numpy() for k, v in r.items()}
        sizes = np.bincount(np.asarray(smp["mol_idx"]))
        fmax = float(np.abs(ref["forces"]).max())
        de = np.abs(r["energy"] - ref["energy"])
        slack, rms_ratio = np.zeros_like(de), None
        if smp["e64"] is not None:
            r64 = smp["e64"]()
            mol = np.asarray(smp["mol_idx"])
            l1 = np.zeros(len(sizes))
            np.add.at(l1, mol, np.abs(ref["_e_atom"][: len(mol)].astype(np.float64) - r64["_e_atom"][: len(mol)]))
            err_ref = np.abs(ref["energy"] - r64["energy"])
            slack = err_ref + 2.0 * l1
            de = np.abs(r["energy"] - r64["energy"])  # against the fp64 energies
            rms_ratio = float(np.sqrt(np.mean(de**2)) / max(1e-30, np.sqrt(np.mean(err_ref**2))))
        e_gate = max(1e-5, 5e-7 * float(sizes.max()))
        d = {"dE": float(de.max()), "dF_max": float(np.abs(r["forces"] - ref["forces"]).max()),
             "dq_max": float(np.abs(r["charges"] - ref["charges"]).max()),
             "dstress_max": float(np.abs(r["stress"] - ref["stress"]).max()) if smp["stress"] else None}
        # the reference's LITERAL force gate, elementwise allclose(rtol 1e-4, atol 1e-5) (tests/test_calculator_gpu.py:137,464): how many
        # force components of the engine lie outside it relative to the oracle - reported next to the global bound that `ok` uses
        # (on these HOT synthetic weights, max|F| 20 - 800 eV/A, two fp32 evaluations differ by more than that gate on small
        # components; the cold-weight goldens, tests/golden/coldw.npz, are held to it with zero violations: profiles/r5_parity_literal.md)
        ratio = np.abs(r["forces"].astype(np.float64) - ref["forces"]) / (1e-5 + 1e-4 * np.abs(ref["forces"].astype(np.float64)))
        d["dF_elementwise_violations"] = {"count": int((ratio > 1).sum()), "of": int(ratio.size), "worst_over_gate": float(ratio.max())}
        gates = {"dE": e_gate, "dE_gate_slack_fp64": float(slack.max()), "dF_max": 1e-5 + 1e-4 * fmax, "dq_max": 1e-4,
                 "dstress_max": 1e-5 if smp["stress"] else None}
        ok = bool((de <= e_gate + slack).all()) and all(d[k] is None or d[k] <= gates[k] for k in ("dF_max", "dq_max", "dstress_max"))
        if rms_ratio is not None:
            ok = ok and rms_ratio <= 1.5 + 1e-5 / max(1e-30, float(np.sqrt(np.mean(err_ref**2))))
            d["dE_vs"], d["rms_ratio"] = "fp64 oracle", rms_ratio
        ok_all = ok_all and ok
        res[name] = dict(d, gates=gates, ok=bool(ok), sample=smp["label"], atoms=int(len(smp["numbers"])), max_abs_force=fmax)
    res["ok"] = bool(ok_all)
    res["oracle"] = "oracle/aimnet2_oracle.py (fp32, pinned to the reference's golden vectors by tests/test_oracle_golden.py)"
    res["note"] = ("hot synthetic weights (max|F| 20 - 800 eV/A): energies of samples with an fp64 anchor are held against the fp64 oracle "
                   "(the fp32 oracle itself sits 1.3e-3 eV from it on the 2 304-atom crystal, the engine 1e-4: tests/tools/pbc2304_margin.py); "
                   "dF_elementwise_violations counts force components outside the reference's literal allclose(1e-4, 1e-5) relative to the "
                   "fp32 oracle - two fp32 evaluations of a hot surface differ by more than that on small components (the fp32 oracle vs the "
                   "reference's own goldens does too: profiles/r5_parity_literal.md); `parity_cold_goldens` holds the literal gates with zero "
                   "violations on weights of realistic force scale")
    return res


def parity_cold_goldens(device):
    """The engine on the COLD variant of the weights against goldens the unmodified reference produced (tests/golden/coldw.npz,
    make_golden.py --only-coldw), at the reference's LITERAL gates: |dE| < 1e-5 eV per molecule, every force component inside
    allclose(rtol 1e-4, atol 1e-5) (tests/test_calculator_gpu.py:137,445,464).  Untimed; committed data only (no oracle)."""
    import torch

    from aimnetcentral_amd import loader
    from aimnetcentral_amd.engine import HipEngine

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "coldw.npz")
    if not os.path.exists(path):
        return None
    gf = dict(np.load(path))
    big = os.path.join(os.path.dirname(path), "coldw_big.npz")  # headline size: 2 304-atom DSF cell + stress, 256-molecule batch
    names = ["taxol", "batch5", "rand8", "pbc96"]
    if os.path.exists(big):
        gf.update({k: v for k, v in np.load(big).items() if not k.startswith("weights_")})
        names += ["pbc2304", "batch256"]
    eng = HipEngine(loader.synthetic_spec(0, cold=True), device)
    out, ok_all = {}, True
    for name in names:
        g = {k[len(name) + 1:]: gf[k] for k in gf if k.startswith(name + "_")}
        g["numbers"] = g["numbers"].astype(np.int64)
        mol = g["mol_idx"].astype(np.int64) if "mol_idx" in g else np.zeros(len(g["numbers"]), dtype=np.int64)
        e_gate = np.maximum(1e-5, 5e-7 * np.bincount(mol))
        kw = dict(cell=torch.from_numpy(g["cell"]).to(device), coulomb="dsf", stress=True, dsf_rc=float(g["dsf_rc"]),
                  dsf_alpha=float(g["dsf_alpha"])) if "cell" in g else dict(coulomb="simple")
        r = eng.eval(torch.from_numpy(g["coord"]).to(device), torch.from_numpy(g["numbers"]).to(device), torch.from_numpy(mol).to(device),
                     torch.from_numpy(np.atleast_1d(g["charge"]).astype(np.float32)).to(device), forces=True, **kw)
        r = {k: v.cpu().numpy() for k, v in r.items()}
        ratio = np.abs(r["forces"].astype(np.float64) - g["forces"]) / (1e-5 + 1e-4 * np.abs(g["forces"].astype(np.float64)))
        d = {"atoms": int(len(g["numbers"])), "max_abs_force": float(np.abs(g["forces"]).max()),
             "dE": float(np.abs(r["energy"] - g["energy"]).max()), "dF_max": float(np.abs(r["forces"] - g["forces"]).max()),
             "dF_elementwise_violations": {"count": int((ratio > 1).sum()), "of": int(ratio.size), "worst_over_gate": float(ratio.max())},
             "dq_max": float(np.abs(r["charges"] - g["charges"]).max())}
        if "stress" in g:
            d["dstress_max"] = float(np.abs(r["stress"] - g["stress"]).max())
        d["dE_over_gate"] = float((np.abs(r["energy"] - g["energy"]) / e_gate).max())
        d["ok"] = bool(d["dE_over_gate"] < 1.0 and d["dF_elementwise_violations"]["count"] == 0 and d["dq_max"] <= 1e-4 and d.get("dstress_max", 0.0) <= 1e-5)
        ok_all = ok_all and d["ok"]
        out[name] = d
    out["ok"] = bool(ok_all)
    out["gates"] = ("reference literal: |dE| < max(1e-5, 5e-7 n) eV per system, forces allclose(rtol 1e-4, atol 1e-5) elementwise, |dq| <= 1e-4, "
                    "|dstress| <= 1e-5; default GEMM path, no fp64 anchor")
    out["golden"] = ("tests/golden/coldw.npz + coldw_big.npz (unmodified reference, cold variant of the seed-0 weights; pbc2304 / batch256 = "
                     "the sizes the headline is quoted on)")
    out["violations_total"] = int(sum(out[n]["dF_elementwise_violations"]["count"] for n in names))
    return out


def cpu_baseline(workload: str, budget_s: float = 10.0, samples=None, oracle_out=None, full_inputs=None):
    """Oracle (kind 'port': torch-CPU eager restatement of the reference op sequence) on the host cores, on a bounded sample of
    the same workload.  The timed step INCLUDES the neighbour lists (k-d tree over the periodic images,
    oracle.neighbor_list_fast), as the GPU step does; measured at every host thread torch will use and at 1 thread
    (SURVEY 8d).  `value` / `cores` are the better-throughput setting, `all` keeps both."""
    import torch

    from aimnetcentral_amd import synth, workloads
    from oracle import aimnet2_oracle as O

    cores = os.cpu_count() or 1
    samples = samples or parity_samples()
    key = {"pbc10k": "pbc2304", "taxol": "taxol113"}.get(workload, "md48x50")
    step = samples[key]["step"]
    z = samples[key]["numbers"]
    sample = samples[key]["label"] + ", neighbour lists rebuilt and timed every step"
    n_atoms = len(z)
    results = []
    # torch CPU eager degrades badly when it spins hundreds of threads on these small tensors (one 2 304-atom evaluation: 2.3 s on
    # 1 thread, 52 s on 256): the all-cores figure is therefore taken from ONE evaluation, the 16-thread and 1-thread figures
    # from a time budget each
    plan = [(min(cores, 16), budget_s * 0.6, True), (1, budget_s * 0.6, True)]
    if budget_s >= 5.0:
        plan.append((cores, 0.0, False))
    for nt, budget, warm in plan:
        if any(r["cores"] == nt for r in results):
            continue
        torch.set_num_threads(nt)
        if warm:
            out = step()
            if oracle_out is not None:
                oracle_out[key] = out
        t0 = time.perf_counter()
        reps = 0
        while True:
            step()
            reps += 1
            dt = time.perf_counter() - t0
            if dt > budget or reps >= 50:
                break
        results.append({"cores": int(nt), "value": n_atoms * reps / dt, "evals": reps, "seconds": round(dt, 2)})
    best = max(results, key=lambda r: r["value"])
    full = None
    if workload == "pbc10k" and budget_s >= 5.0:  # ONE evaluation of the full-size configuration (10 080 atoms, both lists rebuilt)
        try:
            if full_inputs is not None:  # the very frame the GPU was timed on (rank 0)
                c32, z10, cell = full_inputs["coord"].astype(np.float32), np.asarray(full_inputs["numbers"]), full_inputs["cell"].astype(np.float64)
            else:
                c, z10, cell = workloads.glucose_supercell((7, 3, 5))
                c32 = c.astype(np.float32)
            mol = np.zeros(len(z10), dtype=np.int64)
            pbc = np.ones(3, dtype=bool)
            cell32 = cell.astype(np.float32)
            om = O.OracleModel(synth.synthetic_state_dict(0), torch.float32)
            torch.set_num_threads(min(cores, 16))
            t0 = time.perf_counter()
            xw = O.wrap_into_cell(c32, cell32, mol, pbc)
            nb, sh = O.neighbor_list_fast(xw, 5.0, mol, cell, pbc)
            nbl, shl = O.neighbor_list_fast(xw, 15.0, mol, cell, pbc)
            out10 = O.evaluate(om, coord=xw, numbers=z10, charge=np.zeros(1, np.float32), mol_idx=mol, cell=cell32, coulomb="dsf", stress=True,
                               nbmat=nb, shifts=sh, nbmat_lr=nbl, shifts_lr=shl)
            dt = time.perf_counter() - t0
            if oracle_out is not None:
                oracle_out["pbc10080"] = dict(out10, _coord=xw)
            full = {"atoms": int(len(z10)), "cores": int(min(cores, 16)), "evals": 1, "seconds": round(dt, 2), "value": len(z10) / dt,
                    "sample": "the full-size configuration itself: 2019828.cif (7,3,5) supercell, 10080 atoms, DSF 15A, E+F+stress, lists rebuilt"}
        except Exception as exc:  # e.g. host memory: the bounded sample above stands
            full = {"error": f"{type(exc).__name__}: {exc}"}
    return {"value": best["value"], "unit": "atoms*steps/s", "cores": best["cores"], "kind": "port",
            "sample": sample + f"; torch {torch.__version__} CPU eager, {cores} host threads available", "all": results, "full_size": full}


def _pmc_file():
    """The newest committed profiles/rN_pmc.json (written by tests/tools/pmc_bench.sh)."""
    import glob
    import re

    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")):
        m = re.match(r"r(\d+)([a-z]?)_pmc\.json$", os.path.basename(f))
        if m and (best is None or (int(m.group(1)), m.group(2)) > best[0]):
            best = ((int(m.group(1)), m.group(2)), f)
    return best[1] if best else None


PMC_FILE = _pmc_file()


def pmc_record(workload):
    """The whole committed PMC record of this workload (or {}): traffic, MFMA-busy fraction, the commit it was measured at."""
    if workload != "pbc10k" or not PMC_FILE or not os.path.exists(PMC_FILE):
        return {}
    with open(PMC_FILE) as f:
        return json.load(f)


def pmc_traffic(workload):
    """HBM-side bytes per GEMM launch from the committed rocprofv3 PMC pass of this workload (TCC_EA0_RDREQ x 128 B +
    TCC_EA0_WRREQ x 64 B, the gfx950 correction of MI355X_MICROARCH.md).  PMC counters cannot be read from inside the timed
    process, so this is the last measured value: the file records the commit it was measured at (`commit`) and
    tests/tools/pmc_bench.sh regenerates it; None for other workloads or when the file is missing."""
    if workload != "pbc10k" or not PMC_FILE or not os.path.exists(PMC_FILE):
        return None, None
    with open(PMC_FILE) as f:
        d = json.load(f)
    return d.get("traffic_bytes_per_launch"), d.get("commit")


# algorithmic FLOPs per atom and step (energy + forces) of the WHOLE path, F(M) = 8.90e6 + 1.95e4 M with M the mean
# number of neighbours inside the 5 A cutoff: MLP forward + input-gradient backward, conv_a / conv_q forward and backward, AEV, agh
def e2e_flops_per_atom(mean_nb: float) -> float:
    return 2 * (2 * 2_181_760) + 3 * (2048 + 4096) * mean_nb + 2 * (128 + 256) * mean_nb + 300 * mean_nb + 170_000


def hessian_config4(eng):
    """BASELINE configs[3] in the same run (untimed region): the dense Hessian of the 40-atom fixture by the analytic tangent sweep
    (csrc/hvp.hip, 120 directions in one call) against the reference's own double-backward Hessian (tests/golden/hvp40.npz, made by
    the unmodified reference: tests/golden/make_golden.py) at the reference's gate allclose(rtol = atol = 1e-3) (tests/test_hvp.py:75),
    with its cost next to one force evaluation of the same molecule."""
    import numpy as np
    import torch

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "hvp40.npz"))
    dev = eng.device
    t = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a)).to(device=dev, dtype=dt)  # noqa: E731
    args = (t(g["coord"]), t(g["numbers"], torch.int32), torch.zeros(40, dtype=torch.int32, device=dev), t([float(g["charge"])]))
    eye = torch.eye(120, device=dev).view(120, 40, 3)

    def clock(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    H = eng.hvp(*args, eye)["hv"].view(120, 120)
    H = (0.5 * (H + H.T)).cpu().numpy()
    Href = g["hessian"].reshape(120, 120)
    v4 = t(g["v4"])
    hv4 = eng.hvp(*args, v4)["hv"].cpu().numpy()
    ms_h, ms_1 = clock(lambda: eng.hvp(*args, eye), 10), clock(lambda: eng.hvp(*args, v4[:1]), 10)
    ms_f = clock(lambda: eng.eval(*args, forces=True), 10)
    return {"workload": "hvp40: 40 atoms, dense Hessian = 120 directions in one tangent sweep (BASELINE configs[3])",
            "hessian_ms": ms_h, "one_direction_ms": ms_1, "force_eval_ms": ms_f,
            "force_evals_per_direction": {"dense": ms_h / 120.0 / ms_f, "single": ms_1 / ms_f},
            "dH_max": float(np.abs(H - Href).max()), "dHv4_max": float(np.abs(hv4 - g["hv4"]).max()),
            "H_abs_max": float(np.abs(Href).max()), "reference": "tests/golden/hvp40.npz (reference double backward, fp32)",
            "gate": "allclose(rtol=1e-3, atol=1e-3) elementwise (tests/test_hvp.py:75)",
            "ok": bool(np.allclose(H, Href, rtol=1e-3, atol=1e-3) and np.allclose(hv4, g["hv4"], rtol=1e-3, atol=1e-3))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=60,
                    help="untimed steps in front of the timed region (default 60 = ~80 ms: the five regions of timed_region_repeat show the "
                         "first ~40 steps after an idle phase running 2 - 3 % slower than the following ones)")
    ap.add_argument("--workload", default="pbc10k")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-repeat", action="store_true", help="skip the four extra repeats of the timed region (timed_region_repeat)")
    ap.add_argument("--no-exact-f32", action="store_true", help="skip the extra pass with the exact-fp32 GEMM kernels (profiling runs)")
    ap.add_argument("--no-hessian", action="store_true", help="skip the hessian_config4 record (profiling runs: keeps the kernel trace "
                    "and the counter passes to the kernels of the timed workload)")
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of CPU-oracle timing (16 threads + 1 thread); below 5 s the "
                    "single all-host-threads evaluation (51 s on a 256-thread box) is skipped")
    ap.add_argument("--breakdown", action="store_true", help="print a per-kernel-family time table to stderr")
    args = ap.parse_args()

    import torch

    from aimnetcentral_amd import dist as adist
    from aimnetcentral_amd import loader
    from aimnetcentral_amd.engine import HipEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.gpus != world and distributed:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if args.gpus > 1 and not distributed:
        raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dev = torch.device(f"cuda:{local_rank % max(1, torch.cuda.device_count())}")
    torch.cuda.set_device(dev)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # BENCH_BACKEND=gloo: collectives on host tensors - lets several ranks share ONE GPU to exercise the N > 1 control
        # flow on a single-GPU box (RCCL refuses two ranks on one device); never used for reported numbers
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)
    comm_dev = dev if (not distributed or os.environ.get("BENCH_BACKEND", "nccl") == "nccl") else torch.device("cpu")

    eng = HipEngine(loader.synthetic_spec(0), dev)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed_run(workload, steps, warmup, profile=True):
        """warmup checked steps, then `steps` steps enqueued back to back between two barriers; MAX over ranks.
        Returns (elapsed s, atoms on all ranks, this rank's atoms, frames per rank, GEMM profile, workload dict, local step fn)."""
        wl = build_workload(workload, rank, world)
        # inputs resident in HBM in the engine's native dtypes (f32 / i32): no per-step conversion kernels in the timed region
        dt = {"coord": torch.float32, "numbers": torch.int32, "mol_idx": torch.int32, "charge": torch.float32}
        t = {k: torch.from_numpy(np.ascontiguousarray(wl[k])).to(device=dev, dtype=dt[k]) for k in dt}
        cell = torch.from_numpy(wl["cell"]).to(dev) if wl["cell"] is not None else None
        n_atoms = int(t["coord"].shape[0])
        frames = int(wl["frames"])
        statuses = []

        def local_step(sync=True):  # this rank's shard only: no collective
            r = eng.eval(t["coord"], t["numbers"], t["mol_idx"], t["charge"], cell=cell, forces=True, stress=wl["stress"],
                         coulomb=wl["coulomb"], sync=sync)
            if not sync:
                statuses.append(r["status"])
            return r["energy"]

        def step(sync=True):
            e = local_step(sync)
            if distributed:
                return adist.all_gather_energies(e.to(comm_dev), [frames] * world, reuse_buffer=True)
            return e

        for _ in range(max(1, warmup)):  # at least one checked evaluation: it settles the neighbour-row capacities
            step()
        # HIP events at the GEMM <-> rest boundaries of every 4th timed step (the events themselves cost ~3 % of a step)
        if profile:
            eng.set_profiling(1, every=4)
        barrier()
        t0 = time.perf_counter()
        # Timed steps are enqueued back to back, like a device-resident MD driver would: the 32-byte neighbour-overflow status
        # of every step stays on the device and is checked after the closing barrier (the warm-up steps above ran with the
        # per-step check and settled the row capacities); a per-step host read would only add a ~50 us bubble per step.
        for _ in range(steps):
            e_all = step(sync=False)
        barrier()
        elapsed = time.perf_counter() - t0
        st_all = torch.stack(statuses).cpu().numpy()
        assert not st_all[:, [2, 3, 5]].any(), "neighbour-list overflow inside the timed region: results invalid"
        prof = eng.read_profile() if profile else None
        eng.set_profiling(0)
        ranks_seen = world
        if distributed:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            na = torch.tensor([n_atoms], dtype=torch.int64, device=comm_dev)
            dist.all_reduce(na)
            total_atoms = int(na.item())
            ranks_seen = dist.get_world_size()
            assert e_all.numel() == frames * world, "all-gather returned the wrong number of frame energies"
        else:
            total_atoms = n_atoms
        assert bool(torch.isfinite(e_all).all()), "non-finite energies"
        return dict(elapsed=elapsed, total_atoms=total_atoms, n_atoms=n_atoms, frames=frames, prof=prof, wl=wl, local_step=local_step,
                    ranks_seen=ranks_seen)

    run = timed_run(args.workload, args.steps, args.warmup)
    elapsed, total_atoms, n_atoms, frames, prof, wl, local_step = (run[k] for k in ("elapsed", "total_atoms", "n_atoms", "frames", "prof",
                                                                                   "wl", "local_step"))

    # the line's own noise bar: four more repeats of the identical timed region (untouched `value` = the FIRST region, as the driver
    # clocks it); min / median / max of the five regions' ms per step
    repeats = [elapsed / args.steps * 1e3]
    if not args.no_repeat:
        for _ in range(4):
            rr = timed_run(args.workload, args.steps, 1, profile=False)
            repeats.append(rr["elapsed"] / args.steps * 1e3)

    # BASELINE configs[4] in the same run (N > 1): 128 frames x 50 atoms per GPU, per-frame energies all-gathered every step
    md = None
    if distributed and args.workload != "md1024":
        r = timed_run("md1024", args.steps, max(2, args.warmup), profile=False)
        md = {"workload": r["wl"]["label"], "value": r["total_atoms"] * args.steps / r["elapsed"], "unit": "atoms*steps/s",
              "ms_per_step": r["elapsed"] / args.steps * 1e3, "frames_per_gpu": r["frames"], "atoms_per_gpu": r["n_atoms"],
              "ranks_seen": r["ranks_seen"], "steps": args.steps, "scaling": "weak",
              "collective": f"all-gather of {r['frames']} fp64 frame energies per rank and step ({'RCCL' if comm_dev.type == 'cuda' else 'gloo'})"}

    fam = None
    if rank == 0:  # after the timed region: 5 extra steps with an event per kernel-family change (untimed, N = 1 view)
        eng.set_profiling(2)
        for _ in range(5):
            local_step()  # rank 0 alone: must not enter a collective here
        torch.cuda.synchronize(dev)
        fam = {k: v / 5 for k, v in eng.read_profile().items() if k != "evals"}
        eng.set_profiling(0)
        n_pairs = int(eng.debug_view("nb_cnt").sum().item())  # ordered pairs inside the 5 A cutoff
        # the engine's own switches (not the environment: set_option may have changed them); reverse-pair form = engine.hip layout()
        xe = eng.get_option("conv_xe") != 0 and not (eng.get_option("conv_mfma") & 2) and n_atoms > eng.get_option("split_max")
        gather_form = "reverse-pair" if xe else "combined"
        # pass 0 in the reverse-pair form reads only the centre's own species-moment block of the pair (256 B)
        gather_bytes = (2 * (4096 + 256 + 16) + 16 + 256) if xe else (2 * 5376 + 512)
        if args.breakdown:
            print("per-family ms/step: " + "  ".join(f"{k}={v:.3f}" for k, v in fam.items()) + f"  total={sum(fam.values()):.3f}",
                  file=sys.stderr)

    exact = None
    if world == 1 and not args.no_exact_f32:  # the exact-fp32 MFMA kernels on the same workload (fallback record), after the timed region
        eng.set_option("gemm_bf3", 0)
        r = timed_run(args.workload, args.steps, 2)
        g_ms = r["prof"]["gemm"] / max(1.0, r["prof"]["evals"])
        fl = eng.gemm_flops_per_atom(True) * n_atoms
        exact = {"value": r["total_atoms"] * args.steps / r["elapsed"], "unit": "atoms*steps/s", "ms_per_step": r["elapsed"] / args.steps * 1e3,
                 "dtype": "f32 (v_mfma_f32_16x16x4_f32, exact fp32 operands)", "option": "aimnet_engine_set_option(\"gemm_bf3\", 0)",
                 "parity_note": "on the cold relaxed256 fixture this mode sits at 1.05 x the un-widened energy gate (the default split "
                                "kernels at 0.66; the reference's own fp32 golden at 0.75): tests/test_gpu_configs.py, profiles/r3_gemm_bf3.md",
                 "roofline": {"bound": "mfma", "achieved": fl / (g_ms * 1e-3) / 1e12, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                              "frac": fl / (g_ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS, "gemm_ms_per_step": g_ms,
                              "kernel": "gemm_nt_panel_kernel (fp32 MFMA MLP GEMMs)"}}
        eng.set_option("gemm_bf3", 1)

    bf16x3 = None
    if world == 1 and not args.no_exact_f32 and eng.get_option("gemm_h2"):  # the bf16x3-split operands (round 4's default; the range fallback)
        eng.set_option("gemm_h2", 0)
        r = timed_run(args.workload, args.steps, 2)
        g_ms = r["prof"]["gemm"] / max(1.0, r["prof"]["evals"])
        fl = eng.gemm_flops_per_atom(True) * n_atoms
        bf16x3 = {"value": r["total_atoms"] * args.steps / r["elapsed"], "unit": "atoms*steps/s", "ms_per_step": r["elapsed"] / args.steps * 1e3,
                  "dtype": "f32 (bf16x3-split MFMA operands, six products, fp32 accumulate)", "option": "aimnet_engine_set_option(\"gemm_h2\", 0)",
                  "roofline": {"bound": "mfma", "achieved": fl / (g_ms * 1e-3) / 1e12, "peak": PEAK_BF16X3_TFLOPS, "unit": "TFLOP/s",
                               "frac": fl / (g_ms * 1e-3) / 1e12 / PEAK_BF16X3_TFLOPS, "gemm_ms_per_step": g_ms,
                               "kernel": "gemm_bf3a_kernel + head_fused_kernel"}}
        eng.set_option("gemm_h2", 1)

    ewald = None
    if world == 1 and args.workload == "pbc10k" and not args.no_exact_f32:
        # the same frame with Ewald summation instead of DSF (csrc/ewald.hip): its own record, after the timed region
        t_in = {k: torch.from_numpy(np.ascontiguousarray(wl[k])).to(dev) for k in ("coord", "numbers", "mol_idx", "charge")}
        cell_t = torch.from_numpy(wl["cell"]).to(dev)

        def ew_step(sync, method="ewald"):
            return eng.eval(t_in["coord"], t_in["numbers"], t_in["mol_idx"], t_in["charge"], cell=cell_t, forces=True, stress=True,
                            coulomb=method, ewald_accuracy=1e-6, sync=sync)

        for _ in range(3):
            r_ew = ew_step(True)
        k_entries = int(eng.last_status[7])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r_ew = ew_step(False)
        torch.cuda.synchronize()
        ew_ms = (time.perf_counter() - t0) / args.steps * 1e3
        ewald = {"ms_per_step": ew_ms, "value": n_atoms / (ew_ms * 1e-3), "unit": "atoms*steps/s", "ewald_accuracy": 1e-6,
                 "k_box_entries": k_entries, "energy_eV": float(r_ew["energy"][0]),
                 "note": "set_lrcoulomb_method('ewald'): real space on the cell-grid walk, exact structure-factor sum in reciprocal space; "
                         "parity against the oracle's restatement in tests/test_gpu_ewald.py; the oracle is pinned to the reference's "
                         "in-tree pure-PyTorch Ewald (ops.py:196) and to Madelung constants, not to nvalchemiops' kernel (absent)"}
        # and with the reciprocal sum on a mesh (csrc/pme.hip): the same energy to the accuracy, from a different splitting
        for _ in range(3):
            r_pm = ew_step(True, "pme")
        mesh_pts = int(eng.last_status[7])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r_pm = ew_step(False, "pme")
        torch.cuda.synchronize()
        pm_ms = (time.perf_counter() - t0) / args.steps * 1e3
        f_ew, f_pm = r_ew["forces"].cpu().numpy(), r_pm["forces"].cpu().numpy()
        ewald["pme"] = {"ms_per_step": pm_ms, "value": n_atoms / (pm_ms * 1e-3), "unit": "atoms*steps/s", "ewald_accuracy": 1e-6,
                        "mesh_points": mesh_pts, "energy_eV": float(r_pm["energy"][0]),
                        "vs_ewald": {"dE_eV": float(abs(r_pm["energy"][0] - r_ew["energy"][0])), "dF_max": float(np.abs(f_pm - f_ew).max()),
                                     "F_max": float(np.abs(f_ew).max()),
                                     "dstress_max": float(np.abs(r_pm["stress"].cpu().numpy() - r_ew["stress"].cpu().numpy()).max())},
                        "note": "set_lrcoulomb_method('pme'): order-8 B-spline mesh, real-space cutoff 10 A; pinned to the exact Ewald sum "
                                "(oracle/pme.py, tests/test_gpu_pme.py), unpinned against nvalchemiops' particle_mesh_ewald (absent)"}

    if rank == 0:
        # `value` = the MEDIAN of the five identical K-step regions (each bracketed by barriers, MAX over ranks); the first region - what
        # a single timed loop would have reported - is kept as `first_region_ms`
        first_region_ms = elapsed / args.steps * 1e3
        ms_per_step = float(np.median(repeats))
        value = total_atoms / (ms_per_step * 1e-3)
        gemm_ms = prof["gemm"] / max(1.0, prof["evals"])  # average over the sampled timed steps
        flops_step = eng.gemm_flops_per_atom(True) * n_atoms
        achieved = flops_step / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        traffic, traffic_commit = pmc_traffic(args.workload)
        pmc = pmc_record(args.workload)
        presplit = eng.get_option("gemm_presplit") != 0 and eng.get_option("gemm_bf3") != 0 and n_atoms > max(256, eng.get_option("split_max"))
        h2 = presplit and eng.get_option("gemm_h2") != 0  # fp16x2-split operands (csrc/gemm_h2.hip): three products per fp32 product
        peak_split = PEAK_F16X2_TFLOPS if h2 else PEAK_BF16X3_TFLOPS
        n_lay = sum(len(d) - 1 for d in eng.spec.mlp_dims)
        fused = presplit and eng.get_option("head_fused") != 0
        chain = h2 and eng.get_option("gemm_chain") != 0  # one launch per MLP sweep (csrc/gemm_chain.hip; pass 0 forward needs the embedding-bias table)
        gemm_launches = (2 * len(eng.spec.mlp_dims) if chain else 2 * n_lay) + (1 if fused else 2 * (len(eng.spec.head_dims) - 2))
        e2e_flops = e2e_flops_per_atom(n_pairs / n_atoms) * n_atoms
        e2e_tflops = e2e_flops * world / (ms_per_step * 1e-3) / 1e12 / world  # per GPU (weak scaling: every rank runs the same work)
        out = {
            "metric": "atoms*steps/sec (energy+forces)",
            "value": value,
            "unit": "atoms*steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "first_region_ms": first_region_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32 (fp16x2-split MFMA operands: fp32 == hi + lo / 4096 to 2^-24, three fp16 products per fp32 product, fp32 accumulate; "
                      "rms error against fp64 below the fp32 MFMA chain and the bf16x3 split, profiles/r5_gemm_h2.md)") if h2 else
                     "f32 (bf16x3-split MFMA operands, fp32 accumulate)",
            "timed_region_repeat": {"ms_per_step": repeats, "min": float(np.min(repeats)), "median": float(np.median(repeats)),
                                    "max": float(np.max(repeats)), "regions": len(repeats), "steps_per_region": args.steps,
                                    "note": "`value` / `ms_per_step` = the median of these regions; region 0 = `first_region_ms`; the others repeat it after one re-warming evaluation"},
            "data": "synthetic (seeded weights of the real aimnet2 architecture; " + (
                "crystal from 2019828.cif + 0.02 A jitter)" if args.workload == "pbc10k" else
                "taxol.xyz frame 0)" if args.workload == "taxol" else "seeded random organic geometries)"),
            "config": {"workload": wl["label"], "atoms_per_gpu": n_atoms, "frames_per_gpu": frames,
                       "parallelism": f"batch-shard x{world} (independent frames, RCCL all-gather of energies)" if world > 1 else "single GPU"},
            # algorithmic fp32 FLOPs of the MLP GEMMs / their time: against the roof of the kernel that runs them (six bf16 MFMA
            # products per fp32 product: 2.5 PFLOP/s / 6) and against the fp32-matrix peak the exact kernels are bound by
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak_split, "unit": "TFLOP/s",
                         "frac": achieved / peak_split,
                         "peak_note": "2.5 PFLOP/s dense fp16 / 3 products per fp32 product" if h2 else "2.5 PFLOP/s dense bf16 / 6 products per fp32 product",
                         # continuity with rounds 3-4, whose kernels needed six products per fp32 product: the same algorithmic FLOPs
                         # against THAT roof (VERDICT r4's target was >= 0.38 of it)
                         "frac_vs_bf16x3_roof": achieved / PEAK_BF16X3_TFLOPS,
                         "peak_fp32_matrix": PEAK_FP32_MATRIX_TFLOPS, "frac_fp32_matrix": achieved / PEAK_FP32_MATRIX_TFLOPS,
                         "traffic": traffic, "traffic_measured_at_commit": traffic_commit,
                         "mfma_busy_frac_in_kernel": pmc.get("mfma_busy_frac_in_kernel"),
                         "kernel": ("gemm_chain_kernel + head_fused_h2_kernel (one launch per MLP sweep: a block owns 48 rows and the full width of every "
                                    "layer, hidden activations stay in LDS, weights stream L2 -> registers in fragment order; fp16x2-split MFMA "
                                    "operands, fp32 accumulate, bitwise the per-layer gemm_h2_kernel launches; all launches of a step)") if chain else
                                   ("gemm_h2_kernel + head_fused_h2_kernel (fp16x2-split MFMA MLP GEMMs, operands pre-split by their producers, "
                                    "fp32 accumulate; all launches of a step)") if h2 else
                                   ("gemm_bf3a_kernel + head_fused_kernel (bf16x3-split MFMA MLP GEMMs, operands pre-split by their producers, "
                                    "fp32 accumulate; all launches of a step)") if presplit else
                                   "gemm_bf3_kernel (bf16x3-split MFMA MLP GEMMs, fp32 accumulate; all launches of a step)",
                         "gemm_launches_per_step": gemm_launches,
                         "gemm_ms_per_step": gemm_ms, "other_ms_per_step": prof["other"] / max(1.0, prof["evals"]),
                         "sampled_steps": int(prof["evals"]),
                         "algorithmic_flop_per_step": flops_step,
                         # the WHOLE step (roofline_e2e below) inside this block too: against the fp32-matrix roof (north-star target
                         # 0.5) and against the pipes it runs on
                         "frac_e2e_fp32_matrix": e2e_tflops / PEAK_FP32_MATRIX_TFLOPS,
                         "frac_mixed": (flops_step / (peak_split * 1e12) + (e2e_flops - flops_step) / (PEAK_FP32_MATRIX_TFLOPS * 1e12))
                                       / (ms_per_step * 1e-3)},
            # the whole step against the same roof (F(M) x atoms / ms_per_step / 157.3): the north-star target is
            # frac >= 0.5 here, i.e. 7.7e6 atoms*steps/s on the 10 080-atom configuration
            "roofline_e2e": {"bound": "mfma", "achieved": e2e_tflops, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                             "frac": e2e_tflops / PEAK_FP32_MATRIX_TFLOPS, "algorithmic_flop_per_step": e2e_flops,
                             "mean_neighbours": n_pairs / n_atoms, "target_frac": 0.5,
                             # the same step against the pipes it actually runs on: the GEMM FLOPs at the bf16x3 roof (416.7), the
                             # rest (conv / AEV / agh, vector pipe) at 157.3: ideal time of both parts / measured time
                             "frac_mixed": (flops_step / (peak_split * 1e12) + (e2e_flops - flops_step) / (PEAK_FP32_MATRIX_TFLOPS * 1e12))
                                           / (ms_per_step * 1e-3),
                             "frac_mixed_note": "GEMM FLOPs / the split kernels' roof (roofline.peak) + the rest / 157.3 TFLOP/s, over ms_per_step"},
            # second kernel class (SURVEY 8d ii): the gather-contract backward.  Algorithmic gathered bytes per ordered pair:
            # reverse-pair form (default above 1 024 atoms): passes 1, 2 read Sbar_j (4 KiB) + Sqbar_j (256 B) and write F1 into
            # the pair buffer (16 B, read back once when the passes are summed); combined form (AIMNET_CONV_XE=0): a_j (1 KiB) more;
            # pass 0 reads two 256 B species-moment blocks.  These are L2 / Infinity-Cache gathers (the tables are N x 4-5 KiB
            # << 256 MB), so the roof is the L2 figure of MI355X_MICROARCH.md (34.5 TB/s), not HBM.
            "roofline_gather": {"bound": "l2", "kernel": "conv_bwd_kernel + conv_bwd_p0_kernel", "pairs": n_pairs,
                                "bytes_per_pair": gather_bytes, "form": gather_form,
                                "achieved": n_pairs * gather_bytes / (fam["conv_bwd"] * 1e-3) / 1e12, "peak": 34.5,
                                "unit": "TB/s", "frac": n_pairs * gather_bytes / (fam["conv_bwd"] * 1e-3) / 1e12 / 34.5,
                                "conv_bwd_ms_per_step": fam["conv_bwd"]},
            "family_ms_per_step": fam,
        }
        if exact is not None:
            out["exact_f32"] = exact
        if bf16x3 is not None:
            out["bf16x3_split"] = bf16x3
        if ewald is not None:
            out["ewald_config3"] = ewald
        if md is not None:
            out["scaling_md1024"] = md
        if world == 1 and not args.no_cpu_baseline:
            samples, oracle_out = parity_samples(), {}
            out["cpu_baseline"] = cpu_baseline(args.workload, budget_s=args.cpu_budget, samples=samples, oracle_out=oracle_out,
                                               full_inputs=wl if args.workload == "pbc10k" else None)
            ref10 = oracle_out.pop("pbc10080", None)
            if ref10 is not None:  # the full-size frame itself, engine vs oracle (one evaluation each; its own record, not part of parity.ok)
                smp = dict(coord=ref10["_coord"], numbers=np.asarray(wl["numbers"]), mol_idx=np.zeros(n_atoms, dtype=np.int64),
                           charge=np.zeros(1, np.float32), cell=wl["cell"].astype(np.float32), coulomb="dsf", stress=True, e64=None,
                           label=wl["label"] + " (the timed frame of rank 0, wrapped coordinates)")
                out["parity_full_size"] = parity_gate(eng, {"pbc10080": smp}, {"pbc10080": ref10})
            for k, smp in samples.items():  # the sample the baseline was not timed on: one oracle evaluation for the parity gate
                if k not in oracle_out:
                    oracle_out[k] = smp["step"]()
            out["parity"] = parity_gate(eng, samples, oracle_out)
            out["parity_cold_goldens"] = parity_cold_goldens(eng.device)
        if world == 1 and not args.no_hessian:
            out["hessian_config4"] = hessian_config4(eng)
        print(json.dumps(out))
    if distributed:
        dist.barrier()  # the other ranks wait here while rank 0 finishes its per-family pass and prints
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
