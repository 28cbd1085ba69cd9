"""Worker of tests/test_gpu_dd.py: one rank of a domain-decomposed evaluation (aimnetcentral_amd/dd.py).  Launched by
`python -m torch.distributed.run --nproc-per-node W tests/dd_worker.py <case> <out.json>`; the ranks share cuda:0 and talk over
gloo (RCCL refuses two ranks on one device).  Rank 0 also evaluates the whole periodic system on the plain engine and writes
the comparison."""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from aimnetcentral_amd import loader, workloads  # noqa: E402
from aimnetcentral_amd.dd import DomainDecomposedEngine  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402


def case_inputs(case: str):
    if case == "golden2304":  # the reference-produced cold-weight golden at headline-class size (tests/golden/coldw_big.npz)
        g = np.load(os.path.join(ROOT, "tests", "golden", "coldw_big.npz"))
        return dict(coord=g["pbc2304_coord"], numbers=g["pbc2304_numbers"], cell=g["pbc2304_cell"], charge=float(g["pbc2304_charge"]),
                    dsf_rc=float(g["pbc2304_dsf_rc"]), dsf_alpha=float(g["pbc2304_dsf_alpha"]), cold=True, nq=1,
                    ref=dict(energy=g["pbc2304_energy"], forces=g["pbc2304_forces"], charges=g["pbc2304_charges"], stress=g["pbc2304_stress"]))
    if case in ("cube1536", "cube1536_nse", "cube1536_nocoul", "cube1536_d3", "cube1536_d3rc12"):  # a near-cubic (4,2,2) supercell: 20 x 25 x 24 A, jittered, sheared outside the cell
        c, z, cell = workloads.glucose_supercell((4, 2, 2))
        rng = np.random.default_rng(5)
        c = c + rng.normal(0.0, 0.03, c.shape) + np.array([37.0, -61.0, 13.0])  # (atoms start one or two cells outside the box)
        nse = case.endswith("_nse")
        d3 = None
        if "_d3" in case:  # external DFT-D3(BJ): one matrix shared with DSF (cutoff 15 A) or its own (12 A, the list build forms cn)
            g, t = (np.load(os.path.join(ROOT, "tests", "golden", f + ".npz")) for f in ("dftd3", "dftd3_subset"))
            d3 = dict(par=dict(s6=float(g["s6"]), s8=float(g["s8"]), a1=float(g["a1"]), a2=float(g["a2"]),
                               cutoff=12.0 if case.endswith("rc12") else 15.0, smoothing_fraction=0.2),
                      tables={k: t[k] for k in ("c6ab", "cn_ref", "rcov", "r4r2")})
        return dict(d3=d3, coord=c.astype(np.float32), numbers=z, cell=cell.astype(np.float32), charge=(np.array([1.0, 0.0]) if nse else 0.0),
                    dsf_rc=15.0, dsf_alpha=0.2, cold=not nse, nq=2 if nse else 1, coulomb="none" if case.endswith("_nocoul") else "dsf")
    raise SystemExit(f"unknown case {case}")


def main_calculator(out_path: str):
    """The calculator surface (`AIMNet2Calculator.set_domain_decomposition`): the same dict-in / dict-out call as the reference's
    `calc(data, forces=True, stress=True)`, on every rank, against the same calculator without decomposition."""
    from aimnetcentral_amd import AIMNet2Calculator

    backend = os.environ.get("DD_BACKEND", "gloo")
    torch.cuda.set_device(0)
    dist.init_process_group(backend)
    rank = dist.get_rank()
    inp = case_inputs("cube1536")
    calc = AIMNet2Calculator(loader.synthetic_spec(0, cold=True), device="cuda:0")
    calc.set_lrcoulomb_method("dsf", cutoff=12.0)
    # (atoms inside the cell: wrapping an atom that starts outside is not unique to the last fp32 bit - see main())
    from aimnetcentral_amd.dd import wrapped_fractional

    inp["coord"] = (wrapped_fractional(inp["coord"], inp["cell"]) @ np.asarray(inp["cell"], np.float64)).astype(np.float32)
    data = {"coord": inp["coord"], "numbers": inp["numbers"], "charge": 0.0, "cell": inp["cell"]}
    ref = calc(data, forces=True, stress=True)
    calc.set_domain_decomposition(True)
    out = calc(data, forces=True, stress=True)
    out_h = calc.eval(data, forces=True, host_out=True)  # (what the ASE adapter asks for: CPU tensors)
    bad = None
    try:
        calc({"coord": inp["coord"][None], "numbers": inp["numbers"][None], "charge": [0.0], "cell": inp["cell"]}, forces=True)
    except ValueError as exc:
        bad = str(exc)
    calc.set_domain_decomposition(False)
    back = calc(data, forces=True, stress=True)
    if rank == 0:
        f, fr = out["forces"].double().cpu().numpy(), ref["forces"].double().cpu().numpy()
        ratio = np.abs(f - fr) / (1e-5 + 1e-4 * np.abs(fr))
        rec = {"case": "calculator", "world": dist.get_world_size(), "shapes": {k: list(v.shape) for k, v in out.items()},
               "ref_shapes": {k: list(v.shape) for k, v in ref.items()},
               "dE": float((out["energy"] - ref["energy"]).abs().max()), "dF_violations": int((ratio > 1).sum()),
               "dF_worst_ratio": float(ratio.max()), "dq_max": float((out["charges"] - ref["charges"]).abs().max()),
               "ds_max": float((out["stress"] - ref["stress"]).abs().max()), "batch_refused": bad, "host_out_on_cpu": bool(all(not v.is_cuda for v in out_h.values()) and
                                                                        torch.allclose(out_h["forces"], out["forces"].cpu(), rtol=1e-4, atol=1e-5)),
               "off_again_bitwise": bool(torch.equal(back["forces"], ref["forces"]) and torch.equal(back["energy"], ref["energy"]))}
        with open(out_path, "w") as fh:
            json.dump(rec, fh)
        print(json.dumps(rec))
    dist.barrier()
    dist.destroy_process_group()


def main():
    case, out_path = sys.argv[1], sys.argv[2]
    if case == "calculator":
        return main_calculator(out_path)
    grid = tuple(int(v) for v in os.environ["DD_GRID"].split(",")) if os.environ.get("DD_GRID") else None  # bricks instead of slabs
    backend = os.environ.get("DD_BACKEND", "gloo")
    torch.cuda.set_device(0)
    dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    cdev = "cuda:0" if backend == "nccl" else "cpu"  # where the worker's own bookkeeping collectives live
    inp = case_inputs(case)
    coulomb = inp.get("coulomb", "dsf")
    spec = loader.synthetic_spec(0, cold=inp["cold"]) if inp["nq"] == 1 else loader.synthetic_spec(0, num_charge_channels=2)
    eng = HipEngine(spec, "cuda:0")
    d3par = None
    if inp.get("d3"):
        eng.set_dftd3_tables(inp["d3"]["tables"])
        d3par = inp["d3"]["par"]
    dde = DomainDecomposedEngine(eng)
    if os.environ.get("DD_FORCE_OVERFLOW") and rank == 0:
        # ONE rank starts with rows that are too short: its overflow has to make EVERY rank repeat the evaluation (a rank that
        # retried alone would leave the others waiting in a collective)
        eng.max_nb = 16
        eng._max_nb_lr[float(inp["dsf_rc"])] = 64
    res = dde.eval(inp["coord"], inp["numbers"], inp["cell"], charge=inp["charge"], forces=True, stress=True, coulomb=coulomb,
                   dsf_rc=inp["dsf_rc"], dsf_alpha=inp["dsf_alpha"], dftd3=d3par, grid=grid)
    again = dde.eval(inp["coord"], inp["numbers"], inp["cell"], charge=inp["charge"], forces=True, stress=True, coulomb=coulomb,
                     dsf_rc=inp["dsf_rc"], dsf_alpha=inp["dsf_alpha"], dftd3=d3par, grid=grid)
    dom = dde.last_domain
    # every rank holds the same result
    chk = torch.stack([res["energy"].double().cpu(), res["forces"].double().abs().sum().cpu()]).to(cdev)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    rec = {"case": case, "world": world, "backend": backend, "max_nb_after": int(eng.max_nb), "ranks_agree": bool((lo == hi).all()), "axis": dom.axis, "grid": list(grid) if grid else None, "n_owned": dom.n_owned,
           "n_local": dom.n_local, "exchange_calls": {str(k): v for k, v in dde.last_calls.items()},
           "repeat_bitwise": bool(torch.equal(res["forces"], again["forces"]) and torch.equal(res["energy"], again["energy"]) and
                                  torch.equal(res["stress"], again["stress"]))}
    # wall time of one decomposed evaluation (all ranks SHARE this GPU and the exchanges go through the host over gloo: an upper
    # bound that says nothing about a multi-GPU run; recorded so that nobody has to guess)
    import time

    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dde.eval(inp["coord"], inp["numbers"], inp["cell"], charge=inp["charge"], forces=True, coulomb=coulomb, dsf_rc=inp["dsf_rc"],
                 dsf_alpha=inp["dsf_alpha"], dftd3=d3par, grid=grid)
    torch.cuda.synchronize()
    dist.barrier()
    rec["ms_per_decomposed_eval_shared_gpu_gloo"] = (time.perf_counter() - t0) / 3 * 1e3
    owned = torch.tensor([dom.n_owned], dtype=torch.int64, device=cdev)
    dist.all_reduce(owned)
    rec["owned_total"] = int(owned[0])
    if rank == 0:
        dev = eng.device
        n = len(inp["numbers"])
        q_in = (torch.as_tensor(np.asarray(inp["charge"], np.float32).reshape(1, 2), device=dev) if inp["nq"] == 2
                else torch.tensor([float(inp["charge"])], device=dev))
        # (both sides start from the same wrapped positions: wrapping an atom that starts one or two cells outside the box is not
        # unique to the last fp32 bit, and the single-rank engine wraps on the device in fp32 - DESIGN.md 7, noise floor)
        from aimnetcentral_amd.dd import wrapped_fractional

        xw = (wrapped_fractional(inp["coord"], inp["cell"]) @ np.asarray(inp["cell"], np.float64)).astype(np.float32)
        one = eng.eval(torch.as_tensor(xw, device=dev), torch.as_tensor(inp["numbers"], device=dev).int(),
                       torch.zeros(n, dtype=torch.int32, device=dev), q_in, cell=torch.as_tensor(inp["cell"], device=dev), forces=True,
                       stress=True, coulomb=coulomb, dsf_rc=inp["dsf_rc"], dsf_alpha=inp["dsf_alpha"], dftd3=d3par)

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.eval(torch.as_tensor(xw, device=dev), torch.as_tensor(inp["numbers"], device=dev).int(),
                     torch.zeros(n, dtype=torch.int32, device=dev), q_in, cell=torch.as_tensor(inp["cell"], device=dev), forces=True,
                     coulomb=coulomb, dsf_rc=inp["dsf_rc"], dsf_alpha=inp["dsf_alpha"], dftd3=d3par)
        torch.cuda.synchronize()
        rec["ms_per_single_rank_eval"] = (time.perf_counter() - t0) / 3 * 1e3

        def cmp(ref_e, ref_f, ref_q, tag, ref_s=None):
            f, fr = res["forces"].double().cpu().numpy(), np.asarray(ref_f, np.float64)
            ratio = np.abs(f - fr) / (1e-5 + 1e-4 * np.abs(fr))  # the reference's literal gate, allclose(rtol 1e-4, atol 1e-5)
            rec[tag] = {"dE": float(abs(float(res["energy"]) - float(np.asarray(ref_e).reshape(-1)[0]))),
                        "dF_max": float(np.abs(f - fr).max()), "F_max": float(np.abs(fr).max()), "dF_violations": int((ratio > 1).sum()),
                        "dF_worst_ratio": float(ratio.max()),
                        "dq_max": float(np.abs(res["charges"].cpu().numpy() - np.asarray(ref_q, np.float64).reshape(-1)[:n]).max())}
            if ref_s is not None:
                rec[tag]["ds_max"] = float(np.abs(res["stress"].double().cpu().numpy() - np.asarray(ref_s, np.float64).reshape(3, 3)).max())

        q_one = one["charges"].cpu().numpy()
        cmp(one["energy"].cpu().numpy(), one["forces"].cpu().numpy(), q_one, "vs_single_rank", one["stress"].cpu().numpy())
        if "ref" in inp:
            cmp(inp["ref"]["energy"], inp["ref"]["forces"], inp["ref"]["charges"], "vs_reference_golden", inp["ref"]["stress"])
            f1, fr = one["forces"].double().cpu().numpy(), np.asarray(inp["ref"]["forces"], np.float64)
            rec["single_rank_vs_reference_golden"] = {"dE": float(abs(float(one["energy"][0]) - float(inp["ref"]["energy"][0]))),
                                                      "dF_max": float(np.abs(f1 - fr).max())}
        rec["n_atoms"] = n
        with open(out_path, "w") as fh:
            json.dump(rec, fh)
        print(json.dumps(rec))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
