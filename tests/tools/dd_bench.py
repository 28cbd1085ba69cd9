#!/usr/bin/env python
"""Strong scaling of ONE periodic system cut over the ranks (aimnetcentral_amd/dd.py); for a node with several GPUs:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tests/tools/dd_bench.py [reps_x reps_y reps_z]

One rank per GPU over RCCL (DD_BACKEND=gloo + DD_SHARE_GPU=1: all ranks on cuda:0 through the host - the only form a one-GPU box can
run; it measures nothing but the control flow).  System: the config-3 crystal replicated `reps` times (default 7 3 5 = 10 080 atoms;
14 6 10 = 80 640), DSF 15 A, energy + forces.  Rank 0 prints ONE JSON line: ms per decomposed evaluation (max over ranks, barrier on
both sides), atoms*steps/s, local / owned atoms per rank, and - when the system fits - the single-rank periodic evaluation beside it
with the differences in E and F.  bench.py's `value` is never taken from here (its N > 1 line is frame sharding, BASELINE configs[4])."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import loader, workloads  # noqa: E402
from aimnetcentral_amd.dd import DomainDecomposedEngine, brick_grid  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402


def main():
    reps = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (7, 3, 5)
    steps = int(os.environ.get("STEPS", 10))
    backend = os.environ.get("DD_BACKEND", "nccl")
    local = 0 if os.environ.get("DD_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    c, z, cell = workloads.glucose_supercell(reps)
    c = (c + np.random.default_rng(0).normal(0.0, 0.02, c.shape)).astype(np.float32)
    cell = cell.astype(np.float32)
    eng = HipEngine(loader.synthetic_spec(0, cold=True), dev)
    dde = DomainDecomposedEngine(eng)
    grid = brick_grid(cell, world) if os.environ.get("DD_BRICKS") else None

    # positions on the device, as an MD driver holds them: the partition runs there (slab_partition_device)
    c_dev, z_dev = torch.as_tensor(c, device=dev), torch.as_tensor(z, device=dev)

    def step():
        return dde.eval(c_dev, z_dev, cell, charge=0.0, forces=True, coulomb="dsf", grid=grid)

    for _ in range(2):
        res = step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    torch.cuda.synchronize()
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    dom = dde.last_domain
    loc = torch.tensor([dom.n_local, dom.n_owned], dtype=torch.int64, device=el.device)
    lmax = loc.clone()
    dist.all_reduce(lmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        ms = float(el[0]) / steps * 1e3
        rec = {"workload": f"ONE glucose supercell {reps}, DSF 15 A, E + F, cut over {world} ranks ({'bricks ' + str(grid) if grid else 'slabs'})",
               "atoms": len(z), "ranks": world, "backend": backend, "shared_gpu": bool(os.environ.get("DD_SHARE_GPU")), "steps": steps,
               "ms_per_step": ms, "value": len(z) / (ms * 1e-3), "unit": "atoms*steps/s", "scaling": "strong",
               "max_local_atoms_per_rank": int(lmax[0]), "max_owned_atoms_per_rank": int(lmax[1]),
               "exchange_calls_per_step": {str(k): v for k, v in dde.last_calls.items()},
               "note": "device-side partitioner and global-size exchange arrays inside the timed region: aimnetcentral_amd/dd.py"}
        if len(z) <= 100000:
            zt, ct = torch.as_tensor(z, device=dev).int(), torch.as_tensor(cell, device=dev)
            xt, mol, q = torch.as_tensor(c, device=dev), torch.zeros(len(z), dtype=torch.int32, device=dev), torch.zeros(1, device=dev)
            one = eng.eval(xt, zt, mol, q, cell=ct, forces=True, coulomb="dsf")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                eng.eval(xt, zt, mol, q, cell=ct, forces=True, coulomb="dsf", sync=False)
            torch.cuda.synchronize()
            rec["single_rank_ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
            rec["dE_vs_single_rank"] = float(abs(float(res["energy"]) - float(one["energy"][0])))
            rec["dF_max_vs_single_rank"] = float((res["forces"] - one["forces"]).abs().max())
        print(json.dumps(rec), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
