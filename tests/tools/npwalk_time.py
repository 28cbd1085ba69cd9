import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0, cold=True), "cuda:0")
dev = eng.device
for reps in ((2, 3, 4), (7, 3, 5)):
    c, z, _ = workloads.glucose_supercell(reps)
    args = (torch.as_tensor(c.astype(np.float32), device=dev), torch.as_tensor(z, device=dev), torch.zeros(len(z), dtype=torch.int32, device=dev), torch.zeros(1, device=dev))
    for mode in (1, 0):
        eng.set_option("dsf_np_walk", mode)
        for _ in range(3): eng.eval(*args, forces=True, coulomb="dsf")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): eng.eval(*args, forces=True, coulomb="dsf", sync=False)
        torch.cuda.synchronize()
        print(f"non-periodic cluster of {len(z)} atoms, DSF 15 A, E+F: dsf_np_walk={mode}: {(time.perf_counter()-t0)/20*1e3:.3f} ms/step", flush=True)
