import sys, time, cProfile, pstats, numpy as np, torch
sys.path.insert(0, '/root/repo')
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
c, z, cell = workloads.glucose_supercell((7, 3, 5))
dev = eng.device
args = (torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev).int(), torch.zeros(len(z), dtype=torch.int32, device=dev), torch.zeros(1, device=dev))
cl = torch.from_numpy(cell.astype(np.float32)).to(dev)
for _ in range(5): eng.eval(*args, cell=cl, forces=True, stress=True, coulomb="dsf")
# host time of one eval without the sync, split: python prep vs the C call
import aimnetcentral_amd.engine as E
orig = eng.lib.aimnet_engine_eval
tc = [0.0]
def timed(*a):
    t = time.perf_counter(); r = orig(*a); tc[0] += time.perf_counter() - t; return r
class L:  # proxy
    def __getattr__(self, k): return timed if k == "aimnet_engine_eval" else getattr(eng_lib, k)
eng_lib = eng.lib; eng.lib = L()
torch.cuda.synchronize(); n = 300; t0 = time.perf_counter()
for _ in range(n):
    eng.eval(*args, cell=cl, forces=True, stress=True, coulomb="dsf", sync=False)
    torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"per eval: total {tot/n*1e3:.3f} ms; C enqueue {tc[0]/n*1e3:.3f} ms")
eng.lib = eng_lib
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    eng.eval(*args, cell=cl, forces=True, stress=True, coulomb="dsf", sync=False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
