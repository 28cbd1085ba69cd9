"""Is a small-molecule evaluation bound by the host's launch rate or by the GPU's dispatch rate?
Times engine.eval on taxol: synchronous, enqueue-only (sync=False; host cost per eval) and pipelined (one sync per 200)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aimnetcentral_amd import AIMNet2Calculator, loader
calc = AIMNet2Calculator(loader.synthetic_spec(0), device="cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "taxol.npz"))
dev = calc.engine.device
c_t, z_t = torch.from_numpy(g["coord"]).to(dev), torch.from_numpy(g["numbers"]).to(dev).int()
mol = torch.zeros(113, dtype=torch.int32, device=dev); q = torch.zeros(1, device=dev)
e = calc.engine
for _ in range(30): e.eval(c_t, z_t, mol, q, forces=True)
torch.cuda.synchronize()
n = 200
t = time.perf_counter()
for _ in range(n): e.eval(c_t, z_t, mol, q, forces=True)
torch.cuda.synchronize(); t_sync = (time.perf_counter() - t) / n * 1e3
t = time.perf_counter()
for _ in range(n): e.eval(c_t, z_t, mol, q, forces=True, sync=False)
t_enq = (time.perf_counter() - t) / n * 1e3
torch.cuda.synchronize(); t_pipe = (time.perf_counter() - t) / n * 1e3
print(f"synchronous {t_sync:.3f} ms   enqueue only {t_enq:.3f} ms   pipelined (GPU-side) {t_pipe:.3f} ms per eval")
