"""Wall time of one Hessian-vector product (one direction) on the 10 080-atom crystal, DSF 15 A: python tests/tools/hvp_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
c, z, cell = workloads.glucose_supercell()
n = len(z)
t = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a)).to(dt).cuda()
args = (t(c), t(z, torch.int32), torch.zeros(n, dtype=torch.int32, device="cuda"), t([0.0]))
kw = dict(cell=t(cell), coulomb="dsf", dsf_rc=15.0, dsf_alpha=0.2)
torch.manual_seed(0)
v = torch.randn(1, n, 3, device="cuda:0")
for _ in range(3):
    r = eng.hvp(*args, v, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    r = eng.hvp(*args, v, **kw)
torch.cuda.synchronize()
print(os.environ.get("AIMNET_HVP_CONV_BWD", "default"), f"{(time.perf_counter()-t0)/10*1e3:.3f} ms per direction", "checksum", float(r["hv"].double().abs().sum()))
