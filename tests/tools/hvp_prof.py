"""One-direction HVP on the 10 080-atom crystal (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aimnetcentral_amd import loader, workloads  # noqa: E402
from aimnetcentral_amd.engine import HipEngine  # noqa: E402

eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
c, z, cell = workloads.glucose_supercell()
n = len(z)
t = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a)).to(dt).cuda()  # noqa: E731
args = (t(c), t(z, torch.int32), torch.zeros(n, dtype=torch.int32, device="cuda"), t([0.0]))
kw = dict(cell=t(cell), coulomb="dsf", dsf_rc=15.0, dsf_alpha=0.2)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
v = torch.randn(K, n, 3, device="cuda:0")
for _ in range(4):
    eng.hvp(*args, v, **kw)
torch.cuda.synchronize()
