"""Dump the engine's own short-range neighbour matrix / counts / processing order of the default bench workload as raw int32 files
(gpurun_out/lists/) for tests/tools/gather_probe.hip --lists."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aimnetcentral_amd import loader, workloads
from aimnetcentral_amd.engine import HipEngine

c, z, cell = workloads.glucose_supercell((7, 3, 5))
rng = np.random.Generator(np.random.PCG64(1000))
c = (c + rng.standard_normal(c.shape) * 0.02).astype(np.float32)
dev = torch.device("cuda:0")
eng = HipEngine(loader.synthetic_spec(0), dev)
n = len(z)
eng.eval(torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(1, device=dev),
         cell=torch.from_numpy(cell.astype(np.float32)).to(dev), forces=True, stress=True, coulomb="dsf")
nb = eng.debug_view("nb_idx").cpu().numpy().astype(np.int32)
cnt = eng.debug_view("nb_cnt").cpu().numpy().astype(np.int32).ravel()
xw = eng.debug_view("xw").cpu().numpy()
f = (xw.astype(np.float64) @ np.linalg.inv(cell)) % 1.0
h = np.abs(np.linalg.det(cell)) / np.array([np.linalg.norm(np.cross(cell[(k + 1) % 3], cell[(k + 2) % 3])) for k in range(3)])
nbin = np.maximum(1, np.floor(h / 5.0).astype(int))
b = np.minimum((f * nbin).astype(int), nbin - 1)
key = (b[:, 0] * nbin[1] + b[:, 1]) * nbin[2] + b[:, 2]
order = np.argsort(key, kind="stable").astype(np.int32)
os.makedirs("gpurun_out/lists", exist_ok=True)
# alternative processing orders: z-major (longest axis slowest: every XCD's eighth is a thick slab), and z-major over 2.5 A sub-bins
key_z = (b[:, 2] * nbin[1] + b[:, 1]) * nbin[0] + b[:, 0]
np.argsort(key_z, kind="stable").astype(np.int32).tofile("gpurun_out/lists/order_z.bin")
nb2 = nbin * 2
b2 = np.minimum((f * nb2).astype(int), nb2 - 1)
key_z2 = (b2[:, 2] * nb2[1] + b2[:, 1]) * nb2[0] + b2[:, 0]
np.argsort(key_z2, kind="stable").astype(np.int32).tofile("gpurun_out/lists/order_z2.bin")
# z slabs (one per XCD), inside a slab: columns of 2x2 bins in (x, y), z fastest inside a column
slab = np.minimum((f[:, 2] * 8).astype(int), 7)
key_c = ((slab * ((nbin[1] + 1) // 2) + b[:, 1] // 2) * ((nbin[0] + 1) // 2) + b[:, 0] // 2) * 1000 + (f[:, 2] * 999).astype(int)
np.argsort(key_c, kind="stable").astype(np.int32).tofile("gpurun_out/lists/order_col.bin")
nb.tofile("gpurun_out/lists/nb_idx.bin"); cnt.tofile("gpurun_out/lists/nb_cnt.bin"); order.tofile("gpurun_out/lists/order.bin")
open("gpurun_out/lists/meta.txt", "w").write(f"{n} {nb.shape[1]}\n")
print(n, nb.shape, cnt.mean(), nbin)
