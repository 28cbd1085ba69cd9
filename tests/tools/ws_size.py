import sys, ctypes as C
sys.path.insert(0, '/root/repo')
from aimnetcentral_amd import loader, _lib
from aimnetcentral_amd.engine import HipEngine
eng = HipEngine(loader.synthetic_spec(0), "cuda:0")
for flags in (0, 3):
    opt = _lib.EvalOptions(); opt.flags = flags; opt.coulomb = 2; opt.dsf_rc = 15.0; opt.dsf_alpha = 0.2; opt.max_nb = 112; opt.max_nb_lr = 0
    n = 10080
    need = int(eng.lib.aimnet_engine_workspace_bytes(eng._h, n, 1, 1, C.byref(opt)))
    print("flags", flags, "workspace", need / 1e6, "MB =", need / n / 1e3, "KB/atom")
