import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aimnetcentral_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
cfg = int(os.environ.get("CFG", 152))
M, N, K = 160, 128, int(os.environ.get("K", 32))
stream = torch.cuda.current_stream(dev).cuda_stream
torch.manual_seed(0)
A = torch.randn(M, K, device=dev); Bt = torch.randn(N, K, device=dev)
C = torch.full((M, N), -7.0, device=dev)
rc = lib.aimnet_debug_gemm(cfg, 0, A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, None, C.data_ptr(), None, N, stream)
torch.cuda.synchronize()
ref = A.double() @ Bt.double().T
err = (C.double() - ref).abs()
print("rc", rc, "max err", err.max().item(), "unwritten", (C == -7).sum().item())
bad = err > 1e-3
print("bad frac", bad.float().mean().item())
# which 16x16 blocks are bad
bb = bad.view(M // 16, 16, N // 16, 16).any(1).any(2)
print(bb.int())
# per-k contribution test: one-hot k
for k in range(K):
    A1 = torch.zeros_like(A); A1[:, k] = A[:, k]
    C1 = torch.zeros(M, N, device=dev)
    lib.aimnet_debug_gemm(cfg, 0, A1.data_ptr(), K, Bt.data_ptr(), K, M, N, K, None, C1.data_ptr(), None, N, stream)
    torch.cuda.synchronize()
    # find which Bt column it got multiplied with: C1[m,n] = A[m,k]*Bt[n,k'] -> k'
    ratio = C1[0:1, :] / A[0:1, k:k+1]   # should equal Bt[:, k]
    d = (ratio.T - Bt).abs().sum(0)
    print("k", k, "-> matches Bt column", int(d.argmin()), float(d.min()))
