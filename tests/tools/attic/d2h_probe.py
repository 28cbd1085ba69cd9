import torch, time
dev = torch.device("cuda:0")
def T(f, n=30):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for nbytes in (2048, 16384, 65536, 163840, 1 << 20, 8 << 20):
    d = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    f = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
    pin = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    def a(): d.cpu()
    def b():
        pin.copy_(d, non_blocking=True); torch.cuda.current_stream().synchronize()
    def c(): f.cpu()
    h = torch.zeros(nbytes, dtype=torch.uint8)
    def u(): d.copy_(h)
    print(nbytes, "u8 .cpu() %.3f ms | pinned async %.3f ms | f32 .cpu() %.3f ms | H2D pageable %.3f ms" % (T(a), T(b), T(c), T(u)))
