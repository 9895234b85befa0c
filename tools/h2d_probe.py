import torch, time
for mb in (16, 64, 256):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): d.copy_(h, non_blocking=True)
        s.synchronize()
        n = max(4, 2048 // mb)
        t = time.perf_counter()
        for _ in range(n): d.copy_(h, non_blocking=True)
        s.synchronize()
        dt = time.perf_counter() - t
    print("H2D %3d MB chunks: %.1f GB/s" % (mb, n * (mb << 20) / dt / 1e9), flush=True)
