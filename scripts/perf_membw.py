import torch, time
dev = torch.device("cuda")
for mb in (64, 520, 2048):
    n = mb * 1000 * 1000 // 8
    x = torch.randint(0, 1000, (n,), dtype=torch.int64, device=dev)
    y = torch.empty_like(x)
    for name, f, bytes_ in (("sum(read)", lambda: x.sum(), n * 8), ("max(read)", lambda: x.max(), n*8), ("copy(r+w)", lambda: y.copy_(x), 2 * n * 8), ("fill(write)", lambda: y.fill_(1), n * 8)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        print("%5d MB %-12s %.3f ms  %.2f TB/s" % (mb, name, ms, bytes_ / ms / 1e9))
