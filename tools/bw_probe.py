"""Read-bandwidth probe: how fast can ANY kernel stream N bytes once (cold) on this GPU?"""
import torch
dev = torch.device("cuda:0")
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for mb in (67, 134, 268, 1024):
    n = mb * 1000 * 1000 // 2
    bufs = [torch.randn(n, device=dev, dtype=torch.bfloat16) for _ in range(max(2, 1200 // mb))]
    i = [0]
    def rd():
        b = bufs[i[0] % len(bufs)]; i[0] += 1
        return b.view(torch.int32).sum()
    g = torch.cuda.CUDAGraph()
    outs = []
    with torch.cuda.graph(g):
        for b in bufs:
            outs.append(b.view(torch.int32).sum())
    us = t(g.replay, it=5) / len(bufs)
    print(f"int32 sum over {mb} MB: {us:8.2f} us  -> {mb * 1e6 / us / 1e6:6.2f} TB/s")
    dst = torch.empty_like(bufs[0])
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        for b in bufs:
            dst.copy_(b)
    us = t(g2.replay, it=5) / len(bufs)
    print(f"copy {mb} MB: {us:8.2f} us  -> read {mb * 1e6 / us / 1e6:6.2f} TB/s (+ same written)")
    del bufs, dst
