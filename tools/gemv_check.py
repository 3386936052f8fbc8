import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_shell as bs
dev = torch.device("cuda:0")
for (N, K) in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (128256, 4096)]:
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    x = torch.randn(1, K, device=dev).to(torch.bfloat16)
    y = bs.linear(x, w)
    ref = (x.float() @ w.float().t())
    err = (y.float() - ref).abs().max().item()
    def t(fn, it=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / it * 1e3
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(max(2, int(8e8 // (N * K * 2))))]
    i = [0]
    def mine():
        i[0] += 1
        return bs.linear(x, ws[i[0] % len(ws)])
    def blas():
        i[0] += 1
        return torch.matmul(x, ws[i[0] % len(ws)].t())
    a, b = t(mine), t(blas)
    print(f"N={N} K={K}: maxerr {err:.4f} | shell gemv {a:7.1f} us {N*K*2/a/1e6:5.2f} TB/s | hipBLASLt {b:7.1f} us {N*K*2/b/1e6:5.2f} TB/s", flush=True)
