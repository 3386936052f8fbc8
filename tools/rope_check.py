"""shell rope kernels (wave-per-head and vectorised) against a torch fp32 reference."""
import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_shell as bs
dev = torch.device("cuda:0")
S = bs.shell()
g = torch.Generator(device=dev).manual_seed(0)
H, HQ, D = 40, 32, 128
for N in (63, 64, 1000):
    qkv = torch.randn(N, (H + 8) * D, device=dev, generator=g).to(torch.bfloat16)
    pos = torch.randint(0, 5000, (N,), device=dev, generator=g, dtype=torch.int64)
    inv = 1.0 / (500000.0 ** (torch.arange(0, D, 2, device=dev, dtype=torch.float32) / D))
    ang = torch.arange(5000, device=dev, dtype=torch.float32)[:, None] * inv[None, :]
    cs = torch.cat([ang.cos(), ang.sin()], dim=1).contiguous()
    out = torch.empty(N, H, D, dtype=torch.bfloat16, device=dev)
    S.shell_rope(qkv.data_ptr(), qkv.stride(0), out.data_ptr(), pos.data_ptr(), cs.data_ptr(), None, None, N, H, HQ,
                 1e-5, torch.cuda.current_stream().cuda_stream)
    x = qkv[:, : H * D].view(N, H, D).float()
    a, b = x[..., :64], x[..., 64:]
    c, s = cs[pos][:, None, :64], cs[pos][:, None, 64:]
    ref = torch.cat([a * c - b * s, b * c + a * s], dim=-1)
    err = (out.float() - ref).abs().max().item()
    bad = ((out.float() - ref).abs() > 0.02 * ref.abs() + 1e-2).sum().item()
    print(f"N={N}: max abs err {err:.4f}, elements off by more than a bf16 ulp or so: {bad}")
    assert bad == 0
print("rope check ok")
