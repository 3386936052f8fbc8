"""Fixed cost of the decode kernel: launch it on tiny caches (prologue + epilogue only) and on growing ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "compactor-vllm_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import compactor_vllm_amd.attention.sparse_decode_kernel as dk
from microbench import build_cache
dev = torch.device("cuda:0")
for L, S in [(16, 32), (64, 32), (1024, 32), (2048, 32), (4096, 32), (8192, 32), (16384, 32), (16, 1), (16, 8), (2048, 8)]:
    layers = 4
    caches, pt, bm, lens = build_cache(1, 8, 128, 128, max(L, 128), torch.bfloat16, dev, layers)
    lens.fill_(L)
    q = torch.randn(1, 32, 128, device=dev, dtype=torch.bfloat16)
    dk.plan_internal_splits = lambda n_bh, bound, ks, S=S: S
    for it in range(10):
        for kc, vc in caches:
            dk.head_sparse_decode_attention(q, kc, vc, lens, pt, bm, 8, 128)
    torch.cuda.synchronize()
