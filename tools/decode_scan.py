"""Launch the decode kernel over a grid of (L, splits) so a rocprofv3 trace shows T(bytes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "compactor-vllm_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

import compactor_vllm_amd.attention.sparse_decode_kernel as dk
from compactor_vllm_amd import _lib
from microbench import build_cache

dev = torch.device("cuda:0")
variant = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
ident = len(sys.argv) > 2 and sys.argv[2] == 'ident'
_lib.lib().cvllm_debug_set_decode_variant(variant)
for B, L, S in [(1, 2048, 4), (1, 16384, 32), (1, 32768, 64), (8, 16384, 8)]:
    layers = max(2, min(16, int(600e6 // (2 * B * 8 * L * 128 * 2)) + 1))
    caches, pt, bm, lens = build_cache(B, 8, 128, 128, L, torch.bfloat16, dev, layers)
    if ident:
        pt = torch.arange(pt.numel(), dtype=torch.int32, device=dev).view_as(pt)
    q = torch.randn(B, 32, 128, device=dev, dtype=torch.bfloat16)
    dk.plan_internal_splits = lambda n_bh, bound, ks, S=S: S
    for it in range(6):
        for kc, vc in caches:
            dk.head_sparse_decode_attention(q, kc, vc, lens, pt, bm, 8, 128)
    torch.cuda.synchronize()
    del caches
