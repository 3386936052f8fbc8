"""Shared builders for the parity tests (synthetic paged caches with the reference's layout)."""
from __future__ import annotations

import torch

from oracle import ref_cpu as O


def mk_paged(B, HKV, D, PS, lens_bh, dtype, seed, extra_pages=3, bmax_extra=1, min_pages=1):
    """Random paged cache on CPU: shuffled page table, batch_mapping != arange (rows >= 1)."""
    g = torch.Generator().manual_seed(seed)
    P = max(min_pages, int(max(-(-int(x) // PS) for x in lens_bh.flatten().tolist())))
    Bmax = B + bmax_extra
    n_pages = (Bmax + 1) * HKV * P + extra_pages
    perm = torch.randperm(n_pages, generator=g)[: (Bmax + 1) * HKV * P]
    page_table = perm.view(Bmax + 1, HKV, P).to(torch.int32).contiguous()
    bm = (torch.randperm(Bmax, generator=g)[:B] + 1).to(torch.int32)
    k_cache = torch.full((n_pages * PS, D), 0.5, dtype=dtype)
    v_cache = torch.full((n_pages * PS, D), -0.5, dtype=dtype)
    for b in range(B):
        for h in range(HKV):
            L = int(lens_bh[b, h])
            rows = O.cache_rows(page_table[int(bm[b]), h], L, PS)
            k_cache[rows] = torch.randn(L, D, generator=g).to(dtype)
            v_cache[rows] = torch.randn(L, D, generator=g).to(dtype)
    return k_cache, v_cache, page_table, bm, P


def kept_sets_from_lists(kept_idx, new_lens, lens0):
    """[B,H,max] int32 + lengths -> python list of sorted token lists per (b,h)."""
    B, H, _ = kept_idx.shape
    out = []
    for b in range(B):
        for h in range(H):
            n = int(new_lens[b, h]) - int(lens0[b, h])
            out.append(sorted(kept_idx[b, h, :n].tolist()))
    return out


def tol(dtype):
    """Attention tolerances vs the fp32 oracle (SURVEY §8 'Attention tolerance'): the reference's own
    bar is atol 3e-3 in fp16 (tests/test_triton_attention.py:283,403); bf16 has 8x coarser mantissa."""
    return 3e-3 if dtype == torch.float16 else 2e-2
