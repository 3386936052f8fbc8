"""Kernel micro-benchmarks on one MI355X (HIP-event timed on the launch stream).

    python tools/microbench.py decode [--L 16384] [--B 1] [--splits 0]
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "compactor-vllm_amd"))

import torch


def time_fn(fn, iters=50, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def build_cache(B, HKV, D, PS, L, dtype, dev, layers=1, seed=0):
    """`layers` independent caches so consecutive launches do not re-hit the 256 MiB Infinity Cache."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    P = -(-L // PS)
    n_pages = (B + 1) * HKV * P
    caches = []
    for _ in range(layers):
        kc = torch.randn(n_pages * PS, D, device=dev, dtype=dtype)
        vc = torch.randn(n_pages * PS, D, device=dev, dtype=dtype)
        caches.append((kc, vc))
    pt = torch.randperm(n_pages, generator=g).view(B + 1, HKV, P).to(torch.int32).to(dev)
    bm = torch.arange(1, B + 1, dtype=torch.int32, device=dev)
    lens = torch.full((B, HKV), L, dtype=torch.int32, device=dev)
    return caches, pt, bm, lens


def bench_decode(args):
    import compactor_vllm_amd.attention.sparse_decode_kernel as dk

    dev = torch.device("cuda:0")
    dtype = torch.bfloat16
    B, HQ, HKV, D, PS, L = args.B, 32, 8, 128, 128, args.L
    layers = max(1, min(32, int(600e6 // (2 * B * HKV * L * D * 2)) + 1))
    caches, pt, bm, lens = build_cache(B, HKV, D, PS, L, dtype, dev, layers)
    q = torch.randn(B, HQ, D, device=dev, dtype=dtype)
    splits_list = [args.splits] if args.splits else [16, 32, 64]
    bytes_alg = 2 * D * 2 * B * HKV * L + 2 * B * HQ * D * 2
    from compactor_vllm_amd import _lib
    if args.flush:
        junk_a = torch.empty(128 << 20, dtype=torch.uint8, device=dev)
        junk_b = torch.empty_like(junk_a)
    for variant in [0]:
        for S in splits_list:
            dk.plan_internal_splits = lambda n_bh, bound, ks, S=S: S
            # one HIP graph with `layers` launches over distinct caches: no host launch overhead in the timing
            dk.head_sparse_decode_attention(q, caches[0][0], caches[0][1], lens, pt, bm, HKV, PS)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for kc, vc in caches:
                    if args.flush:  # stream 256 MB through the L2s between launches: the metadata lines go cold, as they
                        junk_b.copy_(junk_a)  # do between two layers of a decode step (the GEMVs' weights in between)
                    out = dk.head_sparse_decode_attention(q, kc, vc, lens, pt, bm, HKV, PS)
            us = time_fn(g.replay, iters=50, warmup=max(3, int(40e3 / (25.0 * layers)))) / layers  # ~40 ms of warm replays: settled clocks
            print(f"decode variant={variant} B={B} L={L} splits={S:3d} layers={layers}: {us:8.2f} us/(stage1+merge)  "
                  f"{bytes_alg / us / 1e6:7.3f} TB/s algorithmic", flush=True)


def bench_prefill(args):
    from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache

    dev = torch.device("cuda:0")
    dtype = torch.bfloat16
    B, HQ, HKV, D, PS = args.B, 32, 8, 128, 128
    for L in ([args.L] if args.L != 16384 or True else []):
        N = B * L
        q = torch.randn(N, HQ, D, device=dev, dtype=dtype)
        k = torch.randn(N, HKV, D, device=dev, dtype=dtype)
        v = torch.randn(N, HKV, D, device=dev, dtype=dtype)
        if args.fused_v:  # the engine's layout: V is a strided view of the fused projection output [N, (HQ + 2 HKV) D]
            qkv = torch.randn(N, (HQ + 2 * HKV) * D, device=dev, dtype=dtype)
            v = qkv[:, (HQ + HKV) * D:].view(N, HKV, D)
            if args.fused_v > 1:
                k = qkv[:, HQ * D:(HQ + HKV) * D].view(N, HKV, D)
        kc = torch.zeros(PS, D, device=dev, dtype=dtype)
        lens = torch.zeros(B, HKV, dtype=torch.int32, device=dev)
        pt = torch.zeros(B + 1, HKV, 1, dtype=torch.int32, device=dev)
        bm = torch.arange(1, B + 1, dtype=torch.int32, device=dev)
        cu = torch.arange(0, B + 1, dtype=torch.int32, device=dev) * L
        fn = lambda: causal_sparse_varlen_with_cache(q, k, v, kc, kc, lens, pt, bm, cu, L, 0, HKV, PS)
        # warm until ~40 ms of this kernel have run: with 2 warm launches a 16 K kernel (2 ms) was still timed on a ramping
        # clock (2.30 ms against 2.03 steady, profiles/r02_prefill_workgroup_stamps.txt)
        warm = max(2, min(40, int(40e-3 / (4e-9 * 2 * L * L / 2 * B / 1e3 + 1e-4))))
        us = time_fn(fn, iters=max(5, warm // 2), warmup=warm)
        flops = 4 * D * HQ * B * (L * (L + 1) / 2)
        print(f"prefill B={B} L={L}: {us / 1e3:9.3f} ms  {flops / us / 1e6:8.1f} TFLOP/s (causal flops)", flush=True)


def bench_scoring(args):
    """Stand-alone cost of every store-stream / scoring kernel at the C3 layer shape (nothing else running)."""
    from compactor_vllm_amd.compression.common import extract_and_store_top_kv, select_retained
    from compactor_vllm_amd.compression.compactor import approximate_leverage_scores, non_causal_attn_scores
    from compactor_vllm_amd.compression.snapkv import query_aware_key_scores
    from compactor_vllm_amd.kv_cache.store_kv_cache import prefill_store_all_kv

    dev = torch.device("cuda:0")
    dtype = torch.bfloat16
    B, HQ, HKV, D, PS, L = args.B, 32, 8, 128, 128, args.L
    N = B * L
    q = (torch.randn(N, HQ, D, device=dev) * 0.3).to(dtype)
    k = (torch.randn(N, HKV, D, device=dev) * 0.3).to(dtype)
    v = torch.randn(N, HKV, D, device=dev).to(dtype)
    PHI = (torch.randn(D, 48, device=dev) / 48 ** 0.5).to(dtype)
    cu = (torch.arange(0, B + 1, device=dev) * L).to(torch.int32)
    lens = [L] * B
    P = -(-L // PS)
    n_pages = (B + 1) * HKV * P
    pt = torch.randperm(n_pages, device=dev).view(B + 1, HKV, P).to(torch.int32)
    bm = torch.arange(1, B + 1, dtype=torch.int32, device=dev)
    kc = torch.zeros(n_pages * PS, D, dtype=dtype, device=dev)
    vc = torch.zeros_like(kc)
    pre = approximate_leverage_scores(k, lens, PHI, normalize=True, chunk_size=512)
    sc = non_causal_attn_scores(q, k, v, cu, L, chunk_size=128, sm_scale=1.0, normalize=True, accum_scores=pre,
                                context_lens=lens, protected_first_tokens=[16] * B, protected_last_tokens=[64] * B,
                                accum_blending=0.5)
    retain = torch.tensor([max(int(round(0.5 * (L - 80) * HKV)), 1)] * B, dtype=torch.int32, device=dev)
    zero = torch.zeros(B, HKV, dtype=torch.int32, device=dev)

    def store_top():
        l = zero.clone()
        extract_and_store_top_kv(sc, cu, L, L * HKV, HKV, k, v, retain, pt, bm, l, kc, vc, PS)

    def store_all():
        l = zero.clone()
        prefill_store_all_kv(new_keys=k, new_values=v, cu_seqlens_k=cu, max_seqlen_k=L, k_cache=kc, v_cache=vc,
                             page_table=pt, bh_lens=l, batch_mapping=bm, PAGE_SIZE=PS)

    items = [
        ("leverage (sketch+solve+zscore)", lambda: approximate_leverage_scores(k, lens, PHI, normalize=True, chunk_size=512)),
        ("chunk mass + zscore + blend + protect", lambda: non_causal_attn_scores(
            q, k, v, cu, L, chunk_size=128, sm_scale=1.0, normalize=True, accum_scores=pre, context_lens=lens,
            protected_first_tokens=[16] * B, protected_last_tokens=[64] * B, accum_blending=0.5)),
        ("snapkv scores", lambda: query_aware_key_scores(q, k, cu, cu, w=32, max_seqlen_k=L)),
        ("select (joint + per-head)", lambda: select_retained(sc, cu, L, retain, bm, zero, PS, True)),
        ("select + compaction", store_top),
        ("store_all", store_all),
    ]
    if args.with_producer:
        from compactor_vllm_amd.layers.rotary_embedding import fused_qkv_rope, get_rope
        rope = get_rope(D, D, max(L, 1024), 500000.0, None).to(dev)
        qkv = torch.randn(N, (HQ + 2 * HKV) * D, device=dev).to(dtype)
        pos = torch.arange(L, device=dev).repeat(B)
        items.append(("qkv split + rope producer", lambda: fused_qkv_rope(qkv, pos, rope.cos_sin_cache, HQ, HKV, D)))
    if args.only:
        items = [it for it in items if args.only in it[0]]
    for name, fn in items:
        # default ~30 ms of warm calls: settled clocks (3 warm calls read 5-15 % higher)
        us = time_fn(fn, iters=args.iters or 50, warmup=args.warmup if args.warmup >= 0 else 300)
        print(f"scoring B={B} L={L}: {name:42s} {us:9.1f} us", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what")
    ap.add_argument("--L", type=int, default=16384)
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--splits", type=int, default=0)
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--fused-v", type=int, default=0, help="prefill: 1 = V (2 = K and V) as strided views of a fused qkv buffer")
    ap.add_argument("--flush", action="store_true", help="decode: a 128 MB device copy between launches (cold metadata, as inside a decode step)")
    ap.add_argument("--iters", type=int, default=0)      # scoring: timed calls per item (0 = default)
    ap.add_argument("--warmup", type=int, default=-1)    # scoring: warm calls per item (-1 = default)
    ap.add_argument("--with-producer", action="store_true")  # scoring: add the f-2 producer kernel
    ap.add_argument("--only", default="")                # scoring: substring filter on the item name
    a = ap.parse_args()
    {"decode": bench_decode, "prefill": bench_prefill, "scoring": bench_scoring}[a.what](a)
