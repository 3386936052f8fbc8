"""Generate the committed golden vectors by RUNNING THE UPSTREAM REFERENCE (build container only).

    PYTHONDONTWRITEBYTECODE=1 TRITON_INTERPRET=1 python tests/golden/gen_fixtures.py

Every case stores the exact inputs and the reference's outputs (tests/golden/<case>.npz) and,
while generating, checks oracle/ref_cpu.py against the reference so the oracle is pinned at the
moment the vectors are made.  The vectors are data only: no reference source travels.
"""
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import _ref_loader  # noqa: E402

R = _ref_loader.load()

import torch  # noqa: E402

from golden_io import save_case  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402

torch.set_num_threads(8)


def mk_paged(B, HKV, D, PS, lens_bh, dtype, seed, extra_pages=3, bmax_extra=1):
    """Random paged cache: shuffled page table, batch_mapping != arange (rows >= 1), cache rows of
    valid positions filled with N(0,1), everything else NaN-free garbage (0.5)."""
    g = torch.Generator().manual_seed(seed)
    P = max(1, int(max(-(-int(x) // PS) for x in lens_bh.flatten().tolist())))
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


def maxdiff(a, b):
    return (a.float() - b.float()).abs().max().item()


# ------------------------------------------------------------------------------------------ a1
def gen_prefill():
    cases = [
        # name, dtype, B, HQ, HKV, D, PS, cache lens (per b or 'var'), append lens
        ("prefill_f16_nocache", torch.float16, 2, 8, 2, 128, 128, [0, 0], [70, 13]),
        ("prefill_f16_varhead", torch.float16, 3, 8, 2, 128, 128, "var300", [1, 2, 200]),
        ("prefill_bf16_varhead", torch.bfloat16, 2, 8, 2, 128, 128, "var300", [13, 130]),
        ("prefill_f16_ps256_hq32", torch.float16, 2, 32, 8, 128, 256, "var300", [5, 66]),
        ("prefill_f16_d64", torch.float16, 2, 4, 4, 64, 128, "var140", [1, 90]),
    ]
    for name, dtype, B, HQ, HKV, D, PS, cl, al in cases:
        g = torch.Generator().manual_seed(sum(name.encode()) % 10000)
        if isinstance(cl, str):
            mx = int(cl[3:])
            lens = torch.randint(0, mx + 1, (B, HKV), generator=g, dtype=torch.int32)
            lens[0, 0] = 0
            lens[-1, -1] = mx
            if B > 1:
                lens[1, 0] = PS  # exactly one page
        else:
            lens = torch.tensor(cl, dtype=torch.int32)[:, None].repeat(1, HKV)
        kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=11)
        cu = torch.tensor([0] + list(torch.tensor(al).cumsum(0).tolist()), dtype=torch.int32)
        N = int(cu[-1])
        q = torch.randn(N, HQ, D, generator=g).to(dtype)
        # k and v as strided views of a fused qkv buffer, like the model's qkv.split (llama3.py:96-100)
        qkv = torch.randn(N, (HQ + 2 * HKV) * D, generator=g).to(dtype)
        k = qkv[:, HQ * D : (HQ + HKV) * D].view(N, HKV, D)
        v = qkv[:, (HQ + HKV) * D :].view(N, HKV, D)
        scale = 1.0 / math.sqrt(D)
        ref = R.pk.causal_sparse_varlen_with_cache(
            q, k, v, kc, vc, lens, pt, bm, cu, max(al), int(lens.max()), HKV, PS, scale
        )
        mine = O.prefill_attention(q, k, v, kc, vc, lens, pt, bm, cu, HKV, PS, scale)
        print(f"{name}: oracle vs reference max|d| = {maxdiff(ref, mine):.3e}")
        save_case(name, q=q, qkv=qkv, k_cache=kc, v_cache=vc, seq_lens_bh=lens, page_table=pt,
                  batch_mapping=bm, cu_seqlens_q=cu, HQ=HQ, HKV=HKV, D=D, PAGE_SIZE=PS,
                  sm_scale=scale, out=ref)


# ------------------------------------------------------------------------------------------ a2
def gen_decode():
    cases = [
        ("decode_f16_small", torch.float16, 3, 8, 2, 128, 128, [1, 2, 70], 1),
        ("decode_f16_var_split3", torch.float16, 3, 8, 2, 128, 128, "var1000", 3),
        ("decode_bf16_var_split2", torch.bfloat16, 2, 32, 8, 128, 128, "var400", 2),
        ("decode_f16_ps256", torch.float16, 2, 8, 2, 128, 256, "var600", 1),
        ("decode_f16_d64", torch.float16, 2, 8, 4, 64, 128, "var300", 2),
    ]
    for name, dtype, B, HQ, HKV, D, PS, cl, split in cases:
        g = torch.Generator().manual_seed(sum(name.encode()) % 10000)
        if isinstance(cl, str):
            mx = int(cl[3:])
            lens = torch.randint(1, mx + 1, (B, HKV), generator=g, dtype=torch.int32)
            lens[0, 0] = 1
            lens[-1, -1] = mx
            if B > 1:
                lens[1, 0] = PS
        else:
            lens = torch.tensor(cl, dtype=torch.int32)[:, None].repeat(1, HKV)
        kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=23)
        q = torch.randn(B, HQ, D, generator=g).to(dtype)
        scale = 1.0 / math.sqrt(D)
        ref = R.dk.head_sparse_decode_attention(q, kc, vc, lens, pt, bm, HKV, PS, scale, key_split=split)
        mine = O.decode_attention(q, kc, vc, lens, pt, bm, HKV, PS, scale)
        print(f"{name}: oracle vs reference max|d| = {maxdiff(ref, mine):.3e}")
        save_case(name, q=q, k_cache=kc, v_cache=vc, seq_lens_bh=lens, page_table=pt, batch_mapping=bm,
                  HQ=HQ, HKV=HKV, D=D, PAGE_SIZE=PS, sm_scale=scale, key_split=split, out=ref)


# ------------------------------------------------------------------------------------- a3 / a4
def gen_stores():
    for name, dtype, B, HKV, D, PS, al in [
        ("storeall_f16", torch.float16, 3, 2, 64, 128, [10, 300, 70]),
        ("storeall_bf16_hkv8", torch.bfloat16, 2, 8, 128, 128, [130, 17]),
    ]:
        g = torch.Generator().manual_seed(5)
        lens0 = torch.randint(0, 200, (B, HKV), generator=g, dtype=torch.int32)
        total = lens0 + torch.tensor(al, dtype=torch.int32)[:, None]
        kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, total, dtype, seed=31)
        cu = torch.tensor([0] + torch.tensor(al).cumsum(0).tolist(), dtype=torch.int32)
        N = int(cu[-1])
        HQ = 2 * HKV
        qkv = torch.randn(N, (HQ + 2 * HKV) * D, generator=g).to(dtype)
        k = qkv[:, HQ * D : (HQ + HKV) * D].view(N, HKV, D)
        v = qkv[:, (HQ + HKV) * D :].view(N, HKV, D)
        kc_ref, vc_ref, l_ref = kc.clone(), vc.clone(), lens0.clone()
        R.st.prefill_store_all_kv(new_keys=k, new_values=v, cu_seqlens_k=cu, max_seqlen_k=max(al),
                                  k_cache=kc_ref, v_cache=vc_ref, page_table=pt, bh_lens=l_ref,
                                  batch_mapping=bm, PAGE_SIZE=PS)
        kc_o, vc_o, l_o = kc.clone(), vc.clone(), lens0.clone()
        O.store_all_kv(k, v, cu, kc_o, vc_o, pt, l_o, bm, PS)
        ok = torch.equal(kc_o, kc_ref) and torch.equal(vc_o, vc_ref) and torch.equal(l_o, l_ref)
        print(f"{name}: oracle == reference: {ok}")
        assert ok
        save_case(name, qkv=qkv, HQ=HQ, HKV=HKV, D=D, PAGE_SIZE=PS, cu_seqlens_k=cu, k_cache0=kc, v_cache0=vc,
                  page_table=pt, batch_mapping=bm, bh_lens0=lens0, k_cache=kc_ref, v_cache=vc_ref, bh_lens=l_ref)

    for name, dtype, B, HKV, D, PS in [
        ("decodestore_f16", torch.float16, 4, 2, 64, 128),
        ("decodestore_bf16", torch.bfloat16, 3, 8, 128, 128),
    ]:
        g = torch.Generator().manual_seed(6)
        lens0 = torch.randint(0, 300, (B, HKV), generator=g, dtype=torch.int32)
        lens0[0, 0] = 0
        lens0[1, 1] = PS - 1
        lens0[2, 0] = PS
        kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens0 + 1, dtype, seed=37)
        bm[B - 1] = 0  # RESERVED_BATCH padding row: must be skipped (store_kv_cache.py:395-397)
        key = torch.randn(B, HKV, D, generator=g).to(dtype)
        val = torch.randn(B, HKV, D, generator=g).to(dtype)
        kc_ref, vc_ref, l_ref = kc.clone(), vc.clone(), lens0.clone()
        R.st.decode_store_kv(key=key, value=val, batch_mapping=bm, bh_lens=l_ref, page_table=pt,
                             k_cache=kc_ref, v_cache=vc_ref, PAGE_SIZE=PS)
        kc_o, vc_o, l_o = kc.clone(), vc.clone(), lens0.clone()
        O.decode_store_kv(key, val, bm, l_o, pt, kc_o, vc_o, PS)
        ok = torch.equal(kc_o, kc_ref) and torch.equal(vc_o, vc_ref) and torch.equal(l_o, l_ref)
        print(f"{name}: oracle == reference: {ok}")
        assert ok
        save_case(name, key=key, value=val, HKV=HKV, D=D, PAGE_SIZE=PS, k_cache0=kc, v_cache0=vc, page_table=pt,
                  batch_mapping=bm, bh_lens0=lens0, k_cache=kc_ref, v_cache=vc_ref, bh_lens=l_ref)


# ------------------------------------------------------------------------------------ a9 + a10
def kept_from_cache(kc, new_keys, cu, pt, bm, lens_final, lens0, PS):
    """Recover the retained token set of every (b,h) from the cache contents (rows are unique)."""
    B, H = lens_final.shape
    flat, offs = [], [0]
    for b in range(B):
        s, e = int(cu[b]), int(cu[b + 1])
        for h in range(H):
            L0, L1 = int(lens0[b, h]), int(lens_final[b, h])
            rows = O.cache_rows(pt[int(bm[b]), h], L1, PS)[L0:]
            toks = []
            src = new_keys[s:e, h].float()
            for r in rows.tolist():
                m = (src == kc[r].float()[None, :]).all(-1).nonzero().flatten().tolist()
                assert len(m) == 1, (b, h, r, m)
                toks.append(m[0])
            flat.extend(sorted(toks))
            offs.append(len(flat))
    return torch.tensor(flat, dtype=torch.int32), torch.tensor(offs, dtype=torch.int32)


def gen_select():
    cases = [
        # name, B, HKV, D, PS, lens, ratios, first, last, pad
        ("select_hkv8_pad", 3, 8, 32, 128, [300, 517, 70], [0.5, 0.25, 0.1], 16, 8, True),
        ("select_hkv2_pad", 2, 2, 32, 128, [1300, 30], [0.1, 0.5], 4, 4, True),
        ("select_hkv8_nopad", 2, 8, 64, 128, [300, 140], [0.25, 0.5], 16, 16, False),
        ("select_short_q2", 2, 8, 32, 128, [50, 200], [1.0, 0.5], 16, 64, True),
    ]
    for name, B, H, D, PS, lens, ratios, first, last, pad in cases:
        g = torch.Generator().manual_seed(77)
        dtype = torch.float16
        cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
        N = int(cu[-1])
        scores = torch.randn(N, H, generator=g)
        for b in range(B):  # protected tokens = +inf, as the scoring stage produces them
            s, L = int(cu[b]), lens[b]
            if L <= first + last:
                continue  # all-protected sequences would be one big tie: order undefined upstream
            scores[s : s + first] = float("inf")
            scores[s + L - last : s + L] = float("inf")
        retain = torch.tensor([O.retain_count(r, L, first, last, H) for r, L in zip(ratios, lens)],
                              dtype=torch.int32)
        lens0 = torch.zeros(B, H, dtype=torch.int32)
        full = torch.tensor(lens, dtype=torch.int32)[:, None].repeat(1, H)
        kc, vc, pt, bm, P = mk_paged(B, H, D, PS, full, dtype, seed=41)
        keys = torch.randn(N, H, D, generator=g).to(dtype)
        vals = torch.randn(N, H, D, generator=g).to(dtype)
        kc_ref, vc_ref, l_ref = kc.clone(), vc.clone(), lens0.clone()
        idx = R.cm.scores_to_retain_indices(scores, cu, max(lens), max(lens) * H, H)
        R.st.prefill_store_topk_kv(new_keys=keys, new_vals=vals, indices_topk=idx, num_tokens_to_retain=retain,
                                   page_table=pt, batch_mapping=bm, bh_lens=l_ref, k_cache=kc_ref, v_cache=vc_ref,
                                   PAGE_SIZE=PS, PAD_TO_PAGE_SIZE=pad, cu_seqlens_k=cu)
        kept_flat, kept_offs = kept_from_cache(kc_ref, keys, cu, pt, bm, l_ref, lens0, PS)
        kept_o, l_o = O.retained_sets(scores, cu, retain, lens0, bm, PS, pad)
        flat_o = [t for b in range(B) for h in range(H) for t in sorted(kept_o[b][h])]
        ok = torch.equal(l_o, l_ref) and flat_o == kept_flat.tolist()
        print(f"{name}: retain={retain.tolist()} final lens row0={l_ref[0].tolist()} oracle==reference: {ok}")
        assert ok
        save_case(name, scores=scores, cu_seqlens_k=cu, retain=retain, keys=keys, vals=vals, HKV=H, D=D, PAGE_SIZE=PS,
                  pad=int(pad), page_table=pt, batch_mapping=bm, k_cache0=kc, v_cache0=vc, bh_lens0=lens0,
                  bh_lens=l_ref, kept_flat=kept_flat, kept_offs=kept_offs, ref_indices=idx)


# --------------------------------------------------------------------------- a5 / a6 / a7 / a8
def gen_scoring():
    # a5: leverage scores
    for name, dtype, lens, chunk, H, D in [
        ("leverage_f32_chunk512", torch.float32, [257, 127, 1300], 512, 2, 128),
        ("leverage_f32_whole", torch.float32, [257, 127, 600], -1, 2, 128),
        ("leverage_bf16_chunk512", torch.bfloat16, [600, 1100], 512, 4, 128),
        ("leverage_f16_chunk128", torch.float16, [300, 128], 128, 2, 64),
    ]:
        g = torch.Generator().manual_seed(3)
        N = sum(lens)
        # keys with a low-rank + noise structure so leverage scores are informative
        base = torch.randn(N, H, 16, generator=g) @ torch.randn(16, D, generator=g) * 0.3
        k = (base + torch.randn(N, H, D, generator=g)).to(dtype)
        PHI = (torch.randn(D, 48, generator=g) / math.sqrt(48)).to(dtype)
        for norm in (False, True):
            ref = R.cp.approximate_leverage_scores(k.clone(), lens, PHI, normalize=norm, chunk_size=chunk)
            mine = O.leverage_scores(k, lens, PHI, normalize=norm, chunk_size=chunk)
            print(f"{name} normalize={norm}: oracle vs reference max|d| = {maxdiff(ref, mine):.3e} "
                  f"(|ref| max {ref.float().abs().max():.3f})")
            save_case(f"{name}_norm{int(norm)}", k=k, PHI=PHI, context_lens=lens, chunk_size=chunk,
                      normalize=int(norm), out=ref)

    # a7: Compactor post-RoPE scores (full wrapper: mass, z-score per sequence, blend, protected)
    for name, dtype, lens, HQ, HKV, D, first, last in [
        ("chunkattn_f16", torch.float16, [257, 100, 600], 8, 2, 128, 16, 64),
        ("chunkattn_bf16", torch.bfloat16, [130, 64], 16, 4, 128, 4, 8),
        ("chunkattn_f16_d64", torch.float16, [200], 4, 4, 64, 16, 64),
    ]:
        g = torch.Generator().manual_seed(9)
        N = sum(lens)
        cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
        # 0.3-scaled so that sm_scale=1.0 logits are not a pure arg-max (still peaky, like real data)
        q = (torch.randn(N, HQ, D, generator=g) * 0.3).to(dtype)
        k = (torch.randn(N, HKV, D, generator=g) * 0.3).to(dtype)
        v = torch.zeros(N, HKV, D, dtype=dtype)
        pre = torch.randn(N, HKV, generator=g).to(dtype)
        B = len(lens)
        raw = R.cp.non_causal_attn_scores(q, k, v, cu, max(lens), chunk_size=128, sm_scale=1.0, normalize=False)
        ref = R.cp.non_causal_attn_scores(q, k, v, cu, max(lens), chunk_size=128, sm_scale=1.0, normalize=True,
                                          accum_scores=pre, context_lens=lens,
                                          protected_first_tokens=[first] * B, protected_last_tokens=[last] * B,
                                          accum_blending=0.5)
        raw_o = O.chunk_attn_mass(q, k, cu, 128, 1.0)
        mine = O.compactor_post_scores(q, k, cu, lens, pre, [first] * B, [last] * B)
        fin = torch.isfinite(ref)
        assert torch.equal(fin, torch.isfinite(mine))
        print(f"{name}: mass max|d| = {maxdiff(raw, raw_o):.3e}; final max|d| = {maxdiff(ref[fin], mine[fin]):.3e}")
        save_case(name, q=q, k=k, pre=pre, cu_seqlens=cu, context_lens=lens, first=first, last=last,
                  HQ=HQ, HKV=HKV, D=D, mass=raw, out=ref)

    # a8: SnapKV
    for name, dtype, lens, HQ, HKV, D in [
        ("snapkv_f16", torch.float16, [257, 100, 600], 8, 2, 128),
        ("snapkv_bf16", torch.bfloat16, [300, 40], 16, 4, 128),
    ]:
        g = torch.Generator().manual_seed(10)
        N = sum(lens)
        cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
        q = torch.randn(N, HQ, D, generator=g).to(dtype)
        k = torch.randn(N, HKV, D, generator=g).to(dtype)
        ref = R.sk.query_aware_key_scores(q, k, cu, cu, w=32)
        mine = O.snapkv_scores(q, k, cu, cu, 32)
        fin = torch.isfinite(ref)
        assert torch.equal(fin, torch.isfinite(mine))
        print(f"{name}: oracle vs reference max|d| = {maxdiff(ref[fin], mine[fin]):.3e}")
        save_case(name, q=q, k=k, cu_seqlens=cu, w=32, HQ=HQ, HKV=HKV, D=D, out=ref)


def gen_snapkv_ext():
    """SnapKV with one window per sequence and with the windowed z-score (normalize=True) - the two arguments the
    reference's engine never uses (snapkv.py:279-329, :351-357)."""
    for name, dtype, lens, HQ, HKV, D, w, norm in [
        ("snapkvx_f16_wvar_norm", torch.float16, [257, 100, 600], 8, 2, 128, [32, 8, 16], True),
        ("snapkvx_bf16_wvar", torch.bfloat16, [300, 40, 90], 16, 4, 128, [16, 32, 4], False),
        ("snapkvx_bf16_norm", torch.bfloat16, [200, 333], 16, 4, 128, 32, True),
    ]:
        g = torch.Generator().manual_seed(11)
        N = sum(lens)
        cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
        q = torch.randn(N, HQ, D, generator=g).to(dtype)
        k = torch.randn(N, HKV, D, generator=g).to(dtype)
        wt = w if isinstance(w, int) else torch.tensor(w, dtype=torch.int32)
        ref = R.sk.query_aware_key_scores(q, k, cu, cu, w=wt, normalize=norm)
        mine = O.snapkv_scores(q, k, cu, cu, w, normalize=norm)
        wl = [w] * len(lens) if isinstance(w, int) else w
        ok = torch.zeros(N, dtype=torch.bool)  # rows the reference defines: sequences with L > w
        s0 = 0
        for L, wb in zip(lens, wl):
            if L > wb:
                ok[s0 : s0 + L] = True
            s0 += L
        fin = torch.isfinite(ref) & ok[:, None]
        assert torch.equal(fin, torch.isfinite(mine) & ok[:, None])
        print(f"{name}: oracle vs reference max|d| = {maxdiff(ref[fin], mine[fin]):.3e}")
        save_case(name, q=q, k=k, cu_seqlens=cu, w=torch.tensor(wl, dtype=torch.int32), normalize=int(norm), HQ=HQ,
                  HKV=HKV, D=D, out=ref)


# ------------------------------------------------------------------------------------------ f-2
def gen_producer():
    """qkv split + (Qwen3) q/k RMSNorm + RoPE as the reference's model code runs them: the reference's own RMSNorm and
    RotaryEmbedding modules, eager (their @torch.compile wrappers are bypassed with TORCHDYNAMO_DISABLE=1 or, failing
    that, through __wrapped__-free re-binding below; compile only changes speed)."""
    import torch._dynamo

    torch._dynamo.config.disable = True
    import compactor_vllm.layers.layernorm as ln
    import compactor_vllm.layers.rotary_embedding as re

    for name, dtype, N, HQ, HKV, D, base, scaling, norm in [
        ("producer_bf16_llama", torch.bfloat16, 300, 8, 2, 128, 500000.0, ("llama3", 8.0, 1.0, 4.0, 8192), False),
        ("producer_f16_llama_d64", torch.float16, 120, 4, 4, 64, 10000.0, None, False),
        ("producer_bf16_qwen3", torch.bfloat16, 260, 8, 2, 128, 1000000.0, None, True),
        ("producer_f16_qwen3", torch.float16, 90, 4, 2, 128, 1000000.0, None, True),
    ]:
        g = torch.Generator().manual_seed(21)
        max_pos = 512
        qkv = torch.randn(N, (HQ + 2 * HKV) * D, generator=g).to(dtype)
        positions = torch.randint(0, max_pos, (N,), generator=g)
        positions[:5] = torch.tensor([0, 1, max_pos - 1, 17, 300])
        rope = re.RotaryEmbedding(D, D, max_pos, base, scaling)
        q = qkv[:, : HQ * D].view(N, HQ, D)
        k = qkv[:, HQ * D : (HQ + HKV) * D].view(N, HKV, D)
        qw = kw = None
        eps = 1e-6
        if norm:
            qn, kn = ln.RMSNorm(D, eps=eps), ln.RMSNorm(D, eps=eps)
            with torch.no_grad():
                qn.weight.copy_(1.0 + 0.2 * torch.randn(D, generator=g))
                kn.weight.copy_(1.0 + 0.2 * torch.randn(D, generator=g))
            qn, kn = qn.to(dtype), kn.to(dtype)  # the engine builds the model under default dtype = model dtype
            qw, kw = qn.weight.detach().clone(), kn.weight.detach().clone()
            with torch.no_grad():
                q, k = qn(q), kn(k)
        k_pre = k.clone()
        with torch.no_grad():
            q_rot, k_rot = rope(positions, q, k)
        cs = rope.cos_sin_cache.view(max_pos, D).clone()
        mine = O.qkv_producer(qkv, positions, O.rope_cos_sin_cache(D, max_pos, base, scaling), HQ, HKV, D, qw, kw, eps)
        assert torch.equal(O.rope_cos_sin_cache(D, max_pos, base, scaling), cs), "cos/sin cache restatement differs"
        assert torch.equal(mine[0], q_rot) and torch.equal(mine[1], k_rot) and torch.equal(mine[3], k_pre), name
        print(f"{name}: oracle == reference bit for bit (q, k, pre-RoPE k, cos/sin cache)")
        save_case(name, qkv=qkv, positions=positions, cos_sin=cs, HQ=HQ, HKV=HKV, D=D, eps=eps, base=base,
                  max_pos=max_pos, has_norm=int(norm), has_scaling=int(scaling is not None),
                  **({"q_norm_w": qw, "k_norm_w": kw} if norm else {}), q_rot=q_rot, k_rot=k_rot, k_pre=k_pre)


# ---------------------------------------------------------------------------- a8, the other two autotune outcomes
def gen_snapkv_tiles():
    """SnapKV with the reference's `_scores_from_logits_kernel` pinned to BLOCK_K = 32 and 64 (its autotuner picks one
    of {32, 64, 128}, snapkv.py:160-168; the 5-tap pooling is clipped at BLOCK_K tile edges, :253-262, so the pooled
    scores differ between the three).  The committed snapkv_* / snapkvx_* sets are the BLOCK_K = 128 outcome."""
    import triton

    for bk in (32, 64):
        R.sk._scores_from_logits_kernel.configs = [triton.Config({"BLOCK_Q": 64, "BLOCK_K": bk})]
        R.sk._scores_from_logits_kernel.cache.clear()
        for name, dtype, lens, HQ, HKV, D in [
            (f"snapkvt_f16_bk{bk}", torch.float16, [257, 100, 600], 8, 2, 128),
            (f"snapkvt_bf16_bk{bk}", torch.bfloat16, [300, 40, 161], 16, 4, 128),
        ]:
            g = torch.Generator().manual_seed(12)
            N = sum(lens)
            cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
            q = torch.randn(N, HQ, D, generator=g).to(dtype)
            k = torch.randn(N, HKV, D, generator=g).to(dtype)
            ref = R.sk.query_aware_key_scores(q, k, cu, cu, w=32)
            mine = O.snapkv_scores(q, k, cu, cu, 32, pool_tile=bk)
            other = O.snapkv_scores(q, k, cu, cu, 32, pool_tile=128)
            fin = torch.isfinite(ref)
            assert torch.equal(fin, torch.isfinite(mine))
            print(f"{name}: oracle(pool_tile={bk}) vs reference max|d| = {maxdiff(ref[fin], mine[fin]):.3e}; "
                  f"oracle(pool_tile=128) would differ by {maxdiff(ref[fin], other[fin]):.3e}")
            save_case(name, q=q, k=k, cu_seqlens=cu, w=32, HQ=HQ, HKV=HKV, D=D, pool_tile=bk, out=ref)
    R.sk._scores_from_logits_kernel.configs = [triton.Config({"BLOCK_Q": 64, "BLOCK_K": 128})]
    R.sk._scores_from_logits_kernel.cache.clear()


# -------------------------------------------------------------------- a2 with a RESERVED_BATCH padding row (Q6)
def gen_decode_reserved():
    """Decode attention as the engine's graph path calls it: the batch is padded with RESERVED_BATCH (row 0) entries
    whose lengths are 0 (model_runner.py:468-491; stage 1 returns without writing for L == 0,
    sparse_decode_kernel.py:281-283).  The reference's output rows of the padded entries are uninitialised memory
    (torch.empty): the vector stores which rows those are; every other row is the reference's."""
    for name, dtype, B, HQ, HKV, D, PS, split in [
        ("decoderes_f16", torch.float16, 4, 8, 2, 128, 128, 2),
        ("decoderes_bf16", torch.bfloat16, 3, 32, 8, 128, 128, 3),
    ]:
        g = torch.Generator().manual_seed(sum(name.encode()) % 10000)
        lens = torch.randint(1, 500, (B, HKV), generator=g, dtype=torch.int32)
        lens[0, 0] = 1
        reserved_rows = [1, B - 1]
        for rb in reserved_rows:
            lens[rb] = 0
        kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens.clamp_min(1), dtype, seed=29)
        for rb in reserved_rows:
            bm[rb] = 0
        q = torch.randn(B, HQ, D, generator=g).to(dtype)
        scale = 1.0 / math.sqrt(D)
        ref = R.dk.head_sparse_decode_attention(q, kc, vc, lens, pt, bm, HKV, PS, scale, key_split=split)
        mine = O.decode_attention(q, kc, vc, lens, pt, bm, HKV, PS, scale)
        live = torch.tensor([b not in reserved_rows for b in range(B)])
        print(f"{name}: oracle vs reference on the live rows max|d| = {maxdiff(ref[live], mine[live]):.3e}")
        save_case(name, q=q, k_cache=kc, v_cache=vc, seq_lens_bh=lens, page_table=pt, batch_mapping=bm, HQ=HQ, HKV=HKV,
                  D=D, PAGE_SIZE=PS, sm_scale=scale, key_split=split, live_rows=live.to(torch.int32), out=ref)


# ------------------------------------------------------------- a1 at BASELINE.json configs[0] (C1): 4096 dense
def c1_inputs(seed=1234):
    """Seeded inputs of the C1 case; tests/test_gpu_fullsize.py rebuilds them with the same calls (same torch build)."""
    N, HQ, HKV, D = 4096, 32, 8, 128
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(N, HQ, D, generator=g).to(torch.float16)
    k = torch.randn(N, HKV, D, generator=g).to(torch.float16)
    v = torch.randn(N, HKV, D, generator=g).to(torch.float16)
    return q, k, v


def gen_c1():
    """BASELINE.json configs[0]: HQ 32 / HKV 8 / D 128 / page 128, 4096 tokens, batch 1, dense causal attention with an
    empty cache (the shape of tests/test_triton_attention.py:200-287), fp16 - the reference's own kernel under the
    interpreter (about two minutes).  The 32 MiB output is not committed: 512 sampled (token, head) rows + checksums of
    the seeded inputs are."""
    N, HQ, HKV, D, PS = 4096, 32, 8, 128, 128
    q, k, v = c1_inputs()
    lens = torch.zeros(1, HKV, dtype=torch.int32)
    pt = torch.zeros(2, HKV, 1, dtype=torch.int32)
    bm = torch.tensor([1], dtype=torch.int32)
    kc = torch.zeros(PS, D, dtype=torch.float16)
    cu = torch.tensor([0, N], dtype=torch.int32)
    scale = 1.0 / math.sqrt(D)
    ref = R.pk.causal_sparse_varlen_with_cache(q, k, v, kc, kc.clone(), lens, pt, bm, cu, N, 0, HKV, PS, scale)
    g = torch.Generator().manual_seed(99)
    tok = torch.cat([torch.tensor([0, 1, 63, 64, 127, 128, 4095]), torch.randint(0, N, (505,), generator=g)])
    head = torch.randint(0, HQ, (tok.numel(),), generator=g)
    rows = ref[tok, head]
    mine = O.prefill_attention(q, k, v, kc, kc.clone(), lens, pt, bm, cu, HKV, PS, scale)
    print(f"c1prefill_4096: oracle vs reference max|d| over the whole output = {maxdiff(ref, mine):.3e}")
    chk = lambda t: int(t.view(torch.int16).to(torch.int64).sum())
    save_case("c1prefill_4096", seed=1234, N=N, HQ=HQ, HKV=HKV, D=D, PAGE_SIZE=PS, sm_scale=scale, tok=tok, head=head,
              rows=rows, q_checksum=chk(q), k_checksum=chk(k), v_checksum=chk(v),
              out_abs_mean=float(ref.float().abs().mean()))


# --------------------------------------------------------- f-1 host policy: page allocator and prefill admission
def gen_engine_policy():
    """Traces of the reference's own PagedKVCache (page_table.py:144-291) and Scheduler.get_prefill_batch
    (scheduler.py:65-108) on scripted operation lists: what they return and the state they leave, as JSON."""
    import json

    E = _ref_loader.load_engine_policy()
    L, P, NP, PS, H, MB = 3, 6, 40, 16, 2, 4
    cache = E.ptab.PagedKVCache(num_layers=L, max_logical_pages_per_head=P, num_pages=NP, page_size=PS, H_kv=H,
                                head_dim=8, max_num_batches=MB, dtype=torch.float16, device="cpu")
    g = torch.Generator().manual_seed(5)
    ops, rows = [], {}

    def snap(ret):
        st = {"ret": ret, "free_batches": list(cache.free_batches),
              "free_pages": [sorted(fp) for fp in cache.free_pages],
              "num_pages": cache.bh_num_pages.tolist(), "lens": cache.bh_seq_lens.tolist(), "tables": {}}
        for name, r in rows.items():
            npg = cache.bh_num_pages[:, r]
            st["tables"][name] = [[cache.page_table[l, r, h, : int(npg[l, h])].tolist() for h in range(H)]
                                  for l in range(L)]
        return st

    def do(op, *a):
        if op == "new_batch":
            r = cache.new_batch()
            if r is not None:
                rows[a[0]] = int(r)
            ret = r
        elif op == "reserve":
            ret = cache.reserve_tokens(rows[a[0]], a[1]).name
        elif op == "set_lens":  # what the store kernels do on the device: per-(layer, head) lengths
            cache.bh_seq_lens[:, rows[a[0]]] = torch.tensor(a[1], dtype=torch.int32)
            ret = None
        elif op == "reclaim":
            ret = cache.reclaim_pages(rows[a[0]], a[1])
        elif op == "free":
            ret = cache.free_batch(rows.pop(a[0]))
        ops.append({"op": op, "args": list(a), "after": snap(ret)})

    rl = lambda hi: torch.randint(0, hi + 1, (L, H), generator=g).tolist()
    do("new_batch", "a"); do("reserve", "a", 50); do("new_batch", "b"); do("reserve", "b", 33)
    do("set_lens", "a", rl(40)); do("reclaim", "a", 8)            # compression left per-head lengths, 8 new tokens
    do("reserve", "a", 8); do("reserve", "a", 30)                  # fits / grows
    do("new_batch", "c"); do("reserve", "c", 97)                   # exceeds max pages per head (6 * 16 = 96)
    do("reserve", "c", 96); do("new_batch", "d"); do("reserve", "d", 90)   # d: not enough pages left
    do("set_lens", "b", rl(33)); do("reclaim", "b", 0); do("free", "a")
    do("reserve", "d", 60); do("new_batch", "e"); do("new_batch", "f")    # rows run out at 4
    do("free", "c"); do("new_batch", "f"); do("reserve", "f", 16); do("reclaim", "f", 0)
    do("set_lens", "d", rl(60)); do("reclaim", "d", 5); do("free", "b"); do("free", "d"); do("free", "f"); do("free", "e")
    alloc = {"config": dict(num_layers=L, max_logical_pages_per_head=P, num_pages=NP, page_size=PS, H_kv=H,
                            max_num_batches=MB), "ops": ops}

    class Mgr:  # the five attributes get_prefill_batch reads from its KVCacheManager (scheduler.py:84-103)
        def __init__(s, **k):
            s.__dict__.update(k)

    sched_cases = []
    for cname, mk, prompts in [
        ("token_budget", dict(max_batched_tokens=1000, num_free_batches=8, num_free_pages=10000, page_size=128, num_kv_heads=8),
         [(400, 16), (500, 16), (200, 16), (100, 16), (90, 16)]),
        ("rows", dict(max_batched_tokens=100000, num_free_batches=2, num_free_pages=10000, page_size=128, num_kv_heads=8),
         [(400, 16), (500, 16), (200, 16)]),
        ("pages_strict", dict(max_batched_tokens=100000, num_free_batches=8, num_free_pages=64, page_size=128, num_kv_heads=8),
         [(500, 12), (500, 13), (100, 10), (1000, 24), (120, 8), (100, 28)]),
        ("skip_then_fit", dict(max_batched_tokens=4096, num_free_batches=3, num_free_pages=200, page_size=128, num_kv_heads=8),
         [(4000, 256), (5000, 1), (96, 256), (1, 1), (1, 1)]),
        ("nothing_fits", dict(max_batched_tokens=64, num_free_batches=8, num_free_pages=10000, page_size=128, num_kv_heads=8),
         [(65, 1), (100, 1)]),
    ]:
        seqs = [E.seq.Sequence(prompt_token_ids=[1] * pl, sampling_params=E.SamplingParams(max_new_tokens=mn))
                for pl, mn in prompts]
        sc = E.sched.Scheduler(seqs, Mgr(**mk), use_tqdm=False)
        rounds = []
        for _ in range(4):  # admit, mark running, ask again (pages / rows are the fake manager's: unchanged)
            batch = sc.get_prefill_batch()
            rounds.append([seqs.index(s) for s in batch])
            if not batch:
                break
            sc.add_running_sequence_ids([s.seq_id for s in batch], update_status=True)
        sched_cases.append({"name": cname, "manager": mk, "prompts": prompts, "rounds": rounds,
                            "total_tokens_input": sc.total_tokens_input})
        print(f"scheduler {cname}: admitted {rounds}")
    out = os.path.join(HERE, "engine_policy.json")
    with open(out, "w") as f:
        json.dump({"allocator": alloc, "scheduler": sched_cases}, f, separators=(",", ":"))
    print(f"engine policy: {len(ops)} allocator operations, {len(sched_cases)} admission cases -> {out}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["prefill", "decode", "stores", "select", "scoring", "producer", "snapkv_ext",
                             "snapkv_tiles", "decode_reserved", "c1", "engine_policy"]
    for w in which:
        {"prefill": gen_prefill, "decode": gen_decode, "stores": gen_stores, "select": gen_select,
         "scoring": gen_scoring, "producer": gen_producer, "snapkv_ext": gen_snapkv_ext,
         "snapkv_tiles": gen_snapkv_tiles, "decode_reserved": gen_decode_reserved, "c1": gen_c1,
         "engine_policy": gen_engine_policy}[w]()
