"""GPU parity (through the C ABI): decode attention (a2), cache writes (a3, a4), select + compaction
(a9, a10) against the committed golden vectors and the CPU oracle."""
import math

import pytest
import torch

from golden_io import list_cases, load_case
from helpers import kept_sets_from_lists, mk_paged, tol
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


def _g(c, dev, *names):
    return [c[n].to(dev) for n in names]


# ------------------------------------------------------------------------------------------ a2
@pytest.mark.parametrize("name", list_cases("decode_"))
@pytest.mark.parametrize("key_split", [None, 1, 3])
def test_decode_golden(dev, name, key_split):
    from compactor_vllm_amd.attention.sparse_decode_kernel import head_sparse_decode_attention

    c = load_case(name)
    q, kc, vc, lens, pt, bm = _g(c, dev, "q", "k_cache", "v_cache", "seq_lens_bh", "page_table", "batch_mapping")
    out = head_sparse_decode_attention(q, kc, vc, lens, pt, bm, c["HKV"], c["PAGE_SIZE"], c["sm_scale"],
                                       key_split=key_split)
    torch.cuda.synchronize()
    assert out.shape == q.shape and out.dtype == q.dtype
    ref = c["out"].float()
    assert torch.allclose(out.cpu().float(), ref, rtol=1e-6, atol=tol(q.dtype)), (out.cpu().float() - ref).abs().max()
    # and against the fp32 oracle (tighter in practice than the reference's fp16 partials)
    orc = O.decode_attention(c["q"], c["k_cache"], c["v_cache"], c["seq_lens_bh"], c["page_table"],
                             c["batch_mapping"], c["HKV"], c["PAGE_SIZE"], c["sm_scale"]).float()
    assert torch.allclose(out.cpu().float(), orc, rtol=1e-6, atol=tol(q.dtype))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,HQ,HKV,D,PS,maxlen", [
    (1, 32, 8, 128, 128, 5000),   # long, per-head-varying: exercises many splits
    (8, 32, 8, 128, 256, 700),    # reference test shape (tests/test_triton_attention.py:52-68) w/ page 256
    (3, 8, 8, 64, 128, 300),      # G = 1
    (2, 16, 2, 128, 128, 1),      # L = 1 everywhere
    (2, 16, 2, 256, 128, 200),    # D = 256, G = 8
])
def test_decode_oracle_shapes(dev, dtype, B, HQ, HKV, D, PS, maxlen):
    """Per-head-varying lengths, shuffled pages, batch_mapping != arange, a zero-length head and a
    RESERVED-style empty row (untested upstream)."""
    from compactor_vllm_amd.attention.sparse_decode_kernel import head_sparse_decode_attention

    g = torch.Generator().manual_seed(B * 1000 + D + maxlen)
    lens = torch.randint(1, maxlen + 1, (B, HKV), generator=g, dtype=torch.int32)
    lens[-1, -1] = maxlen
    if maxlen > 1:
        lens[0, 0] = 0  # empty head -> zeros (documented deviation from quirk Q6)
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=7)
    q = torch.randn(B, HQ, D, generator=g).to(dtype)
    scale = 1.0 / math.sqrt(D)
    out = head_sparse_decode_attention(q.to(dev), kc.to(dev), vc.to(dev), lens.to(dev), pt.to(dev), bm.to(dev),
                                       HKV, PS, scale)
    torch.cuda.synchronize()
    ref = O.decode_attention(q, kc, vc, lens, pt, bm, HKV, PS, scale)
    assert torch.allclose(out.cpu().float(), ref.float(), rtol=1e-6, atol=tol(dtype)), \
        (out.cpu().float() - ref.float()).abs().max()
    if maxlen > 1:
        G = HQ // HKV
        assert (out[0, :G].cpu().float() == 0).all()


def test_decode_rejects_bad_args(dev):
    from compactor_vllm_amd.attention.sparse_decode_kernel import head_sparse_decode_attention

    q = torch.zeros(1, 6, 128, dtype=torch.float16, device=dev)  # G = 3 unsupported
    kc = torch.zeros(128, 128, dtype=torch.float16, device=dev)
    lens = torch.ones(1, 2, dtype=torch.int32, device=dev)
    pt = torch.zeros(2, 2, 1, dtype=torch.int32, device=dev)
    bm = torch.ones(1, dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError):
        head_sparse_decode_attention(q, kc, kc, lens, pt, bm, 2, 128)
    with pytest.raises(AssertionError):
        head_sparse_decode_attention(q, kc, kc, lens, pt, bm, 2, 100)  # PAGE_SIZE % 32


def test_num_splits_heuristic_matches_reference_table():
    """Known answers computed with the reference's heuristic (SURVEY App. C occupancy note)."""
    from compactor_vllm_amd.attention.sparse_decode_kernel import num_splits_heuristic

    assert num_splits_heuristic(8, 16384, 256, 12) == 9
    assert num_splits_heuristic(8, 32768, 256, 12) == 9
    assert num_splits_heuristic(64, 65536, 256, 12) == 3
    assert num_splits_heuristic(205, 65536, 256, 12) == 1
    assert num_splits_heuristic(8, 1024, 256, 12) == 1


# ------------------------------------------------------------------------------------- a3 / a4
@pytest.mark.parametrize("name", list_cases("storeall_"))
def test_store_all_golden(dev, name):
    from compactor_vllm_amd.kv_cache.store_kv_cache import prefill_store_all_kv

    c = load_case(name)
    HQ, HKV, D = c["HQ"], c["HKV"], c["D"]
    qkv = c["qkv"].to(dev)
    N = qkv.shape[0]
    k = qkv[:, HQ * D : (HQ + HKV) * D].view(N, HKV, D)  # strided views, like the model's qkv.split
    v = qkv[:, (HQ + HKV) * D :].view(N, HKV, D)
    kc, vc, pt, bm, l, cu = _g(c, dev, "k_cache0", "v_cache0", "page_table", "batch_mapping", "bh_lens0", "cu_seqlens_k")
    prefill_store_all_kv(new_keys=k, new_values=v, cu_seqlens_k=cu, max_seqlen_k=int(cu.diff().max()), k_cache=kc,
                         v_cache=vc, page_table=pt, bh_lens=l, batch_mapping=bm, PAGE_SIZE=c["PAGE_SIZE"])
    torch.cuda.synchronize()
    assert torch.equal(l.cpu(), c["bh_lens"])
    assert torch.equal(kc.cpu(), c["k_cache"]) and torch.equal(vc.cpu(), c["v_cache"])


@pytest.mark.parametrize("name", list_cases("decodestore_"))
def test_decode_store_golden(dev, name):
    from compactor_vllm_amd.kv_cache.store_kv_cache import decode_store_kv

    c = load_case(name)
    key, val, kc, vc, pt, bm, l = _g(c, dev, "key", "value", "k_cache0", "v_cache0", "page_table", "batch_mapping",
                                     "bh_lens0")
    decode_store_kv(key=key, value=val, batch_mapping=bm, bh_lens=l, page_table=pt, k_cache=kc, v_cache=vc,
                    PAGE_SIZE=c["PAGE_SIZE"])
    torch.cuda.synchronize()
    assert torch.equal(l.cpu(), c["bh_lens"])
    assert torch.equal(kc.cpu(), c["k_cache"]) and torch.equal(vc.cpu(), c["v_cache"])


def test_store_then_decode_roundtrip(dev):
    """store_all -> decode_store -> decode attention == dense attention over [prompt rows || new row]."""
    from compactor_vllm_amd.attention.sparse_decode_kernel import head_sparse_decode_attention
    from compactor_vllm_amd.kv_cache.store_kv_cache import decode_store_kv, prefill_store_all_kv

    dtype, B, HQ, HKV, D, PS = torch.bfloat16, 2, 8, 2, 128, 128
    g = torch.Generator().manual_seed(3)
    al = [200, 77]
    lens0 = torch.zeros(B, HKV, dtype=torch.int32)
    cap = torch.tensor(al, dtype=torch.int32)[:, None].repeat(1, HKV) + 1
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, cap, dtype, seed=5)
    cu = torch.tensor([0, al[0], al[0] + al[1]], dtype=torch.int32)
    k = torch.randn(sum(al), HKV, D, generator=g).to(dtype)
    v = torch.randn(sum(al), HKV, D, generator=g).to(dtype)
    k1 = torch.randn(B, HKV, D, generator=g).to(dtype)
    v1 = torch.randn(B, HKV, D, generator=g).to(dtype)
    q = torch.randn(B, HQ, D, generator=g).to(dtype)
    kcd, vcd, ptd, bmd, ld = kc.to(dev), vc.to(dev), pt.to(dev), bm.to(dev), lens0.to(dev)
    prefill_store_all_kv(new_keys=k.to(dev), new_values=v.to(dev), cu_seqlens_k=cu.to(dev), max_seqlen_k=max(al),
                         k_cache=kcd, v_cache=vcd, page_table=ptd, bh_lens=ld, batch_mapping=bmd, PAGE_SIZE=PS)
    decode_store_kv(key=k1.to(dev), value=v1.to(dev), batch_mapping=bmd, bh_lens=ld, page_table=ptd, k_cache=kcd,
                    v_cache=vcd, PAGE_SIZE=PS)
    out = head_sparse_decode_attention(q.to(dev), kcd, vcd, ld, ptd, bmd, HKV, PS)
    torch.cuda.synchronize()
    G = HQ // HKV
    for b in range(B):
        for h in range(HKV):
            K = torch.cat([k[cu[b] : cu[b + 1], h], k1[b, h][None]]).float()
            V = torch.cat([v[cu[b] : cu[b + 1], h], v1[b, h][None]]).float()
            p = torch.softmax(q[b, h * G : (h + 1) * G].float() @ K.T / math.sqrt(D), -1)
            assert torch.allclose(out[b, h * G : (h + 1) * G].cpu().float(), p @ V, atol=2e-2)
    assert torch.equal(ld.cpu(), torch.tensor(al, dtype=torch.int32)[:, None].repeat(1, HKV) + 1)


# ------------------------------------------------------------------------------------ a9 + a10
def _check_cache_sets(kc, vc, keys, vals, kept_sets, cu, pt, bm, lens0, new_lens, PS):
    B, H = new_lens.shape
    i = 0
    for b in range(B):
        for h in range(H):
            toks = kept_sets[i]
            i += 1
            L0, L1 = int(lens0[b, h]), int(new_lens[b, h])
            rows = O.cache_rows(pt[int(bm[b]), h], L1, PS)[L0:]
            src = [int(cu[b]) + t for t in toks]
            # our compaction is token-ordered, so the comparison is exact and ordered
            assert torch.equal(kc[rows], keys[src, h]) and torch.equal(vc[rows], vals[src, h])


@pytest.mark.parametrize("name", list_cases("select_"))
def test_select_compact_golden(dev, name):
    """Bit-exact selection vs the REFERENCE's retained sets (set equality per (b,h) + exact bh_lens)."""
    from compactor_vllm_amd.compression.common import extract_and_store_top_kv

    c = load_case(name)
    H, PS = c["HKV"], c["PAGE_SIZE"]
    sc, cu, ret, keys, vals, pt, bm, kc, vc, l = _g(c, dev, "scores", "cu_seqlens_k", "retain", "keys", "vals",
                                                    "page_table", "batch_mapping", "k_cache0", "v_cache0", "bh_lens0")
    maxL = int(c["cu_seqlens_k"].diff().max())
    kept, new_lens = extract_and_store_top_kv(sc, cu, maxL, maxL * H, H, keys, vals, ret, pt, bm, l, kc, vc, PS,
                                              PAD_TO_PAGE_SIZE=bool(c["pad"]))
    torch.cuda.synchronize()
    assert torch.equal(l.cpu(), c["bh_lens"]) and torch.equal(new_lens.cpu(), c["bh_lens"])
    sets = kept_sets_from_lists(kept.cpu(), new_lens.cpu(), c["bh_lens0"])
    offs = c["kept_offs"].tolist()
    ref_sets = [c["kept_flat"][offs[i] : offs[i + 1]].tolist() for i in range(len(offs) - 1)]
    assert sets == ref_sets
    _check_cache_sets(kc.cpu(), vc.cpu(), c["keys"], c["vals"], sets, c["cu_seqlens_k"], c["page_table"],
                      c["batch_mapping"], c["bh_lens0"], new_lens.cpu(), PS)


@pytest.mark.parametrize("B,H,lens,ratio,ties", [
    (1, 8, [4096], 0.5, False),
    (3, 8, [1000, 37, 2500], 0.3, False),
    (2, 4, [700, 300], 0.5, True),      # heavy ties: quantised scores + inf blocks
    (2, 2, [5, 128], 1.0, False),       # first+last >= L  (quirk Q2: retain = 1)
    (4, 1, [64, 65, 127, 129], 0.7, True),
    (2, 3, [500, 129], 0.4, True),      # H not a power of two: per-element counting fallback
    (1, 6, [1000], 0.3, False),
])
def test_select_oracle_random(dev, B, H, lens, ratio, ties):
    """Against the CPU oracle with the canonical tie rule (score desc, flat index asc), incl. tie-heavy
    inputs, lengths around page boundaries, a RESERVED row, and non-zero starting lengths."""
    from compactor_vllm_amd.compression.common import select_retained

    PS = 128
    g = torch.Generator().manual_seed(sum(lens) + H)
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
    N = int(cu[-1])
    sc = torch.randn(N, H, generator=g)
    if ties:
        sc = (sc * 2).round() / 2
    first, last = 16, 64
    for b in range(B):
        s, L = int(cu[b]), lens[b]
        sc[s : s + min(first, L)] = float("inf")
        sc[s + max(L - last, 0) : s + L] = float("inf")
    retain = torch.tensor([O.retain_count(ratio, L, first, last, H) for L in lens], dtype=torch.int32)
    lens0 = torch.randint(0, 3, (B, H), generator=g, dtype=torch.int32) * 5
    bm = torch.arange(1, B + 1, dtype=torch.int32)
    if B >= 3:
        bm[1] = 0  # RESERVED_BATCH: keeps nothing
    kept_o, lens_o = O.retained_sets(sc, cu, retain, lens0, bm, PS, True)
    kept, new_lens = select_retained(sc.to(dev), cu.to(dev), max(lens), retain.to(dev), bm.to(dev), lens0.to(dev), PS,
                                     True)
    torch.cuda.synchronize()
    assert torch.equal(new_lens.cpu(), lens_o)
    sets = kept_sets_from_lists(kept.cpu(), new_lens.cpu(), lens0)
    ref = [sorted(kept_o[b][h]) for b in range(B) for h in range(H)]
    assert sets == ref


@pytest.mark.parametrize("name", list_cases("select_"))
def test_ranked_store_golden(dev, name):
    """prefill_store_topk_kv with the reference's own rank list (tests/test_store_kv.py semantics:
    exact bh_lens + multiset of cached rows per (b,h))."""
    from compactor_vllm_amd.kv_cache.store_kv_cache import prefill_store_topk_kv

    c = load_case(name)
    H, PS = c["HKV"], c["PAGE_SIZE"]
    idx, cu, ret, keys, vals, pt, bm, kc, vc, l = _g(c, dev, "ref_indices", "cu_seqlens_k", "retain", "keys", "vals",
                                                     "page_table", "batch_mapping", "k_cache0", "v_cache0", "bh_lens0")
    prefill_store_topk_kv(new_keys=keys, new_vals=vals, indices_topk=idx, num_tokens_to_retain=ret, page_table=pt,
                          batch_mapping=bm, bh_lens=l, k_cache=kc, v_cache=vc, PAGE_SIZE=PS,
                          PAD_TO_PAGE_SIZE=bool(c["pad"]), cu_seqlens_k=cu)
    torch.cuda.synchronize()
    assert torch.equal(l.cpu(), c["bh_lens"])
    kcc, vcc, B = kc.cpu(), vc.cpu(), l.shape[0]
    offs = c["kept_offs"].tolist()
    i = 0
    for b in range(B):
        for h in range(H):
            L1 = int(c["bh_lens"][b, h])
            toks = c["kept_flat"][offs[i] : offs[i + 1]].tolist()  # the REFERENCE's retained set of (b,h)
            i += 1
            rows = O.cache_rows(c["page_table"][int(c["batch_mapping"][b]), h], L1, PS)
            src = [int(c["cu_seqlens_k"][b]) + t for t in toks]
            for cache, new in ((kcc, c["keys"]), (vcc, c["vals"])):
                a = sorted(map(tuple, cache[rows].float().tolist()))
                r = sorted(map(tuple, new[src, h].float().tolist()))
                assert a == r


def test_fused_decode_step_equals_four_call_sequence(dev):
    """cvllm_decode_append_attn == index_select + decode_store_kv + decode attention + index_copy_, incl. a
    RESERVED_BATCH padding row (skipped by the store, attends to nothing)."""
    from compactor_vllm_amd.attention.sparse_decode_kernel import fused_decode_step, head_sparse_decode_attention
    from compactor_vllm_amd.kv_cache.store_kv_cache import decode_store_kv

    dtype, B, HQ, HKV, D, PS = torch.bfloat16, 4, 32, 8, 128, 128
    g = torch.Generator().manual_seed(9)
    lens = torch.randint(0, 700, (B, HKV), generator=g, dtype=torch.int32)
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens + 1, dtype, seed=3, bmax_extra=2)
    bm[2] = 0  # RESERVED padding row
    table = torch.zeros(pt.shape[0], HKV, dtype=torch.int32)
    for b in range(B):
        if int(bm[b]) != 0:
            table[int(bm[b])] = lens[b]
    q = torch.randn(B, HQ, D, generator=g).to(dtype).to(dev)
    k1 = torch.randn(B, HKV, D, generator=g).to(dtype).to(dev)
    v1 = torch.randn(B, HKV, D, generator=g).to(dtype).to(dev)
    ptd, bmd = pt.to(dev), bm.to(dev)
    # reference order
    kc_a, vc_a, tab_a = kc.to(dev), vc.to(dev), table.to(dev)
    sl = tab_a.index_select(0, bmd).contiguous()
    decode_store_kv(key=k1, value=v1, batch_mapping=bmd, bh_lens=sl, page_table=ptd, k_cache=kc_a, v_cache=vc_a,
                    PAGE_SIZE=PS)
    o_a = head_sparse_decode_attention(q, kc_a, vc_a, sl, ptd, bmd, HKV, PS)
    keep = bmd != 0
    tab_a.index_copy_(0, bmd[keep].long(), sl[keep])
    # fused
    kc_b, vc_b, tab_b = kc.to(dev), vc.to(dev), table.to(dev)
    o_b = fused_decode_step(q, k1, v1, kc_b, vc_b, tab_b, ptd, bmd, HKV, PS)
    torch.cuda.synchronize()
    assert torch.equal(tab_a, tab_b) and torch.equal(kc_a, kc_b) and torch.equal(vc_a, vc_b)
    assert torch.equal(o_a[keep], o_b[keep])
    assert (o_b[2] == 0).all()
