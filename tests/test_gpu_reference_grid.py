"""GPU: the reference's OWN test matrices (tests/test_triton_attention.py:33-68, tests/test_store_kv.py:27-45),
restated against an independent fp32 torch evaluation on the GPU (flash-attn, the reference's oracle, is not
installable here) and the CPU oracle for the store semantics.  Same shapes, page size 256 with 256 logical
pages, identity page table phys = (b*HKV + h)*P + lp, seed 1234, fp16, atol 3e-3 (:283, :403)."""
import collections
import math

import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu

HQ, HKV, D, PS, NLP = 32, 8, 128, 256, 256


def _build_cache(B, lens, dev, dtype=torch.float16):
    g = torch.Generator(device=dev).manual_seed(1234)
    n_pages = B * HKV * NLP
    kc = torch.zeros(n_pages * PS, D, device=dev, dtype=dtype)
    vc = torch.zeros_like(kc)
    pt = torch.arange(n_pages, dtype=torch.int32, device=dev).view(B, HKV, NLP)
    for b in range(B):
        for h in range(HKV):
            L = lens[b]
            if L:
                base = int(pt[b, h, 0]) * PS  # identity table: logical rows are physically contiguous
                kc[base : base + L] = torch.randn(L, D, device=dev, generator=g).to(dtype)
                vc[base : base + L] = torch.randn(L, D, device=dev, generator=g).to(dtype)
    return kc, vc, pt


def _dense_ref(q, K, V, causal_offset):
    """q [HQ, Lq, D], K/V [HKV, Lk, D] fp32; query t sees keys <= causal_offset + t."""
    G = HQ // HKV
    Lq, Lk = q.shape[1], K.shape[1]
    out = torch.empty_like(q)
    t = torch.arange(Lq, device=q.device)[:, None]
    j = torch.arange(Lk, device=q.device)[None, :]
    mask = j <= (causal_offset + t)
    for h in range(HKV):
        for c0 in range(0, Lq, 2048):  # chunk the queries to bound the logits matrix
            qs = q[h * G : (h + 1) * G, c0 : c0 + 2048]
            logits = torch.einsum("gqd,kd->gqk", qs, K[h]) / math.sqrt(D)
            logits = logits.masked_fill(~mask[c0 : c0 + 2048][None], float("-inf"))
            out[h * G : (h + 1) * G, c0 : c0 + 2048] = torch.softmax(logits, -1) @ V[h]
    return out


@pytest.mark.parametrize("B", [1, 2, 3, 8])
@pytest.mark.parametrize("cache_len", [0, 1, 70, 128, 8193])
@pytest.mark.parametrize("append_len", [1, 2, 13, 8000])
def test_prefill_reference_grid(dev, B, cache_len, append_len):
    from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache

    dtype = torch.float16
    kc, vc, pt = _build_cache(B, [cache_len] * B, dev)
    lens = torch.full((B, HKV), cache_len, dtype=torch.int32, device=dev)
    cu = (torch.arange(B + 1, device=dev) * append_len).to(torch.int32)
    N = B * append_len
    q = torch.randn(N, HQ, D, device=dev, dtype=dtype)
    k = torch.randn(N, HKV, D, device=dev, dtype=dtype)
    v = torch.randn_like(k)
    bm = torch.arange(B, device=dev, dtype=torch.int32)
    out = causal_sparse_varlen_with_cache(q, k, v, kc, vc, lens, pt, bm, cu, append_len, cache_len, HKV, PS,
                                          1.0 / math.sqrt(D))
    check = range(B) if append_len < 8000 else [0, B - 1]  # the 8000-token cases: first and last sequence
    for b in check:
        s, e = b * append_len, (b + 1) * append_len
        Kf = torch.stack([torch.cat([kc[int(pt[b, h, 0]) * PS :][:cache_len], k[s:e, h]]) for h in range(HKV)]).float()
        Vf = torch.stack([torch.cat([vc[int(pt[b, h, 0]) * PS :][:cache_len], v[s:e, h]]) for h in range(HKV)]).float()
        ref = _dense_ref(q[s:e].transpose(0, 1).float(), Kf, Vf, cache_len)
        got = out[s:e].transpose(0, 1).float()
        assert torch.allclose(got, ref, rtol=1e-6, atol=3e-3), (got - ref).abs().max()


@pytest.mark.parametrize("B", [1, 2, 3, 8])
@pytest.mark.parametrize("cache_len", [1, 2, 70, 128, 8000])
def test_decode_reference_grid(dev, B, cache_len):
    from compactor_vllm_amd.attention.sparse_decode_kernel import head_sparse_decode_attention

    dtype = torch.float16
    kc, vc, pt = _build_cache(B, [cache_len] * B, dev)
    lens = torch.full((B, HKV), cache_len, dtype=torch.int32, device=dev)
    q = torch.randn(B, HQ, D, device=dev, dtype=dtype)
    bm = torch.arange(B, device=dev, dtype=torch.int32)
    out = head_sparse_decode_attention(q, kc, vc, lens, pt, bm, HKV, PS, 1.0 / math.sqrt(D))  # key_split: auto
    for b in range(B):
        Kf = torch.stack([kc[int(pt[b, h, 0]) * PS :][:cache_len] for h in range(HKV)]).float()
        Vf = torch.stack([vc[int(pt[b, h, 0]) * PS :][:cache_len] for h in range(HKV)]).float()
        ref = _dense_ref(q[b][:, None].float(), Kf, Vf, cache_len)[:, 0]
        assert torch.allclose(out[b].float(), ref, rtol=1e-6, atol=3e-3)


_STORE_GRID = [(B, frac, H, Dh, L, ps) for B in [1, 2, 3, 8] for frac in [0.10, 0.20, 0.30, 0.40] for H in [2, 4, 8]
               for Dh in [32, 64, 128] for L in [10, 20, 30, 70, 1000] for ps in [128, 256]]


@pytest.mark.parametrize("B,frac,H,Dh,L,ps", _STORE_GRID[::7])  # every 7th of the reference's 1440 combinations
def test_store_topk_reference_grid(dev, B, frac, H, Dh, L, ps):
    """tests/test_store_kv.py:47-173: scores_to_retain_indices + prefill_store_topk_kv(PAD_TO_PAGE_SIZE=False,
    TRITON_RESERVED_BATCH=-1); lengths == per-head counts of the selected indices, cached rows == selected
    source rows as multisets."""
    from compactor_vllm_amd.compression.common import scores_to_retain_indices
    from compactor_vllm_amd.kv_cache.store_kv_cache import prefill_store_topk_kv

    dtype = torch.float16
    TOP_K = int(L * H * frac)
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    cu = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    cu[1:] = torch.cumsum(lens, 0)
    N = B * L
    keys = torch.randn(N, H, Dh, dtype=dtype, device=dev)
    vals = torch.randn_like(keys)
    scores = torch.randn(N, H, dtype=torch.float32, device=dev)
    top_k_eff = max(0, min(TOP_K, L * H))
    if top_k_eff == 0:
        pytest.skip("reference builds an empty selection here")
    idx = scores_to_retain_indices(scores, cu, L, top_k_eff, H)
    assert idx.shape == (B, top_k_eff) and idx.dtype == torch.int64
    LP = max(1, (top_k_eff + ps - 1) // ps)
    n_pages = B * H * LP + 32
    kc = torch.empty(n_pages * ps, Dh, dtype=dtype, device=dev)
    vc = torch.empty_like(kc)
    pt = torch.arange(B * H * LP, dtype=torch.int32, device=dev).view(B, H, LP)
    local_lens = torch.zeros(B, H, dtype=torch.int32, device=dev)
    bm = torch.arange(B, dtype=torch.int32, device=dev)
    retain = torch.full((B,), top_k_eff, dtype=torch.int32, device=dev)
    prefill_store_topk_kv(new_keys=keys, new_vals=vals, indices_topk=idx, num_tokens_to_retain=retain, page_table=pt,
                          batch_mapping=bm, bh_lens=local_lens, PAGE_SIZE=ps, k_cache=kc, v_cache=vc,
                          PAD_TO_PAGE_SIZE=False, TRITON_RESERVED_BATCH=-1)
    torch.cuda.synchronize()
    idx_c, lens_c = idx.cpu(), local_lens.cpu()
    kcc, vcc, kcpu, vcpu, ptc = kc.cpu(), vc.cpu(), keys.cpu(), vals.cpu(), pt.cpu()
    # the ranking itself: descending scores, exactly top_k_eff entries of this sequence
    sc = scores.cpu().reshape(-1)
    for b in range(B):
        ranked = sc[idx_c[b]]
        assert (ranked[:-1] >= ranked[1:]).all()
        assert ((idx_c[b] // H >= int(cu[b])) & (idx_c[b] // H < int(cu[b + 1]))).all()
        hed = (idx_c[b] % H).tolist()
        counts = collections.Counter(hed)
        per_head = collections.defaultdict(list)
        for t, h in zip((idx_c[b] // H).tolist(), hed):
            per_head[h].append(t)
        for h in range(H):
            Lh = int(lens_c[b, h])
            assert Lh == counts.get(h, 0)
            if Lh == 0:
                continue
            rows = O.cache_rows(ptc[b, h], Lh, ps)
            exp_k = sorted(map(tuple, kcpu[per_head[h], h].tolist()))
            exp_v = sorted(map(tuple, vcpu[per_head[h], h].tolist()))
            assert sorted(map(tuple, kcc[rows].tolist())) == exp_k
            assert sorted(map(tuple, vcc[rows].tolist())) == exp_v
