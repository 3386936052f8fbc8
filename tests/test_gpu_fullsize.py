"""GPU: BASELINE.json full sizes through size-independent properties (the CPU oracle cannot run these in
seconds): identities, round trips and an independent torch-on-GPU evaluation of the same math."""
import math

import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


def _identity_cache(B, HKV, D, PS, L, dtype, dev):
    P = -(-L // PS)
    n_pages = (B + 1) * HKV * P
    pt = torch.randperm(n_pages, device=dev).view(B + 1, HKV, P).to(torch.int32)
    bm = torch.arange(1, B + 1, dtype=torch.int32, device=dev)
    kc = torch.zeros(n_pages * PS, D, dtype=dtype, device=dev)
    vc = torch.zeros_like(kc)
    return kc, vc, pt, bm, P


@pytest.mark.parametrize("L", [32768, 131072 - 256])
def test_select_full_size_exact_vs_topk(dev, L):
    """C3 size (32768 tokens x 8 heads) and the 128 K context of configs[4], retain 50 %: the kept set must equal
    torch.topk's set (tie-free random scores) plus, per head, that head's next-best tokens up to a page boundary."""
    from compactor_vllm_amd.compression.common import select_retained

    H, PS = 8, 128
    g = torch.Generator(device=dev).manual_seed(3)
    sc = torch.randn(L, H, device=dev, generator=g)
    sc[:16] = float("inf")
    sc[-64:] = float("inf")
    retain = O.retain_count(0.5, L, 16, 64, H)
    cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
    lens0 = torch.zeros(1, H, dtype=torch.int32, device=dev)
    kept, new_lens = select_retained(sc, cu, L, torch.tensor([retain], dtype=torch.int32, device=dev),
                                     torch.ones(1, dtype=torch.int32, device=dev), lens0, PS, True)
    torch.cuda.synchronize()
    flat = sc.reshape(-1)
    top = torch.topk(flat, retain).indices  # inf entries all lie inside the kept set: no tie at the cut
    joint = torch.zeros(L * H, dtype=torch.bool, device=dev)
    joint[top] = True
    joint = joint.view(L, H)
    assert (new_lens % PS == 0).all()
    assert int(new_lens.sum()) >= retain and int(new_lens.sum()) < retain + H * PS
    for h in range(H):
        n = int(new_lens[0, h])
        mine = torch.zeros(L, dtype=torch.bool, device=dev)
        toks = kept[0, h, :n].long()
        assert (toks[1:] > toks[:-1]).all()  # ascending token order = deterministic slots
        mine[toks] = True
        c_h = int(joint[:, h].sum())
        assert (mine & joint[:, h]).sum() == c_h  # superset of the joint top-k of this head
        exp_n = c_h if c_h % PS == 0 else c_h + (PS - c_h % PS)
        assert n == min(exp_n, L)
        # the padding tokens are exactly the head's next-best ones
        col = sc[:, h].clone()
        col[joint[:, h]] = float("-inf")
        pad = torch.topk(col, n - c_h).indices if n > c_h else torch.empty(0, dtype=torch.long, device=dev)
        assert torch.equal(torch.sort(pad).values, torch.sort(toks[~joint[toks, h]]).values)


def test_store_all_roundtrip_32k(dev):
    """store_all at C2/C3 size, then read every row back through the page table: identity, lengths += L."""
    from compactor_vllm_amd.kv_cache.store_kv_cache import prefill_store_all_kv

    B, HKV, D, PS, L = 1, 8, 128, 128, 32768
    dtype = torch.bfloat16
    kc, vc, pt, bm, P = _identity_cache(B, HKV, D, PS, L, dtype, dev)
    k = torch.randn(L, HKV, D, device=dev).to(dtype)
    v = torch.randn(L, HKV, D, device=dev).to(dtype)
    lens = torch.zeros(B, HKV, dtype=torch.int32, device=dev)
    cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
    prefill_store_all_kv(new_keys=k, new_values=v, cu_seqlens_k=cu, max_seqlen_k=L, k_cache=kc, v_cache=vc, page_table=pt,
                         bh_lens=lens, batch_mapping=bm, PAGE_SIZE=PS)
    torch.cuda.synchronize()
    assert (lens == L).all()
    i = torch.arange(L, device=dev)
    for h in range(HKV):
        rows = pt[1, h].long()[i // PS] * PS + i % PS
        assert torch.equal(kc[rows], k[:, h]) and torch.equal(vc[rows], v[:, h])


@pytest.mark.parametrize("B,L", [(1, 131072), (8, 65536)])
def test_decode_long_context_vs_torch(dev, B, L):
    """128K context (C5 per-sequence size) and an 8-sequence batch: against an fp32 torch evaluation on the GPU."""
    from compactor_vllm_amd.attention.sparse_decode_kernel import head_sparse_decode_attention

    HQ, HKV, D, PS = 32, 8, 128, 128
    dtype = torch.bfloat16
    kc, vc, pt, bm, P = _identity_cache(B, HKV, D, PS, L, dtype, dev)
    kc.normal_()
    vc.normal_()
    g = torch.Generator(device=dev).manual_seed(1)
    lens = torch.randint(L // 2, L + 1, (B, HKV), device=dev, generator=g, dtype=torch.int32)
    lens[0, 0] = L
    q = torch.randn(B, HQ, D, device=dev, generator=g).to(dtype)
    out = head_sparse_decode_attention(q, kc, vc, lens, pt, bm, HKV, PS)
    torch.cuda.synchronize()
    G = HQ // HKV
    i = torch.arange(L, device=dev)
    for b in range(B):
        for h in (0, 3, 7):
            n = int(lens[b, h])
            rows = (pt[int(bm[b]), h].long()[i // PS] * PS + i % PS)[:n]
            K, V = kc[rows].float(), vc[rows].float()
            p = torch.softmax(q[b, h * G : (h + 1) * G].float() @ K.T / math.sqrt(D), -1)
            assert torch.allclose(out[b, h * G : (h + 1) * G].float(), p @ V, atol=2e-2)


@pytest.mark.parametrize("N", [32768, 131072 - 256])
def test_prefill_full_size_properties(dev, N):
    """C3-size (32 K) and C5-size (128 K - 256, BASELINE.json configs[4]) dense causal prefill: V == const -> output ==
    const; sampled rows against the fp32 definition."""
    from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache

    dtype, B, HQ, HKV, D, PS = torch.bfloat16, 1, 32, 8, 128, 128
    g = torch.Generator(device=dev).manual_seed(7)
    q = torch.randn(N, HQ, D, device=dev, generator=g).to(dtype)
    k = torch.randn(N, HKV, D, device=dev, generator=g).to(dtype)
    v = torch.randn(N, HKV, D, device=dev, generator=g).to(dtype)
    kc = torch.zeros(PS, D, dtype=dtype, device=dev)
    lens = torch.zeros(B, HKV, dtype=torch.int32, device=dev)
    pt = torch.zeros(2, HKV, 1, dtype=torch.int32, device=dev)
    bm = torch.ones(1, dtype=torch.int32, device=dev)
    cu = torch.tensor([0, N], dtype=torch.int32, device=dev)
    args = (kc, kc, lens, pt, bm, cu, N, 0, HKV, PS)
    out = causal_sparse_varlen_with_cache(q, k, v, *args)
    G = HQ // HKV
    for t in (0, 63, 64, 8191, 20000, N - 1):
        for hq in (0, 13, 31):
            kk = k[: t + 1, hq // G].float()
            p = torch.softmax(q[t, hq].float() @ kk.T / math.sqrt(D), -1)
            assert torch.allclose(out[t, hq].float(), p @ v[: t + 1, hq // G].float(), atol=2e-2), (t, hq)
    ones = torch.full_like(v, 0.5)
    out1 = causal_sparse_varlen_with_cache(q, k, ones, *args)
    assert torch.allclose(out1.float(), torch.full_like(out1, 0.5).float(), atol=4e-3)


def test_compactor_pipeline_32k_shapes_and_invariants(dev):
    """Full C3 scoring -> select -> compaction -> decode on one layer: lengths are page multiples, protected
    tokens are always retained, the decode over the compacted cache equals attention over the kept rows."""
    from compactor_vllm_amd.attention.sparse_decode_kernel import head_sparse_decode_attention
    from compactor_vllm_amd.compression.common import extract_and_store_top_kv
    from compactor_vllm_amd.compression.compactor import approximate_leverage_scores, non_causal_attn_scores

    dtype, HQ, HKV, D, PS, L = torch.bfloat16, 32, 8, 128, 128, 32768
    g = torch.Generator(device=dev).manual_seed(11)
    q = (torch.randn(L, HQ, D, device=dev, generator=g) * 0.3).to(dtype)
    k = (torch.randn(L, HKV, D, device=dev, generator=g) * 0.3).to(dtype)
    v = torch.randn(L, HKV, D, device=dev, generator=g).to(dtype)
    PHI = (torch.randn(D, 48, device=dev, generator=g) / math.sqrt(48)).to(dtype)
    cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
    pre = approximate_leverage_scores(k, [L], PHI, normalize=True, chunk_size=512)
    sc = non_causal_attn_scores(q, k, v, cu, L, chunk_size=128, sm_scale=1.0, normalize=True, accum_scores=pre,
                                context_lens=[L], protected_first_tokens=[16], protected_last_tokens=[64],
                                accum_blending=0.5)
    assert sc.shape == (L, HKV) and sc.dtype == torch.float32
    assert torch.isinf(sc[:16]).all() and torch.isinf(sc[-64:]).all() and torch.isfinite(sc[16:-64]).all()
    fin = sc[16:-64]
    assert abs(float(fin.mean())) < 0.05 and 0.8 < float(fin.std()) < 1.4  # z(mass) + 0.5 z(leverage)
    kc, vc, pt, bm, P = _identity_cache(1, HKV, D, PS, L, dtype, dev)
    lens = torch.zeros(1, HKV, dtype=torch.int32, device=dev)
    retain = torch.tensor([O.retain_count(0.5, L, 16, 64, HKV)], dtype=torch.int32, device=dev)
    kept, new_lens = extract_and_store_top_kv(sc, cu, L, L * HKV, HKV, k, v, retain, pt, bm, lens, kc, vc, PS)
    torch.cuda.synchronize()
    assert torch.equal(lens, new_lens) and (lens % PS == 0).all()
    assert retain.item() <= int(lens.sum()) < retain.item() + HKV * PS
    for h in range(HKV):
        toks = kept[0, h, : int(lens[0, h])].long()
        assert (toks[:16] == torch.arange(16, device=dev)).all() and (toks[-64:] == torch.arange(L - 64, L, device=dev)).all()
    q1 = torch.randn(1, HQ, D, device=dev, generator=g).to(dtype)
    o = head_sparse_decode_attention(q1, kc, vc, lens, pt, bm, HKV, PS)
    G = HQ // HKV
    for h in (0, 5):
        toks = kept[0, h, : int(lens[0, h])].long()
        p = torch.softmax(q1[0, h * G : (h + 1) * G].float() @ k[toks, h].float().T / math.sqrt(D), -1)
        assert torch.allclose(o[0, h * G : (h + 1) * G].float(), p @ v[toks, h].float(), atol=2e-2)
