"""GPU parity sweeps over seeded random shapes (through the C ABI) against the CPU oracle: decode attention (plain
and with the fused append), decode store, prefill attention.  The fixed-shape tests pin the reference's own matrices
and the golden vectors; these walk the corners in between (page sizes 32..256, G 1..8, D 64/128, lengths around page
and split boundaries, empty heads, shuffled pages, batch_mapping != arange)."""
import math
import random

import pytest
import torch

from helpers import mk_paged, tol
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


def _cfg(seed):
    r = random.Random(seed)
    HKV = r.choice([1, 2, 4, 8])
    G = r.choice([1, 2, 4, 8])
    D = r.choice([64, 128])
    PS = r.choice([32, 64, 128, 256])
    B = r.choice([1, 2, 3, 5])
    dtype = r.choice([torch.float16, torch.bfloat16])
    return r, B, HKV * G, HKV, D, PS, dtype


def _lens(r, B, HKV, PS, maxlen):
    """lengths that sit on / next to page and split boundaries, plus zeros"""
    special = [0, 1, PS - 1, PS, PS + 1, 2 * PS, 16, 63, 64, 65, 255, 256, 257, maxlen]
    out = torch.zeros(B, HKV, dtype=torch.int32)
    for b in range(B):
        for h in range(HKV):
            out[b, h] = min(maxlen, r.choice(special) if r.random() < 0.5 else r.randint(0, maxlen))
    return out


@pytest.mark.parametrize("seed", range(24))
def test_decode_random(dev, seed):
    from compactor_vllm_amd.attention.sparse_decode_kernel import head_sparse_decode_attention

    r, B, HQ, HKV, D, PS, dtype = _cfg(1000 + seed)
    maxlen = r.choice([40, 300, 1500, 4000])
    lens = _lens(r, B, HKV, PS, maxlen)
    lens[-1, -1] = maxlen
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=seed)
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, HQ, D, generator=g).to(dtype)
    scale = 1.0 / math.sqrt(D)
    ks = r.choice([None, 1, 2, 5])
    out = head_sparse_decode_attention(q.to(dev), kc.to(dev), vc.to(dev), lens.to(dev), pt.to(dev), bm.to(dev), HKV,
                                       PS, scale, key_split=ks)
    torch.cuda.synchronize()
    ref = O.decode_attention(q, kc, vc, lens, pt, bm, HKV, PS, scale)
    d = (out.cpu().float() - ref.float()).abs().max()
    assert torch.allclose(out.cpu().float(), ref.float(), rtol=1e-6, atol=tol(dtype)), (seed, d)
    G = HQ // HKV
    for b in range(B):
        for h in range(HKV):
            if int(lens[b, h]) == 0:
                assert (out[b, h * G : (h + 1) * G].cpu().float() == 0).all()


@pytest.mark.parametrize("seed", range(16))
def test_fused_decode_random(dev, seed):
    """fused append + attention == oracle store + oracle attention; cache rows and the length table bit-exact."""
    from compactor_vllm_amd.attention.sparse_decode_kernel import fused_decode_step

    r, B, HQ, HKV, D, PS, dtype = _cfg(2000 + seed)
    maxlen = r.choice([40, 300, 1500])
    lens = _lens(r, B, HKV, PS, maxlen)
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens + 1, dtype, seed=seed, bmax_extra=2)
    if B > 1 and r.random() < 0.5:
        bm[r.randrange(B)] = 0  # RESERVED padding row
    table = torch.zeros(pt.shape[0], HKV, dtype=torch.int32)
    for b in range(B):
        if int(bm[b]) != 0:
            table[int(bm[b])] = lens[b]
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, HQ, D, generator=g).to(dtype)
    k1 = torch.randn(B, HKV, D, generator=g).to(dtype)
    v1 = torch.randn(B, HKV, D, generator=g).to(dtype)
    scale = 1.0 / math.sqrt(D)
    # oracle: store then attend on the gathered lengths
    kc_o, vc_o = kc.clone(), vc.clone()
    sl = table.index_select(0, bm.long()).contiguous()
    O.decode_store_kv(k1, v1, bm, sl, pt, kc_o, vc_o, PS)
    ref = O.decode_attention(q, kc_o, vc_o, sl, pt, bm, HKV, PS, scale)
    tab_o = table.clone()
    keep = bm != 0
    tab_o.index_copy_(0, bm[keep].long(), sl[keep])
    # device
    kc_d, vc_d, tab_d = kc.to(dev), vc.to(dev), table.to(dev)
    out = fused_decode_step(q.to(dev), k1.to(dev), v1.to(dev), kc_d, vc_d, tab_d, pt.to(dev), bm.to(dev), HKV, PS,
                            scale)
    torch.cuda.synchronize()
    assert torch.equal(tab_d.cpu(), tab_o), seed
    assert torch.equal(kc_d.cpu(), kc_o) and torch.equal(vc_d.cpu(), vc_o), seed
    d = (out.cpu().float()[keep] - ref.float()[keep]).abs().max()
    assert torch.allclose(out.cpu().float()[keep], ref.float()[keep], rtol=1e-6, atol=tol(dtype)), (seed, d)
    assert (out.cpu()[~keep] == 0).all()


@pytest.mark.parametrize("seed", range(16))
def test_prefill_random(dev, seed):
    from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache

    r, B, HQ, HKV, D, PS, dtype = _cfg(3000 + seed)
    cache_max = r.choice([0, 0, 70, 300, 900])
    lens = _lens(r, B, HKV, PS, cache_max) if cache_max else torch.zeros(B, HKV, dtype=torch.int32)
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=seed)
    append = [r.choice([1, 2, 31, 64, 65, 127, 200, 513]) for _ in range(B)]
    cu = torch.tensor([0] + torch.tensor(append).cumsum(0).tolist(), dtype=torch.int32)
    N = int(cu[-1])
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(N, HQ, D, generator=g).to(dtype)
    qkv = torch.randn(N, (HQ + 2 * HKV) * D, generator=g).to(dtype)  # k, v: strided views of a fused projection
    k = qkv[:, HQ * D : (HQ + HKV) * D].view(N, HKV, D)
    v = qkv[:, (HQ + HKV) * D :].view(N, HKV, D)
    scale = 1.0 / math.sqrt(D)
    qkv_d = qkv.to(dev)
    out = causal_sparse_varlen_with_cache(
        q.to(dev), qkv_d[:, HQ * D : (HQ + HKV) * D].view(N, HKV, D), qkv_d[:, (HQ + HKV) * D :].view(N, HKV, D),
        kc.to(dev), vc.to(dev), lens.to(dev), pt.to(dev), bm.to(dev), cu.to(dev), max(append), int(lens.max()), HKV, PS,
        scale)
    torch.cuda.synchronize()
    ref = O.prefill_attention(q, k, v, kc, vc, lens, pt, bm, cu, HKV, PS, scale)
    d = (out.cpu().float() - ref.float()).abs().max()
    assert torch.allclose(out.cpu().float(), ref.float(), rtol=1e-6, atol=tol(dtype)), (seed, d)


@pytest.mark.parametrize("seed", range(20))
def test_select_random(dev, seed):
    """Joint top-k + page padding: kept sets and new lengths bit-exact against the oracle's canonical rule (score desc,
    flat index asc) on adversarial score tensors: heavy ties, +-inf blocks, -0.0 vs +0.0, denormals, huge / tiny
    magnitudes, retain from 1 to everything, PAD on and off, page sizes 32..256, non-zero starting lengths."""
    from compactor_vllm_amd.compression.common import select_retained
    from helpers import kept_sets_from_lists

    r = random.Random(4000 + seed)
    B = r.choice([1, 2, 3, 4])
    H = r.choice([1, 2, 3, 4, 6, 8])
    PS = r.choice([32, 64, 128, 256])
    pad = r.random() < 0.7
    lens = [r.choice([1, 5, 63, 64, 65, 129, 300, 777, 2049]) for _ in range(B)]
    g = torch.Generator().manual_seed(seed)
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
    N = int(cu[-1])
    kind = r.choice(["normal", "quantised", "tiny", "huge", "zeros"])
    sc = torch.randn(N, H, generator=g)
    if kind == "quantised":
        sc = (sc * 2).round() / 2
    elif kind == "tiny":
        sc = sc * 1e-41  # denormals
    elif kind == "huge":
        sc = sc * 1e30
    elif kind == "zeros":
        sc = torch.where(torch.rand(N, H, generator=g) < 0.5, torch.zeros(()), -torch.zeros(()))  # +0.0 / -0.0 ties
    for b in range(B):
        s, L = int(cu[b]), lens[b]
        if r.random() < 0.6:
            sc[s : s + min(16, L)] = float("inf")
            sc[s + max(L - 64, 0) : s + L] = float("inf")
        if r.random() < 0.3 and L > 3:
            sc[s + r.randrange(L), r.randrange(H)] = float("-inf")
    retain = torch.tensor([r.choice([1, max(1, L * H // 3), max(1, L * H - 1), L * H]) for L in lens], dtype=torch.int32)
    lens0 = torch.randint(0, 3, (B, H), generator=g, dtype=torch.int32) * r.choice([0, 5, PS])
    bm = torch.arange(1, B + 1, dtype=torch.int32)
    if B >= 3 and r.random() < 0.5:
        bm[1] = 0  # RESERVED_BATCH: keeps nothing
    kept_o, lens_o = O.retained_sets(sc, cu, retain, lens0, bm, PS, pad)
    kept, new_lens = select_retained(sc.to(dev), cu.to(dev), max(lens), retain.to(dev), bm.to(dev), lens0.to(dev), PS,
                                     pad)
    torch.cuda.synchronize()
    assert torch.equal(new_lens.cpu(), lens_o), (seed, kind)
    sets = kept_sets_from_lists(kept.cpu(), new_lens.cpu(), lens0)
    ref = [sorted(kept_o[b][h]) for b in range(B) for h in range(H)]
    assert sets == ref, (seed, kind)


@pytest.mark.parametrize("seed", range(12))
def test_select_and_compaction_random(dev, seed):
    """extract_and_store_top_kv end to end: lengths, kept sets and the cache rows written (token-ordered, bit-exact,
    nothing outside the new rows touched) against the oracle, on shuffled pages with non-zero starting lengths."""
    from compactor_vllm_amd.compression.common import extract_and_store_top_kv
    from helpers import kept_sets_from_lists

    r = random.Random(5000 + seed)
    B = r.choice([1, 2, 3])
    H = r.choice([1, 2, 4, 8])
    D = r.choice([64, 128])
    PS = r.choice([32, 128, 256])
    dtype = r.choice([torch.float16, torch.bfloat16])
    lens = [r.choice([10, 70, 129, 300, 517, 1200]) for _ in range(B)]
    g = torch.Generator().manual_seed(seed)
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
    N = int(cu[-1])
    sc = torch.randn(N, H, generator=g)
    if r.random() < 0.5:
        sc = (sc * 4).round() / 4
    for b in range(B):
        s, L = int(cu[b]), lens[b]
        sc[s : s + min(16, L)] = float("inf")
        sc[s + max(L - 64, 0) : s + L] = float("inf")
    ratio = r.choice([0.1, 0.25, 0.5, 0.9])
    retain = torch.tensor([O.retain_count(ratio, L, 16, 64, H) for L in lens], dtype=torch.int32)
    lens0 = torch.randint(0, 2, (B, H), generator=g, dtype=torch.int32) * r.choice([0, 7, PS])
    total = lens0 + torch.tensor(lens, dtype=torch.int32)[:, None]
    kc, vc, pt, bm, P = mk_paged(B, H, D, PS, total, dtype, seed=seed)
    keys = torch.randn(N, H, D, generator=g).to(dtype)
    vals = torch.randn(N, H, D, generator=g).to(dtype)
    kept_o, lens_o = O.retained_sets(sc, cu, retain, lens0, bm, PS, True)
    kcd, vcd, ld = kc.to(dev), vc.to(dev), lens0.to(dev)
    kept, new_lens = extract_and_store_top_kv(sc.to(dev), cu.to(dev), max(lens), max(lens) * H, H, keys.to(dev),
                                              vals.to(dev), retain.to(dev), pt.to(dev), bm.to(dev), ld, kcd, vcd, PS)
    torch.cuda.synchronize()
    assert torch.equal(new_lens.cpu(), lens_o) and torch.equal(ld.cpu(), lens_o), seed
    sets = kept_sets_from_lists(kept.cpu(), new_lens.cpu(), lens0)
    assert sets == [sorted(kept_o[b][h]) for b in range(B) for h in range(H)], seed
    kcc, vcc = kcd.cpu(), vcd.cpu()
    touched = torch.zeros(kc.shape[0], dtype=torch.bool)
    i = 0
    for b in range(B):
        for h in range(H):
            L0, L1 = int(lens0[b, h]), int(lens_o[b, h])
            rows = O.cache_rows(pt[int(bm[b]), h], L1, PS)[L0:]
            src = [int(cu[b]) + t for t in sets[i]]
            i += 1
            assert torch.equal(kcc[rows], keys[src, h]) and torch.equal(vcc[rows], vals[src, h]), (seed, b, h)
            touched[rows] = True
    assert torch.equal(kcc[~touched], kc[~touched]) and torch.equal(vcc[~touched], vc[~touched]), seed


@pytest.mark.parametrize("seed", range(10))
def test_scoring_random(dev, seed):
    """Leverage (raw fp32, before the 16-bit rounding), chunk attention mass and SnapKV scores against the fp32 oracle on
    random batches: ragged lengths incl. sub-chunk / sub-window sequences, G 1..8, D 64 / 128."""
    from compactor_vllm_amd import _lib
    from compactor_vllm_amd.compression.compactor import _cu_from_lens, non_causal_attn_scores, split_into_chunks
    from compactor_vllm_amd.compression.snapkv import query_aware_key_scores

    r = random.Random(6000 + seed)
    B = r.choice([1, 2, 3])
    HKV = r.choice([1, 2, 4, 8])
    G = r.choice([1, 2, 4, 8])
    D = r.choice([64, 128])
    HQ = HKV * G
    lens = [r.choice([20, 33, 49, 127, 128, 129, 511, 513, 700, 1300]) for _ in range(B)]
    N = sum(lens)
    g = torch.Generator().manual_seed(seed)
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
    q = (torch.randn(N, HQ, D, generator=g) * 0.5).to(torch.bfloat16)
    k = (torch.randn(N, HKV, D, generator=g) * 0.5).to(torch.bfloat16)
    v = torch.zeros_like(k)
    # a7: chunk attention mass (sm_scale = 1.0 as the reference passes)
    mass = non_causal_attn_scores(q.to(dev), k.to(dev), v.to(dev), cu.to(dev), max(lens), chunk_size=128, sm_scale=1.0,
                                  normalize=False).cpu()
    ref = O.chunk_attn_mass(q, k, cu, 128, 1.0)
    assert torch.allclose(mass, ref, rtol=2e-4, atol=2e-4), (seed, (mass - ref).abs().max())
    # a8: SnapKV (w = 32 only when w * G <= 256 rows)
    if 32 * G <= 256:
        out = query_aware_key_scores(q.to(dev), k.to(dev), cu.to(dev), cu.to(dev), w=32, max_seqlen_k=max(lens)).cpu()
        refs = O.snapkv_scores(q, k, cu, cu, 32)
        fin = torch.isfinite(refs)
        assert torch.equal(torch.isfinite(out), fin), seed
        assert torch.allclose(out[fin], refs[fin], rtol=2e-4, atol=2e-5), (seed, (out[fin] - refs[fin]).abs().max())
    # a5: leverage scores, fp32 out of the kernel
    if D == 128:
        PHI = (torch.randn(D, 48, generator=g) / math.sqrt(48)).to(torch.bfloat16)
        _, chunks = split_into_chunks(lens, 512)
        cuc = _cu_from_lens(chunks, dev)
        kd, pd = k.to(dev), PHI.to(dev)
        scores = torch.empty(N, HKV, dtype=torch.float32, device=dev)
        L = _lib.lib()
        nb = L.cvllm_leverage_workspace_bytes(N, HKV, 48)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        st = L.cvllm_leverage_scores(kd.data_ptr(), kd.stride(0), kd.stride(1), pd.data_ptr(), scores.data_ptr(),
                                     cuc.data_ptr(), len(chunks), N, HKV, D, 48, 5e-3, 1, max(chunks), ws.data_ptr(), nb,
                                     torch.cuda.current_stream().cuda_stream)
        assert st == 0
        refl = O.leverage_scores(k, lens, PHI, normalize=False, chunk_size=512, out_dtype=torch.float32)
        assert torch.allclose(scores.cpu(), refl, rtol=1e-3, atol=2e-4), (seed, (scores.cpu() - refl).abs().max())


def test_many_short_sequences(dev):
    """A batch of 64 short sequences (the batch size of BASELINE.json configs[4]): decode over ragged short caches and
    a packed prefill of 48 prompts of 1..70 tokens, against the oracle."""
    from compactor_vllm_amd.attention.sparse_decode_kernel import head_sparse_decode_attention
    from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache

    dtype, HQ, HKV, D, PS = torch.bfloat16, 32, 8, 128, 128
    g = torch.Generator().manual_seed(64)
    B = 64
    lens = torch.randint(0, 600, (B, HKV), generator=g, dtype=torch.int32)
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=64)
    q = torch.randn(B, HQ, D, generator=g).to(dtype)
    scale = 1.0 / math.sqrt(D)
    out = head_sparse_decode_attention(q.to(dev), kc.to(dev), vc.to(dev), lens.to(dev), pt.to(dev), bm.to(dev), HKV,
                                       PS, scale)
    torch.cuda.synchronize()
    ref = O.decode_attention(q, kc, vc, lens, pt, bm, HKV, PS, scale)
    assert torch.allclose(out.cpu().float(), ref.float(), rtol=1e-6, atol=tol(dtype))

    B2 = 48
    append = torch.randint(1, 71, (B2,), generator=g).tolist()
    lens2 = torch.randint(0, 140, (B2, HKV), generator=g, dtype=torch.int32)
    kc2, vc2, pt2, bm2, _ = mk_paged(B2, HKV, D, PS, lens2, dtype, seed=65)
    cu = torch.tensor([0] + torch.tensor(append).cumsum(0).tolist(), dtype=torch.int32)
    N = int(cu[-1])
    q2 = torch.randn(N, HQ, D, generator=g).to(dtype)
    k2 = torch.randn(N, HKV, D, generator=g).to(dtype)
    v2 = torch.randn(N, HKV, D, generator=g).to(dtype)
    out2 = causal_sparse_varlen_with_cache(q2.to(dev), k2.to(dev), v2.to(dev), kc2.to(dev), vc2.to(dev), lens2.to(dev),
                                           pt2.to(dev), bm2.to(dev), cu.to(dev), max(append), int(lens2.max()), HKV, PS,
                                           scale)
    torch.cuda.synchronize()
    ref2 = O.prefill_attention(q2, k2, v2, kc2, vc2, lens2, pt2, bm2, cu, HKV, PS, scale)
    assert torch.allclose(out2.cpu().float(), ref2.float(), rtol=1e-6, atol=tol(dtype))


@pytest.mark.parametrize("kind", ["normal", "quantised", "zeros", "inf_heavy"])
@pytest.mark.parametrize("H", [8, 6])
def test_select_long_sequences_multi_workgroup_path(dev, kind, H):
    """L * H >= 32768 keys per sequence takes the multi-workgroup joint selection (slice histograms + scan per radix
    pass); mixed with a short sequence and a RESERVED row in the same batch.  Heavy ties exercise the partial-tie walk."""
    from compactor_vllm_amd.compression.common import select_retained
    from helpers import kept_sets_from_lists

    PS = 128
    lens = [5000, 300, 9000, 4100]
    B = len(lens)
    g = torch.Generator().manual_seed(77 + H)
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
    N = int(cu[-1])
    sc = torch.randn(N, H, generator=g)
    if kind == "quantised":
        sc = (sc * 2).round() / 2
    elif kind == "zeros":
        sc = torch.where(torch.rand(N, H, generator=g) < 0.5, torch.zeros(()), -torch.zeros(()))
    elif kind == "inf_heavy":
        sc[torch.rand(N, H, generator=g) < 0.4] = float("inf")
    for b in range(B):
        s, L = int(cu[b]), lens[b]
        sc[s : s + 16] = float("inf")
        sc[s + L - 64 : s + L] = float("inf")
    retain = torch.tensor([O.retain_count(0.5, L, 16, 64, H) for L in lens], dtype=torch.int32)
    retain[3] = lens[3] * H - 5  # nearly everything
    lens0 = torch.zeros(B, H, dtype=torch.int32)
    bm = torch.tensor([1, 2, 0, 4], dtype=torch.int32)  # the 9000-token sequence sits on the RESERVED row
    kept_o, lens_o = O.retained_sets(sc, cu, retain, lens0, bm, PS, True)
    kept, new_lens = select_retained(sc.to(dev), cu.to(dev), max(lens), retain.to(dev), bm.to(dev), lens0.to(dev), PS,
                                     True)
    torch.cuda.synchronize()
    assert torch.equal(new_lens.cpu(), lens_o), kind
    sets = kept_sets_from_lists(kept.cpu(), new_lens.cpu(), lens0)
    assert sets == [sorted(kept_o[b][h]) for b in range(B) for h in range(H)], kind
