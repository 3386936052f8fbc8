"""GPU parity (through the C ABI): scoring kernels a5-a8, full ranking a9, and the Attention boundary a11."""
import math

import pytest
import torch

from golden_io import list_cases, load_case
from helpers import kept_sets_from_lists, mk_paged, tol
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------ a6
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_zscore_segments(dev, dtype):
    from compactor_vllm_amd.compression.compactor import zscore_segments_

    g = torch.Generator().manual_seed(0)
    lens = [512, 512, 37, 1, 2000, 128]
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + n)
    x = (torch.randn(cu[-1], 8, generator=g) * 3 + 1).to(dtype)
    ref = O.zscore_segments(x, cu)
    out = zscore_segments_(x.to(dev).clone(), torch.tensor(cu, dtype=torch.int32, device=dev)).cpu()
    ok = torch.isfinite(ref.float())  # the 1-row segment has var > 0 here; keep the mask for safety
    atol = 2e-5 if dtype == torch.float32 else (2e-2 if dtype == torch.bfloat16 else 3e-3)
    assert torch.allclose(out.float()[ok], ref.float()[ok], rtol=0, atol=atol)


# ------------------------------------------------------------------------------------------ a5
@pytest.mark.parametrize("name", [n for n in list_cases("leverage_") if "f32" not in n])
def test_leverage_golden(dev, name):
    """vs the reference's own output (16-bit Gram + SVD noise => SURVEY P3 tolerances) and, tighter,
    vs the fp32 oracle of the same closed form."""
    from compactor_vllm_amd.compression.compactor import approximate_leverage_scores

    c = load_case(name)
    lens = c["context_lens"].tolist()
    k, PHI = c["k"], c["PHI"]
    out = approximate_leverage_scores(k.to(dev), lens, PHI.to(dev), normalize=bool(c["normalize"]),
                                      chunk_size=c["chunk_size"]).cpu()
    assert out.dtype == k.dtype and out.shape == c["out"].shape
    chunks = O.split_into_chunks(lens, c["chunk_size"]) if c["chunk_size"] > 0 else lens
    # chunks of >= 96 rows: SURVEY P3 bar vs the reference (0.1 z-scored, 0.01 raw; measured need 0.078 / 0.0032);
    # shorter tails are "parity unpinned" upstream (tests/test_oracle_golden.py::test_leverage_short_tails_are_unpinned)
    mask = torch.zeros(out.shape[0], dtype=torch.bool)
    s = 0
    for L in chunks:
        if L >= 96:
            mask[s : s + L] = True
        s += L
    atol_ref = 0.1 if c["normalize"] else 0.01
    d = (out.float()[mask] - c["out"].float()[mask]).abs().max()
    assert torch.allclose(out.float()[mask], c["out"].float()[mask], rtol=0, atol=atol_ref), float(d)
    # against the fp32 oracle of the same closed form EVERY chunk counts, tails included: a few output ulps
    orc = O.leverage_scores(k, lens, PHI, normalize=bool(c["normalize"]), chunk_size=c["chunk_size"])
    ulp = 2.0 ** -7 if k.dtype == torch.bfloat16 else 2.0 ** -10
    atol_o = (6 * ulp * 5) if c["normalize"] else 4 * ulp  # a few output ulps (|z| <= ~5, scores <= ~1)
    if c["normalize"]:  # a tail's z-scores amplify one-ulp differences of the rounded raw scores by 1/std
        assert torch.allclose(out.float()[mask], orc.float()[mask], rtol=0, atol=atol_o), \
            (out.float()[mask] - orc.float()[mask]).abs().max()
        assert torch.isfinite(out.float()).all()
    else:
        assert torch.allclose(out.float(), orc.float(), rtol=0, atol=atol_o), (out.float() - orc.float()).abs().max()


def test_leverage_unnormalised_fp32_accuracy(dev):
    """The kernel's fp32 pipeline vs the fp32 oracle on 16-bit inputs, before the final rounding."""
    from compactor_vllm_amd import _lib
    from compactor_vllm_amd.compression.compactor import _cu_from_lens, split_into_chunks

    g = torch.Generator().manual_seed(4)
    lens, H, D = [700, 130, 49], 4, 128
    N = sum(lens)
    k = torch.randn(N, H, D, generator=g).to(torch.bfloat16)
    PHI = (torch.randn(D, 48, generator=g) / math.sqrt(48)).to(torch.bfloat16)
    _, chunks = split_into_chunks(lens, 512)
    cu = _cu_from_lens(chunks, dev)
    kd, pd = k.to(dev), PHI.to(dev)
    scores = torch.empty(N, H, dtype=torch.float32, device=dev)
    L = _lib.lib()
    nb = L.cvllm_leverage_workspace_bytes(N, H, 48)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    import os

    both = []
    # <= 512 rows per chunk: ONE kernel, sketch in registers + fp32 MFMA Gram / solve (default) or sketch resident in LDS
    # (CVLLM_LEVERAGE=resident); 0: sketch kernel + solve kernel through the workspace
    for longest, env in ((max(chunks), None), (max(chunks), "resident"), (0, None)):
        scores.fill_(float("nan"))
        if env:
            os.environ["CVLLM_LEVERAGE"] = env
        try:
            st = L.cvllm_leverage_scores(kd.data_ptr(), kd.stride(0), kd.stride(1), pd.data_ptr(), scores.data_ptr(),
                                         cu.data_ptr(), len(chunks), N, H, D, 48, 5e-3, 1, longest, ws.data_ptr(), nb,
                                         torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("CVLLM_LEVERAGE", None)
        assert st == 0
        both.append(scores.cpu().clone())
    # the same fp32 arithmetic in three summation orders (and W = L^-1 applied instead of a forward substitution)
    assert torch.allclose(both[0], both[2], rtol=1e-4, atol=1e-6), (both[0] - both[2]).abs().max()
    assert torch.allclose(both[1], both[2], rtol=1e-4, atol=1e-6), (both[1] - both[2]).abs().max()
    ref = O.leverage_scores(k, lens, PHI, normalize=False, chunk_size=512, out_dtype=torch.float32)
    assert torch.allclose(scores.cpu(), ref, rtol=1e-3, atol=2e-4), (scores.cpu() - ref).abs().max()


# ------------------------------------------------------------------------------------------ a7
@pytest.mark.parametrize("name", list_cases("chunkattn_"))
def test_chunk_attn_golden(dev, name):
    from compactor_vllm_amd.compression.compactor import non_causal_attn_scores

    c = load_case(name)
    lens = c["context_lens"].tolist()
    B = len(lens)
    q, k, pre, cu = c["q"].to(dev), c["k"].to(dev), c["pre"].to(dev), c["cu_seqlens"].to(dev)
    v = torch.zeros_like(k)
    mass = non_causal_attn_scores(q, k, v, cu, max(lens), chunk_size=128, sm_scale=1.0, normalize=False).cpu()
    assert torch.allclose(mass, c["mass"], rtol=2e-4, atol=2e-4), (mass - c["mass"]).abs().max()
    out = non_causal_attn_scores(q, k, v, cu, max(lens), chunk_size=128, sm_scale=1.0, normalize=True,
                                 accum_scores=pre, context_lens=lens, protected_first_tokens=[c["first"]] * B,
                                 protected_last_tokens=[c["last"]] * B, accum_blending=0.5).cpu()
    fin = torch.isfinite(c["out"])
    assert torch.equal(fin, torch.isfinite(out))
    assert torch.allclose(out[fin], c["out"][fin], rtol=2e-4, atol=5e-4), (out[fin] - c["out"][fin]).abs().max()


def test_chunk_attn_protected_slices_quirk_q9(dev):
    """L < protected_last makes the reference's python slice start negative / spill: reproduced."""
    from compactor_vllm_amd.compression.compactor import non_causal_attn_scores

    g = torch.Generator().manual_seed(2)
    lens, HQ, HKV, D = [40, 200, 20], 8, 2, 128
    N = sum(lens)
    cu = torch.tensor([0, 40, 240, 260], dtype=torch.int32)
    q = (torch.randn(N, HQ, D, generator=g) * 0.3).to(torch.float16)
    k = (torch.randn(N, HKV, D, generator=g) * 0.3).to(torch.float16)
    first, last = [16] * 3, [64] * 3
    out = non_causal_attn_scores(q.to(dev), k.to(dev), k.to(dev), cu.to(dev), max(lens), chunk_size=128, sm_scale=1.0,
                                 normalize=True, context_lens=lens, protected_first_tokens=first,
                                 protected_last_tokens=last).cpu()
    ref = O.compactor_post_scores(q, k, cu, lens, None, first, last)
    assert torch.equal(torch.isfinite(out), torch.isfinite(ref))
    fin = torch.isfinite(ref)
    assert torch.allclose(out[fin], ref[fin], rtol=2e-4, atol=5e-4)


# ------------------------------------------------------------------------------------------ a8
@pytest.mark.parametrize("name", list_cases("snapkv_"))
def test_snapkv_golden(dev, name):
    from compactor_vllm_amd.compression.snapkv import query_aware_key_scores

    c = load_case(name)
    cu = c["cu_seqlens"].to(dev)
    out = query_aware_key_scores(c["q"].to(dev), c["k"].to(dev), cu, cu, w=c["w"]).cpu()
    ref = c["out"]
    s = 0
    for L in c["cu_seqlens"].diff().tolist():
        a, r = out[s : s + L], ref[s : s + L]
        if L > c["w"]:
            fin = torch.isfinite(r)
            assert torch.equal(fin, torch.isfinite(a))
            assert torch.allclose(a[fin], r[fin], rtol=2e-4, atol=2e-5), (a[fin] - r[fin]).abs().max()
        else:
            assert torch.isinf(a).all()  # documented: all-protected (reference leaves these uninitialised)
        s += L


@pytest.mark.parametrize("name", list_cases("snapkvx_"))
def test_snapkv_golden_windows_and_normalize(dev, name):
    """`w` as a [B] tensor and normalize=True (the two arguments the reference's engine never uses) against vectors
    of the reference, and against the same call with the windows passed one sequence at a time."""
    from compactor_vllm_amd.compression.snapkv import query_aware_key_scores

    c = load_case(name)
    cu = c["cu_seqlens"].to(dev)
    wl = c["w"].tolist()
    norm = bool(c["normalize"])
    w_arg = wl[0] if len(set(wl)) == 1 else c["w"].to(dev)
    out = query_aware_key_scores(c["q"].to(dev), c["k"].to(dev), cu, cu, w=w_arg, normalize=norm).cpu()
    ref = c["out"]
    s = 0
    for L, wb in zip(c["cu_seqlens"].diff().tolist(), wl):
        a, r = out[s : s + L], ref[s : s + L]
        if L > wb:
            fin = torch.isfinite(r)
            assert torch.equal(fin, torch.isfinite(a))
            assert torch.allclose(a[fin], r[fin], rtol=2e-4, atol=2e-4), (a[fin] - r[fin]).abs().max()
            assert torch.isinf(a[L - wb :]).all() and (a[L - wb :] > 0).all()
        else:
            assert torch.isinf(a).all()
        s += L


def test_snapkv_oracle_long(dev):
    from compactor_vllm_amd.compression.snapkv import query_aware_key_scores

    g = torch.Generator().manual_seed(8)
    lens, HQ, HKV, D = [1500, 33, 20, 640], 32, 8, 128
    N = sum(lens)
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
    q = torch.randn(N, HQ, D, generator=g).to(torch.bfloat16)
    k = torch.randn(N, HKV, D, generator=g).to(torch.bfloat16)
    out = query_aware_key_scores(q.to(dev), k.to(dev), cu.to(dev), cu.to(dev), w=32, max_seqlen_k=max(lens)).cpu()
    ref = O.snapkv_scores(q, k, cu, cu, 32)
    assert torch.equal(torch.isfinite(out), torch.isfinite(ref))
    fin = torch.isfinite(ref)
    assert torch.allclose(out[fin], ref[fin], rtol=2e-4, atol=2e-5), (out[fin] - ref[fin]).abs().max()


# ------------------------------------------------------------------------------------------ a9
@pytest.mark.parametrize("name", list_cases("select_"))
def test_rank_indices_golden(dev, name):
    """scores_to_retain_indices: same shape/dtype as the reference; identical wherever scores are tie-free
    (tie ORDER is implementation-defined upstream), canonical (index-ascending) inside ties."""
    from compactor_vllm_amd.compression.common import scores_to_retain_indices

    c = load_case(name)
    H = c["HKV"]
    maxL = int(c["cu_seqlens_k"].diff().max())
    out = scores_to_retain_indices(c["scores"].to(dev), c["cu_seqlens_k"].to(dev), maxL, maxL * H, H).cpu()
    ref = c["ref_indices"]
    assert out.shape == ref.shape and out.dtype == torch.int64
    mine = O.rank_indices(c["scores"], c["cu_seqlens_k"], maxL, maxL * H, H)
    assert torch.equal(out, mine)
    sc = c["scores"].reshape(-1)
    for b in range(out.shape[0]):
        nb = int(c["cu_seqlens_k"][b + 1] - c["cu_seqlens_k"][b]) * H
        a, r = out[b, :nb], ref[b, :nb]
        assert torch.equal(sc[a], sc[r])
        fin = torch.isfinite(sc[a])
        assert torch.equal(a[fin], r[fin])


# ------------------------------------------------------------------------------------------ a11
def _run_layer(dev, method, ratio, lens, dtype=torch.bfloat16, HQ=8, HKV=2, D=128, PS=128, streams=True):
    """One attention layer end to end through the reference's call order (models/llama3.py:101-110):
    pre-RoPE scoring -> (no RoPE: identity) -> post-RoPE scoring -> Attention.forward (store on the second
    stream, attention on the main stream) -> one decode step."""
    from compactor_vllm_amd.compression import (CompressionMethod, apply_postrope_compression,
                                                apply_prerope_compression)
    from compactor_vllm_amd.kv_cache.page_table import PagedKVCache
    from compactor_vllm_amd.layers.attention import Attention
    from compactor_vllm_amd.utils.context import CompressionContext, get_context, set_context

    g = torch.Generator().manual_seed(11)
    B = len(lens)
    N = sum(lens)
    first, last = 4, 8
    cache = PagedKVCache(num_layers=1, max_logical_pages_per_head=-(-(max(lens) + 8) // PS), num_pages=64,
                         page_size=PS, H_kv=HKV, head_dim=D, max_num_batches=B + 2, dtype=dtype, device=dev)
    rows = []
    for L in lens:
        bi = cache.new_batch()
        assert cache.reserve_tokens(bi, L + 4).name == "SUCCESS"
        rows.append(bi)
    attn = Attention(HQ, D, 1.0 / math.sqrt(D), HKV)
    attn.k_cache, attn.v_cache, attn.page_table, attn.bh_seq_lens = cache.layer_slices(0)
    attn.page_size = PS
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32, device=dev)
    bm = torch.tensor(rows, dtype=torch.int32, device=dev)
    q = (torch.randn(N, HQ, D, generator=g) * 0.5).to(dtype).to(dev)
    qkv = (torch.randn(N, (HQ + 2 * HKV) * D, generator=g) * 0.5).to(dtype).to(dev)
    k = qkv[:, HQ * D : (HQ + HKV) * D].view(N, HKV, D)
    v = qkv[:, (HQ + HKV) * D :].view(N, HKV, D)
    retain = torch.tensor([O.retain_count(ratio, L, first, last, HKV) for L in lens], dtype=torch.int32, device=dev)
    PHI = (torch.randn(D, 48, generator=g) / math.sqrt(48)).to(dtype).to(dev)
    cc = CompressionContext(compression_method=method, compression_chunk_size=512, batch_tokens_to_retain=retain,
                            max_tokens_to_retain=max(lens) * HKV, context_lens=list(lens), PHI=PHI,
                            protected_first_tokens=[first] * B, protected_last_tokens=[last] * B)
    store = torch.cuda.Stream() if streams else None
    set_context(is_prefill=True, do_compression=method != CompressionMethod.NONE, cu_seqlens_q=cu, cu_seqlens_k=cu,
                max_seqlen_q=max(lens), max_seqlen_k=max(lens), batch_mapping=bm, max_bh_len=0,
                compression_context=cc, STORE_STREAM=store)
    ctx = get_context()
    scores = None
    if ctx.do_compression:
        scores = apply_prerope_compression(q, k, v, ctx)
        scores = apply_postrope_compression(q, k, v, scores, ctx)
    o = attn(q, k, v, scores)
    if store is not None:
        torch.cuda.current_stream().wait_stream(store)
    torch.cuda.synchronize()
    return dict(o=o, q=q, k=k, v=v, cu=cu, bm=bm, attn=attn, cache=cache, scores=scores, retain=retain, rows=rows,
                lens=lens, HQ=HQ, HKV=HKV, D=D, PS=PS, dtype=dtype)


@pytest.mark.parametrize("method_name,ratio", [("NONE", 1.0), ("COMPACTOR", 0.5), ("SNAPKV", 0.25)])
def test_attention_boundary_end_to_end(dev, method_name, ratio):
    from compactor_vllm_amd.compression import CompressionMethod
    from compactor_vllm_amd.utils.context import set_context

    method = CompressionMethod[method_name]
    lens = [300, 45, 700]
    r = _run_layer(dev, method, ratio, lens)
    HQ, HKV, D, PS, dtype = r["HQ"], r["HKV"], r["D"], r["PS"], r["dtype"]
    qc, kc, vc = r["q"].cpu(), r["k"].cpu(), r["v"].cpu()
    cu = r["cu"].cpu()
    # (1) prefill output = dense causal attention (the cache was empty)
    lens0 = torch.zeros(len(lens), HKV, dtype=torch.int32)
    dummy = torch.zeros(PS, D, dtype=dtype)
    ref_o = O.prefill_attention(qc, kc, vc, dummy, dummy, lens0, torch.zeros(8, HKV, 1, dtype=torch.int32),
                                r["bm"].cpu(), cu, HKV, PS, 1.0 / math.sqrt(D))
    assert torch.allclose(r["o"].cpu().float(), ref_o.float(), atol=tol(dtype))
    # (2) the cache holds exactly the oracle's retained set given the SAME score tensor (P3), exact lengths
    bh = r["attn"].bh_seq_lens.index_select(0, r["bm"].long()).cpu()
    pt = r["attn"].page_table.cpu()
    if method == CompressionMethod.NONE:
        assert torch.equal(bh, torch.tensor(lens, dtype=torch.int32)[:, None].repeat(1, HKV))
        kept = [[list(range(L)) for _ in range(HKV)] for L in lens]
    else:
        kept, lens_o = O.retained_sets(r["scores"].cpu(), cu, r["retain"].cpu(), lens0, r["bm"].cpu(), PS, True)
        assert torch.equal(bh, lens_o)
        assert (bh % PS == 0).logical_or(bh == torch.tensor(lens)[:, None]).all()  # padded to a page or everything
    kcache, vcache = r["attn"].k_cache.cpu(), r["attn"].v_cache.cpu()
    for b, L in enumerate(lens):
        for h in range(HKV):
            rows = O.cache_rows(pt[r["rows"][b], h], int(bh[b, h]), PS)
            src = [int(cu[b]) + t for t in sorted(kept[b][h])]
            assert torch.equal(kcache[rows], kc[src, h]) and torch.equal(vcache[rows], vc[src, h])
    # (3) one decode step: store-then-attend over {retained rows} U {new token}
    g = torch.Generator().manual_seed(5)
    B = len(lens)
    q1 = torch.randn(B, HQ, D, generator=g).to(dtype)
    k1 = torch.randn(B, HKV, D, generator=g).to(dtype)
    v1 = torch.randn(B, HKV, D, generator=g).to(dtype)
    set_context(is_prefill=False, batch_mapping=r["bm"])
    o1 = r["attn"](q1.to(dev), k1.to(dev), v1.to(dev)).cpu()
    torch.cuda.synchronize()
    G = HQ // HKV
    for b in range(B):
        for h in range(HKV):
            src = [int(cu[b]) + t for t in sorted(kept[b][h])]
            K = torch.cat([kc[src, h], k1[b, h][None]]).float()
            V = torch.cat([vc[src, h], v1[b, h][None]]).float()
            p = torch.softmax(q1[b, h * G : (h + 1) * G].float() @ K.T / math.sqrt(D), -1)
            assert torch.allclose(o1[b, h * G : (h + 1) * G].float(), p @ V, atol=tol(dtype))
    bh2 = r["attn"].bh_seq_lens.index_select(0, r["bm"].long()).cpu()
    assert torch.equal(bh2, bh + 1)
    # (4) reclaim: pages beyond ceil((len + future)/PS) return to the allocator
    freed = r["cache"].reclaim_pages(r["rows"][2], future_reserve_tokens=4)
    if method != CompressionMethod.NONE:
        assert freed > 0


def test_stream_overlap_matches_serial(dev):
    """Store stream + main stream == everything on one stream (debug mode of SURVEY §5)."""
    from compactor_vllm_amd.compression import CompressionMethod

    a = _run_layer(dev, CompressionMethod.COMPACTOR, 0.5, [513, 129], streams=True)
    b = _run_layer(dev, CompressionMethod.COMPACTOR, 0.5, [513, 129], streams=False)
    assert torch.equal(a["o"], b["o"])
    assert torch.equal(a["scores"], b["scores"])
    assert torch.equal(a["attn"].bh_seq_lens, b["attn"].bh_seq_lens)
    la = a["attn"].bh_seq_lens.index_select(0, a["bm"].long()).cpu()
    for bi in range(2):
        for h in range(a["HKV"]):
            ra = O.cache_rows(a["attn"].page_table[a["rows"][bi], h].cpu(), int(la[bi, h]), a["PS"])
            rb = O.cache_rows(b["attn"].page_table[b["rows"][bi], h].cpu(), int(la[bi, h]), b["PS"])
            assert torch.equal(a["attn"].k_cache.cpu()[ra], b["attn"].k_cache.cpu()[rb])
