"""GPU: chunked prefill with compression on the last chunk.

A prompt fed through ONE attention layer in chunks - each chunk attends to [its cached prefix || itself] and is written to
the cache uncompressed, the scoring state is carried across chunks, the cache is compacted in place after the last chunk -
must end in exactly the state of the one-shot prefill of the same q / k / v: the same score tensor (bit for bit), the
same retained set per (sequence, head) as the CPU ORACLE's one-shot selection, the same cache rows and lengths; the
attention outputs agree within the attention tolerance (same sums in a different tile order)."""
import math

import pytest
import torch

from helpers import tol
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


def _layer(dev, HQ, HKV, D, PS, total, dtype):
    from compactor_vllm_amd.kv_cache.page_table import PagedKVCache
    from compactor_vllm_amd.layers.attention import Attention

    cache = PagedKVCache(num_layers=1, max_logical_pages_per_head=-(-(total + 8) // PS), num_pages=HKV * (-(-(total + 8) // PS)) * 3,
                         page_size=PS, H_kv=HKV, head_dim=D, max_num_batches=3, dtype=dtype, device=dev)
    cache.new_batch()  # burn row 1 so that batch_mapping != arange
    row = cache.new_batch()
    assert cache.reserve_tokens(row, total + 4).name == "SUCCESS"
    attn = Attention(HQ, D, 1.0 / math.sqrt(D), HKV)
    attn.k_cache, attn.v_cache, attn.page_table, attn.bh_seq_lens = cache.layer_slices(0)
    attn.page_size = PS
    return cache, attn, row


def _run(dev, method, q_pre, k_pre, q, k, v, cuts, retain, PHI, first, last, HQ, HKV, D, PS, dtype):
    """Feed the token ranges [cuts[i], cuts[i+1]) through the layer (one range = one-shot).  Returns outputs, scores of
    the last call, the layer."""
    from compactor_vllm_amd.compression import apply_postrope_compression, apply_prerope_compression
    from compactor_vllm_amd.utils.chunked import ChunkedPrefillState, PrefillChunk
    from compactor_vllm_amd.utils.context import CompressionContext, get_context, set_context

    total = q.shape[0]
    cache, attn, row = _layer(dev, HQ, HKV, D, PS, total, dtype)
    bm = torch.tensor([row], dtype=torch.int32, device=dev)
    store = torch.cuda.Stream()
    chunked = len(cuts) > 2
    state = ChunkedPrefillState(total, HKV, [attn]) if chunked else None
    outs, scores = [], None
    for ci in range(len(cuts) - 1):
        s0, s1 = cuts[ci], cuts[ci + 1]
        n = s1 - s0
        cu = torch.tensor([0, n], dtype=torch.int32, device=dev)
        cc = CompressionContext(compression_method=method, compression_chunk_size=512 if method.name == "COMPACTOR" else -1,
                                batch_tokens_to_retain=retain, max_tokens_to_retain=total * HKV, context_lens=[n], PHI=PHI,
                                protected_first_tokens=[first], protected_last_tokens=[last])
        if chunked:
            state.begin_chunk()
        set_context(is_prefill=True, do_compression=True, cu_seqlens_q=cu, cu_seqlens_k=cu, max_seqlen_q=n, max_seqlen_k=n,
                    batch_mapping=bm, max_bh_len=s0, compression_context=cc, STORE_STREAM=store,
                    chunk=PrefillChunk(s0, n, ci == len(cuts) - 2, state) if chunked else None)
        ctx = get_context()
        sc = apply_prerope_compression(q_pre[s0:s1], k_pre[s0:s1], v[s0:s1], ctx)
        sc = apply_postrope_compression(q[s0:s1], k[s0:s1], v[s0:s1], sc, ctx)
        outs.append(attn(q[s0:s1], k[s0:s1], v[s0:s1], sc))
        torch.cuda.current_stream().wait_stream(store)
        torch.cuda.synchronize()
        scores = sc
    return torch.cat(outs), scores, attn, row, cache


@pytest.mark.parametrize("method_name,ratio", [("COMPACTOR", 0.5), ("SNAPKV", 0.25)])
@pytest.mark.parametrize("total,chunk", [(2700, 1024), (1536, 512), (4100, 2048)])
def test_chunked_prefill_equals_one_shot_and_oracle(dev, method_name, ratio, total, chunk):
    from compactor_vllm_amd.compression import CompressionMethod
    from compactor_vllm_amd.utils.chunked import chunk_boundaries

    method = CompressionMethod[method_name]
    dtype, HQ, HKV, D, PS, first, last = torch.bfloat16, 8, 2, 128, 128, 4, 8
    g = torch.Generator().manual_seed(total + chunk)
    mk = lambda h, s=0.5: (torch.randn(total, h, D, generator=g) * s).to(dtype).to(dev)  # noqa: E731
    q_pre, k_pre, q, k, v = mk(HQ), mk(HKV), mk(HQ), mk(HKV), mk(HKV, 1.0)
    PHI = (torch.randn(D, 48, generator=g) / math.sqrt(48)).to(dtype).to(dev)
    retain = torch.tensor([O.retain_count(ratio, total, first, last, HKV)], dtype=torch.int32, device=dev)
    cuts = chunk_boundaries(total, chunk)
    assert len(cuts) > 2 and all(c % 512 == 0 for c in cuts[:-1])
    args = (q_pre, k_pre, q, k, v)
    o1, s1, a1, r1, _ = _run(dev, method, *args, [0, total], retain, PHI, first, last, HQ, HKV, D, PS, dtype)
    oc, sc, ac, rc, _ = _run(dev, method, *args, cuts, retain, PHI, first, last, HQ, HKV, D, PS, dtype)
    # (1) the score tensor the selection sees: identical bits
    assert sc.shape == s1.shape == (total, HKV) and torch.equal(sc, s1)
    # (2) lengths and cache rows: identical to the one-shot prefill ...
    l1, lc = a1.bh_seq_lens[r1].cpu(), ac.bh_seq_lens[rc].cpu()
    assert torch.equal(l1, lc)
    # ... and to the CPU oracle's one-shot selection + compaction from the same scores
    cu = torch.tensor([0, total], dtype=torch.int32)
    kept, lens_o = O.retained_sets(s1.cpu(), cu, retain.cpu(), torch.zeros(1, HKV, dtype=torch.int32),
                                   torch.tensor([rc], dtype=torch.int32), PS, True)
    assert torch.equal(lc[None], lens_o)
    for attn, row in ((a1, r1), (ac, rc)):
        kc, vc, pt = attn.k_cache.cpu(), attn.v_cache.cpu(), attn.page_table.cpu()
        for h in range(HKV):
            rows = O.cache_rows(pt[row, h], int(lens_o[0, h]), PS)
            src = sorted(kept[0][h])
            assert torch.equal(kc[rows], k.cpu()[src, h]) and torch.equal(vc[rows], v.cpu()[src, h]), h
    # (3) attention outputs: the oracle's one-shot dense causal attention, within the attention tolerance
    dummy = torch.zeros(PS, D, dtype=dtype)
    ref = O.prefill_attention(q.cpu(), k.cpu(), v.cpu(), dummy, dummy, torch.zeros(1, HKV, dtype=torch.int32),
                              torch.zeros(2, HKV, 1, dtype=torch.int32), torch.ones(1, dtype=torch.int32), cu, HKV, PS,
                              1.0 / math.sqrt(D)).float()
    assert torch.allclose(o1.cpu().float(), ref, atol=tol(dtype)) and torch.allclose(oc.cpu().float(), ref, atol=tol(dtype))


def test_compact_cache_inplace_equals_compaction_from_packed(dev):
    """The in-place kernel on a cache holding every row, against cvllm_compact_store fed with the packed tensors: two
    sequences, 20 000 and 777 tokens, 8 heads, random scores (long lists: many 128-row tiles per workgroup, kept rows
    that stay put at the start, sources inside later destination tiles)."""
    from compactor_vllm_amd.compression.common import compact_cache_inplace, extract_and_store_top_kv, select_retained
    from compactor_vllm_amd.kv_cache.store_kv_cache import prefill_store_all_kv

    dtype, HKV, D, PS = torch.bfloat16, 8, 128, 128
    lens = [20000, 777]
    B, N = len(lens), sum(lens)
    g = torch.Generator(device=dev).manual_seed(2)
    k = torch.randn(N, HKV, D, device=dev, generator=g).to(dtype)
    v = torch.randn(N, HKV, D, device=dev, generator=g).to(dtype)
    sc = torch.randn(N, HKV, device=dev, generator=g)
    sc[:300] = float("inf")  # a long prefix that does not move
    cu = torch.tensor([0, lens[0], N], dtype=torch.int32, device=dev)
    P = -(-max(lens) // PS)
    n_pages = (B + 1) * HKV * P
    pt = torch.randperm(n_pages, device=dev).view(B + 1, HKV, P).to(torch.int32)
    bm = torch.tensor([2, 1], dtype=torch.int32, device=dev)
    retain = torch.tensor([int(0.4 * lens[0] * HKV), int(0.7 * lens[1] * HKV)], dtype=torch.int32, device=dev)
    zero = torch.zeros(B, HKV, dtype=torch.int32, device=dev)
    # reference path: compaction from the packed tensors into an empty cache
    kc1, vc1 = torch.zeros(n_pages * PS, D, dtype=dtype, device=dev), torch.zeros(n_pages * PS, D, dtype=dtype, device=dev)
    l1 = zero.clone()
    kept1, new1 = extract_and_store_top_kv(sc, cu, max(lens), max(lens) * HKV, HKV, k, v, retain, pt, bm, l1, kc1, vc1, PS)
    # in-place path: everything stored first, then compacted inside the cache
    kc2, vc2 = torch.zeros_like(kc1), torch.zeros_like(vc1)
    l2 = zero.clone()
    prefill_store_all_kv(new_keys=k, new_values=v, cu_seqlens_k=cu, max_seqlen_k=max(lens), k_cache=kc2, v_cache=vc2,
                         page_table=pt, bh_lens=l2, batch_mapping=bm, PAGE_SIZE=PS)
    kept2, new2 = select_retained(sc, cu, max(lens), retain, bm, zero, PS, True)
    compact_cache_inplace(kept2, new2, zero, zero, pt, bm, kc2, vc2, PS)
    torch.cuda.synchronize()
    assert torch.equal(new1, new2)
    i = torch.arange(max(lens), device=dev)
    for b in range(B):
        for h in range(HKV):
            n = int(new1[b, h])
            rows = (pt[int(bm[b]), h].long()[i // PS] * PS + i % PS)[:n]
            assert torch.equal(kc1[rows], kc2[rows]) and torch.equal(vc1[rows], vc2[rows]), (b, h)


@pytest.mark.parametrize("method_name", ["COMPACTOR", "SNAPKV"])
def test_engine_chunked_prefill_long_prompt(dev, method_name):
    """Through `LLM.generate`: a 3 000-token prompt with a 1 024-token prefill budget (three chunks) next to short
    prompts; page accounting returns to zero, per-layer retained counts obey the budget, and a repeat of the same call
    returns the same tokens (the chunked path is deterministic)."""
    from compactor_vllm_amd import (LLM, BatchCompressionParams, CompressionMethod, LLMConfig, SamplingParams,
                                    SequenceCompressionParams)
    from tiny_model import TinyConfig, TinyModel

    cfg = TinyConfig()
    conf = LLMConfig(model="tiny", max_num_seqs=4, max_model_len=4096, hf_config=cfg, eos=-1, kvcache_page_size=128,
                     enforce_eager=True, show_progress_bar=False)
    llm = LLM(conf, TinyModel(cfg, dev), device=dev, num_pages=200, max_batched_tokens=1024)
    runner = llm.master_model_runner
    seen = []
    orig = runner.run_decode_loop

    def spy(batch, pending=None):
        seen.append(runner.kv_manager.paged_cache.bh_seq_lens[:, batch.batch_mapping.long()].clone())
        return orig(batch, pending)

    runner.run_decode_loop = spy
    g = torch.Generator().manual_seed(4)
    prompts = [torch.randint(0, 512, (n,), generator=g).tolist() for n in (3000, 200, 640)]
    sp = SamplingParams(temperature=0.0, max_new_tokens=6)
    bcp = BatchCompressionParams(CompressionMethod[method_name])
    mk = lambda: [SequenceCompressionParams(0.5, 4, 16) for _ in prompts]  # noqa: E731
    out1 = llm.generate(prompts, sp, bcp, per_sequence_compression_params=mk())
    out2 = llm.generate(prompts, sp, bcp, per_sequence_compression_params=mk())
    assert out1 == out2 and all(len(o) == 7 for o in out1)
    assert runner.kv_manager.num_free_pages == 200 and runner.kv_manager.num_free_batches == 4
    lens = seen[0]  # [layers, B, HKV] when the first decode loop starts: the long prompt is in it, compacted
    long_row = lens.sum(-1).max(dim=1).values  # per layer
    retain = round(0.5 * (3000 - 20) * 2)
    assert all(retain <= int(x) < retain + 2 * 128 for x in long_row), long_row
