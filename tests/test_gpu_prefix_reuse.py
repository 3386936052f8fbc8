"""GPU: prefix reuse on the chunked-prefill machinery (second half; core/prefix_cache.py).

A prompt whose first 1024 tokens were registered with `LLM.cache_prefix` is prefilled from token 1024 on, over a COPY of
the prefix's uncompressed K/V rows, with the prefix's scoring stash - and must end exactly where a cold chunked prefill
of the whole prompt (same chunk boundaries) ends: same tokens, same per-head lengths, same cache rows; the suffix
chunk's attention equals the CPU oracle's `prefill_attention` with the prefix in the cache."""
import math

import pytest
import torch

from helpers import tol
from oracle import ref_cpu as O
from tiny_model import TinyConfig, TinyModel

pytestmark = pytest.mark.gpu


def _llm(dev, num_pages=96):
    from compactor_vllm_amd import LLM, LLMConfig

    cfg = TinyConfig()
    conf = LLMConfig(model="tiny", max_num_seqs=4, max_model_len=4096, hf_config=cfg, eos=-1, kvcache_page_size=128,
                     enforce_eager=True, show_progress_bar=False)
    return LLM(conf, TinyModel(cfg, dev), device=dev, num_pages=num_pages, max_batched_tokens=1024)


def _final_state(runner, seq_row):
    cache = runner.kv_manager.paged_cache
    lens = cache.bh_seq_lens[:, seq_row].cpu()
    rows = []
    for l in range(cache.num_layers):
        kc, vc, pt, _ = cache.layer_slices(l)
        for h in range(cache.H_kv):
            r = O.cache_rows(pt[seq_row, h].cpu(), int(lens[l, h]), cache.page_size)
            rows.append((kc.cpu()[r], vc.cpu()[r]))
    return lens, rows


@pytest.mark.parametrize("method_name,ratio", [("COMPACTOR", 0.5), ("SNAPKV", 0.25), ("NONE", 1.0)])
def test_prefix_reuse_equals_cold_chunked_prefill_and_oracle(dev, method_name, ratio):
    from compactor_vllm_amd import BatchCompressionParams, CompressionMethod, SamplingParams, SequenceCompressionParams
    from compactor_vllm_amd.core.memory_manager import attention_modules
    from compactor_vllm_amd.utils.context import get_context

    method = CompressionMethod[method_name]
    bcp = BatchCompressionParams(compression_method=method)
    scp = lambda: SequenceCompressionParams(ratio, protected_first_tokens=4, protected_last_tokens=16)  # noqa: E731
    sp = SamplingParams(temperature=0.0, max_new_tokens=6)
    g = torch.Generator().manual_seed(23)
    prompt = torch.randint(0, 512, (2348,), generator=g).tolist()
    other = prompt[:1024] + torch.randint(0, 512, (700,), generator=g).tolist()   # same prefix, another suffix
    fresh = torch.randint(0, 512, (1500,), generator=g).tolist()                  # no registered prefix

    def run(llm, p, spy_layer0=None):
        runner = llm.master_model_runner
        seen = {}
        orig = runner.run_decode_loop

        def grab(batch, pending=None):  # state right after the prefill, before the first decode step
            if "lens" not in seen:
                seen["row"] = int(batch.batch_mapping[0])
                seen["lens"], seen["rows"] = _final_state(runner, seen["row"])
            return orig(batch, pending)

        runner.run_decode_loop = grab
        out = llm.generate([p], sp, bcp, per_sequence_compression_params=scp())
        runner.run_decode_loop = orig
        return out[0], seen

    cold = _llm(dev)
    tok_cold, st_cold = run(cold, prompt)
    tok_cold2, st_cold2 = run(cold, other)
    tok_fresh, _ = run(cold, fresh)

    warm = _llm(dev)
    runner = warm.master_model_runner
    free0 = runner.kv_manager.num_free_pages
    assert warm.cache_prefix(prompt[:1100], bcp) == 1024  # rounded down to a multiple of 512
    assert warm.cache_prefix(prompt[:1024], bcp) == 1024  # already known: no second row
    assert len(runner.prefix_cache.entries) == 1
    assert runner.kv_manager.num_free_pages == free0 - 2 * (1024 // 128)  # HKV = 2 heads x 8 pages per layer
    # spy on layer 0 during the FIRST suffix chunk: what the attention kernel was handed, and the cache under it
    attn0 = attention_modules(runner.model)[0]
    rec = {}
    orig_fwd = attn0.forward

    def spy(q, k, v, scores=None):
        ctx = get_context()
        if ctx.is_prefill and "q" not in rec:
            torch.cuda.synchronize()
            bm = ctx.batch_mapping
            rec.update(q=q.cpu(), k=k.cpu(), v=v.cpu(), bm=bm.cpu(), cu=ctx.cu_seqlens_q.cpu(),
                       lens=attn0.bh_seq_lens.index_select(0, bm.long()).cpu(), kc=attn0.k_cache.cpu().clone(),
                       vc=attn0.v_cache.cpu().clone(), pt=attn0.page_table.cpu().clone())
            out = orig_fwd(q, k, v, scores)
            torch.cuda.synchronize()
            rec["out"] = out.cpu()
            return out
        return orig_fwd(q, k, v, scores)

    attn0.forward = spy
    tok_warm, st_warm = run(warm, prompt)
    attn0.forward = orig_fwd
    # (1) identical to the cold chunked prefill with the same chunk boundaries: tokens, lengths, cache rows
    assert tok_warm == tok_cold
    assert torch.equal(st_warm["lens"], st_cold["lens"])
    for (ka, va), (kb, vb) in zip(st_warm["rows"], st_cold["rows"]):
        assert torch.equal(ka, kb) and torch.equal(va, vb)
    if method != CompressionMethod.NONE:
        assert int(st_warm["lens"].max()) < 2348
    # (2) the suffix chunk's attention == oracle prefill attention over [prefix in the cache || chunk]
    cfg = runner.model.cfg
    assert (rec["lens"] == 1024).all() and rec["cu"].tolist() == [0, 1324]  # the chunk [1024, 2348) over a 1024-row prefix
    ref = O.prefill_attention(rec["q"], rec["k"], rec["v"], rec["kc"], rec["vc"], rec["lens"], rec["pt"], rec["bm"],
                              rec["cu"], cfg.num_key_value_heads, 128, 1.0 / math.sqrt(cfg.head_dim))
    assert torch.allclose(rec["out"].float(), ref.float(), atol=tol(torch.bfloat16))
    # (3) the prefix stays resident and serves another suffix; an unrelated prompt takes the cold path
    tok_warm2, st_warm2 = run(warm, other)
    assert tok_warm2 == tok_cold2 and torch.equal(st_warm2["lens"], st_cold2["lens"])
    tok_fresh_w, _ = run(warm, fresh)
    assert tok_fresh_w == tok_fresh
    entry = next(iter(runner.prefix_cache.entries.values()))
    assert entry.hits == 2
    # (4) releasing the prefix returns its pages
    runner.drop_prefixes()
    assert runner.kv_manager.num_free_pages == free0 and not runner.prefix_cache.entries


def test_prefix_reuse_needs_matching_method_and_suffix(dev):
    """A prefix registered for one scoring method does not serve another; SnapKV needs its 32-query window inside the
    suffix; a prompt equal to the prefix has no suffix to run."""
    from compactor_vllm_amd import BatchCompressionParams, CompressionMethod, SamplingParams, SequenceCompressionParams

    llm = _llm(dev)
    runner = llm.master_model_runner
    g = torch.Generator().manual_seed(5)
    prompt = torch.randint(0, 512, (1024 + 20,), generator=g).tolist()
    comp = BatchCompressionParams(compression_method=CompressionMethod.COMPACTOR)
    snap = BatchCompressionParams(compression_method=CompressionMethod.SNAPKV)
    assert llm.cache_prefix(prompt, comp) == 1024 and llm.cache_prefix(prompt, snap) == 1024
    assert len(runner.prefix_cache.entries) == 2
    entries = {e.method: e for e in runner.prefix_cache.entries.values()}
    sp = SamplingParams(temperature=0.0, max_new_tokens=2)
    scp = SequenceCompressionParams(0.5, 4, 16)
    llm.generate([prompt], sp, comp, per_sequence_compression_params=scp)       # 20-token suffix: fine for Compactor
    assert entries[CompressionMethod.COMPACTOR].hits == 1 and entries[CompressionMethod.SNAPKV].hits == 0
    llm.generate([prompt], sp, snap, per_sequence_compression_params=scp)       # 20 < 32: SnapKV takes the cold path
    assert entries[CompressionMethod.SNAPKV].hits == 0
    llm.generate([prompt[:1024]], sp, comp, per_sequence_compression_params=scp)  # no suffix: cold path
    assert entries[CompressionMethod.COMPACTOR].hits == 1
    llm.generate([prompt + prompt[:40]], sp, snap, per_sequence_compression_params=scp)
    assert entries[CompressionMethod.SNAPKV].hits == 1
