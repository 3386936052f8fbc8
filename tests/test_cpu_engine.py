"""CPU tests of the engine loop: scheduler policy, cache manager bookkeeping, continuous batching
with the stash / occupancy rule, EOS and length limits, page reclamation - driven by a model stub whose next token is a
closed-form function of (previous token, position), so any mix-up between sequences changes the output.  No GPU kernel
runs here; the GPU counterpart with the real kernels is tests/test_gpu_engine.py."""
import math
from types import SimpleNamespace

import pytest
import torch

from compactor_vllm_amd import LLM, BatchCompressionParams, CompressionMethod, LLMConfig, SamplingParams, \
    SequenceCompressionParams
from compactor_vllm_amd.core.memory_manager import KVCacheManager
from compactor_vllm_amd.core.scheduler import Scheduler
from compactor_vllm_amd.layers.attention import Attention
from compactor_vllm_amd.layers.sampler import Sampler
from compactor_vllm_amd.utils.arguments import DecodeBatchArguments, make_phi, tokens_to_retain
from compactor_vllm_amd.utils.context import get_context
from compactor_vllm_amd.utils.sequence import Sequence, SequenceStatus

VOCAB = 997


def _hf(layers=2, hkv=2, heads=4, D=64, max_pos=4096):
    return SimpleNamespace(num_hidden_layers=layers, num_key_value_heads=hkv, num_attention_heads=heads, head_dim=D,
                           hidden_size=heads * D, max_position_embeddings=max_pos, torch_dtype=torch.bfloat16,
                           model_type="stub")


def _next(tok, pos):
    return (tok * 31 + pos * 7 + 3) % VOCAB


class StubModel:
    """Keeps the cache tables honest (per-head lengths after a 'compressing' prefill, +1 per decode step) without
    computing anything: token t+1 = _next(token t, position t)."""

    def __init__(self, hf, keep_ratio=0.5):
        self.attn = [Attention(hf.num_attention_heads, hf.head_dim, 1.0, hf.num_key_value_heads)
                     for _ in range(hf.num_hidden_layers)]
        self.keep_ratio = keep_ratio
        self.prefill_batches = []
        self.decode_batches = []

    def attention_modules(self):
        return self.attn

    def __call__(self, input_ids, positions):
        ctx = get_context()
        rows = ctx.batch_mapping.long()
        if ctx.is_prefill:
            lens = ctx.cu_seqlens_q.diff()
            self.prefill_batches.append(lens.tolist())
            kept = lens if not ctx.do_compression else (lens.float() * self.keep_ratio).ceil().to(torch.int32)
            for a in self.attn:
                a.bh_seq_lens[rows] = kept[:, None].expand(-1, a.bh_seq_lens.shape[1]).to(torch.int32)
        else:
            self.decode_batches.append(int(input_ids.numel()))
            for a in self.attn:
                a.bh_seq_lens[rows] += 1
        return input_ids, positions

    def compute_logits(self, hidden):
        ids, pos = hidden
        ctx = get_context()
        if ctx.is_prefill:
            last = (ctx.cu_seqlens_q[1:] - 1).long()
            ids, pos = ids[last], pos[last]
        return torch.nn.functional.one_hot(_next(ids, pos), VOCAB).float()


def _expected(prompt, n_new, eos=-1):
    tok, pos, out = prompt[-1], len(prompt) - 1, []
    for _ in range(n_new + 1):  # quirk Q11: max_new_tokens + 1 tokens
        tok = _next(tok, pos)
        pos += 1
        out.append(tok)
        if tok == eos:
            break
    return out


def _engine(num_pages, max_num_seqs=8, max_model_len=1024, eos=-1, max_batched_tokens=None, hf=None):
    hf = hf or _hf()
    cfg = LLMConfig(model="stub", max_num_seqs=max_num_seqs, max_model_len=max_model_len, hf_config=hf, eos=eos,
                    kvcache_page_size=32, enforce_eager=True, show_progress_bar=False)
    model = StubModel(hf)
    return LLM(cfg, model, device="cpu", num_pages=num_pages, max_batched_tokens=max_batched_tokens), model


def _prompts(n, seed=0, lo=5, hi=300):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, VOCAB, (int(torch.randint(lo, hi, (1,), generator=g)),), generator=g).tolist()
            for _ in range(n)]


def test_continuous_batching_mixed_lengths_matches_closed_form():
    """16 requests, prompt lengths 5..300 and 1..12 new tokens, a cache too small for all of them at once (several
    prefill waves, stashing with the 0.66 occupancy rule, pages reclaimed after the 'compressing' prefill): every
    sequence gets exactly its own continuation; rows and pages all return to the pool."""
    llm, model = _engine(num_pages=64, max_num_seqs=6, max_batched_tokens=512)
    prompts = _prompts(16)
    g = torch.Generator().manual_seed(1)
    sp = [SamplingParams(temperature=0.0, max_new_tokens=int(torch.randint(1, 13, (1,), generator=g))) for _ in prompts]
    out, seqs = llm.generate(prompts, sp, BatchCompressionParams(CompressionMethod.COMPACTOR),
                             per_sequence_compression_params=SequenceCompressionParams(0.5), return_sequences=True)
    for p, s, o, seq in zip(prompts, sp, out, seqs):
        assert o == _expected(p, s.max_new_tokens), (len(p), s.max_new_tokens)
        assert seq.status == SequenceStatus.FINISHED and seq.num_tokens_processed == len(p) + len(o)
    assert len(model.prefill_batches) > 2, "the cache was supposed to force several prefill waves"
    assert max(model.decode_batches) > 1, "decode batches were supposed to mix sequences"
    mgr = llm.master_model_runner.kv_manager
    assert mgr.num_free_batches == 6 and mgr.num_free_pages == 64 and not mgr.seq_id_to_batch
    assert int(mgr.paged_cache.bh_seq_lens.abs().sum()) == 0
    sched = llm.master_model_runner.last_scheduler
    assert sched.total_tokens_input == sum(map(len, prompts)) and sched.total_tokens_generated == sum(map(len, out))
    assert llm.last_throughput > 0


def test_eos_stops_a_sequence_and_is_emitted():
    prompts = _prompts(6, seed=3, lo=20, hi=60)
    full = [_expected(p, 10) for p in prompts]
    eos = full[2][4]  # the fifth token of request 2
    llm, _ = _engine(num_pages=256, eos=eos)
    out = llm.generate(prompts, SamplingParams(temperature=0.0, max_new_tokens=10),
                       BatchCompressionParams(CompressionMethod.NONE))
    for p, o in zip(prompts, out):
        assert o == _expected(p, 10, eos=eos)
    assert out[2][-1] == eos and len(out[2]) <= 5


def test_prompt_that_never_fits_raises():
    llm, _ = _engine(num_pages=8)
    with pytest.raises(RuntimeError):
        llm.generate([list(range(500))], SamplingParams(temperature=0.0, max_new_tokens=4))


def test_string_prompts_need_a_tokenizer_and_short_prompts_are_not_compressed():
    llm, _ = _engine(num_pages=64)
    with pytest.raises(ValueError):
        llm.generate("hello", SamplingParams(temperature=0.0, max_new_tokens=2))
    cp = SequenceCompressionParams(0.25, protected_first_tokens=16, protected_last_tokens=64)
    _, seqs = llm.generate([list(range(50))], SamplingParams(temperature=0.0, max_new_tokens=2),
                           BatchCompressionParams(), per_sequence_compression_params=cp, return_sequences=True)
    assert seqs[0].compression_params.compression_ratio == 1.0  # reference llm_engine.py:144-145


def test_scheduler_admission_rules():
    """Token budget, free rows, and STRICTLY fewer pages than free (reference scheduler.py:92-101)."""
    mgr = SimpleNamespace(max_batched_tokens=100, num_free_batches=2, num_free_pages=9, page_size=32, num_kv_heads=2)
    mk = lambda L, new: Sequence(list(range(L)), sampling_params=SamplingParams(max_new_tokens=new))  # noqa: E731
    a, b, c, d = mk(60, 4), mk(50, 4), mk(30, 2), mk(5, 1)
    s = Scheduler([a, b, c, d], mgr)
    got = s.get_prefill_batch()
    # a: 2 pages x 2 heads = 4 < 9 ok (tokens 60); b: 60 + 50 > 100 no; c: 30 ok -> pages 4 + 2*1... rows then exhausted
    assert [x.seq_id for x in got] == [a.seq_id, c.seq_id]
    mgr.num_free_pages = 4  # a and b need 4 pages each: 4 < 4 is false; c takes 2 of the 4; d needs 2 < 2: no
    assert [x.seq_id for x in s.get_prefill_batch()] == [c.seq_id]
    s.add_running_sequence_ids([a.seq_id], update_status=True)
    assert a.status == SequenceStatus.RUNNING and s.total_tokens_input == 60 and s.any_pending_sequences()
    assert s.get_finished_sequence_ids_from_unfinished([]) == {a.seq_id}
    s.record_finished_sequence_ids([a.seq_id], update_status=True)
    assert a.status == SequenceStatus.FINISHED and not s.is_finished()


def test_cache_manager_rows_pages_and_reclaim():
    cfg = LLMConfig(model="stub", max_num_seqs=3, max_model_len=256, hf_config=_hf(), kvcache_page_size=32,
                    show_progress_bar=False)
    model = StubModel(cfg.hf_config)
    mgr = KVCacheManager(cfg, "cpu", num_pages=40)
    mgr.init_cache(model)
    assert all(a.k_cache is not None and a.page_size == 32 for a in model.attn)
    assert mgr.max_batched_tokens == 256  # min(cache share of one head, max_model_len), page multiple
    ok, rows = mgr.allocate_sequences([7, 9], [100, 33])
    assert ok and rows.tolist() == [1, 2] and mgr.num_free_batches == 1
    assert mgr.num_free_pages == 40 - (4 + 2) * 2  # ceil(100/32) and ceil(33/32) pages for each of 2 heads
    ok2, _ = mgr.allocate_sequences([11], [300])  # longer than max_model_len
    assert not ok2
    mgr.free_sequences([11])
    for a in model.attn:
        a.bh_seq_lens[1] = 40  # "compressed" to 40 rows per head
    freed = mgr.reclaim_pages([7], [8])
    assert freed > 0 and mgr.num_free_pages == 40 - (2 + 2) * 2
    mgr.free_sequences([7, 9])
    assert mgr.num_free_batches == 3 and mgr.num_free_pages == 40


def test_retain_formula_phi_and_sequence_record():
    assert tokens_to_retain(0.5, 32768, 16, 64, 8) == round(0.5 * (32768 - 80) * 8)
    assert tokens_to_retain(0.3, 50, 16, 64, 8) == 1  # quirk Q2: first + last >= L
    assert tokens_to_retain(0.25, 101, 0, 0, 2) == 50  # python banker's rounding of 50.5
    phi = make_phi(128, 48, torch.float32, "cpu", seed=42)
    assert phi.shape == (128, 48) and torch.equal(phi, make_phi(128, 48, torch.float32, "cpu", seed=42))
    assert abs(float(phi.std()) - 1 / math.sqrt(48)) < 0.01
    s = Sequence([1, 2, 3], sampling_params=SamplingParams(max_new_tokens=2))
    t = Sequence([4])
    assert t.seq_id == s.seq_id + 1 and s.prompt_len == 3 and s.completion_len == 0
    s.add_new_token(9)
    s.add_new_token(8)
    assert s.num_tokens_processed == 5 and s.completion_token_ids == [9, 8]
    assert s.tokens_to_retain_per_layer(8) == 24


def test_sampler_greedy_and_temperature():
    torch.manual_seed(0)
    logits = torch.tensor([[0.0, 5.0, 1.0], [2.0, 0.0, 0.0]]).repeat(2000, 1)
    temps = torch.tensor([0.0, 1.0]).repeat(2000)
    out = Sampler()(logits, temps)
    assert (out[0::2] == 1).all()  # greedy rows
    p0 = float((out[1::2] == 0).float().mean())
    expect = math.exp(2.0) / (math.exp(2.0) + 2.0)
    assert abs(p0 - expect) < 0.03


def test_decode_batch_record_update_and_select():
    b = DecodeBatchArguments()
    b.update(torch.tensor([1, 2]), torch.tensor([10, 20]), torch.tensor([5, 6]), torch.tensor([9, 9]), torch.tensor([0, 1]),
             torch.tensor([0.0, 0.0]), 1)
    b.update(torch.tensor([3]), torch.tensor([30]), torch.tensor([7]), torch.tensor([9]), torch.tensor([2]),
             torch.tensor([1.0]))
    assert len(b) == 3 and b.desired_batch_occupancy == 1 and b.seq_ids.tolist() == [0, 1, 2]
    b.select(torch.tensor([0, 2]))
    assert b.token_ids.tolist() == [10, 30] and b.temps.tolist() == [0.0, 1.0]
