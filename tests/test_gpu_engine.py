"""GPU: the engine loop on the real kernels.  A mixed-length 16-request run through `LLM.generate`
(packed prefill waves under memory pressure, overlapped Compactor / SnapKV scoring + eviction, page reclamation,
continuous batching with stashing, HIP-graph decode buckets with RESERVED padding rows, finished sequences leaving the
batch) must equal per-request runs token for token (greedy).  The model is tests/tiny_model.py, whose row-wise reductions
are batch-invariant by construction; everything on the attention path is the product's HIP library."""
import pytest
import torch

from tiny_model import TinyConfig, TinyModel

pytestmark = pytest.mark.gpu


def _llm(dev, *, eager, num_pages, max_num_seqs=16, max_batched_tokens=None, eos=-1):
    from compactor_vllm_amd import LLM, LLMConfig

    cfg = TinyConfig()
    conf = LLMConfig(model="tiny", max_num_seqs=max_num_seqs, max_model_len=1024, hf_config=cfg, eos=eos,
                     kvcache_page_size=128, enforce_eager=eager, show_progress_bar=False)
    model = TinyModel(cfg, dev)
    return LLM(conf, model, device=dev, num_pages=num_pages, max_batched_tokens=max_batched_tokens)


def _requests(n, seed):
    from compactor_vllm_amd import SamplingParams

    g = torch.Generator().manual_seed(seed)
    lens = [int(x) for x in torch.randint(40, 700, (n,), generator=g)]
    lens[0], lens[1], lens[2] = 700, 33, 257
    prompts = [torch.randint(0, 512, (L,), generator=g).tolist() for L in lens]
    new = [int(x) for x in torch.randint(1, 24, (n,), generator=g)]
    return prompts, [SamplingParams(temperature=0.0, max_new_tokens=k) for k in new]


@pytest.mark.parametrize("method_name,ratio", [("COMPACTOR", 0.5), ("SNAPKV", 0.25), ("NONE", 1.0)])
@pytest.mark.parametrize("eager", [False, True])
def test_mixed_batch_equals_per_request_runs(dev, method_name, ratio, eager):
    from compactor_vllm_amd import BatchCompressionParams, CompressionMethod, SequenceCompressionParams
    from compactor_vllm_amd.attention import sparse_decode_kernel as dk

    method = CompressionMethod[method_name]
    prompts, sp = _requests(16, seed=11)
    bcp = BatchCompressionParams(compression_method=method)
    scp = lambda: SequenceCompressionParams(ratio, protected_first_tokens=4, protected_last_tokens=16)  # noqa: E731
    # pages: 2 kv-heads x ceil((L + new) / 128) per sequence; 60 pages hold only a few sequences at a time
    llm = _llm(dev, eager=eager, num_pages=60, max_num_seqs=8, max_batched_tokens=2048)
    out, seqs = llm.generate(prompts, sp, bcp, per_sequence_compression_params=[scp() for _ in prompts],
                             return_sequences=True)
    runner = llm.master_model_runner
    mgr = runner.kv_manager
    assert mgr.num_free_batches == 8 and mgr.num_free_pages == 60
    assert all(len(o) == s.max_new_tokens + 1 for o, s in zip(out, sp))  # quirk Q11
    if not eager:
        assert len(runner.captured_graphs) >= 2  # several batch buckets were replayed
    # the same requests one at a time through a fresh engine of the same model
    solo = _llm(dev, eager=eager, num_pages=60, max_num_seqs=8, max_batched_tokens=2048)
    for i, (p, s) in enumerate(zip(prompts, sp)):
        o1 = solo.generate([p], s, bcp, per_sequence_compression_params=scp())
        assert o1[0] == out[i], (i, len(p), s.max_new_tokens, o1[0], out[i])
    assert dk.merge_status(dev) == 0
    torch.cuda.synchronize()


def test_engine_compression_really_evicts_and_reclaims(dev):
    """Page accounting of one compressed sequence: during decode it holds ceil((kept + new) / page) pages per head, far
    fewer than the uncompressed reservation."""
    from compactor_vllm_amd import BatchCompressionParams, CompressionMethod, SamplingParams, SequenceCompressionParams
    from compactor_vllm_amd.core.model_runner import ModelRunner

    llm = _llm(dev, eager=True, num_pages=40, max_num_seqs=2)
    runner: ModelRunner = llm.master_model_runner
    seen = {}
    orig = runner.run_decode_loop

    def spy(batch, pending=None):
        seen["free_pages"] = runner.kv_manager.num_free_pages
        seen["lens"] = runner.kv_manager.paged_cache.bh_seq_lens[:, batch.batch_mapping.long()].clone()
        return orig(batch, pending)

    runner.run_decode_loop = spy
    g = torch.Generator().manual_seed(5)
    prompt = torch.randint(0, 512, (900,), generator=g).tolist()
    llm.generate([prompt], SamplingParams(temperature=0.0, max_new_tokens=8),
                 BatchCompressionParams(CompressionMethod.COMPACTOR),
                 per_sequence_compression_params=SequenceCompressionParams(0.25, 4, 16))
    lens = seen["lens"]  # [layers, 1, HKV]
    assert int(lens.max()) < 900 and (lens % 128 == 0).logical_or(lens == 900).all()
    retain = round(0.25 * (900 - 20) * 2)
    for layer in range(lens.shape[0]):
        assert retain <= int(lens[layer].sum()) < retain + 2 * 128
    # uncompressed: ceil(908 / 128) = 8 pages x 2 heads = 16 per layer held; after reclaim far fewer
    assert seen["free_pages"] > 40 - 16
    assert runner.kv_manager.num_free_pages == 40


@pytest.mark.parametrize("method_name,ratio", [("COMPACTOR", 0.5), ("SNAPKV", 0.25), ("NONE", 1.0)])
def test_engine_prefill_cache_and_first_decode_step_equal_oracle(dev, method_name, ratio):
    """The oracle leg of the engine loop (f-1): what `LLM.generate` leaves behind is checked against the CPU oracle,
    layer by layer, not against another run of the engine.  Spies on every layer's `Attention.forward` record what the
    engine fed it; then, per layer:
      * prefill output == oracle.prefill_attention on the recorded q / k / v (empty cache);
      * per-head lengths after the prefill == oracle.retained_sets fed with THAT layer's score tensor and the engine's
        retain counts, and the cache rows of every (sequence, head) == the retained tokens' K / V rows, in token order;
      * the first decode step's attention output == dense softmax attention over {retained rows} U {new token}, and the
        lengths grew by one."""
    import math

    from compactor_vllm_amd import BatchCompressionParams, CompressionMethod, SamplingParams, SequenceCompressionParams
    from compactor_vllm_amd.core.memory_manager import attention_modules
    from compactor_vllm_amd.utils.context import get_context
    from helpers import tol
    from oracle import ref_cpu as O

    method = CompressionMethod[method_name]
    llm = _llm(dev, eager=True, num_pages=60, max_num_seqs=4, max_batched_tokens=2048)
    runner = llm.master_model_runner
    attns = attention_modules(runner.model)
    prefill, decode = {}, {}

    def install(li, attn):
        orig = attn.forward

        def spy(q, k, v, scores=None):
            ctx = get_context()
            torch.cuda.synchronize()  # the score tensor is produced on the store stream
            bm = ctx.batch_mapping.long()
            before = attn.bh_seq_lens.index_select(0, bm).cpu()
            rec = dict(q=q.cpu(), k=k.cpu(), v=v.cpu(), bm=ctx.batch_mapping.cpu(), before=before,
                       scores=None if scores is None else scores.float().cpu())
            if ctx.is_prefill:
                rec["cu"] = ctx.cu_seqlens_k.cpu()
                cc = ctx.compression_context
                rec["retain"] = None if cc is None or not ctx.do_compression else cc.batch_tokens_to_retain.cpu()
            out = orig(q, k, v, scores)
            torch.cuda.synchronize()
            rec["out"] = out.cpu()
            rec["after"] = attn.bh_seq_lens.index_select(0, bm).cpu()
            if ctx.is_prefill:
                pt, PS = attn.page_table.cpu(), attn.page_size
                kc, vc = attn.k_cache.cpu(), attn.v_cache.cpu()
                rows = [[O.cache_rows(pt[int(rec["bm"][b]), h], int(rec["after"][b, h]), PS)
                         for h in range(attn.num_kv_heads)] for b in range(bm.numel())]
                rec["k_rows"] = [[kc[r] for r in per_b] for per_b in rows]
                rec["v_rows"] = [[vc[r] for r in per_b] for per_b in rows]
                assert li not in prefill, "one prefill wave expected"
                prefill[li] = rec
            elif li not in decode:
                decode[li] = rec  # first decode step only
            return out

        attn.forward = spy

    for li, attn in enumerate(attns):
        install(li, attn)
    g = torch.Generator().manual_seed(17)
    lens = [700, 300, 45]
    prompts = [torch.randint(0, 512, (L,), generator=g).tolist() for L in lens]
    first, last = 4, 16
    llm.generate(prompts, SamplingParams(temperature=0.0, max_new_tokens=3), BatchCompressionParams(compression_method=method),
                 per_sequence_compression_params=[SequenceCompressionParams(ratio, first, last) for _ in prompts])
    assert len(prefill) == len(attns) and len(decode) == len(attns)
    cfg = runner.model.cfg
    HQ, HKV, D, PS = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, 128
    G = HQ // HKV
    dtype = torch.bfloat16
    B = len(lens)
    for li in range(len(attns)):
        p, d = prefill[li], decode[li]
        cu = p["cu"]
        assert cu.diff().tolist() == lens and (p["before"] == 0).all()
        lens0 = torch.zeros(B, HKV, dtype=torch.int32)
        dummy = torch.zeros(PS, D, dtype=dtype)
        ref_o = O.prefill_attention(p["q"], p["k"], p["v"], dummy, dummy, lens0, torch.zeros(8, HKV, 1, dtype=torch.int32),
                                    p["bm"], cu, HKV, PS, 1.0 / math.sqrt(D))
        assert torch.allclose(p["out"].float(), ref_o.float(), atol=tol(dtype)), li
        if method == CompressionMethod.NONE:
            kept = [[list(range(L)) for _ in range(HKV)] for L in lens]
            assert torch.equal(p["after"], torch.tensor(lens, dtype=torch.int32)[:, None].repeat(1, HKV))
        else:
            want = [O.retain_count(ratio, L, first, last, HKV) for L in lens]
            assert p["retain"].tolist() == want, li  # the engine's retain counts are the reference formula's
            kept, lens_o = O.retained_sets(p["scores"], cu, p["retain"], lens0, p["bm"], PS, True)
            assert torch.equal(p["after"], lens_o), (li, p["after"].tolist(), lens_o.tolist())
            assert torch.isinf(p["scores"]).any()  # protected tokens present in the engine's score tensor
        for b in range(B):
            for h in range(HKV):
                src = [int(cu[b]) + t for t in sorted(kept[b][h])]
                assert torch.equal(p["k_rows"][b][h], p["k"][src, h]), (li, b, h)
                assert torch.equal(p["v_rows"][b][h], p["v"][src, h]), (li, b, h)
        # first decode step: the decode batch holds the same cache rows (possibly in another order)
        order = [p["bm"].tolist().index(r) for r in d["bm"].tolist()]
        assert sorted(order) == list(range(B))
        assert torch.equal(d["before"], p["after"][order]) and torch.equal(d["after"], d["before"] + 1)
        for i, b in enumerate(order):
            for h in range(HKV):
                src = [int(cu[b]) + t for t in sorted(kept[b][h])]
                K = torch.cat([p["k"][src, h], d["k"][i, h][None]]).float()
                V = torch.cat([p["v"][src, h], d["v"][i, h][None]]).float()
                pr = torch.softmax(d["q"][i, h * G : (h + 1) * G].float() @ K.T / math.sqrt(D), -1)
                assert torch.allclose(d["out"][i, h * G : (h + 1) * G].float(), pr @ V, atol=tol(dtype)), (li, b, h)


_FALLBACK_CHILD = r'''
import os, sys, torch
ROOT = sys.argv[1]
for p in (ROOT, os.path.join(ROOT, "compactor-vllm_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from tiny_model import TinyConfig, TinyModel
from compactor_vllm_amd import LLM, LLMConfig, SamplingParams, BatchCompressionParams, CompressionMethod, SequenceCompressionParams
from compactor_vllm_amd.attention import sparse_decode_kernel as dk
dev = torch.device("cuda:0")
cfg = TinyConfig()
def llm(eager):
    conf = LLMConfig(model="tiny", max_num_seqs=2, max_model_len=4096, hf_config=cfg, eos=-1, kvcache_page_size=128,
                     enforce_eager=eager, show_progress_bar=False)
    return LLM(conf, TinyModel(cfg, dev), device=dev, num_pages=80, max_batched_tokens=4096)
g = torch.Generator().manual_seed(3)
prompt = torch.randint(0, 512, (3000,), generator=g).tolist()   # 2 kv heads x splits: the grid fits the chip -> in-launch merge
sp = SamplingParams(temperature=0.0, max_new_tokens=4)
bcp = BatchCompressionParams(compression_method=CompressionMethod.NONE)
eng = llm(False)
try:
    eng.generate([prompt], sp, bcp)
    print("NO_ERROR")
except RuntimeError as e:
    print("RAISED" if "in-launch split merge timed out" in str(e) else "OTHER " + str(e)[:200])
assert dk.set_merge_mode("default") == "two-kernel"      # the runner switched the process over ...
dk.set_merge_mode("two-kernel")
assert len(eng.master_model_runner.captured_graphs) == 0  # ... and dropped the graphs that captured the in-launch form
eng.master_model_runner.kv_manager.free_sequences(list(eng.master_model_runner.kv_manager.seq_id_to_batch))
out2 = eng.generate([prompt], sp, bcp)                     # the same engine keeps working on the two-kernel merge
ref = llm(True).generate([prompt], sp, bcp)
print("FALLBACK_OK" if out2 == ref and dk.merge_status(dev) == 0 else "FALLBACK_MISMATCH")
'''


def test_engine_reports_in_launch_merge_timeout_and_falls_back(dev):
    """The in-launch split merge assumes co-resident workgroups; when a sibling split never delivers, the merging
    workgroups give up (bounded wait), the engine's per-decode-loop health check raises, switches the process to the
    two-kernel merge and drops the captured graphs - and the next call works.  Driven with the debug build
    tools/dbg/libcvllm_dec_withhold.so (split 1 withholds its numerators) in a child process (CVLLM_LIB_PATH)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "tools", "dbg", "libcvllm_dec_withhold.so")
    if not os.path.exists(lib):
        pytest.fail("tools/dbg/libcvllm_dec_withhold.so missing: run __graft_entry__.build()")
    env = dict(os.environ, CVLLM_LIB_PATH=lib, CVLLM_MERGE_PROBE="0")  # (the start-up probe would catch it first: below)
    env.pop("CVLLM_DECODE_MERGE", None)
    r = subprocess.run([sys.executable, "-c", _FALLBACK_CHILD, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert "RAISED" in lines and "FALLBACK_OK" in lines, r.stdout[-2000:] + r.stderr[-2000:]


_PROBE_CHILD = r'''
import os, sys, warnings, torch
ROOT = sys.argv[1]
for p in (ROOT, os.path.join(ROOT, "compactor-vllm_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from tiny_model import TinyConfig, TinyModel
from compactor_vllm_amd import LLM, LLMConfig, SamplingParams, BatchCompressionParams, CompressionMethod
from compactor_vllm_amd.attention import sparse_decode_kernel as dk
dev = torch.device("cuda:0")
cfg = TinyConfig()
conf = LLMConfig(model="tiny", max_num_seqs=2, max_model_len=4096, hf_config=cfg, eos=-1, kvcache_page_size=128,
                 enforce_eager=False, show_progress_bar=False)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    eng = LLM(conf, TinyModel(cfg, dev), device=dev, num_pages=80, max_batched_tokens=4096)
print("WARNED" if any("start-up probe" in str(x.message) for x in w) else "NO_WARNING")
prev = dk.set_merge_mode("two-kernel")
print("SWITCHED" if prev == "two-kernel" else "MODE " + prev)
g = torch.Generator().manual_seed(3)
prompt = torch.randint(0, 512, (3000,), generator=g).tolist()
out = eng.generate([prompt], SamplingParams(temperature=0.0, max_new_tokens=4),
                   BatchCompressionParams(compression_method=CompressionMethod.NONE))
print("GENERATED" if len(out) == 1 and dk.merge_status(dev) == 0 else "BAD")
'''


def test_engine_startup_probe_moves_to_two_kernel_merge(dev):
    """ModelRunner's start-up probe (one synthetic chip-filling decode launch + the status word): with the debug build in
    which split 1 withholds its numerators - the stand-in for a GPU whose CUs are not the process's alone - the engine
    warns, runs on the two-kernel merge from the first token on, and generate() works."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "tools", "dbg", "libcvllm_dec_withhold.so")
    if not os.path.exists(lib):
        pytest.fail("tools/dbg/libcvllm_dec_withhold.so missing: run __graft_entry__.build()")
    env = dict(os.environ, CVLLM_LIB_PATH=lib)
    env.pop("CVLLM_DECODE_MERGE", None)
    env.pop("CVLLM_MERGE_PROBE", None)
    r = subprocess.run([sys.executable, "-c", _PROBE_CHILD, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert "WARNED" in lines and "SWITCHED" in lines and "GENERATED" in lines, r.stdout[-2000:] + r.stderr[-2000:]
