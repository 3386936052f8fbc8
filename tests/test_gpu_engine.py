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
