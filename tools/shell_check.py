"""Consistency of the bench shell model under the product's engine: fused one-token decode path == unfused path, and a
packed 3-sequence prefill == one-by-one prefill (same weights, greedy)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_shell as bs
from compactor_vllm_amd import (LLM, BatchCompressionParams, CompressionMethod, LLMConfig, SamplingParams,
                                SequenceCompressionParams)
dev = torch.device("cuda:0")
# TINY exercises the generic GEMV kernels; MID (hidden 2048, K % 2048 == 0) the pipelined ones incl. the RoPE epilogue
MID = bs.ShellConfig(name="mid-shape", hidden=2048, layers=2, heads=16, kv_heads=4, intermediate=4096, vocab=2048,
                     max_pos=8192)


def engine(cfg, max_len, seqs, eager, budget=None):
    m = bs.ModelShell(cfg, dev, max_model_len=max_len, seed=0)
    conf = LLMConfig(model=cfg.name, max_num_seqs=seqs, max_model_len=max_len, hf_config=m.hf_config, eos=-1,
                     enforce_eager=eager, show_progress_bar=False)
    return LLM(conf, m, device=dev, num_pages=seqs * cfg.kv_heads * (-(-max_len // 128)) + 8, max_batched_tokens=budget)


def gen(llm, prompts, n):
    return llm.generate(prompts, SamplingParams(0.0, n), BatchCompressionParams(CompressionMethod.COMPACTOR),
                        per_sequence_compression_params=[SequenceCompressionParams(0.5) for _ in prompts])


for cfg in (bs.TINY, MID):
    g = torch.Generator().manual_seed(1)
    prompt = torch.randint(0, cfg.vocab, (500,), generator=g).tolist()
    outs = []
    for fused in (True, False):
        bs.USE_SHELL_GEMV = fused
        outs.append(gen(engine(cfg, 600, 1, eager=not fused), [prompt], 23)[0])
    bs.USE_SHELL_GEMV = True
    print(cfg.name, "fused tokens  :", outs[0][:12])
    print(cfg.name, "unfused tokens:", outs[1][:12])
    # greedy decoding of a random-weight model: one near-tie flips a token and everything after it, so the check is
    # the common prefix, not the overall agreement
    first_diff = next((i for i in range(len(outs[0])) if outs[0][i] != outs[1][i]), len(outs[0]))
    print("common prefix", first_diff)
    assert first_diff >= 8, "fused and unfused decode paths diverge early"
# packed varlen prefill of 3 sequences == one-by-one prefill (token budget of one prompt) + the same batched decode
g = torch.Generator().manual_seed(2)
prompts = [torch.randint(0, MID.vocab, (n,), generator=g).tolist() for n in (700, 333, 512)]
a = gen(engine(MID, 900, 3, eager=False), prompts, 11)
b = gen(engine(MID, 900, 3, eager=False, budget=700), prompts, 11)
print("packed    :", [x[:6] for x in a])
print("sequential:", [x[:6] for x in b])
assert [x[0] for x in a] == [x[0] for x in b], "first tokens differ between packed and one-by-one prefill"
pref = min(next((i for i in range(len(x)) if x[i] != y[i]), len(x)) for x, y in zip(a, b))
print("common prefix", pref)
assert pref >= 6
print("shell check ok")
