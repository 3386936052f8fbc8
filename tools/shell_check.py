"""Consistency of the shell's fused one-token path against its unfused path (same weights, same cache state)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_shell as bs
from compactor_vllm_amd.compression import CompressionMethod
dev = torch.device("cuda:0")
# TINY exercises the generic GEMV kernels; MID (hidden 2048, K % 2048 == 0) the pipelined ones incl. the RoPE epilogue
MID = bs.ShellConfig(name="mid-shape", hidden=2048, layers=2, heads=16, kv_heads=4, intermediate=4096, vocab=2048,
                     max_pos=8192)
for cfg in (bs.TINY, MID):
    m = bs.ModelShell(cfg, dev, max_model_len=600, max_seqs=1, seed=0)
    g = torch.Generator().manual_seed(1)
    prompt = torch.randint(0, cfg.vocab, (500,), generator=g)
    outs = []
    for fused in (True, False):
        bs.USE_SHELL_GEMV = fused
        outs.append(m.generate([prompt], 24, CompressionMethod.COMPACTOR, 0.5, use_graph=fused).cpu())
    print(cfg.name, "fused tokens  :", outs[0][0, :12].tolist())
    print(cfg.name, "unfused tokens:", outs[1][0, :12].tolist())
    agree = (outs[0] == outs[1]).float().mean().item()
    print(f"agreement {agree:.3f}")
    # greedy decoding of a random-weight model: one near-tie flips a token and everything after it, so the check is
    # the common prefix, not the overall agreement
    first_diff = next((i for i in range(outs[0].shape[1]) if outs[0][0, i] != outs[1][0, i]), outs[0].shape[1])
    print("common prefix", first_diff)
    assert first_diff >= 8, "fused and unfused decode paths diverge early"
    del m
# packed varlen prefill of 3 sequences == one-by-one prefill followed by the same batched decode
m = bs.ModelShell(MID, dev, max_model_len=900, max_seqs=3, seed=0)
g = torch.Generator().manual_seed(2)
prompts = [torch.randint(0, MID.vocab, (n,), generator=g) for n in (700, 333, 512)]
a = m.generate(prompts, 12, CompressionMethod.COMPACTOR, 0.5).cpu()
b = m.generate(prompts, 12, CompressionMethod.COMPACTOR, 0.5, max_prefill_tokens=1).cpu()
print("packed    :", a[:, :6].tolist())
print("sequential:", b[:, :6].tolist())
assert torch.equal(a[:, 0], b[:, 0]), "first tokens differ between packed and one-by-one prefill"
pref = min(next((i for i in range(a.shape[1]) if a[r, i] != b[r, i]), a.shape[1]) for r in range(3))
print("common prefix", pref)
assert pref >= 6
print("shell check ok")
