"""Consistency of the shell's fused one-token path against its unfused path (same weights, same cache state)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_shell as bs
from compactor_vllm_amd.compression import CompressionMethod
dev = torch.device("cuda:0")
cfg = bs.TINY
m = bs.ModelShell(cfg, dev, max_model_len=600, max_seqs=1, seed=0)
g = torch.Generator().manual_seed(1)
prompt = torch.randint(0, cfg.vocab, (500,), generator=g)
outs = []
for fused in (True, False):
    bs.USE_SHELL_GEMV = fused
    outs.append(m.generate([prompt], 24, CompressionMethod.COMPACTOR, 0.5, use_graph=fused).cpu())
print("fused tokens  :", outs[0][0, :12].tolist())
print("unfused tokens:", outs[1][0, :12].tolist())
agree = (outs[0] == outs[1]).float().mean().item()
print(f"agreement {agree:.3f}")
assert outs[0][0, 0] == outs[1][0, 0]
assert agree > 0.8, "fused and unfused decode paths diverge (bf16 argmax ties aside)"
print("shell check ok")
