"""A tiny Llama-shaped decoder for engine tests: random bf16 weights, plain torch for everything around the attention
boundary, the product's `Attention` module and compression hooks on it (call order of models/llama3.py:96-110).

Every row-wise op with a reduction (linear layers, RMSNorm) runs on fixed 16-row blocks, so the kernel torch picks - and
with it the summation order - never depends on how many rows a batch has: a sequence's activations are bit-identical
whether it is prefilled / decoded alone or packed with others.  That is what lets the engine test demand token-for-token
equality between a continuously batched run and per-request runs.
"""
from __future__ import annotations

import math

import torch

from compactor_vllm_amd.compression import apply_postrope_compression, apply_prerope_compression
from compactor_vllm_amd.layers.attention import Attention
from compactor_vllm_amd.layers.rotary_embedding import fused_qkv_rope
from compactor_vllm_amd.utils.context import get_context

BLOCK = 16


def _blocked(fn, x: torch.Tensor) -> torch.Tensor:
    n = x.shape[0]
    pad = (-n) % BLOCK
    xp = torch.cat([x, x.new_zeros((pad,) + x.shape[1:])]) if pad else x
    outs = [fn(xp[i : i + BLOCK]) for i in range(0, xp.shape[0], BLOCK)]
    return torch.cat(outs)[:n]


def linear(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    return _blocked(lambda b: b @ w.t(), x)


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    def f(b):
        bf = b.float()
        return (bf * torch.rsqrt(bf.pow(2).mean(-1, keepdim=True) + eps)).to(b.dtype) * w

    return _blocked(f, x)


class TinyConfig:
    def __init__(self, hidden=256, layers=2, heads=4, kv_heads=2, head_dim=64, intermediate=512, vocab=512,
                 max_pos=4096, theta=10000.0, eps=1e-5):
        self.hidden_size, self.num_hidden_layers = hidden, layers
        self.num_attention_heads, self.num_key_value_heads, self.head_dim = heads, kv_heads, head_dim
        self.intermediate_size, self.vocab_size = intermediate, vocab
        self.max_position_embeddings, self.rope_theta, self.rms_norm_eps = max_pos, theta, eps
        self.torch_dtype = torch.bfloat16
        self.model_type = "tiny"


class TinyModel:
    def __init__(self, cfg: TinyConfig, device, seed=0, scale=0.06):
        self.cfg, self.dev = cfg, device
        g = torch.Generator(device=device).manual_seed(seed)
        dt = torch.bfloat16

        def w(*shape):
            return (torch.randn(*shape, device=device, generator=g) * scale).to(dt)

        H, D = cfg.hidden_size, cfg.head_dim
        self.qsz, self.kvsz = cfg.num_attention_heads * D, cfg.num_key_value_heads * D
        self.embed = w(cfg.vocab_size, H) * 8
        self.lm_head = w(cfg.vocab_size, H)
        self.final_norm = torch.ones(H, device=device, dtype=dt)
        self.layers = [dict(wqkv=w(self.qsz + 2 * self.kvsz, H), wo=w(H, self.qsz), wgu=w(2 * cfg.intermediate_size, H),
                            wd=w(H, cfg.intermediate_size), n1=torch.ones(H, device=device, dtype=dt),
                            n2=torch.ones(H, device=device, dtype=dt)) for _ in range(cfg.num_hidden_layers)]
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, device=device, dtype=torch.float32) / D))
        ang = torch.arange(cfg.max_position_embeddings, device=device, dtype=torch.float32)[:, None] * inv[None, :]
        self.cos, self.sin = ang.cos(), ang.sin()
        self.cos_sin = torch.cat([self.cos, self.sin], dim=-1).contiguous()  # the reference's cos_sin_cache layout
        self.attn = [Attention(cfg.num_attention_heads, D, 1.0 / math.sqrt(D), cfg.num_key_value_heads)
                     for _ in range(cfg.num_hidden_layers)]

    def attention_modules(self):
        return self.attn

    def _rope(self, x: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:  # neox style, fp32 math, elementwise
        c, s = self.cos[positions][:, None, :], self.sin[positions][:, None, :]
        x1, x2 = x.float().chunk(2, dim=-1)
        return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1).to(x.dtype)

    def __call__(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        cfg = self.cfg
        ctx = get_context()
        N, D = input_ids.numel(), cfg.head_dim
        h = self.embed.index_select(0, input_ids)
        for li, L in enumerate(self.layers):
            x = rmsnorm(h, L["n1"], cfg.rms_norm_eps)
            qkv = linear(x, L["wqkv"])
            # the product's fused producer step: split + RoPE in one launch (k_pre = the pre-RoPE keys for the scoring)
            q, k, v, k_pre = fused_qkv_rope(qkv, positions, self.cos_sin, cfg.num_attention_heads,
                                            cfg.num_key_value_heads, D, want_prerope_k=True)
            scores = None
            compress = ctx.is_prefill and ctx.do_compression
            if compress:
                scores = apply_prerope_compression(qkv[:, : self.qsz].view(N, cfg.num_attention_heads, D), k_pre, v, ctx)
            if compress:
                scores = apply_postrope_compression(q, k, v, scores, ctx)
            o = self.attn[li](q, k, v, scores)
            h = h + linear(o.reshape(N, self.qsz), L["wo"])
            x = rmsnorm(h, L["n2"], cfg.rms_norm_eps)
            gu = linear(x, L["wgu"])
            gate, up = gu.chunk(2, dim=-1)
            h = h + linear((torch.nn.functional.silu(gate.float()) * up.float()).to(h.dtype), L["wd"])
        return h

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        ctx = get_context()
        if ctx.is_prefill:
            hidden = hidden.index_select(0, (ctx.cu_seqlens_q[1:] - 1).long())
        return linear(rmsnorm(hidden, self.final_norm, self.cfg.rms_norm_eps), self.lm_head)
