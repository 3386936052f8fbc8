"""Thin random-weight model shell used ONLY to measure the hot path in situ.

The reference's model zoo is out of scope; measuring the BASELINE.json metric (prompt+generated
tokens per second of one `LLM.generate` call, scheduler.py:203-205) still needs *something* that produces q/k/v around
the attention boundary with the real shapes and issues the calls in the reference's order (models/llama3.py:90-112,
qwen3.py:82-104).  This file is that something: Llama-3.1-8B / Qwen3-8B shaped bf16 weights ~N(0, 0.02^2), dense
projections through torch.matmul (hipBLASLt), the glue kernels of tools/shell/shell_ops.hip, the compactor_vllm_amd
Attention module and compression hooks.  It follows the model protocol of compactor_vllm_amd.core.model_runner
(`model(input_ids, positions)`, `compute_logits`, `attention_modules`), so the product's engine (`LLM`) drives it:
scheduling, the KV cache, prefill / decode orchestration and the HIP-graph decode buckets are the engine's.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
import sys
from dataclasses import dataclass

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "compactor-vllm_amd"))

from compactor_vllm_amd.compression import (  # noqa: E402
    CompressionMethod,
    apply_postrope_compression,
    apply_prerope_compression,
)
from compactor_vllm_amd.layers.attention import Attention  # noqa: E402
from compactor_vllm_amd.layers.rotary_embedding import fused_qkv_rope  # noqa: E402
from compactor_vllm_amd.utils.context import get_context  # noqa: E402

SHELL_DIR = os.path.join(ROOT, "tools", "shell")
SHELL_LIB = os.path.join(SHELL_DIR, "libbench_shell.so")


def build_shell_lib(force: bool = False) -> str:
    src = os.path.join(SHELL_DIR, "shell_ops.hip")
    if force or not os.path.exists(SHELL_LIB) or os.path.getmtime(src) > os.path.getmtime(SHELL_LIB):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", src,
                        "-o", SHELL_LIB], check=True)
    return SHELL_LIB


_shell = None


def shell():
    global _shell
    if _shell is None:
        if not os.path.exists(SHELL_LIB):
            raise RuntimeError(f"{SHELL_LIB} missing: run __graft_entry__.build()")
        L = ctypes.CDLL(SHELL_LIB)
        P, I, F, L64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
        L.shell_add_rmsnorm.argtypes = [P, P, P, P, I, I, F, P]
        L.shell_silu_mul.argtypes = [P, P, ctypes.c_long, I, P]
        L.shell_gemv.argtypes = [P, P, P, I, I, I, P]
        L.shell_gemv.restype = I
        L.shell_gemv_norm.argtypes = [P, P, P, P, P, P, I, I, F, P]
        L.shell_gemv_norm.restype = I
        L.shell_gemv_norm_rope.argtypes = [P, P, P, P, P, P, I, I, F, P, P, I, P]
        L.shell_gemv_norm_rope.restype = I
        L.shell_gemv_silu.argtypes = [P, P, P, I, I, P]
        L.shell_gemv_silu.restype = I
        for f in (L.shell_add_rmsnorm, L.shell_silu_mul):
            f.restype = None
        _shell = L
    return _shell


def _st():
    return torch.cuda.current_stream().cuda_stream


USE_SHELL_GEMV = True


def linear(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """x [N, K] @ w[M, K]^T.  Prefill (N large): hipBLASLt through torch.matmul.  Decode (N <= 4): a plain
    bandwidth-bound GEMV kernel of the shell (hipBLASLt's M=1 kernels reach ~3.9 TB/s on these shapes)."""
    N, K = x.shape
    if USE_SHELL_GEMV and N <= 4 and K * N * 2 <= 65536 and x.is_contiguous() and w.is_contiguous():
        y = torch.empty((N, w.shape[0]), dtype=x.dtype, device=x.device)
        if shell().shell_gemv(w.data_ptr(), x.data_ptr(), y.data_ptr(), N, w.shape[0], K, _st()) == 0:
            return y
    return torch.matmul(x, w.t())


@dataclass
class ShellConfig:
    name: str = "llama-3.1-8b-shape"
    hidden: int = 4096
    layers: int = 32
    heads: int = 32
    kv_heads: int = 8
    head_dim: int = 128
    intermediate: int = 14336
    vocab: int = 128256
    rope_theta: float = 500000.0
    rms_eps: float = 1e-5
    qk_norm: bool = False
    max_pos: int = 131072


LLAMA31_8B = ShellConfig()
QWEN3_8B = ShellConfig(name="qwen3-8b-shape", layers=36, intermediate=12288, vocab=151936, rope_theta=1000000.0,
                       rms_eps=1e-6, qk_norm=True, max_pos=40960)
TINY = ShellConfig(name="tiny-shape", hidden=512, layers=2, heads=8, kv_heads=2, intermediate=1024, vocab=1024,
                   max_pos=8192)


class ModelShell:
    def __init__(self, cfg: ShellConfig, device, max_model_len: int, max_seqs: int = 1, page_size: int = 128,
                 seed: int = 0):
        del max_seqs, page_size  # the KV cache belongs to the engine now
        self.cfg, self.dev = cfg, device
        g = torch.Generator(device=device).manual_seed(seed)
        dt = torch.bfloat16

        def w(*shape):
            return (torch.randn(*shape, device=device, dtype=torch.float32, generator=g) * 0.02).to(dt)

        H, D = cfg.hidden, cfg.head_dim
        self.qsz, self.kvsz = cfg.heads * D, cfg.kv_heads * D
        self.embed = w(cfg.vocab, H)
        self.lm_head = w(cfg.vocab, H)
        self.final_norm = torch.ones(H, device=device, dtype=dt)
        self.layers = []
        for _ in range(cfg.layers):
            self.layers.append(dict(
                wqkv=w(self.qsz + 2 * self.kvsz, H), wo=w(H, self.qsz), wgu=w(2 * cfg.intermediate, H),
                wd=w(H, cfg.intermediate), n1=torch.ones(H, device=device, dtype=dt),
                n2=torch.ones(H, device=device, dtype=dt),
                qn=torch.ones(D, device=device, dtype=dt) if cfg.qk_norm else None,
                kn=torch.ones(D, device=device, dtype=dt) if cfg.qk_norm else None,
            ))
        # rope table [pos, cos(64) | sin(64)] fp32 (plain theta; llama3 frequency scaling does not change cost)
        npos = min(cfg.max_pos, max_model_len + 8)
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, device=device, dtype=torch.float32) / D))
        ang = torch.arange(npos, device=device, dtype=torch.float32)[:, None] * inv[None, :]
        self.rope_cs = torch.cat([ang.cos(), ang.sin()], dim=1).contiguous()
        self.attn = [Attention(cfg.heads, D, 1.0 / math.sqrt(D), cfg.kv_heads) for _ in range(cfg.layers)]

    # ---- the engine's model protocol (compactor_vllm_amd.core.model_runner) --------------------------------------
    @property
    def hf_config(self):
        """The fields `LLMConfig` / `KVCacheManager` read from a Hugging Face config, built locally (no network)."""
        from types import SimpleNamespace

        c = self.cfg
        return SimpleNamespace(model_type=c.name, num_hidden_layers=c.layers, num_key_value_heads=c.kv_heads,
                               num_attention_heads=c.heads, head_dim=c.head_dim, hidden_size=c.hidden,
                               max_position_embeddings=c.max_pos, torch_dtype=torch.bfloat16, vocab_size=c.vocab)

    def attention_modules(self):
        return self.attn

    def __call__(self, input_ids: torch.Tensor, positions: torch.Tensor):
        return self.forward(input_ids, positions)

    def compute_logits(self, hidden) -> torch.Tensor:
        """`hidden` is what `forward` returned: ("logits", t) when the fused one-token path already multiplied by the
        LM head, else ("residual", h, delta) = the two halves of the last residual add, still un-normalised."""
        if hidden[0] == "logits":
            return hidden[1]
        _, h, delta = hidden
        ctx = get_context()
        if ctx.is_prefill:  # only the last token of every sequence feeds the LM head (reference embed_head.py)
            last = (ctx.cu_seqlens_q[1:] - 1).to(torch.int64)
            h, delta = h.index_select(0, last), delta.index_select(0, last)
        xf = torch.empty_like(h)
        shell().shell_add_rmsnorm(h.data_ptr(), delta.data_ptr(), self.final_norm.data_ptr(), xf.data_ptr(), h.shape[0],
                                  self.cfg.hidden, self.cfg.rms_eps, _st())
        return linear(xf, self.lm_head)

    # ---- one forward pass over N packed tokens (prefill) or B single tokens (decode) -----------------------
    def forward(self, tokens: torch.Tensor, positions: torch.Tensor):
        cfg, S = self.cfg, shell()
        N = tokens.numel()
        H, D, I = cfg.hidden, cfg.head_dim, cfg.intermediate
        ctx = get_context()
        h = self.embed.index_select(0, tokens)
        if N == 1 and USE_SHELL_GEMV and not ctx.is_prefill:
            return self._forward_one_token(h, positions)
        x = torch.empty_like(h)
        delta = None
        for li, L in enumerate(self.layers):
            S.shell_add_rmsnorm(h.data_ptr(), None if delta is None else delta.data_ptr(), L["n1"].data_ptr(),
                                x.data_ptr(), N, H, cfg.rms_eps, _st())
            qkv = linear(x, L["wqkv"])
            # the product's fused producer step: qkv split + (Qwen3) per-head q/k RMSNorm + RoPE in ONE
            # launch into a fresh [N, HQ+HKV, D] buffer; k_pre = the pre-RoPE keys the Compactor scoring wants (a view of
            # the projection for Llama, the normed keys for Qwen3, qwen3.py:88-94); v stays a strided view
            compress = ctx.is_prefill and ctx.do_compression
            q, k, v, k_pre = fused_qkv_rope(qkv, positions, self.rope_cs, cfg.heads, cfg.kv_heads, D, L["qn"], L["kn"],
                                            cfg.rms_eps, want_prerope_k=compress)
            scores = None
            if compress:
                scores = apply_prerope_compression(qkv[:, : self.qsz].view(N, cfg.heads, D), k_pre, v, ctx)
            if compress:
                scores = apply_postrope_compression(q, k, v, scores, ctx)
            o = self.attn[li](q, k, v, scores)
            delta = linear(o.view(N, self.qsz), L["wo"])
            S.shell_add_rmsnorm(h.data_ptr(), delta.data_ptr(), L["n2"].data_ptr(), x.data_ptr(), N, H, cfg.rms_eps,
                                _st())
            gu = linear(x, L["wgu"])
            act = torch.empty((N, I), dtype=h.dtype, device=h.device)
            S.shell_silu_mul(gu.data_ptr(), act.data_ptr(), N, I, _st())
            del gu
            delta = linear(act, L["wd"])
            del act
        return ("residual", h, delta)

    def _forward_one_token(self, h: torch.Tensor, positions: torch.Tensor):
        """Decode step for one sequence: 8 launches per layer (norm+qkv GEMV, RoPE, attention stream, split merge,
        o GEMV, norm+gate_up GEMV, silu+down GEMV) - the RMSNorm / residual add / SiLU*mul producers are folded
        into the x-load of the GEMV that consumes them."""
        cfg, S = self.cfg, shell()
        H, D, I = cfg.hidden, cfg.head_dim, cfg.intermediate
        dt, dev = h.dtype, h.device
        ha, hb = h, torch.empty_like(h)  # residual stream ping-pong (the fused kernels must not alias in/out)
        delta = None
        for li, L in enumerate(self.layers):
            qkv = torch.empty((1, self.qsz + 2 * self.kvsz), dtype=dt, device=dev)
            dptr = None if delta is None else delta.data_ptr()
            fused_rope = L["qn"] is None and S.shell_gemv_norm_rope(
                L["wqkv"].data_ptr(), ha.data_ptr(), hb.data_ptr(), dptr, L["n1"].data_ptr(), qkv.data_ptr(),
                self.qsz + 2 * self.kvsz, H, cfg.rms_eps, positions.data_ptr(), self.rope_cs.data_ptr(),
                self.qsz + self.kvsz, _st()) == 0
            if not fused_rope:
                S.shell_gemv_norm(L["wqkv"].data_ptr(), ha.data_ptr(), hb.data_ptr(), dptr, L["n1"].data_ptr(),
                                  qkv.data_ptr(), self.qsz + 2 * self.kvsz, H, cfg.rms_eps, _st())
            ha, hb = hb, ha
            v = qkv[:, self.qsz + self.kvsz :].view(1, cfg.kv_heads, D)
            if fused_rope:  # q and k were rotated in the GEMV's epilogue, in place in the projection buffer
                qk = qkv[:, : self.qsz + self.kvsz].view(1, cfg.heads + cfg.kv_heads, D)
                qh, kh = qk[:, : cfg.heads], qk[:, cfg.heads :]
            else:
                qh, kh, _, _ = fused_qkv_rope(qkv, positions, self.rope_cs, cfg.heads, cfg.kv_heads, D, L["qn"], L["kn"],
                                              cfg.rms_eps)
            o = self.attn[li](qh, kh, v, None)
            delta = linear(o.view(1, self.qsz), L["wo"])
            gu = torch.empty((1, 2 * I), dtype=dt, device=dev)
            S.shell_gemv_norm(L["wgu"].data_ptr(), ha.data_ptr(), hb.data_ptr(), delta.data_ptr(), L["n2"].data_ptr(),
                              gu.data_ptr(), 2 * I, H, cfg.rms_eps, _st())
            ha, hb = hb, ha
            delta = torch.empty((1, H), dtype=dt, device=dev)
            S.shell_gemv_silu(L["wd"].data_ptr(), gu.data_ptr(), delta.data_ptr(), H, I, _st())
        logits = torch.empty((1, cfg.vocab), dtype=dt, device=dev)
        S.shell_gemv_norm(self.lm_head.data_ptr(), ha.data_ptr(), hb.data_ptr(), delta.data_ptr(),
                          self.final_norm.data_ptr(), logits.data_ptr(), cfg.vocab, H, cfg.rms_eps, _st())
        return ("logits", logits)
