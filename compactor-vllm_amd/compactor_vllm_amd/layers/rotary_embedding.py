"""Rotary position embedding and the fused producer step.

`RotaryEmbedding` / `get_rope` keep the reference's constructor, cache layout and `forward(positions, query, key)`
(`compactor_vllm/layers/rotary_embedding.py:20-94`, llama3 frequency scaling included); the rotation itself runs in the
HIP producer kernel.  `fused_qkv_rope` is the whole producer step of the model code in one launch:
qkv split + optional per-head q/k RMSNorm (Qwen3, `layers/layernorm.py:15-25`) + RoPE, optionally also the normed
pre-RoPE keys (Compactor's pre-RoPE scoring input for q/k-norm models, `models/qwen3.py:88-94`) and, without
compression, the cache write of `prefill_store_all_kv` - replacing `models/llama3.py:96-110` / `qwen3.py:88-102` up to the
`Attention` call.
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import NamedTuple, Optional

import torch
from torch import nn

from .. import _lib


def _inv_freq(rotary_dim: int, base: float, rope_scaling: Optional[tuple]) -> torch.Tensor:
    inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
    if rope_scaling is None:
        return inv_freq
    rope_type, factor, low_freq_factor, high_freq_factor, old_len = rope_scaling
    assert rope_type == "llama3"
    low_wavelen, high_wavelen = old_len / low_freq_factor, old_len / high_freq_factor
    wavelen = 2 * math.pi / inv_freq
    scaled = torch.where(wavelen > low_wavelen, inv_freq / factor, inv_freq)
    smooth = (old_len / wavelen - low_freq_factor) / (high_freq_factor - low_freq_factor)
    smoothed = (1 - smooth) * scaled / factor + smooth * scaled
    medium = ~(wavelen < high_wavelen) * ~(wavelen > low_wavelen)
    return torch.where(medium, smoothed, scaled)


class RotaryEmbedding(nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float,
                 rope_scaling: Optional[tuple]) -> None:
        super().__init__()
        self.head_size = head_size
        assert rotary_dim == head_size
        t = torch.arange(max_position_embeddings, dtype=torch.float)
        freqs = torch.einsum("i,j -> ij", t, _inv_freq(rotary_dim, base, rope_scaling))
        cache = torch.cat((freqs.cos(), freqs.sin()), dim=-1).unsqueeze_(1)  # [max_pos, 1, D] fp32: cos | sin
        self.register_buffer("cos_sin_cache", cache, persistent=False)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        """query [N, HQ, D], key [N, HKV, D] (any token stride, heads contiguous) -> rotated copies."""
        N, HQ, D = query.shape
        HKV = key.shape[1]
        assert query.stride(1) == D and key.stride(1) == D and query.stride(2) == 1 and key.stride(2) == 1
        q_out = torch.empty((N, HQ, D), dtype=query.dtype, device=query.device)
        k_out = torch.empty((N, HKV, D), dtype=key.dtype, device=key.device)
        # two launches of the producer kernel: once over the q heads, once over the k heads presented as "q" rows
        # (without norm weights q rows and k rows go through the same arithmetic)
        _producer(query, query.stride(0), positions, self.cos_sin_cache, None, None, 0.0, q_out, q_out, None, HQ, 0, D)
        _producer(key, key.stride(0), positions, self.cos_sin_cache, None, None, 0.0, k_out, k_out, None, HKV, 0, D)
        return q_out, k_out


@lru_cache(1)
def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float, rope_scaling: tuple | None = None):
    return RotaryEmbedding(head_size, rotary_dim, max_position, base, rope_scaling)


class CacheWrite(NamedTuple):
    """Arguments of `prefill_store_all_kv` for the no-compression case: the producer appends K/V to the paged cache."""
    k_cache: torch.Tensor
    v_cache: torch.Tensor
    cu_seqlens_k: torch.Tensor
    batch_mapping: torch.Tensor
    bh_lens: torch.Tensor  # [B, HKV] int32, updated in place (+= sequence length)
    page_table: torch.Tensor
    PAGE_SIZE: int


def _producer(src, s_n, positions, cos_sin, qw, kw, eps, q_out, k_out, k_pre, HQ, HKV, D, *, cache: CacheWrite = None):
    _lib.require_cuda(src, positions, cos_sin)
    pos = positions if positions.dtype == torch.int64 else positions.to(torch.int64)
    cs = cos_sin.view(cos_sin.shape[0], -1)
    assert cs.dtype == torch.float32 and cs.is_contiguous() and cs.shape[1] == D
    c = cache
    st = _lib.lib().cvllm_qkv_rope_producer(
        src.data_ptr(), s_n, pos.data_ptr(), cs.data_ptr(), _lib.ptr(qw), _lib.ptr(kw), float(eps), q_out.data_ptr(),
        q_out.stride(0), k_out.data_ptr(), k_out.stride(0), _lib.ptr(k_pre),
        None if c is None else c.k_cache.data_ptr(), None if c is None else c.v_cache.data_ptr(),
        None if c is None else _lib.i32(c.cu_seqlens_k).data_ptr(),
        None if c is None else _lib.i32(c.batch_mapping).data_ptr(),
        None if c is None else c.bh_lens.data_ptr(), None if c is None else c.page_table.data_ptr(),
        0 if c is None else c.cu_seqlens_k.numel() - 1, 0 if c is None else int(c.PAGE_SIZE),
        0 if c is None else c.page_table.shape[-1], src.shape[0], HQ, HKV, D, cs.shape[0], _lib.dtype_code(src.dtype),
        _lib.stream())
    _lib.check(st, "cvllm_qkv_rope_producer")


def fused_qkv_rope(qkv: torch.Tensor, positions: torch.Tensor, cos_sin_cache: torch.Tensor, num_heads: int,
                   num_kv_heads: int, head_dim: int, q_norm_weight: Optional[torch.Tensor] = None,
                   k_norm_weight: Optional[torch.Tensor] = None, eps: float = 1e-6, *, want_prerope_k: bool = False,
                   cache_write: Optional[CacheWrite] = None):
    """One launch for the producer step.  qkv [N, (HQ + 2 HKV) * D] (last dim contiguous, any row stride).
    Returns (q [N,HQ,D], k [N,HKV,D], v [N,HKV,D] view of qkv, k_pre) where q and k are views of ONE
    [N, HQ + HKV, D] buffer and k_pre is the normed pre-RoPE key tensor (None unless requested; without norm weights the
    pre-RoPE keys are simply `qkv[:, HQ*D:(HQ+HKV)*D]`)."""
    N = qkv.shape[0]
    HQ, HKV, D = num_heads, num_kv_heads, head_dim
    assert qkv.shape[1] == (HQ + 2 * HKV) * D and qkv.stride(1) == 1
    assert (q_norm_weight is None) == (k_norm_weight is None)
    qk = torch.empty((N, HQ + HKV, D), dtype=qkv.dtype, device=qkv.device)
    q, k = qk[:, :HQ], qk[:, HQ:]
    k_pre = None
    if want_prerope_k and k_norm_weight is not None:
        k_pre = torch.empty((N, HKV, D), dtype=qkv.dtype, device=qkv.device)
    _producer(qkv, qkv.stride(0), positions, cos_sin_cache, q_norm_weight, k_norm_weight, eps, q, k, k_pre, HQ, HKV, D,
              cache=cache_write)
    v = qkv[:, (HQ + HKV) * D :].view(N, HKV, D)
    if want_prerope_k and k_pre is None:
        k_pre = qkv[:, HQ * D : (HQ + HKV) * D].view(N, HKV, D)
    return q, k, v, k_pre
