"""Causal varlen prefill attention over [paged KV prefix || appended block] (MI355X MFMA kernel).

Mirror of the reference module `compactor_vllm/attention/sparse_varlen_kernel.py`
(wrapper :11-197): same name, argument order and assertions; the Triton kernel (:277-519) is
replaced by `cvllm_prefill_attn`.
"""
from __future__ import annotations

import math

import torch

from .. import _lib


def causal_sparse_varlen_with_cache(
    q,
    k,
    v,
    k_cache,
    v_cache,
    seq_lens_bh,
    global_page_table,
    batch_mapping,
    cu_seqlens_q,
    max_seqlen_q: int,
    max_seqlen_k_cache: int,
    HKV: int,
    PAGE_SIZE: int,
    sm_scale=None,
):
    """out = softmax_causal(q [K_prefix || k]^T * scale) [V_prefix || v] per sequence and kv-head.

    q [N, HQ, D]; k, v [N, HKV, D] (last dim contiguous, arbitrary token stride — the model passes
    strided views of the fused qkv buffer); k_cache / v_cache [CACHE_SIZE, D]; seq_lens_bh [B, HKV] =
    cached prefix lengths BEFORE this step (per kv-head); global_page_table [MAX_BATCHES, HKV, P];
    batch_mapping [B]; cu_seqlens_q [B+1].  `max_seqlen_k_cache` only keys the reference's autotuner
    (quirk Q10) and is ignored.  Returns [N, HQ, D] contiguous in q.dtype.
    """
    assert q.ndim == 3, "q should be [N, HQ, D]"
    N, HQ, D = q.shape
    assert (D & (D - 1)) == 0, "D must be power of two"
    B = cu_seqlens_q.numel() - 1
    assert B > 0
    assert HQ % HKV == 0, "Number of query heads must divide number of keys heads"
    assert k.shape == (N, HKV, D) and v.shape == (N, HKV, D)
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1, "final dimension must be contiguous"
    assert q.stride(1) == D, "query heads must be contiguous"
    CACHE_SIZE = k_cache.shape[0]
    assert v_cache.shape[0] == CACHE_SIZE
    assert k_cache.shape[1] == D and v_cache.shape[1] == D
    assert PAGE_SIZE > 0 and CACHE_SIZE % PAGE_SIZE == 0
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and global_page_table.is_contiguous()
    _lib.require_cuda(q, k, v, k_cache, v_cache, seq_lens_bh, global_page_table, batch_mapping, cu_seqlens_q)
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)

    cu = _lib.i32(cu_seqlens_q)
    lens = _lib.i32(seq_lens_bh)
    bm = _lib.i32(batch_mapping)
    pt = _lib.i32(global_page_table)
    out = torch.empty((N, HQ, D), dtype=q.dtype, device=q.device)
    st = _lib.lib().cvllm_prefill_attn(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
        k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(), lens.data_ptr(), pt.data_ptr(), bm.data_ptr(),
        cu.data_ptr(), B, N, int(max_seqlen_q), HQ, HKV, D, int(PAGE_SIZE), pt.shape[-1], float(sm_scale),
        _lib.dtype_code(q.dtype), _lib.stream(),
    )
    _lib.check(st, "cvllm_prefill_attn")
    return out
