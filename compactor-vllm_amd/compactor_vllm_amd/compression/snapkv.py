"""SnapKV query-aware key scoring on MI355X HIP kernels.

Mirror of the reference module `compactor_vllm/compression/snapkv.py`: `SnapKVCompression` (:12-36)
and `query_aware_key_scores` (:332-448) keep their names, argument order and defaults.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .. import _lib
from ..utils.helpers import maybe_execute_in_stream
from .common import BaseCompressionMethod
from .compactor import zscore_segments_


class SnapKVCompression(BaseCompressionMethod):
    @staticmethod
    def pre_rope_scoring(q, k, v, context) -> Optional[torch.Tensor]:
        return None

    @staticmethod
    def post_rope_scoring(q, k, v, pre_rope_scores, context) -> Optional[torch.Tensor]:
        if getattr(context, "chunk", None) is not None:
            return maybe_execute_in_stream(_chunked_post_rope, q, k, context, STORE_STREAM=context.STORE_STREAM)
        return maybe_execute_in_stream(
            query_aware_key_scores,
            q,
            k,
            context.cu_seqlens_q,
            context.cu_seqlens_k,
            w=32,
            max_seqlen_k=context.max_seqlen_k,
            STORE_STREAM=context.STORE_STREAM,
        )


def _chunked_post_rope(q, k, context):
    """Chunked prefill (utils/chunked.py): SnapKV scores once, on the last chunk - the window's queries are in this
    chunk, the keys are [rows of the earlier chunks read back from this layer's cache || this chunk's keys]."""
    ch = context.chunk
    li = ch.state.next_layer()
    if not ch.is_last:
        return None
    attn = ch.state.attn[li]
    HKV, D = k.shape[1], k.shape[2]
    k_all = torch.empty((ch.total_len, HKV, D), dtype=k.dtype, device=k.device)
    if ch.start > 0:
        pos = torch.arange(ch.start, device=k.device)
        pages = attn.page_table[context.batch_mapping[0].long()].long()  # [HKV, P] of the sequence's row
        rows = pages[:, pos // attn.page_size] * attn.page_size + (pos % attn.page_size)[None, :]  # [HKV, start]
        k_all[: ch.start] = attn.k_cache[rows].transpose(0, 1)
    k_all[ch.start :] = k
    cu_k = torch.tensor([0, ch.total_len], dtype=torch.int32).to(k.device, non_blocking=True)
    return query_aware_key_scores(q, k_all, context.cu_seqlens_q, cu_k, w=32, max_seqlen_k=ch.total_len)


def query_aware_key_scores(
    q: torch.Tensor,  # [N_q, Hq, D]
    k: torch.Tensor,  # [N_k, Hk, D]
    cu_seqlens_q: torch.Tensor,  # [B+1], int32
    cu_seqlens_k: torch.Tensor,  # [B+1], int32
    w: "torch.Tensor | int",  # [B] int32 or one window for every sequence
    sm_scale: float = None,  # defaults to 1/sqrt(D)
    *,
    accum_scores: torch.Tensor = None,
    accum_blending: float = None,
    normalize: bool = False,
    max_seqlen_k: int = None,
    pool_tile: int = 128,
) -> Optional[torch.Tensor]:
    """s_j = sum over the last `w` queries x G heads of softmax_row(q k^T / sqrt D) restricted to keys
    [0, L-w); trailing 5-tap mean clipped at the start of the key's `pool_tile`-wide block (counted from the sequence
    start): `pool_tile` in {32, 64, 128} is the BLOCK_K the reference's autotuner picked for its
    `_scores_from_logits_kernel` (snapkv.py:160-168, 253-262) - the pooled values depend on it, so a caller that must
    match a given reference box passes that box's outcome (default 128; extension argument);
    last w keys <- +inf.  fp32 [N_k, Hk].  `w` is an int (the engine passes 32) or a [B] int32 tensor of per-sequence
    windows (one host sync for its maximum, like the reference's `w.max().item()`).  normalize=True z-scores every
    sequence's scored rows [0, L-w) over all heads (biased variance, eps 1e-12 inside the square root; reference
    :279-329).  `max_seqlen_k` avoids a host sync; when omitted it is read from cu_seqlens_k.  Sequences with L <= w
    come back all +inf (reference: uninitialised)."""
    assert q.stride(-1) == 1 and k.stride(-1) == 1, "last dim must be contiguous"
    assert pool_tile in (32, 64, 128), "pool_tile must be one of the reference's BLOCK_K configurations"
    _lib.require_cuda(q, k, cu_seqlens_q, cu_seqlens_k)
    w_b = None
    if not isinstance(w, int):
        _lib.require_cuda(w)
        w_b = _lib.i32(w)
        assert w_b.numel() == cu_seqlens_q.numel() - 1
        w = int(w_b.max().item())
    N_q, Hq, D = q.shape
    N_k, Hk, Dk = k.shape
    assert (Hq % Hk) == 0, "Hq must be a multiple of Hk"
    assert q.stride(1) == D
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)
    B = cu_seqlens_q.numel() - 1
    assert B == cu_seqlens_k.numel() - 1
    if w * (Hq // Hk) == 0:
        return torch.zeros((N_k, Hk), dtype=torch.float32, device=q.device)
    cq, ck = _lib.i32(cu_seqlens_q), _lib.i32(cu_seqlens_k)
    if max_seqlen_k is None:
        max_seqlen_k = int(ck.diff().max().item())
    out = torch.empty((N_k, Hk), dtype=torch.float32, device=q.device)
    L = _lib.lib()
    ws_bytes = L.cvllm_snapkv_workspace_bytes(B, Hk, w, int(max_seqlen_k))
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=q.device)
    st = L.cvllm_snapkv_scores_wb(
        q.data_ptr(), k.data_ptr(), q.stride(0), k.stride(0), k.stride(1), out.data_ptr(), cq.data_ptr(), ck.data_ptr(),
        _lib.ptr(w_b), B, Hq, Hk, D, int(w), float(sm_scale), 5, int(pool_tile), int(max_seqlen_k),
        _lib.dtype_code(q.dtype),
        ws.data_ptr(), ws_bytes, _lib.stream(),
    )
    _lib.check(st, "cvllm_snapkv_scores_wb")
    if normalize:
        zb = L.cvllm_zscore_workspace_bytes(B)
        zws = torch.empty(max(zb, 4), dtype=torch.uint8, device=q.device)
        st = L.cvllm_zscore_windowed(out.data_ptr(), 2, ck.data_ptr(), _lib.ptr(w_b), int(w), B, Hk, 1e-12, N_k,
                                     zws.data_ptr(), zb, _lib.stream())
        _lib.check(st, "cvllm_zscore_windowed")
    if accum_scores is not None:
        if accum_blending is not None:
            accum_scores.mul_(accum_blending)
        accum_scores.add_(out)
        return accum_scores
    return out
