"""Scoring-method interface, top-k selection and compaction (MI355X HIP kernels).

Mirror of the reference module `compactor_vllm/compression/common.py`:
`BaseCompressionMethod` / `NoCompression` (:9-123), `extract_and_store_top_kv` (:126-168) and
`scores_to_retain_indices` (:171-243) keep their names, argument order and defaults.

`extract_and_store_top_kv` does NOT materialise the int64 rank matrix: it runs the exact radix
select (`cvllm_select_topk`) and the ordered compaction (`cvllm_compact_store`), which give the
same retained set per (sequence, head) and the same `bh_lens` as rank-then-scatter-then-pad
(SURVEY P1); ties are broken by (score desc, flat index asc) (P2).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional

import torch

from .. import _lib
from ..config.constants import RESERVED_BATCH
from ..kv_cache.store_kv_cache import prefill_store_topk_kv  # noqa: F401  (re-export, as the reference)


class BaseCompressionMethod(ABC):
    """Two optional scoring phases around RoPE (reference common.py:9-101).  Both return a
    `[total_tokens, HKV]` score tensor or None (= phase is a no-op / no compression)."""

    @staticmethod
    @abstractmethod
    def pre_rope_scoring(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, context) -> Optional[torch.Tensor]:
        pass

    @staticmethod
    @abstractmethod
    def post_rope_scoring(
        q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, pre_rope_scores: Optional[torch.Tensor], context
    ) -> Optional[torch.Tensor]:
        pass


class NoCompression(BaseCompressionMethod):
    """Disables KV-cache compression (reference common.py:104-123)."""

    @staticmethod
    def pre_rope_scoring(q, k, v, context) -> Optional[torch.Tensor]:
        return None

    @staticmethod
    def post_rope_scoring(q, k, v, pre_rope_scores, context) -> Optional[torch.Tensor]:
        return pre_rope_scores


def select_retained(
    scores: torch.Tensor,  # [N, H]
    cu_seqlens_k: torch.Tensor,
    max_k_len: int,
    num_tokens_to_retain: torch.Tensor,  # [B] int32
    batch_mapping: torch.Tensor,
    bh_lens: torch.Tensor,  # [B, H] lengths before the store (not modified)
    PAGE_SIZE: int,
    PAD_TO_PAGE_SIZE: bool = True,
    reserved_batch: int = RESERVED_BATCH,
):
    """Exact joint top-k + per-head page padding.  Returns (kept_idx [B,H,max_k_len] int32 local token
    indices in ascending token order, new_lens [B,H] int32)."""
    _lib.require_cuda(scores, cu_seqlens_k, num_tokens_to_retain, batch_mapping, bh_lens)
    N, H = scores.shape
    B = cu_seqlens_k.numel() - 1
    sc = scores if scores.dtype == torch.float32 else scores.float()
    sc = sc if sc.is_contiguous() else sc.contiguous()
    cu = _lib.i32(cu_seqlens_k)
    ret = _lib.i32(num_tokens_to_retain)
    bm = _lib.i32(batch_mapping)
    l0 = _lib.i32(bh_lens)
    kept = torch.empty((B, H, max(int(max_k_len), 1)), dtype=torch.int32, device=scores.device)
    new_lens = torch.empty((B, H), dtype=torch.int32, device=scores.device)
    L = _lib.lib()
    ws_bytes = L.cvllm_select_workspace_bytes(B, H, int(max_k_len))
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=scores.device)
    st = L.cvllm_select_topk(
        sc.data_ptr(), cu.data_ptr(), ret.data_ptr(), l0.data_ptr(), bm.data_ptr(), kept.data_ptr(),
        new_lens.data_ptr(), B, H, int(max_k_len), int(PAGE_SIZE), 1 if PAD_TO_PAGE_SIZE else 0,
        int(reserved_batch), ws.data_ptr(), ws_bytes, _lib.stream(),
    )
    _lib.check(st, "cvllm_select_topk")
    return kept, new_lens


def select_status() -> int:
    """Health check of the selection kernels on the current device (synchronises the current stream): 0 = fine,
    1 = a look-back wait of the per-head ordered write timed out since the last check (`cvllm_select_status`; the word
    is cleared).  The engine calls it once per prefill, after the store stream has been joined, and raises."""
    st = int(_lib.lib().cvllm_select_status(_lib.stream()))
    if st < 0:
        _lib.check(st, "cvllm_select_status")
    return st


def extract_and_store_top_kv(
    scores: torch.Tensor,
    cu_seqlens_k: torch.Tensor,
    max_k_len: int,
    top_k: int,
    H: int,
    new_keys: torch.Tensor,  # [N_total, H, D]
    new_vals: torch.Tensor,  # [N_total, H, D]
    num_tokens_to_retain: torch.Tensor,  # [B] int32
    page_table: torch.Tensor,  # [B_total, H, N_LOGICAL_PAGES_MAX] int32
    batch_mapping: torch.Tensor,  # [B] int32 (local -> true batch rows)
    bh_lens: torch.Tensor,  # [B, H] int32 (contiguous), UPDATED
    k_cache: torch.Tensor,  # [N_PAGES * PAGE_SIZE, D]
    v_cache: torch.Tensor,  # [N_PAGES * PAGE_SIZE, D]
    PAGE_SIZE: int,
    PAD_TO_PAGE_SIZE: bool = True,
    K_TILE: int = 16,
    padding: float = -float("inf"),
):
    """scores -> retained (token, head) pairs -> paged cache, on the current stream (reference
    common.py:126-168).  `top_k` (the reference passes max_len*H, i.e. "rank everything") and
    `padding` are accepted for signature parity; the select needs neither."""
    assert scores.shape[1] == H and new_keys.shape[1] == H
    assert new_keys.stride(-1) == 1 and new_vals.stride(-1) == 1
    assert page_table.is_contiguous() and bh_lens.is_contiguous() and bh_lens.dtype == torch.int32
    assert k_cache.is_contiguous() and v_cache.is_contiguous()
    kept, new_lens = select_retained(
        scores, cu_seqlens_k, max_k_len, num_tokens_to_retain, batch_mapping, bh_lens, PAGE_SIZE, PAD_TO_PAGE_SIZE
    )
    B = cu_seqlens_k.numel() - 1
    D = new_keys.shape[-1]
    cu = _lib.i32(cu_seqlens_k)
    bm = _lib.i32(batch_mapping)
    sk_n, sk_h, _ = new_keys.stride()
    sv_n, sv_h, _ = new_vals.stride()
    st = _lib.lib().cvllm_compact_store(
        new_keys.data_ptr(), new_vals.data_ptr(), sk_n, sk_h, sv_n, sv_h, kept.data_ptr(), new_lens.data_ptr(),
        cu.data_ptr(), bh_lens.data_ptr(), page_table.data_ptr(), bm.data_ptr(), k_cache.data_ptr(),
        v_cache.data_ptr(), B, H, D, int(kept.shape[-1]), int(PAGE_SIZE), page_table.shape[-1],
        _lib.dtype_code(new_keys.dtype), _lib.stream(),
    )
    _lib.check(st, "cvllm_compact_store")
    bh_lens.copy_(new_lens)  # lengths become visible after the rows are enqueued (same stream)
    return kept, new_lens


def compact_cache_inplace(
    kept_idx: torch.Tensor,  # [B, H, max_len] int32 ascending token indices (select_retained)
    new_lens: torch.Tensor,  # [B, H] int32 = dst_base + kept count
    dst_base: torch.Tensor,  # [B, H] int32 first destination row
    src_base: torch.Tensor,  # [B, H] int32 logical row of token 0 (>= dst_base)
    page_table: torch.Tensor,
    batch_mapping: torch.Tensor,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    PAGE_SIZE: int,
) -> None:
    """Move the kept rows of every (b, h) down inside the paged cache (last chunk of a chunked prefill): the same final
    cache as `extract_and_store_top_kv` fed with the whole sequence's packed keys / values."""
    _lib.require_cuda(kept_idx, new_lens, dst_base, src_base, page_table, batch_mapping, k_cache, v_cache)
    B, H, max_len = kept_idx.shape
    st = _lib.lib().cvllm_compact_cache_inplace(
        kept_idx.data_ptr(), new_lens.data_ptr(), _lib.i32(dst_base).data_ptr(), _lib.i32(src_base).data_ptr(),
        page_table.data_ptr(), _lib.i32(batch_mapping).data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), B, H,
        k_cache.shape[-1], int(max_len), int(PAGE_SIZE), page_table.shape[-1], _lib.dtype_code(k_cache.dtype),
        _lib.stream())
    _lib.check(st, "cvllm_compact_cache_inplace")


def scores_to_retain_indices(
    scores: torch.Tensor,
    cu_seqlens_k: torch.Tensor,
    max_k_len: int,
    top_k: int,
    H: int,
    padding: float = -float("inf"),
) -> torch.Tensor:
    """Full ranking, kept for API parity (reference common.py:171-243): int64 [B, min(top_k,
    max_k_len*H)] global flat indices `token*H + head`, ordered by (score desc, flat index asc);
    shorter sequences are followed by their padding slots in index order, exactly what a stable
    sort of the reference's padded matrix yields.  Not used by `extract_and_store_top_kv`."""
    _lib.require_cuda(scores, cu_seqlens_k)
    assert scores.shape[1] == H
    assert padding == -float("inf"), "only -inf padding is supported"
    B = cu_seqlens_k.numel() - 1
    k_eff = min(int(top_k), int(max_k_len) * H)
    sc = scores if scores.dtype == torch.float32 else scores.float()
    sc = sc if sc.is_contiguous() else sc.contiguous()
    cu = _lib.i32(cu_seqlens_k)
    out = torch.empty((B, k_eff), dtype=torch.int64, device=scores.device)
    L = _lib.lib()
    ws_bytes = L.cvllm_rank_workspace_bytes(B, H, int(max_k_len))
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=scores.device)
    st = L.cvllm_rank_indices(sc.data_ptr(), cu.data_ptr(), out.data_ptr(), B, H, int(max_k_len), k_eff,
                              ws.data_ptr(), ws_bytes, _lib.stream())
    _lib.check(st, "cvllm_rank_indices")
    return out
