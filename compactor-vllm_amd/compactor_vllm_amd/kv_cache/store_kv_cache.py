"""Paged KV-cache write paths (MI355X HIP kernels).

Mirror of the reference module `compactor_vllm/kv_cache/store_kv_cache.py`: same keyword-only
signatures and assertions for `prefill_store_all_kv` (:322-371), `prefill_store_topk_kv`
(:81-175) and `decode_store_kv` (:419-466).
"""
from __future__ import annotations

import torch

from .. import _lib
from ..config.constants import RESERVED_BATCH as _RESERVED_BATCH


def _check_common(page_table, bh_lens, batch_mapping, k_cache, v_cache, D):
    assert page_table.is_contiguous(), "page table must be contiguous."
    assert bh_lens.is_contiguous(), "bh_lens must be contiguous."
    assert batch_mapping.is_contiguous(), "batch mapping must be contiguous."
    assert k_cache.is_contiguous() and v_cache.is_contiguous()
    assert (D & (D - 1)) == 0, "D must be a power of 2"
    assert page_table.dtype == torch.int32 and bh_lens.dtype == torch.int32, "index tensors must be int32"


def prefill_store_all_kv(
    *,
    new_keys: torch.Tensor,
    new_values: torch.Tensor,  # [N, H_kv, D]
    cu_seqlens_k: torch.Tensor,  # [B + 1] int32
    max_seqlen_k: int,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    page_table: torch.Tensor,  # [B_total, H_kv, N_LOGICAL_PAGES_MAX] int32
    bh_lens: torch.Tensor,  # [B, H_kv] int32 (UPDATED)
    batch_mapping: torch.Tensor,  # [B] int32 (local->true)
    PAGE_SIZE: int,
    K_TILE: int = 32,  # accepted for signature parity; the HIP kernel tiles by rows per wave
):
    assert new_keys.stride(-1) == 1 and new_values.stride(-1) == 1, "last dim must be contiguous"
    N, HKV, D = new_keys.shape
    _check_common(page_table, bh_lens, batch_mapping, k_cache, v_cache, D)
    _lib.require_cuda(new_keys, new_values, cu_seqlens_k, k_cache, v_cache, page_table, bh_lens, batch_mapping)
    B = batch_mapping.shape[0]
    sk_n, sk_h, _ = new_keys.stride()
    sv_n, sv_h, _ = new_values.stride()
    cu = _lib.i32(cu_seqlens_k)
    bm = _lib.i32(batch_mapping)
    st = _lib.lib().cvllm_store_all_kv(
        new_keys.data_ptr(), new_values.data_ptr(), sk_n, sk_h, sv_n, sv_h, cu.data_ptr(),
        bm.data_ptr(), bh_lens.data_ptr(), page_table.data_ptr(), k_cache.data_ptr(),
        v_cache.data_ptr(), B, N, HKV, D, PAGE_SIZE, page_table.shape[-1], _lib.dtype_code(new_keys.dtype),
        _lib.stream(),
    )
    _lib.check(st, "cvllm_store_all_kv")


def decode_store_kv(
    *,
    key: torch.Tensor,  # [B, HKV, D]
    value: torch.Tensor,  # [B, HKV, D]
    batch_mapping: torch.Tensor,  # [B] int32
    bh_lens: torch.Tensor,  # [B, HKV] or flattened [B*HKV] int32
    page_table: torch.Tensor,  # [B_total, HKV, N_LOGICAL_PAGES_MAX] int32
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,  # [N_PAGES*PAGE_SIZE, D]
    PAGE_SIZE: int,
    TRITON_RESERVED_BATCH: int = None,
):
    assert key.shape == value.shape and key.ndim == 3, "key/value must be [B, HKV, D]"
    B, HKV, D = key.shape
    assert key.stride(-1) == 1 and value.stride(-1) == 1, "key/value last dim must be contiguous."
    _check_common(page_table, bh_lens, batch_mapping, k_cache, v_cache, D)
    _lib.require_cuda(key, value, batch_mapping, bh_lens, page_table, k_cache, v_cache)
    assert batch_mapping.dtype == torch.int32
    sk_b, sk_h, _ = key.stride()
    sv_b, sv_h, _ = value.stride()
    reserved = _RESERVED_BATCH if TRITON_RESERVED_BATCH is None else int(TRITON_RESERVED_BATCH)
    st = _lib.lib().cvllm_store_decode_kv(
        key.data_ptr(), value.data_ptr(), sk_b, sk_h, sv_b, sv_h, batch_mapping.data_ptr(), bh_lens.data_ptr(),
        page_table.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), int(batch_mapping.shape[0]), HKV, D,
        PAGE_SIZE, page_table.shape[2], reserved, _lib.dtype_code(key.dtype), _lib.stream(),
    )
    _lib.check(st, "cvllm_store_decode_kv")


def prefill_store_topk_kv(
    *,
    new_keys: torch.Tensor,  # [N_total, H, D]
    new_vals: torch.Tensor,  # [N_total, H, D]
    indices_topk: torch.Tensor,  # [B, MAX_SEL] (global flattened token*H + head), ranked
    num_tokens_to_retain: torch.Tensor,  # [B] int32
    page_table: torch.Tensor,  # [B_total, H, N_LOGICAL_PAGES_MAX] int32
    batch_mapping: torch.Tensor,  # [B] int32 (local -> true batch rows)
    bh_lens: torch.Tensor,  # [B, H] int32 (contiguous), UPDATED
    k_cache: torch.Tensor,  # [N_PAGES * PAGE_SIZE, D]
    v_cache: torch.Tensor,  # [N_PAGES * PAGE_SIZE, D]
    PAGE_SIZE: int,
    PAD_TO_PAGE_SIZE: bool = True,
    cu_seqlens_k: torch.Tensor | None = None,
    K_TILE: int = 16,  # signature parity only
    TRITON_RESERVED_BATCH: int = None,
):
    """Rank-list store with the reference's argument list.  Deterministic: the slot of a row inside
    its head is its rank order (the reference's atomics make it arbitrary; contract = multiset)."""
    assert new_keys.shape == new_vals.shape
    N_total, H, D = new_keys.shape
    B = indices_topk.shape[0]
    assert page_table.shape[1] == H
    assert bh_lens.shape == (B, H)
    assert new_keys.stride(-1) == 1 and new_vals.stride(-1) == 1, "new_keys/new_vals last dim must be contiguous."
    _check_common(page_table, bh_lens, batch_mapping, k_cache, v_cache, D)
    _lib.require_cuda(new_keys, new_vals, indices_topk, num_tokens_to_retain, page_table, batch_mapping, bh_lens,
                      k_cache, v_cache)
    if PAD_TO_PAGE_SIZE:
        assert cu_seqlens_k is not None
    idx = _lib.i32(indices_topk)
    ret = _lib.i32(num_tokens_to_retain)
    cu = None if cu_seqlens_k is None else _lib.i32(cu_seqlens_k)
    bm = _lib.i32(batch_mapping)
    sk_n, sk_h, _ = new_keys.stride()
    sv_n, sv_h, _ = new_vals.stride()
    reserved = _RESERVED_BATCH if TRITON_RESERVED_BATCH is None else int(TRITON_RESERVED_BATCH)
    st = _lib.lib().cvllm_store_topk_ranked(
        new_keys.data_ptr(), new_vals.data_ptr(), sk_n, sk_h, sv_n, sv_h, idx.data_ptr(), ret.data_ptr(),
        page_table.data_ptr(), bm.data_ptr(), bh_lens.data_ptr(), k_cache.data_ptr(),
        v_cache.data_ptr(), _lib.ptr(cu), B, H, D, int(idx.shape[-1]), PAGE_SIZE, page_table.shape[2],
        1 if PAD_TO_PAGE_SIZE else 0, reserved, _lib.dtype_code(new_keys.dtype), _lib.stream(),
    )
    _lib.check(st, "cvllm_store_topk_ranked")
