"""Cross-GPU split-KV decode attention for ONE very long sequence.

The only place on this path where a collective is justified (sequences otherwise shard across GPUs with no data-path
communication): the rows of every (layer, kv-head) of a sequence are partitioned over the W ranks of a process group,
each rank keeping its slice in its own `PagedKVCache` with its own per-head lengths.  A decode step is then

    every rank:  (out_r, lse_r) = head_sparse_decode_attention(q, its slice, return_lse=True)     # HBM-bound, local
    all ranks:   ONE all-gather of the packed (out_r | lse_r) record: B*HQ*(2D + 4) bytes per rank and layer (8.3 KB
                 at B = 1, HQ = 32, D = 128) - RCCL over xGMI; latency-, not bandwidth-bound at this size
    every rank:  out = sum_r exp(lse_r - max) out_r / sum_r exp(lse_r - max)                         # merge kernel

which is the math of the reference's split-K stage 2 (`attention/sparse_decode_kernel.py:391-435`) with the split axis
moved across devices.  `q` is replicated (it is B*HQ*D values); the new token's K/V row is appended by ONE rank per
step (`append_rank`), so the slices stay balanced when the caller rotates it.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import _lib
from .sparse_decode_kernel import fused_decode_step, head_sparse_decode_attention


def merge_shards(out_all: torch.Tensor, lse_all: torch.Tensor) -> torch.Tensor:
    """out_all [W, B, HQ, D] (model dtype), lse_all [W, B, HQ] fp32 -> [B, HQ, D]: LSE-weighted merge (HIP kernel)."""
    _lib.require_cuda(out_all, lse_all)
    W, B, HQ, D = out_all.shape
    assert lse_all.shape == (W, B, HQ) and lse_all.dtype == torch.float32
    out_all = out_all if out_all.is_contiguous() else out_all.contiguous()
    lse_all = lse_all if lse_all.is_contiguous() else lse_all.contiguous()
    out = torch.empty((B, HQ, D), dtype=out_all.dtype, device=out_all.device)
    st = _lib.lib().cvllm_decode_merge_shards(out_all.data_ptr(), lse_all.data_ptr(), out.data_ptr(), W, B, HQ, D,
                                              _lib.dtype_code(out_all.dtype), _lib.stream())
    _lib.check(st, "cvllm_decode_merge_shards")
    return out


def pack_record(out: torch.Tensor, lse: torch.Tensor) -> torch.Tensor:
    """(out [B,HQ,D] 16-bit, lse [B,HQ] fp32) -> one uint8 record [B*HQ*(2D+4)]: out bytes, then lse bytes."""
    return torch.cat([out.contiguous().view(torch.uint8).reshape(-1), lse.contiguous().view(torch.uint8).reshape(-1)])


def unpack_records(buf: torch.Tensor, W: int, B: int, HQ: int, D: int, dtype: torch.dtype):
    rec = buf.view(W, -1)
    n_out = B * HQ * D * 2
    out_all = rec[:, :n_out].contiguous().view(dtype).view(W, B, HQ, D)
    lse_all = rec[:, n_out:].contiguous().view(torch.float32).view(W, B, HQ)
    return out_all, lse_all


def split_kv_decode_attention(q, k_cache, v_cache, seq_lens_bh, page_table, batch_mapping, HKV: int, PAGE_SIZE: int,
                              group=None, sm_scale: Optional[float] = None, *, key: Optional[torch.Tensor] = None,
                              value: Optional[torch.Tensor] = None, bh_seq_lens_layer: Optional[torch.Tensor] = None,
                              local_attention=None, merge=None) -> torch.Tensor:
    """One decode step of a sequence whose KV rows are split over the ranks of `group`.

    This rank's slice: (k_cache, v_cache, page_table, batch_mapping) with lengths `seq_lens_bh` [B, HKV] - or, when
    `key` / `value` / `bh_seq_lens_layer` are given, this rank is the step's append rank: the new row goes into its
    slice first (fused append on the layer's [Bmax+1, HKV] table, like `Attention.forward`).
    `local_attention` / `merge` exist for the CPU protocol test (gloo): they replace the two HIP calls."""
    import torch.distributed as dist

    W = dist.get_world_size(group)
    B, HQ, D = q.shape
    if local_attention is not None:
        out, lse = local_attention(q)
    elif key is not None:
        # append + attend, then read the LSE with a second, attention-only pass over the (now longer) slice would cost a
        # second sweep; instead store first, then ONE attention call that also returns the LSE
        from ..kv_cache.store_kv_cache import decode_store_kv

        lens = bh_seq_lens_layer.index_select(0, batch_mapping.long()).contiguous()
        decode_store_kv(key=key, value=value, batch_mapping=batch_mapping, bh_lens=lens, page_table=page_table,
                        k_cache=k_cache, v_cache=v_cache, PAGE_SIZE=PAGE_SIZE)
        bh_seq_lens_layer.index_copy_(0, batch_mapping.long(), lens)
        out, lse = head_sparse_decode_attention(q, k_cache, v_cache, lens, page_table, batch_mapping, HKV, PAGE_SIZE,
                                                sm_scale, return_lse=True)
    else:
        out, lse = head_sparse_decode_attention(q, k_cache, v_cache, seq_lens_bh, page_table, batch_mapping, HKV,
                                                PAGE_SIZE, sm_scale, return_lse=True)
    rec = pack_record(out, lse)
    gathered = torch.empty((W * rec.numel(),), dtype=torch.uint8, device=rec.device)
    dist.all_gather_into_tensor(gathered, rec, group=group)
    out_all, lse_all = unpack_records(gathered, W, B, HQ, D, out.dtype)
    return (merge or merge_shards)(out_all, lse_all)


__all__ = ["split_kv_decode_attention", "merge_shards", "pack_record", "unpack_records", "fused_decode_step"]
