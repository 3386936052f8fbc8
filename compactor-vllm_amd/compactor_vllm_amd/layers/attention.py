"""The boundary orchestrator: which kernel runs when, on which stream, and the `bh_seq_lens` write-back.

Mirror of the reference module `compactor_vllm/layers/attention.py:17-161` (same class, constructor, module
attributes injected by the cache manager, and call order), with explicit stream dependencies:

  prefill:  STORE_STREAM (after main):  extract_and_store_top_kv | prefill_store_all_kv
            main stream, concurrently:    causal_sparse_varlen_with_cache with the PRE-store lengths
            STORE_STREAM:                 bh_seq_lens[layer][batch_mapping] = new lengths
  decode:   decode_store_kv ; head_sparse_decode_attention ; write-back        (all on the current stream)

The FLASH_ATTENTION backend of the reference (an optional third-party alternative) is not provided.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from ..attention.sparse_decode_kernel import fused_decode_step, head_sparse_decode_attention
from ..attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache
from ..compression.common import compact_cache_inplace, extract_and_store_top_kv, select_retained
from ..config.engine_config import AttentionBackend
from ..kv_cache.store_kv_cache import decode_store_kv, prefill_store_all_kv
from ..utils.context import Context, get_context
from ..utils.helpers import maybe_execute_in_stream


class Attention(nn.Module):
    """Constructor signature and the attributes a cache manager injects are the reference's (`:17-36`); `forward`
    dispatches to one method per phase."""

    def __init__(self, num_heads, head_dim, scale, num_kv_heads):
        super().__init__()
        self.num_heads: int = num_heads
        self.head_dim = head_dim
        self.scale: float = scale
        self.num_kv_heads = int(num_kv_heads)
        # views into the PagedKVCache buffers, set per layer by whoever owns the cache (reference
        # memory_manager.py:82-90); None until then
        for name in ("k_cache", "v_cache", "page_table", "bh_seq_lens", "page_size"):
            setattr(self, name, None)
        self.fused_decode: bool = True  # False = the reference's four-call decode sequence

    # ------------------------------------------------------------------------------------------------ helpers
    def _cache_args(self) -> dict:
        return dict(page_table=self.page_table, k_cache=self.k_cache, v_cache=self.v_cache, PAGE_SIZE=self.page_size)

    def _gathered_lengths(self, batch_mapping) -> Optional[torch.Tensor]:
        """[B, HKV] copy of this layer's lengths for the rows of the batch (the reference's `seq_lens`)."""
        if self.bh_seq_lens is None:
            return None
        return self.bh_seq_lens.index_select(0, batch_mapping).contiguous()

    def _write_back_lengths(self, context: Context, batch_mapping, seq_lens) -> None:
        if self.bh_seq_lens is None:
            return
        maybe_execute_in_stream(self.bh_seq_lens.index_copy_, 0, batch_mapping.to(torch.long), seq_lens,
                                STORE_STREAM=context.STORE_STREAM if context.is_prefill else None)

    # ------------------------------------------------------------------------------------------------ phases
    def _prefill(self, context: Context, q, k, v, scores, seq_lens):
        """Cache write (compacted or full) on the store stream, attention over [old prefix || new block] on the main
        stream with the lengths as they were BEFORE the write."""
        if context.attention_backend != AttentionBackend.COMPACTOR_TRITON:
            raise NotImplementedError("only the native (COMPACTOR_TRITON-slot) backend exists in this build")
        assert self.k_cache is not None, "KV cache must be initialised"
        batch_mapping = context.batch_mapping
        lengths_before = seq_lens.clone()
        common = dict(self._cache_args(), batch_mapping=batch_mapping, bh_lens=seq_lens,
                      STORE_STREAM=context.STORE_STREAM)
        chunk = getattr(context, "chunk", None)
        compress = context.do_compression and scores is not None and chunk is None
        if compress:
            cc = context.compression_context
            assert cc is not None
            maybe_execute_in_stream(
                extract_and_store_top_kv, scores=scores, cu_seqlens_k=context.cu_seqlens_k,
                max_k_len=context.max_seqlen_k, top_k=cc.max_tokens_to_retain, H=int(self.num_kv_heads), new_keys=k,
                new_vals=v, num_tokens_to_retain=cc.batch_tokens_to_retain, PAD_TO_PAGE_SIZE=True, **common)
        else:
            maybe_execute_in_stream(
                prefill_store_all_kv, new_keys=k, new_values=v, cu_seqlens_k=context.cu_seqlens_k,
                max_seqlen_k=context.max_seqlen_k, **common)
        out = causal_sparse_varlen_with_cache(
            q, k, v, self.k_cache, self.v_cache, seq_lens_bh=lengths_before, global_page_table=self.page_table,
            batch_mapping=batch_mapping, cu_seqlens_q=context.cu_seqlens_q, max_seqlen_q=context.max_seqlen_q,
            max_seqlen_k_cache=context.max_bh_len, HKV=int(self.num_kv_heads), PAGE_SIZE=self.page_size,
            sm_scale=self.scale)
        if chunk is not None and chunk.is_last and context.do_compression and scores is not None:
            # last chunk of a chunked prefill: the whole prompt now sits in the cache uncompressed; select over the
            # whole sequence's scores and compact the cache in place.  Enqueued on the store stream BEHIND the attention
            # above (maybe_execute_in_stream makes that stream wait for the caller's stream first): the attention reads
            # the very rows the compaction rewrites.
            maybe_execute_in_stream(self._compact_whole_sequence, context, chunk, scores, seq_lens,
                                    STORE_STREAM=context.STORE_STREAM)
        return out

    def _compact_whole_sequence(self, context: Context, chunk, scores, seq_lens) -> None:
        cc = context.compression_context
        dev = scores.device
        cu = torch.tensor([0, chunk.total_len], dtype=torch.int32).to(dev, non_blocking=True)
        base = torch.zeros_like(seq_lens)  # the sequence's cache was empty before its first chunk
        kept, new_lens = select_retained(scores, cu, chunk.total_len, cc.batch_tokens_to_retain, context.batch_mapping,
                                         base, self.page_size, True)
        compact_cache_inplace(kept, new_lens, base, base, self.page_table, context.batch_mapping, self.k_cache,
                              self.v_cache, self.page_size)
        seq_lens.copy_(new_lens)  # written back to the layer's table by forward() on the same stream

    def _decode_reference_order(self, context: Context, q, k, v, seq_lens):
        """append the token's K/V row, then attend over the whole cache (reference `:127-150`)"""
        assert self.k_cache is not None, "KV Cache must be initialized for decoding"
        batch_mapping = context.batch_mapping
        decode_store_kv(key=k, value=v, batch_mapping=batch_mapping, bh_lens=seq_lens, **self._cache_args())
        return head_sparse_decode_attention(q, self.k_cache, self.v_cache, seq_lens, self.page_table, batch_mapping,
                                            int(self.num_kv_heads), self.page_size, self.scale,
                                            key_split=context.key_split)

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scores: Optional[torch.Tensor] = None):
        context: Context = get_context()
        batch_mapping = context.batch_mapping
        if not context.is_prefill and self.fused_decode and self.bh_seq_lens is not None:
            # store-then-attend directly on the layer's length table: one C-ABI call instead of
            # index_select + decode_store_kv + attention + index_copy_ (same results, 3 fewer launches)
            assert self.k_cache is not None, "KV Cache must be initialized for decoding"
            return fused_decode_step(q, k, v, self.k_cache, self.v_cache, self.bh_seq_lens, self.page_table,
                                     batch_mapping, int(self.num_kv_heads), self.page_size, self.scale,
                                     key_split=context.key_split, max_len_hint=context.decode_len_hint)
        seq_lens = self._gathered_lengths(batch_mapping)
        if context.is_prefill:
            o = self._prefill(context, q, k, v, scores, seq_lens)
        else:
            o = self._decode_reference_order(context, q, k, v, seq_lens)
        self._write_back_lengths(context, batch_mapping, seq_lens)
        return o
