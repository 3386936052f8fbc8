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
from ..compression.common import extract_and_store_top_kv
from ..config.engine_config import AttentionBackend
from ..kv_cache.store_kv_cache import decode_store_kv, prefill_store_all_kv
from ..utils.context import Context, get_context
from ..utils.helpers import maybe_execute_in_stream


class Attention(nn.Module):
    def __init__(self, num_heads, head_dim, scale, num_kv_heads):
        super().__init__()
        self.num_heads: int = num_heads
        self.head_dim = head_dim
        self.scale: float = scale
        self.num_kv_heads = int(num_kv_heads)

        # injected by the cache manager (reference memory_manager.py:82-90): views into PagedKVCache buffers
        self.k_cache: Optional[torch.Tensor] = None
        self.v_cache: Optional[torch.Tensor] = None
        self.page_table: Optional[torch.Tensor] = None
        self.bh_seq_lens: Optional[torch.Tensor] = None
        self.page_size: Optional[int] = None
        self.fused_decode: bool = True  # False = the reference's four-call decode sequence

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scores: Optional[torch.Tensor] = None):
        context: Context = get_context()
        batch_mapping = context.batch_mapping
        if not context.is_prefill and self.fused_decode and self.bh_seq_lens is not None:
            # store-then-attend on the layer's own length table: one C-ABI call instead of
            # index_select + decode_store_kv + attention + index_copy_ (same results, 3 fewer launches)
            assert self.k_cache is not None, "KV Cache must be initialized for decoding"
            return fused_decode_step(
                q, k, v, self.k_cache, self.v_cache, self.bh_seq_lens, self.page_table, batch_mapping,
                int(self.num_kv_heads), self.page_size, self.scale, key_split=context.key_split,
            )
        seq_lens = (
            None if self.bh_seq_lens is None else self.bh_seq_lens.index_select(0, batch_mapping).contiguous()
        )
        if context.is_prefill:
            if context.attention_backend != AttentionBackend.COMPACTOR_TRITON:
                raise NotImplementedError("only the native (COMPACTOR_TRITON-slot) backend exists in this build")
            assert self.k_cache is not None, "KV cache must be initialised"
            seq_lens_copy = seq_lens.clone()  # pre-store lengths: what the attention kernel must see
            if context.do_compression and scores is not None:
                cc = context.compression_context
                assert cc is not None
                maybe_execute_in_stream(
                    extract_and_store_top_kv,
                    scores=scores,
                    cu_seqlens_k=context.cu_seqlens_k,
                    max_k_len=context.max_seqlen_k,
                    top_k=cc.max_tokens_to_retain,
                    H=int(self.num_kv_heads),
                    new_keys=k,
                    new_vals=v,
                    num_tokens_to_retain=cc.batch_tokens_to_retain,
                    page_table=self.page_table,
                    batch_mapping=batch_mapping,
                    bh_lens=seq_lens,
                    k_cache=self.k_cache,
                    v_cache=self.v_cache,
                    PAGE_SIZE=self.page_size,
                    PAD_TO_PAGE_SIZE=True,
                    STORE_STREAM=context.STORE_STREAM,
                )
            else:
                maybe_execute_in_stream(
                    prefill_store_all_kv,
                    new_keys=k,
                    new_values=v,
                    cu_seqlens_k=context.cu_seqlens_k,
                    max_seqlen_k=context.max_seqlen_k,
                    k_cache=self.k_cache,
                    v_cache=self.v_cache,
                    page_table=self.page_table,
                    bh_lens=seq_lens,
                    batch_mapping=batch_mapping,
                    PAGE_SIZE=self.page_size,
                    STORE_STREAM=context.STORE_STREAM,
                )
            o = causal_sparse_varlen_with_cache(
                q,
                k,
                v,
                self.k_cache,
                self.v_cache,
                seq_lens_bh=seq_lens_copy,
                global_page_table=self.page_table,
                batch_mapping=batch_mapping,
                cu_seqlens_q=context.cu_seqlens_q,
                max_seqlen_q=context.max_seqlen_q,
                max_seqlen_k_cache=context.max_bh_len,
                HKV=int(self.num_kv_heads),
                PAGE_SIZE=self.page_size,
                sm_scale=self.scale,
            )
        else:
            assert self.k_cache is not None, "KV Cache must be initialized for decoding"
            decode_store_kv(
                key=k,
                value=v,
                batch_mapping=batch_mapping,
                bh_lens=seq_lens,
                page_table=self.page_table,
                k_cache=self.k_cache,
                v_cache=self.v_cache,
                PAGE_SIZE=self.page_size,
            )
            o = head_sparse_decode_attention(
                q,
                self.k_cache,
                self.v_cache,
                seq_lens,
                self.page_table,
                batch_mapping,
                int(self.num_kv_heads),
                self.page_size,
                self.scale,
                key_split=context.key_split,
            )
        if self.bh_seq_lens is not None:
            maybe_execute_in_stream(
                self.bh_seq_lens.index_copy_,
                0,
                batch_mapping.to(torch.long),
                seq_lens,
                STORE_STREAM=context.STORE_STREAM if context.is_prefill else None,
            )
        return o
