"""Owner of the paged KV cache: sizes it, maps request ids to rows of the cache tables, hands the per-layer views to the
model's attention modules.

Public surface as `compactor_vllm/core/memory_manager.py:14-182` (`allocate_sequences`, `free_sequences`,
`reclaim_pages`, `init_cache`, `num_free_batches`, `num_free_pages`, `estimate_max_batched_tokens`).  Sizing is explicit
here: the number of pages comes from `num_pages` when given, otherwise from `gpu_memory_utilization` x the device's
memory minus what is allocated already (288 GB on MI355X: a 128 K-context Llama-3.1-8B sequence reserves 16 GiB of pages
during prefill and keeps 8 GiB after a 50 % compaction).
"""
from __future__ import annotations

import logging
from typing import Iterable, List, Optional, Sequence as Seq

import torch

from ..kv_cache.page_table import KVAllocationStatus, PagedKVCache

logger = logging.getLogger(__name__)


def attention_modules(model) -> list:
    """The `Attention` module of every layer: `model.attention_modules` when the model offers it, else the reference's
    `model.model.layers[i].self_attn.attn` walk (memory_manager.py:82-90)."""
    mods = getattr(model, "attention_modules", None)
    if mods is not None:
        return list(mods() if callable(mods) else mods)
    return [layer.self_attn.attn for layer in model.model.layers]


class KVCacheManager:
    def __init__(self, config, device, *, num_pages: Optional[int] = None, max_batched_tokens: Optional[int] = None):
        hf = config.hf_config
        self.device = torch.device(device)
        self.gpu_frac = config.gpu_memory_utilization
        self.page_size = int(config.kvcache_page_size)
        self.max_num_batches = int(config.max_num_seqs)
        self.max_model_len = int(config.max_model_len)
        self.num_layers = int(hf.num_hidden_layers)
        self.model_dtype = getattr(hf, "torch_dtype", None) or getattr(hf, "dtype", torch.bfloat16)
        self.head_dim = int(getattr(hf, "head_dim", None) or hf.hidden_size // hf.num_attention_heads)
        self.num_kv_heads = int(hf.num_key_value_heads)
        self.max_pages_per_batch = -(-self.max_model_len // self.page_size)
        self.num_pages = num_pages
        self.paged_cache: Optional[PagedKVCache] = None
        self.max_batched_tokens = max_batched_tokens
        self.seq_id_to_batch: dict = {}
        self.chunked_prefill = True  # prompts longer than max_batched_tokens are prefilled in chunks (scheduler.py)

    # ---- sizing ------------------------------------------------------------------------------------------------
    def bytes_per_page_all_layers(self) -> int:
        elt = torch.empty((), dtype=self.model_dtype).element_size()
        return self.num_layers * 2 * self.page_size * self.head_dim * elt

    def get_num_pages(self) -> int:
        if self.device.type != "cuda":
            raise RuntimeError("num_pages must be given explicitly when the cache is not on a GPU")
        free, total = torch.cuda.mem_get_info(self.device)
        budget = int(total * self.gpu_frac * 0.9) - (total - free)
        tables = self.num_layers * (self.max_num_batches + 1) * self.num_kv_heads * (self.max_pages_per_batch + 2) * 4
        budget -= tables
        if budget <= 0:
            raise RuntimeError("Insufficient memory for the KV cache: raise gpu_memory_utilization or lower max_num_seqs")
        return max(1, budget // self.bytes_per_page_all_layers())

    def init_cache(self, model) -> None:
        if self.num_pages is None:
            self.num_pages = self.get_num_pages()
        self.paged_cache = PagedKVCache(
            num_layers=self.num_layers, H_kv=self.num_kv_heads, head_dim=self.head_dim, page_size=self.page_size,
            num_pages=int(self.num_pages), max_num_batches=self.max_num_batches, device=self.device,
            dtype=self.model_dtype, max_logical_pages_per_head=int(self.max_pages_per_batch))
        if self.max_batched_tokens is None:
            self.max_batched_tokens = self.estimate_max_batched_tokens()
        for layer_index, attn in enumerate(attention_modules(model)):
            attn.k_cache, attn.v_cache, attn.page_table, attn.bh_seq_lens = self.paged_cache.layer_slices(layer_index)
            attn.page_size = self.page_size

    def estimate_max_batched_tokens(self) -> int:
        """Tokens one packed prefill may hold: what the cache can take for one kv-head's worth of pages, rounded down to
        a page (reference :141-160; the activation-memory leg of that estimate needs a warm-up pass and is left to the
        caller's `max_batched_tokens` override)."""
        in_cache = (int(self.num_pages) * self.page_size) // self.num_kv_heads
        return max(self.page_size, (min(in_cache, self.max_model_len) // self.page_size) * self.page_size)

    # ---- rows and pages ------------------------------------------------------------------------------------------
    def allocate_sequences(self, seq_ids: Seq[int], max_positions: Seq[int]):
        rows: List[int] = []
        for seq_id, need in zip(seq_ids, max_positions):
            if seq_id not in self.seq_id_to_batch:
                row = self.paged_cache.new_batch()
                if row is None:
                    logger.warning("Failed to allocate batch!")
                    return False, None
                self.seq_id_to_batch[seq_id] = int(row)
            row = self.seq_id_to_batch[seq_id]
            rows.append(row)
            status = self.paged_cache.reserve_tokens(row, int(need))
            if status != KVAllocationStatus.SUCCESS:
                logger.warning(f"Failed to allocate pages ({status})!")
                return False, None
        return True, torch.as_tensor(rows, dtype=torch.int32, device=self.device)

    def free_sequences(self, seq_ids: Iterable[int]) -> None:
        for seq_id in seq_ids:
            row = self.seq_id_to_batch.pop(seq_id, None)
            if row is not None:
                self.paged_cache.free_batch(row)

    def reclaim_pages(self, seq_ids_to_reclaim: Iterable[int], future_reserved_buffer) -> int:
        freed = 0
        for i, seq_id in enumerate(seq_ids_to_reclaim):
            freed += self.paged_cache.reclaim_pages(self.seq_id_to_batch[seq_id], int(future_reserved_buffer[i]))
        return freed

    @property
    def num_free_batches(self) -> int:
        return len(self.paged_cache.free_batches)

    @property
    def num_free_pages(self) -> int:
        return min(len(pool) for pool in self.paged_cache.free_pages)
