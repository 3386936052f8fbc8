"""Global paged KV cache: buffers with the reference's block layout + the host page allocator.

Mirror of `compactor_vllm/kv_cache/page_table.py:28-313` (same class, constructor arguments, attribute
names, methods and return values) and of `write_page_table.scatter_to_page_table` (:6-95), which is plain
torch indexing here (host-driven bookkeeping off the timed kernel path).

Layout (bit-compatible with the reference):
  kv_cache    [2, L, n_pages * page_size, head_dim]   (K = 0, V = 1)   one page = page_size rows of ONE kv-head
  page_table  [L, max_num_batches + 1, H_kv, max_pages_per_head] int32  (batch row 0 = RESERVED_BATCH)
  bh_seq_lens / bh_num_pages [L, max_num_batches + 1, H_kv] int32
"""
from __future__ import annotations

import heapq
from enum import Enum, auto
from typing import List, Optional, Union

import torch

from ..config.constants import RESERVED_BATCH


def cdiv(a, b):
    return (a + b - 1) // b


def next_multiple(a, b):
    return cdiv(a, b) * b


class KVAllocationStatus(Enum):
    EXCEEDS_MAX_SEQUENCE_LENGTH = auto()
    EXCEEDS_CURRENTLY_AVAILABLE_PAGES = auto()
    EXCEEDS_MAX_NUM_BATCHES = auto()
    SUCCESS = auto()


def scatter_to_page_table(add_pages, new_phys_pages, curr_pages, page_table, max_pages_per_head: int):
    """Append `add_pages[l,h]` new physical page ids (concatenated in (l,h) row-major order in
    `new_phys_pages`) at logical pages curr_pages[l,h].. of page_table[l,h,:] (reference :6-95)."""
    L, H = add_pages.shape
    if L == 0 or H == 0 or new_phys_pages.numel() == 0:
        return
    add_flat = add_pages.reshape(-1).to(torch.int64)
    curr_flat = curr_pages.reshape(-1).to(torch.int64)
    lh = torch.repeat_interleave(torch.arange(L * H, device=add_flat.device), add_flat)
    starts = torch.cumsum(add_flat, 0) - add_flat
    within = torch.arange(lh.numel(), device=add_flat.device) - starts[lh]
    lp = curr_flat[lh] + within
    ok = lp < max_pages_per_head
    page_table[(lh // H)[ok], (lh % H)[ok], lp[ok]] = new_phys_pages.to(page_table.dtype)[ok]


class PagedKVCache(torch.nn.Module):
    def __init__(
        self,
        num_layers: int,
        max_logical_pages_per_head: int,
        num_pages: int,
        page_size: int,
        H_kv: int,
        head_dim: int,
        max_num_batches: int,
        dtype: torch.dtype,
        device: Union[str, torch.device, int] = "cuda",
    ):
        super().__init__()
        self.n_pages = num_pages
        self.num_layers = num_layers
        self.page_size: int = int(page_size)
        self.H_kv = int(H_kv)
        self.max_pages_per_head = max_logical_pages_per_head
        max_num_batches += 1
        self.max_num_batches = max_num_batches
        self.head_dim = head_dim
        self.kv_cache = torch.empty((2, num_layers, num_pages * page_size, head_dim), dtype=dtype, device=device)
        self.page_table = torch.zeros(
            (num_layers, max_num_batches, H_kv, self.max_pages_per_head), device=device, dtype=torch.int32
        )
        self.bh_seq_lens = torch.zeros((num_layers, max_num_batches, H_kv), device=device, dtype=torch.int32)
        self.bh_num_pages = torch.zeros((num_layers, max_num_batches, H_kv), device=device, dtype=torch.int32)
        self.free_pages: List[List[int]] = [list(range(num_pages)) for _ in range(num_layers)]
        for fp in self.free_pages:
            heapq.heapify(fp)
        self.free_batches: List[int] = list(reversed(range(max_num_batches)))
        self.free_batches.remove(RESERVED_BATCH)
        self.pages_indices_per_batch: List[List[set]] = [
            [set() for _ in range(num_layers)] for _ in range(max_num_batches)
        ]

    def new_batch(self) -> Optional[int]:
        if self.free_batches and all(self.H_kv <= len(fp) for fp in self.free_pages):
            return self.free_batches.pop()
        return None

    def reserve_tokens(self, batch_index: int, add_tokens: int) -> KVAllocationStatus:
        """Make room for `add_tokens` more tokens in every (layer, head) of the batch row (reference :144-198)."""
        cur_bh_lens = self.bh_seq_lens[:, batch_index]
        curr_pages = self.bh_num_pages[:, batch_index]
        curr_cap_tokens = curr_pages * self.page_size
        need_tokens = cur_bh_lens + add_tokens
        if (need_tokens <= curr_cap_tokens).all():
            return KVAllocationStatus.SUCCESS
        missing = (need_tokens - curr_cap_tokens).clamp_min(0)
        add_pages = cdiv(missing, self.page_size)
        new_total_pages = curr_pages + add_pages
        if (new_total_pages > self.max_pages_per_head).any():
            return KVAllocationStatus.EXCEEDS_MAX_SEQUENCE_LENGTH
        pages_per_layer = add_pages.sum(dim=-1).tolist()
        for layer in range(self.num_layers):
            if pages_per_layer[layer] > len(self.free_pages[layer]):
                return KVAllocationStatus.EXCEEDS_CURRENTLY_AVAILABLE_PAGES
        new_phys: List[int] = []
        for layer in range(self.num_layers):
            pages = [heapq.heappop(self.free_pages[layer]) for _ in range(pages_per_layer[layer])]
            self.pages_indices_per_batch[batch_index][layer] |= set(pages)
            new_phys.extend(pages)
        new_phys_t = torch.tensor(new_phys, dtype=torch.int32, device=self.page_table.device)
        scatter_to_page_table(add_pages, new_phys_t, curr_pages, self.page_table[:, batch_index],
                              self.max_pages_per_head)
        self.bh_num_pages[:, batch_index, :] = new_total_pages.to(self.bh_num_pages.dtype)
        return KVAllocationStatus.SUCCESS

    def reclaim_pages(self, batch_index: int, future_reserve_tokens: int = 0):
        """Free the tail pages beyond ceil((len + future)/page) per (layer, head); returns ~bytes freed (K+V)
        (reference :200-267)."""
        Lnum, Bn, H = self.bh_seq_lens.shape
        assert 0 <= batch_index < Bn
        seq = self.bh_seq_lens[:, batch_index, :] + future_reserve_tokens
        alloc = self.bh_num_pages[:, batch_index, :]
        pt = self.page_table[:, batch_index, :, :].reshape(-1)
        used = torch.minimum(cdiv(seq, self.page_size), alloc)
        p = torch.arange(self.max_pages_per_head, device=pt.device, dtype=torch.int32).view(1, 1, -1)
        free_mask = (p < alloc.unsqueeze(-1)) & (p >= used.unsqueeze(-1))
        flat = free_mask.reshape(-1)
        if not bool(flat.any()):
            return 0
        idx = flat.nonzero(as_tuple=False).squeeze(-1)
        freed = pt[idx].tolist()
        layers = (idx // (H * self.max_pages_per_head)).tolist()
        self.bh_num_pages[:, batch_index, :] = used
        for page, layer in zip(freed, layers):
            self.pages_indices_per_batch[batch_index][layer].remove(page)
            heapq.heappush(self.free_pages[layer], page)
        return len(freed) * (self.page_size * self.head_dim * self.kv_cache.element_size()) * 2

    def _free_batch_layer(self, layer_index: int, batch_index: int) -> None:
        for phys in self.pages_indices_per_batch[batch_index][layer_index]:
            heapq.heappush(self.free_pages[layer_index], int(phys))
        self.pages_indices_per_batch[batch_index][layer_index] = set()

    def free_batch(self, batch_index: int) -> None:
        for layer in range(self.num_layers):
            self._free_batch_layer(layer, batch_index)
        self.bh_seq_lens[:, batch_index].zero_()
        self.bh_num_pages[:, batch_index].zero_()
        self.free_batches.append(batch_index)

    def layer_slices(self, layer: int):
        """(k, v, page_table, bh_seq_lens) views of one layer — what `Attention` modules hold (:293-313)."""
        assert 0 <= layer < self.num_layers
        return self.kv_cache[0, layer], self.kv_cache[1, layer], self.page_table[layer], self.bh_seq_lens[layer]
