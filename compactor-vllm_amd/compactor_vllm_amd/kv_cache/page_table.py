"""Paged KV cache: the device buffers in the reference's block layout plus the host-side page bookkeeping.

Public surface as `compactor_vllm/kv_cache/page_table.py:28-313` (class and status names, constructor arguments,
tensor attributes, `new_batch / reserve_tokens / reclaim_pages / free_batch / layer_slices`, return values) and
`write_page_table.scatter_to_page_table` (:6-95).  The bookkeeping itself is organised differently: one `_LayerPool`
per layer owns that layer's free list (a min-heap, so the lowest page ids are handed out first like upstream) and the
pages every batch row holds; page-table updates are plain torch indexing (host-driven, off the timed kernel path).

Device layout (bit-compatible with the reference kernels' addressing):
  kv_cache     [2, L, n_pages * page_size, head_dim]      K = 0, V = 1; a page = page_size rows of ONE kv-head
  page_table   [L, max_num_batches + 1, H_kv, max_pages_per_head] int32      batch row 0 = RESERVED_BATCH
  bh_seq_lens, bh_num_pages   [L, max_num_batches + 1, H_kv] int32
"""
from __future__ import annotations

import enum
import heapq
from typing import Dict, Iterable, List, Optional, Set, Union

import torch

from ..config.constants import RESERVED_BATCH

KVAllocationStatus = enum.Enum(
    "KVAllocationStatus",
    ["EXCEEDS_MAX_SEQUENCE_LENGTH", "EXCEEDS_CURRENTLY_AVAILABLE_PAGES", "EXCEEDS_MAX_NUM_BATCHES", "SUCCESS"],
    module=__name__,
)


def cdiv(a, b):
    """ceil(a / b) for ints and integer tensors"""
    return -(-a // b)


def next_multiple(a, b):
    return b * cdiv(a, b)


class _LayerPool:
    """Free pages of one layer + which pages each batch row owns in it."""

    def __init__(self, n_pages: int):
        self.free: List[int] = list(range(n_pages))  # already a valid min-heap
        self.owned: Dict[int, Set[int]] = {}

    def __len__(self) -> int:  # number of free pages (tests and callers use len(cache.free_pages[layer]))
        return len(self.free)

    def take(self, owner: int, count: int) -> List[int]:
        got = [heapq.heappop(self.free) for _ in range(count)]
        self.owned.setdefault(owner, set()).update(got)
        return got

    def give_back(self, owner: int, pages: Iterable[int]) -> None:
        mine = self.owned.get(owner, set())
        for pg in pages:
            mine.remove(pg)
            heapq.heappush(self.free, int(pg))

    def release(self, owner: int) -> None:
        for pg in self.owned.pop(owner, ()):
            heapq.heappush(self.free, int(pg))


def scatter_to_page_table(add_pages, new_phys_pages, curr_pages, page_table, max_pages_per_head: int):
    """page_table[l, h, curr_pages[l,h] + j] = j-th new page of (l, h); `new_phys_pages` lists the new pages of all
    (l, h) in row-major order, add_pages[l, h] of them each (reference write_page_table.py:6-95)."""
    n_l, n_h = add_pages.shape
    if n_l * n_h == 0 or new_phys_pages.numel() == 0:
        return
    counts = add_pages.reshape(-1).long()
    first_slot = curr_pages.reshape(-1).long()
    cell = torch.repeat_interleave(torch.arange(n_l * n_h, device=counts.device), counts)  # (l, h) of every new page
    rank_in_cell = torch.arange(cell.numel(), device=counts.device) - (counts.cumsum(0) - counts)[cell]
    slot = first_slot[cell] + rank_in_cell
    keep = slot < max_pages_per_head
    layer_of, head_of = torch.div(cell, n_h, rounding_mode="floor"), cell % n_h
    page_table[layer_of[keep], head_of[keep], slot[keep]] = new_phys_pages.to(page_table.dtype)[keep]


class PagedKVCache(torch.nn.Module):
    def __init__(self, num_layers: int, max_logical_pages_per_head: int, num_pages: int, page_size: int, H_kv: int,
                 head_dim: int, max_num_batches: int, dtype: torch.dtype,
                 device: Union[str, torch.device, int] = "cuda"):
        super().__init__()
        rows = max_num_batches + 1  # + the reserved padding row
        self.num_layers, self.n_pages, self.head_dim = num_layers, num_pages, head_dim
        self.page_size: int = int(page_size)
        self.H_kv = int(H_kv)
        self.max_pages_per_head = max_logical_pages_per_head
        self.max_num_batches = rows
        self.kv_cache = torch.empty((2, num_layers, num_pages * self.page_size, head_dim), dtype=dtype, device=device)
        per_head = dict(device=device, dtype=torch.int32)
        self.page_table = torch.zeros((num_layers, rows, self.H_kv, max_logical_pages_per_head), **per_head)
        self.bh_seq_lens = torch.zeros((num_layers, rows, self.H_kv), **per_head)
        self.bh_num_pages = torch.zeros((num_layers, rows, self.H_kv), **per_head)
        self.free_pages: List[_LayerPool] = [_LayerPool(num_pages) for _ in range(num_layers)]
        # rows are handed out lowest first; RESERVED_BATCH is never handed out
        self.free_batches: List[int] = [r for r in range(rows - 1, -1, -1) if r != RESERVED_BATCH]

    # ---- batch rows ---------------------------------------------------------------------------------------------
    def new_batch(self) -> Optional[int]:
        """A free row of the tables, or None when no row is left or some layer cannot even give one page per head."""
        if not self.free_batches or any(len(pool) < self.H_kv for pool in self.free_pages):
            return None
        return self.free_batches.pop()

    def free_batch(self, batch_index: int) -> None:
        for pool in self.free_pages:
            pool.release(batch_index)
        self.bh_seq_lens[:, batch_index] = 0
        self.bh_num_pages[:, batch_index] = 0
        self.free_batches.append(batch_index)

    # ---- pages --------------------------------------------------------------------------------------------------
    def reserve_tokens(self, batch_index: int, add_tokens: int) -> KVAllocationStatus:
        """Room for `add_tokens` more tokens in EVERY (layer, head) of the row (reference :144-198): the same number
        of tokens everywhere, whatever the current per-head lengths are."""
        have = self.bh_num_pages[:, batch_index]                                   # [L, H] pages now
        want = cdiv(self.bh_seq_lens[:, batch_index] + add_tokens, self.page_size)  # [L, H] pages needed
        extra = (want - have).clamp_min(0)
        if not bool(extra.any()):
            return KVAllocationStatus.SUCCESS
        total = have + extra
        if bool((total > self.max_pages_per_head).any()):
            return KVAllocationStatus.EXCEEDS_MAX_SEQUENCE_LENGTH
        per_layer = extra.sum(dim=-1).tolist()
        if any(n > len(pool) for n, pool in zip(per_layer, self.free_pages)):
            return KVAllocationStatus.EXCEEDS_CURRENTLY_AVAILABLE_PAGES
        fresh: List[int] = []
        for n, pool in zip(per_layer, self.free_pages):
            fresh += pool.take(batch_index, int(n))
        scatter_to_page_table(extra, torch.tensor(fresh, dtype=torch.int32, device=self.page_table.device), have,
                              self.page_table[:, batch_index], self.max_pages_per_head)
        self.bh_num_pages[:, batch_index] = total.to(self.bh_num_pages.dtype)
        return KVAllocationStatus.SUCCESS

    def reclaim_pages(self, batch_index: int, future_reserve_tokens: int = 0):
        """After compression: give back the tail pages beyond ceil((len + future) / page_size) of every (layer, head);
        returns the bytes released (K + V) (reference :200-267)."""
        assert 0 <= batch_index < self.bh_seq_lens.shape[1]
        have = self.bh_num_pages[:, batch_index]
        keep = torch.minimum(cdiv(self.bh_seq_lens[:, batch_index] + future_reserve_tokens, self.page_size), have)
        logical = torch.arange(self.max_pages_per_head, device=have.device, dtype=torch.int32)
        surplus = (logical >= keep[..., None]) & (logical < have[..., None])      # [L, H, P]
        if not bool(surplus.any()):
            return 0
        released = 0
        rows = self.page_table[:, batch_index]
        for layer, pool in enumerate(self.free_pages):
            pages = rows[layer][surplus[layer]].tolist()
            pool.give_back(batch_index, pages)
            released += len(pages)
        self.bh_num_pages[:, batch_index] = keep
        return released * self.page_size * self.head_dim * self.kv_cache.element_size() * 2

    # ---- views --------------------------------------------------------------------------------------------------
    def layer_slices(self, layer: int):
        """(k_cache, v_cache, page_table, bh_seq_lens) of one layer: what an `Attention` module holds (:293-313)."""
        assert 0 <= layer < self.num_layers
        return self.kv_cache[0, layer], self.kv_cache[1, layer], self.page_table[layer], self.bh_seq_lens[layer]
