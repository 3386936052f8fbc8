"""State of a chunked prefill: a prompt longer than one prefill launch may hold is fed through
the model in chunks over its own cached prefix (the "cached prefix || appended block" form of
`attention/sparse_varlen_kernel.py:362-401`), every chunk is written to the cache uncompressed, the per-layer scoring
state is carried across chunks, and compression is applied once, in place, after the last chunk.

Compactor: the pre-RoPE leverage scores are local to 512-token chunks and the post-RoPE attention mass to 128-token
chunks, so with chunk boundaries at multiples of 512 the per-chunk kernels produce exactly the values of a one-shot
prefill; only the per-sequence z-score + blend + protected fill needs the whole sequence and runs on the last chunk.
SnapKV: only the last `w` queries score, over all keys - the keys of the earlier chunks are read back from the cache.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import torch


@dataclass
class ChunkedPrefillState:
    """Lives as long as one sequence's chunked prefill.  `attn` = the model's Attention modules (the scoring hooks are
    not told which layer calls them; they take the next layer in call order)."""
    total_len: int
    num_kv_heads: int
    attn: List[Any]
    pre: Dict[int, torch.Tensor] = field(default_factory=dict)   # layer -> [total_len, HKV] model dtype
    mass: Dict[int, torch.Tensor] = field(default_factory=dict)  # layer -> [total_len, HKV] fp32
    cursor: int = 0

    def begin_chunk(self) -> None:
        self.cursor = 0

    def next_layer(self) -> int:
        li = self.cursor
        self.cursor += 1
        return li

    def buffers(self, li: int, like_pre: Optional[torch.Tensor], device):
        if li not in self.mass:
            self.mass[li] = torch.empty((self.total_len, self.num_kv_heads), dtype=torch.float32, device=device)
            if like_pre is not None:
                self.pre[li] = torch.empty((self.total_len, self.num_kv_heads), dtype=like_pre.dtype, device=device)
        return self.pre.get(li), self.mass[li]


@dataclass
class PrefillChunk:
    start: int        # tokens of the sequence already cached = index of this chunk's first token
    length: int
    is_last: bool
    state: ChunkedPrefillState

    @property
    def total_len(self) -> int:
        return self.state.total_len


def chunk_boundaries(total_len: int, chunk_tokens: int, align: int = 512) -> List[int]:
    """Chunk starts / end: [0, c, 2c, ..., total_len] with c = chunk_tokens rounded down to a multiple of `align`
    (>= align); a tail shorter than `align` is merged into the previous chunk (SnapKV's window and the leverage tail
    stay inside one chunk)."""
    c = max(align, (chunk_tokens // align) * align)
    cuts = list(range(0, total_len, c))
    if len(cuts) > 1 and total_len - cuts[-1] < align:
        cuts.pop()
    return cuts + [total_len]
