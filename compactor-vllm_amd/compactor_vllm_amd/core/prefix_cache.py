"""Prefix reuse on the chunked-prefill machinery (second half; the reference only announces the
feature: README.md:29-30, :182 "prefix caching - coming soon"; what makes it possible is that its prefill kernel attends
over [cached prefix || appended block], attention/sparse_varlen_kernel.py:362-401).

A registered prefix is the first P tokens of some prompt (P a multiple of 512: the boundary at which Compactor's 512-token
leverage chunks and 128-token attention-mass chunks of the prefix equal those of any longer prompt) prefilled ONCE,
UNCOMPRESSED, into a cache row of its own that stays resident, together with the per-layer scoring state a chunked
prefill carries (`utils/chunked.ChunkedPrefillState`: raw leverage scores and attention mass of the prefix's tokens).

A later request whose prompt starts with those P tokens gets its own row, the prefix's K/V pages are COPIED into it
(device-to-device: a 32 K-token Llama-3-8B prefix is 4 GiB, about a millisecond - against 250 ms of prefill; sharing the
pages instead would not survive the request's own in-place compaction), its per-head lengths start at P, and only its
suffix goes through the model - as the remaining chunk(s) of a chunked prefill whose state starts from the stash.  The
request then ends exactly like a chunked prefill of the whole prompt with a chunk boundary at P: the same scores, the
same selection, the same compacted cache.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

PREFIX_ALIGN = 512


def _digest(tokens: Sequence[int]) -> bytes:
    return hashlib.blake2b(torch.tensor(list(tokens), dtype=torch.int64).numpy().tobytes(), digest_size=16).digest()


@dataclass
class PrefixEntry:
    tokens: Tuple[int, ...]
    length: int                      # P
    row: int                         # cache row that holds the prefix's K/V (all layers), uncompressed
    owner_id: int                    # synthetic sequence id under which the KV manager tracks the row
    method: object                   # CompressionMethod the scoring state was computed for
    chunk_size: int                  # compression chunk size of that state
    pre: Dict[int, torch.Tensor] = field(default_factory=dict)   # layer -> [P, HKV] (Compactor)
    mass: Dict[int, torch.Tensor] = field(default_factory=dict)  # layer -> [P, HKV] fp32 (Compactor)
    hits: int = 0


class PrefixCache:
    """Registered prefixes by (length, digest of the tokens, scoring method, compression chunk size); lookups try the
    longest registered length first."""

    def __init__(self):
        self.entries: Dict[tuple, PrefixEntry] = {}
        self._next_owner = -1

    def next_owner_id(self) -> int:
        oid = self._next_owner
        self._next_owner -= 1
        return oid

    def add(self, entry: PrefixEntry) -> None:
        self.entries[(entry.length, _digest(entry.tokens), entry.method, entry.chunk_size)] = entry

    def lengths(self) -> List[int]:
        return sorted({k[0] for k in self.entries}, reverse=True)

    def lookup(self, prompt: Sequence[int], method, chunk_size: int, min_suffix: int) -> Optional[PrefixEntry]:
        n = len(prompt)
        for P in self.lengths():
            if P + min_suffix > n:
                continue
            e = self.entries.get((P, _digest(prompt[:P]), method, chunk_size))
            if e is not None and tuple(prompt[:P]) == e.tokens:
                return e
        return None

    def pop_all(self) -> List[PrefixEntry]:
        out = list(self.entries.values())
        self.entries.clear()
        return out
