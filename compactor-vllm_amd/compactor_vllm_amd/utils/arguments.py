"""Batch records the engine hands to the model runner.

Parity with `compactor_vllm/utils/arguments.py`: `PrefillBatchArguments` (:15-41) and `DecodeBatchArguments` (:317-386)
keep their field names; the formulas a user-visible result depends on are the reference's -
`tokens_to_retain` (:109-121, quirk Q1: protected tokens are inside the budget computed from L - first - last) and
`make_phi` (:81-86, N(0,1) * 1/sqrt(sketch) in the model dtype from a seeded device generator).  The reference packs
everything into two broadcast buffers for its tensor-parallel peers; this build runs one process per GPU with whole
sequences per process, so the records are built directly.
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence as Seq

import torch

from ..compression.compression_config import BatchCompressionParams, CompressionMethod
from .sequence import Sequence


def tokens_to_retain(ratio: float, prompt_len: int, first: int, last: int, num_kv_heads: int) -> int:
    """(token, head) pairs a sequence keeps per layer: max(round(ratio * (L - first - last) * HKV), 1)."""
    return max(int(round(ratio * (prompt_len - first - last) * num_kv_heads)), 1)


def make_phi(head_dim: int, sketch_dim: int, dtype: torch.dtype, device, seed: int = 42) -> torch.Tensor:
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.randn((head_dim, sketch_dim), device=device, generator=g).to(dtype) * (1 / math.sqrt(sketch_dim))


@dataclass
class PrefillBatchArguments:
    B: int
    N: int
    do_compression: bool
    compression_method: CompressionMethod
    compression_chunk_size: int
    seq_ids: torch.Tensor          # [B] int64 (host)
    input_ids: torch.Tensor        # [N] int64 (device)
    positions: torch.Tensor        # [N] int64 (device)
    cu_seqlens_q: torch.Tensor     # [B+1] int32 (device)
    cu_seqlens_k: torch.Tensor
    max_seqlen_q: int
    max_seqlen_k: int
    batch_tokens_to_retain: Optional[torch.Tensor]  # [B] int32 (device)
    max_tokens_to_retain: Optional[int]
    protected_first: Optional[List[int]]
    protected_last: Optional[List[int]]
    PHI: Optional[torch.Tensor]
    context_lens: torch.Tensor     # [B] int32 (host): prompt lengths
    max_new_tokens: torch.Tensor   # [B] int64 (host)


def build_prefill_args(seqs: Seq[Sequence], params: BatchCompressionParams, num_kv_heads: int, PHI: Optional[torch.Tensor],
                       device) -> PrefillBatchArguments:
    lens = [s.prompt_len for s in seqs]
    first = [s.compression_params.protected_first_tokens for s in seqs]
    last = [s.compression_params.protected_last_tokens for s in seqs]
    do_compression = (any(s.compression_params.compression_ratio < 1.0 for s in seqs)
                      and params.compression_method != CompressionMethod.NONE)
    retain = [tokens_to_retain(s.compression_params.compression_ratio, L, f, l, num_kv_heads)
              for s, L, f, l in zip(seqs, lens, first, last)]
    cu = torch.tensor(list(itertools.accumulate(lens, initial=0)), dtype=torch.int32)
    ids = torch.tensor([t for s in seqs for t in s.prompt_token_ids], dtype=torch.int64)
    pos = torch.cat([torch.arange(L, dtype=torch.int64) for L in lens])
    cu_dev = cu.to(device, non_blocking=True)
    return PrefillBatchArguments(
        B=len(seqs), N=sum(lens), do_compression=do_compression, compression_method=params.compression_method,
        compression_chunk_size=params.chunk_size if params.do_chunked_compression else -1,
        seq_ids=torch.tensor([s.seq_id for s in seqs], dtype=torch.int64),
        input_ids=ids.to(device, non_blocking=True), positions=pos.to(device, non_blocking=True),
        cu_seqlens_q=cu_dev, cu_seqlens_k=cu_dev, max_seqlen_q=max(lens), max_seqlen_k=max(lens),
        batch_tokens_to_retain=torch.tensor(retain, dtype=torch.int32).to(device, non_blocking=True),
        max_tokens_to_retain=max(lens) * num_kv_heads, protected_first=first, protected_last=last, PHI=PHI,
        context_lens=cu.diff(), max_new_tokens=torch.tensor([s.sampling_params.max_new_tokens for s in seqs],
                                                            dtype=torch.int64))


@dataclass
class DecodeBatchOutput:
    output_tokens: Optional[torch.Tensor]
    output_seq_ids: Optional[torch.Tensor]


@dataclass
class DecodeBatchArguments:
    """The running decode batch: one entry per live sequence (device tensors unless noted).  `update` appends the
    sequences of a finished prefill; `desired_batch_occupancy` is the size at or below which the decode loop hands
    control back to the scheduler while prompts are still pending; `num_stashed_batches` counts the entries that were
    already in the batch when it was stashed."""

    batch_mapping: Optional[torch.Tensor] = None
    token_ids: Optional[torch.Tensor] = None
    positions: Optional[torch.Tensor] = None
    max_ctx_lens: Optional[torch.Tensor] = None
    seq_ids: Optional[torch.Tensor] = None
    temps: Optional[torch.Tensor] = None
    desired_batch_occupancy: int = -1
    num_stashed_batches: int = 0

    _FIELDS = ("batch_mapping", "token_ids", "positions", "max_ctx_lens", "seq_ids", "temps")

    def update(self, batch_mapping, token_ids, positions, max_ctx_lens, seq_ids, temps=None,
               desired_batch_occupancy: Optional[int] = None):
        new = dict(batch_mapping=batch_mapping, token_ids=token_ids, positions=positions, max_ctx_lens=max_ctx_lens,
                   seq_ids=seq_ids, temps=temps)
        for name in self._FIELDS:
            add, cur = new[name], getattr(self, name)
            if add is None:
                continue
            setattr(self, name, add.clone() if cur is None else torch.cat([cur, add], dim=0))
        if desired_batch_occupancy is not None:
            self.desired_batch_occupancy = desired_batch_occupancy
        return self

    def select(self, keep: torch.Tensor) -> None:
        """Keep the entries listed in `keep` (int64 indices, ascending)."""
        for name in self._FIELDS:
            cur = getattr(self, name)
            if cur is not None:
                setattr(self, name, cur.index_select(0, keep.to(cur.device)))

    def __len__(self) -> int:
        return 0 if self.token_ids is None else int(self.token_ids.shape[0])
