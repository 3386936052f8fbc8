"""Process-wide forward-pass state shared by the model code and the attention boundary.

API parity with `compactor_vllm/utils/context.py:9-83`: the two record types `CompressionContext` and `Context` expose
the same attribute names and defaults, and `get_context / set_context / reset_context` behave the same way
(`set_context` takes keywords only and REPLACES the whole record).  The records are generated from the field tables
below, so the list of names and defaults is data - one place to compare against the reference.
"""
from __future__ import annotations

from dataclasses import field, make_dataclass
from typing import Any, List, Optional

import torch

from ..compression.compression_config import CompressionMethod
from ..config.engine_config import AttentionBackend

# (attribute, annotation, default) - scoring inputs of one prefill batch (reference :9-27)
_COMPRESSION_FIELDS = (
    ("compression_method", CompressionMethod, CompressionMethod.COMPACTOR),
    ("compression_chunk_size", int, -1),                 # leverage-score chunk; -1 = whole sequence
    ("batch_tokens_to_retain", Optional[torch.Tensor], None),   # [B] int32, (token, head) pairs kept per sequence
    ("max_tokens_to_retain", int, 0),
    ("context_lens", Optional[List[int]], None),         # host copy of the prompt lengths
    ("PHI", Optional[torch.Tensor], None),               # [D, sketch] Gaussian sketch, model dtype
    ("protected_first_tokens", Optional[List[int]], None),
    ("protected_last_tokens", Optional[List[int]], None),
)

# state of the current forward pass (reference :30-52)
_CONTEXT_FIELDS = (
    ("is_prefill", bool, False),
    ("do_compression", bool, False),
    ("cu_seqlens_q", Optional[torch.Tensor], None),
    ("cu_seqlens_k", Optional[torch.Tensor], None),
    ("max_seqlen_q", int, 0),
    ("max_seqlen_k", int, 0),
    ("batch_mapping", Optional[torch.Tensor], None),     # [B] int32: local row -> row of the cache tables
    ("max_bh_len", int, 0),
    ("compression_context", Any, None),                  # CompressionContext | None
    ("STORE_STREAM", Any, None),                         # torch.cuda.Stream | None
    ("key_split", Optional[int], None),
    ("attention_backend", AttentionBackend, AttentionBackend.COMPACTOR_TRITON),
    # extension (not in the reference): an upper bound of the longest cached sequence of a decode batch, 0 = unknown.
    # Only tunes the decode kernel's split count (short contexts take fewer splits); never affects results.
    ("decode_len_hint", int, 0),
    # extension: utils.chunked.PrefillChunk while a long prompt is prefilled in chunks, else None
    ("chunk", Any, None),
)


def _record(name: str, table, doc: str):
    cls = make_dataclass(name, [(n, t, field(default=d)) for n, t, d in table])
    cls.__doc__ = doc
    cls.__module__ = __name__
    return cls


CompressionContext = _record("CompressionContext", _COMPRESSION_FIELDS,
                             "What the scoring kernels of one prefill batch need (method, chunking, retain counts, "
                             "protected ranges, sketch matrix).")
Context = _record("Context", _CONTEXT_FIELDS,
                  "Everything `Attention.forward` reads besides its tensors: phase, packed-batch geometry, the "
                  "batch-row mapping, the store stream, the optional compression record.")

_CONTEXT_NAMES = tuple(n for n, _, _ in _CONTEXT_FIELDS)
_CONTEXT = Context()


def get_context():
    """The record installed by the latest `set_context` (a default-constructed one before that)."""
    return _CONTEXT


def set_context(*, is_prefill, **state):
    """Install a NEW record: unnamed attributes fall back to their defaults, unknown names are an error."""
    unknown = set(state) - set(_CONTEXT_NAMES)
    if unknown:
        raise TypeError(f"set_context() got unexpected keyword argument(s): {sorted(unknown)}")
    global _CONTEXT
    _CONTEXT = Context(is_prefill=is_prefill, **state)


def reset_context():
    global _CONTEXT
    _CONTEXT = Context()
