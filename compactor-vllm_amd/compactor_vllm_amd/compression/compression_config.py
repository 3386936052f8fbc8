"""User-facing compression parameters.

Names, attributes and defaults follow `compactor_vllm/compression/compression_config.py:8-44`:
`CompressionMethod` (COMPACTOR, SNAPKV, NONE - `auto()` values in that order), per-sequence ratio / protected ranges,
per-batch method and chunking (SnapKV needs every query of the window to see the whole sequence, so it switches
chunked scoring off with a warning, exactly like the reference).
"""
import dataclasses
import enum
import logging

_log = logging.getLogger(__name__)

CompressionMethod = enum.Enum("CompressionMethod", ["COMPACTOR", "SNAPKV", "NONE"], module=__name__)
CompressionMethod.__doc__ = "Which scoring rule ranks the (token, kv-head) pairs of a prompt."


@dataclasses.dataclass
class SequenceCompressionParams:
    """compression_ratio = share of the unprotected (token, head) pairs that stay; the first / last tokens of a
    prompt are always kept (attention sinks and the recent window)."""

    compression_ratio: float = 1.0
    protected_first_tokens: int = 16
    protected_last_tokens: int = 64


@dataclasses.dataclass
class BatchCompressionParams:
    compression_method: CompressionMethod = CompressionMethod.COMPACTOR
    do_chunked_compression: bool = True
    chunk_size: int = 512  # leverage scores are computed per chunk of this many tokens

    def __post_init__(self):
        wants_chunks = self.do_chunked_compression
        if wants_chunks and self.compression_method is CompressionMethod.SNAPKV:
            _log.warning("CompressionMethod.SNAPKV is not compatible with chunked compression. Disabling it.")
            self.do_chunked_compression = False
