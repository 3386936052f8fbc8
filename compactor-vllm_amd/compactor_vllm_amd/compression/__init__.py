"""Entry points of the scoring side of the boundary.

Parity with `compactor_vllm/compression/__init__.py:13-29`: `COMPRESSION_REGISTRY` maps a `CompressionMethod` to the
class whose static `pre_rope_scoring` / `post_rope_scoring` produce the [N, HKV] score tensor, and the two
`apply_*_compression` hooks are what the model code calls around RoPE.
"""
from . import common as _common
from . import compactor as _compactor
from . import snapkv as _snapkv
from .compression_config import BatchCompressionParams, CompressionMethod, SequenceCompressionParams

__all__ = ["apply_prerope_compression", "apply_postrope_compression", "CompressionMethod", "BatchCompressionParams",
           "SequenceCompressionParams", "COMPRESSION_REGISTRY"]

COMPRESSION_REGISTRY: dict = {}
for _method, _impl in ((CompressionMethod.NONE, _common.NoCompression),
                       (CompressionMethod.SNAPKV, _snapkv.SnapKVCompression),
                       (CompressionMethod.COMPACTOR, _compactor.CompactorCompression)):
    COMPRESSION_REGISTRY[_method] = _impl


def _scorer(context):
    return COMPRESSION_REGISTRY[context.compression_context.compression_method]


def apply_prerope_compression(q, k, v, context):
    """Scores that need the PRE-RoPE projections (Compactor's leverage scores); None for the other methods."""
    return _scorer(context).pre_rope_scoring(q, k, v, context=context)


def apply_postrope_compression(q, k, v, prerope_scores, context):
    """Final per-(token, kv-head) scores from the rotated q / k, blended with the pre-RoPE ones where the method
    has them."""
    return _scorer(context).post_rope_scoring(q, k, v, prerope_scores, context=context)
