"""compactor_vllm_amd — MI355X-native drop-in for the compactor-vllm hot path.

Module paths mirror the reference package `compactor_vllm` for everything on the path:
`attention.*`, `compression.*`, `kv_cache.*`, `layers.attention`, `utils.{context,helpers}`, `config.*`.
The engine (`LLM`, scheduler, model runner, models) is out of scope: the reference's own
engine keeps calling these modules; `bench.py` carries the thin driver used for measurement.
"""
from .compression import BatchCompressionParams, CompressionMethod, SequenceCompressionParams
from .config import AttentionBackend, LLMConfig, SamplingParams

__all__ = [
    "LLMConfig",
    "SamplingParams",
    "AttentionBackend",
    "CompressionMethod",
    "BatchCompressionParams",
    "SequenceCompressionParams",
]
