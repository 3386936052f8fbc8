"""compactor_vllm_amd — MI355X-native drop-in for the compactor-vllm hot path.

Module paths mirror the reference package `compactor_vllm` for everything on the path:
`attention.*`, `compression.*`, `kv_cache.*`, `layers.attention`, `utils.{context,helpers}`, `config.*`.
The engine loop (`LLM`, scheduler, KV-cache manager, model runner) lives in `core.*` and drives any
model object that follows the small protocol of `core.model_runner`; weights, tokenizers and the model zoo stay out of
scope - `bench.py` supplies a random-weight model of the benchmark's shape.
"""
from .compression import BatchCompressionParams, CompressionMethod, SequenceCompressionParams
from .config import AttentionBackend, LLMConfig, SamplingParams
from .core.llm_engine import LLM, LLMEngine

__all__ = [
    "LLM",
    "LLMEngine",
    "LLMConfig",
    "SamplingParams",
    "AttentionBackend",
    "CompressionMethod",
    "BatchCompressionParams",
    "SequenceCompressionParams",
]
