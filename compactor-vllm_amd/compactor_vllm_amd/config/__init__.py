from .constants import RESERVED_BATCH
from .engine_config import AttentionBackend, LLMConfig
from .sampling_params import SamplingParams

__all__ = ["RESERVED_BATCH", "AttentionBackend", "LLMConfig", "SamplingParams"]
