"""Engine configuration types (`compactor_vllm/config/engine_config.py:9-94`).

`LLMConfig` keeps every field and default of the reference.  The one behavioural difference: when
`hf_config` is None the reference calls `AutoConfig.from_pretrained(model)` (a network fetch); this
build only does so when `transformers` can resolve the name locally, and otherwise raises — always
pass a locally built `hf_config` (there is no network on the target machines).
"""
import os
from dataclasses import dataclass
from enum import Enum, auto
from typing import Any, Optional


class AttentionBackend(Enum):
    FLASH_ATTENTION = auto()
    COMPACTOR_TRITON = auto()  # name kept for drop-in compatibility: here it selects the HIP kernels


@dataclass
class LLMConfig:
    model: str
    path: Optional[str] = None
    nccl_port: Optional[int] = 1218
    max_num_seqs: int = 256
    max_model_len: int = 40960
    gpu_memory_utilization: float = 0.9
    tensor_parallel_size: int = 1
    enforce_eager: bool = False
    hf_config: Any | None = None
    eos: int = -1
    kvcache_page_size: int = 128
    leverage_sketch_size: int = 48
    attention_backend: AttentionBackend = AttentionBackend.COMPACTOR_TRITON
    show_progress_bar: bool = True

    def __post_init__(self):
        if self.path is not None and not os.path.isdir(self.path):
            raise NotADirectoryError(f"Engine config dir {self.path} does not exist")
        if self.tensor_parallel_size <= 0 or self.tensor_parallel_size > 8:
            raise ValueError("tensor_parallel_size must be >= 1 and <= 8")
        if self.hf_config is None:
            from transformers import AutoConfig

            self.hf_config = AutoConfig.from_pretrained(self.model, local_files_only=True)
        self.max_model_len = min(self.max_model_len, self.hf_config.max_position_embeddings)
