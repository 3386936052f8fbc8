"""One request travelling through the engine.

Same record as `compactor_vllm/utils/sequence.py:9-83`: prompt / completion token ids, per-request sampling and
compression parameters, a status, a process-unique `seq_id`, and the running token count the scheduler's throughput
figure is built from.
"""
from __future__ import annotations

import enum
import itertools
from dataclasses import dataclass, field
from typing import List

from ..compression.compression_config import SequenceCompressionParams
from ..config.sampling_params import SamplingParams

SequenceStatus = enum.Enum("SequenceStatus", ["WAITING", "RUNNING", "FINISHED"], module=__name__)

_ids = itertools.count()


@dataclass
class Sequence:
    prompt_token_ids: List[int]
    completion_token_ids: List[int] = field(default_factory=list)
    sampling_params: SamplingParams = field(default_factory=SamplingParams)
    compression_params: SequenceCompressionParams = field(default_factory=SequenceCompressionParams)
    status: SequenceStatus = SequenceStatus.WAITING
    seq_id: int = field(default_factory=lambda: next(_ids), init=False)
    num_tokens_processed: int = 0

    # ---- sizes -------------------------------------------------------------------------------------------------
    @property
    def prompt_len(self) -> int:
        return len(self.prompt_token_ids)

    num_prompt_tokens = prompt_len

    @property
    def completion_len(self) -> int:
        return len(self.completion_token_ids)

    num_generated_tokens = completion_len

    # ---- updates -----------------------------------------------------------------------------------------------
    def add_new_token(self, token_id: int) -> None:
        """The first generated token also accounts for the prompt having been processed (reference :44-48)."""
        if not self.completion_token_ids:
            self.num_tokens_processed += self.prompt_len
        self.completion_token_ids.append(int(token_id))
        self.num_tokens_processed += 1

    def tokens_to_retain_per_layer(self, num_kv_heads: int) -> int:
        return max(1, int(self.compression_params.compression_ratio * self.prompt_len * num_kv_heads))

    # ---- pickling (spawned workers): everything but the id counter ------------------------------------------
    def __getstate__(self):
        return {k: (list(v) if isinstance(v, list) else v) for k, v in self.__dict__.items()}

    def __setstate__(self, state):
        self.__dict__.update(state)
