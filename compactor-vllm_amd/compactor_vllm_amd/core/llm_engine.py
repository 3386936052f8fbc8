"""`LLM` / `LLMEngine`: the user-facing entry point (`compactor_vllm/core/llm_engine.py:34-294`, exported as `LLM` by
`compactor_vllm/__init__.py`).

Same methods and argument meaning: `generate(prompts, sampling_params, batch_compression_params, *,
per_sequence_compression_params, tokenizer_kwargs, detokenizer_kwargs, return_sequences)`, `generate_chat`,
`generate_from_sequences`, `tokenize_prompt`, `detokenize_prompt`; a prompt whose protected ranges cover it entirely has
its ratio forced to 1.0 (`:144-145`).  What differs: the reference downloads weights and a tokenizer by model name; this
build is handed the model object (anything `ModelRunner` accepts) and, optionally, a tokenizer - without one, prompts
must be token-id lists and the results are token-id lists.
"""
from __future__ import annotations

from typing import Any, List, Optional, Union

from ..compression.compression_config import BatchCompressionParams, SequenceCompressionParams
from ..config.engine_config import LLMConfig
from ..config.sampling_params import SamplingParams
from ..utils.sequence import Sequence
from .model_runner import ModelRunner

PromptLike = Union[str, List[int]]


class LLMEngine:
    def __init__(self, config: LLMConfig, model=None, tokenizer=None, *, device=None, num_pages: Optional[int] = None,
                 max_batched_tokens: Optional[int] = None):
        if model is None:
            raise NotImplementedError(
                "pass the model object: loading weights by name (reference llm_engine.py:41-48, models/*.py) is outside "
                "the hot path this package rebuilds")
        self.config = config
        self.tokenizer = tokenizer
        if self.config.eos == -1 and tokenizer is not None and getattr(tokenizer, "eos_token_id", None) is not None:
            self.config.eos = tokenizer.eos_token_id
        self.master_model_runner = ModelRunner(config, model, device, num_pages=num_pages,
                                               max_batched_tokens=max_batched_tokens)

    # ---- text <-> ids --------------------------------------------------------------------------------------------
    def tokenize_prompt(self, prompt: PromptLike, **tokenizer_kwargs) -> List[int]:
        if isinstance(prompt, str):
            if self.tokenizer is None:
                raise ValueError("string prompts need a tokenizer; pass token-id lists instead")
            return self.tokenizer(prompt, **tokenizer_kwargs)["input_ids"]
        return list(prompt)

    def detokenize_prompt(self, sequences: List[Sequence], **detokenizer_kwargs):
        ids = [s.completion_token_ids for s in sequences]
        if self.tokenizer is None:
            return ids
        return self.tokenizer.batch_decode(ids, **detokenizer_kwargs)

    # ---- requests ------------------------------------------------------------------------------------------------
    def _build_sequences(self, prompts, sampling_params, per_sequence_compression_params=None,
                         tokenizer_kwargs: Optional[dict] = None) -> List[Sequence]:
        tokenizer_kwargs = tokenizer_kwargs or {}
        if not isinstance(prompts, list) or (prompts and isinstance(prompts[0], int)):
            prompts = [prompts]
        n = len(prompts)
        sp = [sampling_params] * n if isinstance(sampling_params, SamplingParams) else list(sampling_params)
        assert len(sp) == n, "sampling_params list must match prompts length"
        if per_sequence_compression_params is None:
            cp = [SequenceCompressionParams(1.0) for _ in range(n)]
        elif isinstance(per_sequence_compression_params, SequenceCompressionParams):
            cp = [per_sequence_compression_params] * n
        else:
            cp = list(per_sequence_compression_params)
            assert len(cp) == n, "per_sequence_compression_params list must match prompts length"
        seqs = []
        for prompt, s, c in zip(prompts, sp, cp):
            ids = self.tokenize_prompt(prompt, **tokenizer_kwargs)
            if c.protected_first_tokens + c.protected_last_tokens >= len(ids):
                c.compression_ratio = 1.0
            seqs.append(Sequence(prompt_token_ids=ids, sampling_params=s, compression_params=c))
        return seqs

    def generate(self, prompts, sampling_params, batch_compression_params: Optional[BatchCompressionParams] = None, *,
                 per_sequence_compression_params=None, tokenizer_kwargs: Optional[dict] = None,
                 detokenizer_kwargs: Optional[dict] = None, return_sequences: bool = False):
        seqs = self._build_sequences(prompts, sampling_params, per_sequence_compression_params, tokenizer_kwargs)
        self.master_model_runner.generate(seqs, batch_compression_params)
        out = self.detokenize_prompt(seqs, **(detokenizer_kwargs or {}))
        return (out, seqs) if return_sequences else out

    def generate_chat(self, messages_batch: List[List[dict]], sampling_params, batch_compression_params,
                      per_sequence_compression_params=None, *, tokenizer_kwargs: Optional[dict] = None,
                      detokenizer_kwargs: Optional[dict] = None, return_sequences: bool = False):
        if self.tokenizer is None:
            raise ValueError("generate_chat needs a tokenizer with a chat template")
        ids = [self.tokenizer.apply_chat_template(m, tokenize=True, **(tokenizer_kwargs or {})) for m in messages_batch]
        return self.generate(ids, sampling_params, batch_compression_params,
                             per_sequence_compression_params=per_sequence_compression_params,
                             detokenizer_kwargs=detokenizer_kwargs, return_sequences=return_sequences)

    def cache_prefix(self, prompt: PromptLike, batch_compression_params: Optional[BatchCompressionParams] = None, **tokenizer_kwargs) -> int:
        """Register the first `len // 512 * 512` tokens of `prompt` as a reusable prefix (extension; the reference lists
        prefix caching as "coming soon", README.md:29-30): later `generate` calls with the same
        `batch_compression_params` method whose prompts start with those tokens prefill only their suffix.  Returns the
        number of tokens registered."""
        return self.master_model_runner.cache_prefix(self.tokenize_prompt(prompt, **tokenizer_kwargs), batch_compression_params)

    def generate_from_sequences(self, seqs: List[Sequence], batch_compression_params: BatchCompressionParams):
        self.master_model_runner.generate(seqs, batch_compression_params)
        return seqs

    @property
    def last_throughput(self) -> float:
        """tokens/s of the latest `generate` as the reference's scheduler defines it (scheduler.py:203-205)."""
        s = self.master_model_runner.last_scheduler
        return 0.0 if s is None else s.throughput()


LLM = LLMEngine
