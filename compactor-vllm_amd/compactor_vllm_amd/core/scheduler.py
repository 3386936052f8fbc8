"""Which waiting prompts are prefilled next, and the bookkeeping of who is running / finished.

Same policy and public methods as `compactor_vllm/core/scheduler.py:14-212`:
a pending sequence joins the next prefill batch if (a) the packed batch stays within `max_batched_tokens`, (b) a row of
the cache tables is free, and (c) its prompt + max_new_tokens worth of pages (per kv-head) is strictly less than the
pages still free in the tightest layer (`:65-108`); throughput = (prompt + generated tokens) / wall time since the
scheduler was created (`:203-205`) - the definition of the benchmark's tokens/s.
Pending sequences are visited in submission order (the reference iterates a `set` of small ints, which is the same
order in CPython).
"""
from __future__ import annotations

import time
from typing import Dict, Iterable, List

from ..utils.sequence import Sequence, SequenceStatus


def cdiv(a: int, b: int) -> int:
    return -(-a // b)


class Scheduler:
    def __init__(self, all_sequences: Iterable[Sequence], kv_manager, *, use_tqdm: bool = False):
        self.allseq_mapping: Dict[int, Sequence] = {s.seq_id: s for s in all_sequences}
        self.pending_sequence_ids: Dict[int, None] = dict.fromkeys(self.allseq_mapping)  # ordered set
        self.active_sequence_ids: set = set()
        self.finished_sequence_ids: set = set()
        self.manager = kv_manager
        self.use_tqdm = use_tqdm
        self.start_time = time.perf_counter()
        self.total_tokens_generated = 0
        self.total_tokens_input = 0
        self.pbar = None
        if use_tqdm:
            from tqdm import tqdm

            self.pbar = tqdm(total=len(self.pending_sequence_ids), desc="Completed Batches")

    # ---- admission ---------------------------------------------------------------------------------------------
    def get_prefill_batch(self) -> List[Sequence]:
        budget = self.manager.max_batched_tokens
        rows, pages = self.manager.num_free_batches, self.manager.num_free_pages
        page_size, heads = self.manager.page_size, self.manager.num_kv_heads
        chosen: List[Sequence] = []
        used = 0
        chunked_ok = bool(getattr(self.manager, "chunked_prefill", False))
        goes_alone = getattr(self.manager, "goes_alone", None)  # set by the runner: prompt starts with a registered prefix
        for seq_id in self.pending_sequence_ids:
            seq = self.allseq_mapping[seq_id]
            need = cdiv(seq.prompt_len + seq.sampling_params.max_new_tokens, page_size) * heads
            alone = seq.prompt_len > budget or (goes_alone is not None and goes_alone(seq))
            if chunked_ok and alone:
                # extension: a prompt longer than one prefill launch, or one that starts with a registered
                # prefix, goes alone and is prefilled in chunks (over its cached prefix)
                if not chosen and rows > 0 and need < pages:
                    return [seq]
                continue
            if seq.prompt_len + used <= budget and rows > 0 and need < pages:
                chosen.append(seq)
                used += seq.prompt_len
                pages -= need
                rows -= 1
        return chosen

    def can_prefill_another_batch(self) -> bool:
        return len(self.get_prefill_batch()) > 0

    # ---- state -------------------------------------------------------------------------------------------------
    def is_finished(self) -> bool:
        return not self.pending_sequence_ids and not self.active_sequence_ids

    def any_pending_sequences(self) -> bool:
        return bool(self.pending_sequence_ids)

    def add_running_sequence_ids(self, active_sequence_ids: Iterable[int], *, update_status: bool = False):
        ids = list(active_sequence_ids)
        self.active_sequence_ids.update(ids)
        for i in ids:
            self.pending_sequence_ids.pop(i, None)
        if update_status:
            for i in ids:
                self.allseq_mapping[i].status = SequenceStatus.RUNNING
                self.total_tokens_input += self.allseq_mapping[i].prompt_len

    def get_finished_sequence_ids_from_unfinished(self, unfinished_sequence_ids: Iterable[int]) -> set:
        return self.active_sequence_ids.difference(unfinished_sequence_ids)

    def record_finished_sequence_ids(self, finished_sequence_ids: Iterable[int], *, update_status: bool = False):
        ids = list(finished_sequence_ids)
        self.active_sequence_ids.difference_update(ids)
        self.finished_sequence_ids.update(ids)
        if update_status:
            for i in ids:
                self.allseq_mapping[i].status = SequenceStatus.FINISHED
                if self.pbar is not None:
                    self.pbar.update(1)

    def update_sequences(self, tokens: Iterable[int], seq_ids: Iterable[int]):
        for tok, seq_id in zip(tokens, seq_ids):
            self.allseq_mapping[seq_id].add_new_token(tok)
            self.total_tokens_generated += 1
        if self.pbar is not None:
            self.pbar.set_description(f"Throughput: {self.throughput():.2f} tok/s")

    def throughput(self) -> float:
        """(prompt tokens started + tokens generated) / seconds since construction (reference :203-205)."""
        return (self.total_tokens_generated + self.total_tokens_input) / (time.perf_counter() - self.start_time)

    def close(self):
        if self.pbar is not None:
            self.pbar.close()
