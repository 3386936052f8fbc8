"""Execution loop of one engine process (one process per GPU; whole sequences per process).

What the reference's `ModelRunner` does for one rank (`compactor_vllm/core/model_runner.py`): owns the KV-cache manager
and the store stream (`:70-78`), runs packed prefills with the compression context installed (`:198-236`), keeps ONE
continuously batched decode batch alive across prefill waves with the reference's stash / occupancy policy
(`:266-338`), runs the decode loop with finished sequences dropped from the batch (`:391-470`), replays captured graphs
keyed by (batch bucket, context bucket) with `RESERVED_BATCH` padding rows (`:472-555`).  Built differently:

* the model is any object with `model(input_ids, positions) -> hidden`, `model.compute_logits(hidden) -> [B, vocab]`
  (last token of every sequence during prefill) and the per-layer `Attention` modules (see
  `memory_manager.attention_modules`); weights, tokenizer and tensor parallelism are out of scope;
* positions, length limits and sequence ids of the decode batch live on the HOST (they are deterministic), so a decode
  step costs no device->host round trip unless an EOS id is configured (the reference syncs on `masked_select` every
  step); sampled tokens stay on the device until the loop hands control back;
* HIP graphs are captured lazily, on first use of a bucket, with an all-`RESERVED_BATCH` mapping (the kernels skip such
  rows), so capturing never needs free cache rows; buckets are powers of two in the batch and
  {1K, 4K, 8K, 16K, 32K, 64K, 128K} in the context - the context bucket only bounds the decode kernel's split plan;
* the store stream is joined before page reclamation reads the per-head lengths (hazard H2 of SURVEY section 3.1).

Quirk kept (Q11): a sequence emits `max_new_tokens + 1` tokens - the one sampled from the prefill logits plus one per
decode step while `position < prompt_len + max_new_tokens` (`:408-411`).
"""
from __future__ import annotations

import logging
import os
from typing import Dict, List, Optional, Tuple

import torch

from ..compression.compression_config import BatchCompressionParams
from ..config.constants import RESERVED_BATCH
from ..config.engine_config import LLMConfig
from ..layers.sampler import Sampler
from ..utils.arguments import (DecodeBatchArguments, DecodeBatchOutput, PrefillBatchArguments, build_prefill_args,
                               make_phi)
from ..utils.chunked import ChunkedPrefillState, PrefillChunk, chunk_boundaries
from ..utils.context import CompressionContext, reset_context, set_context
from ..utils.sequence import Sequence
from .memory_manager import KVCacheManager, attention_modules
from .prefix_cache import PREFIX_ALIGN, PrefixCache, PrefixEntry
from .scheduler import Scheduler

logger = logging.getLogger(__name__)

CONTEXT_BUCKETS = (1024, 4096, 8192, 16384, 32768, 65536, 131072)


class _GraphSlot:
    __slots__ = ("graph", "input_ids", "positions", "batch_mapping", "logits")


class ModelRunner:
    def __init__(self, config: LLMConfig, model, device=None, *, num_pages: Optional[int] = None,
                 max_batched_tokens: Optional[int] = None, seed: int = 42):
        if config.tensor_parallel_size != 1:
            raise NotImplementedError("one process per GPU with whole sequences per process; tensor parallelism is out "
                                      "of scope (SURVEY section 2.2)")
        self.config = config
        self.model = model
        self.device = torch.device(device if device is not None else
                                   (f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cpu"))
        self.on_gpu = self.device.type == "cuda"
        self.enforce_eager = bool(config.enforce_eager) or not self.on_gpu
        self.max_num_batches = int(config.max_num_seqs)
        self.max_model_len = int(config.max_model_len)
        self.kv_manager = KVCacheManager(config, self.device, num_pages=num_pages, max_batched_tokens=max_batched_tokens)
        self.kv_manager.init_cache(model)
        self.max_batched_tokens = self.kv_manager.max_batched_tokens
        self.store_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.sampler = Sampler()
        hf = config.hf_config
        self.PHI = make_phi(self.kv_manager.head_dim, int(config.leverage_sketch_size), self.kv_manager.model_dtype,
                            self.device, seed)
        self.num_kv_heads = int(hf.num_key_value_heads)
        self.captured_graphs: Dict[Tuple[int, int], _GraphSlot] = {}
        # ONE capture stream for every bucket: per-stream scratch (the decode workspace) is then shared by all graphs
        # and its key can never be recycled.  Replays must stay on one stream at a time (they do: the engine is
        # single-threaded and replays on the current stream).
        self._capture_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.last_scheduler: Optional[Scheduler] = None
        self.prefix_cache = PrefixCache()
        self._batch_params: Optional[BatchCompressionParams] = None
        self.kv_manager.goes_alone = self._has_prefix_hit
        if self.on_gpu:
            self._probe_in_launch_merge()

    # ------------------------------------------------------------------------------------------------ prefill
    @torch.inference_mode()
    def run_prefill(self, a: PrefillBatchArguments, batch_mapping: torch.Tensor) -> torch.Tensor:
        assert a.B > 0 and a.N > 0
        cc = CompressionContext(
            compression_method=a.compression_method, compression_chunk_size=a.compression_chunk_size,
            batch_tokens_to_retain=a.batch_tokens_to_retain, max_tokens_to_retain=a.max_tokens_to_retain,
            context_lens=a.context_lens.tolist(), PHI=a.PHI, protected_first_tokens=a.protected_first,
            protected_last_tokens=a.protected_last)
        set_context(is_prefill=True, do_compression=a.do_compression, cu_seqlens_q=a.cu_seqlens_q,
                    cu_seqlens_k=a.cu_seqlens_k, max_seqlen_q=a.max_seqlen_q, max_seqlen_k=a.max_seqlen_k,
                    batch_mapping=batch_mapping, max_bh_len=0, compression_context=cc, STORE_STREAM=self.store_stream,
                    attention_backend=self.config.attention_backend)
        try:
            return self.model.compute_logits(self.model(a.input_ids, a.positions))
        finally:
            reset_context()

    @torch.inference_mode()
    def run_prefill_chunked(self, a: PrefillBatchArguments, batch_mapping: torch.Tensor, *, start: int = 0,
                            state: Optional[ChunkedPrefillState] = None, finish: bool = True) -> torch.Tensor:
        """One sequence whose prompt exceeds `max_batched_tokens`: chunks of a multiple of 512 tokens, each
        attending to [its own cached prefix || itself] and written to the cache uncompressed; scoring state carried in
        a `ChunkedPrefillState`; compression applied in place after the last chunk (layers/attention.py).  Returns the
        logits of the prompt's last token.  `start` / `state`: the first `start` tokens are in the cache already and
        `state` holds their scoring stash (prefix reuse); `finish=False`: no chunk is the last one (a prefix being
        registered: nothing is compressed, the state is the caller's to keep)."""
        assert a.B == 1
        total = int(a.context_lens[0])
        cuts = [start + c for c in chunk_boundaries(total - start, self.max_batched_tokens)]
        if state is None:
            state = ChunkedPrefillState(total, self.num_kv_heads, attention_modules(self.model))
        logits = None
        for ci in range(len(cuts) - 1):
            s0, s1 = cuts[ci], cuts[ci + 1]
            n = s1 - s0
            cu = torch.tensor([0, n], dtype=torch.int32, device=self.device)
            cc = CompressionContext(
                compression_method=a.compression_method, compression_chunk_size=a.compression_chunk_size,
                batch_tokens_to_retain=a.batch_tokens_to_retain, max_tokens_to_retain=a.max_tokens_to_retain,
                context_lens=[n], PHI=a.PHI, protected_first_tokens=a.protected_first,
                protected_last_tokens=a.protected_last)
            last = finish and ci == len(cuts) - 2
            state.begin_chunk()
            set_context(is_prefill=True, do_compression=a.do_compression, cu_seqlens_q=cu, cu_seqlens_k=cu,
                        max_seqlen_q=n, max_seqlen_k=n, batch_mapping=batch_mapping, max_bh_len=s0,
                        compression_context=cc, STORE_STREAM=self.store_stream,
                        attention_backend=self.config.attention_backend,
                        chunk=PrefillChunk(start=s0, length=n, is_last=last, state=state))
            try:
                hidden = self.model(a.input_ids[s0:s1], a.positions[s0:s1])
                if last:
                    logits = self.model.compute_logits(hidden)
            finally:
                reset_context()
            self._join_store_stream()  # the next chunk's attention reads this chunk's rows and lengths
        return logits

    # ------------------------------------------------------------------------------------------------ prefix reuse
    def _scoring_key(self, params: BatchCompressionParams):
        return params.compression_method, (params.chunk_size if params.do_chunked_compression else -1)

    @torch.inference_mode()
    def cache_prefix(self, token_ids: List[int], batch_compression_params: Optional[BatchCompressionParams] = None) -> int:
        """Register the first P = len // 512 * 512 tokens as a reusable prefix (core/prefix_cache.py): prefilled once,
        uncompressed, into a resident cache row, with the scoring state of `batch_compression_params`' method.  Returns
        P (0: nothing registered - too short, already known, or no room)."""
        from ..compression.compression_config import CompressionMethod
        from ..compression.compression_config import SequenceCompressionParams
        from ..config.sampling_params import SamplingParams

        params = batch_compression_params if batch_compression_params is not None else BatchCompressionParams()
        method, chunk = self._scoring_key(params)
        P = (len(token_ids) // PREFIX_ALIGN) * PREFIX_ALIGN
        if P == 0 or P + 1 > self.max_model_len:
            return 0
        toks = tuple(int(t) for t in token_ids[:P])
        if self.prefix_cache.lookup(list(toks) + [0], method, chunk, 1) is not None:
            return P
        owner = self.prefix_cache.next_owner_id()
        ok, rows = self.kv_manager.allocate_sequences([owner], [P])
        if not ok:
            self.kv_manager.free_sequences([owner])
            return 0
        # ratio < 1 only so that the scoring hooks run; nothing is ever evicted from a prefix row (finish=False)
        seq = Sequence(list(toks), sampling_params=SamplingParams(0.0, 1),
                       compression_params=SequenceCompressionParams(0.5 if method != CompressionMethod.NONE else 1.0, 0, 0))
        a = build_prefill_args([seq], params, self.num_kv_heads, self.PHI, self.device)
        state = ChunkedPrefillState(P, self.num_kv_heads, attention_modules(self.model))
        self.run_prefill_chunked(a, rows, state=state, finish=False)
        self._join_store_stream()
        torch.cuda.current_stream(self.device).synchronize() if self.on_gpu else None
        self.prefix_cache.add(PrefixEntry(tokens=toks, length=P, row=int(rows[0]), owner_id=owner, method=method,
                                          chunk_size=chunk, pre=dict(state.pre), mass=dict(state.mass)))
        return P

    def drop_prefixes(self) -> None:
        """Release every registered prefix's cache row."""
        for e in self.prefix_cache.pop_all():
            self.kv_manager.free_sequences([e.owner_id])

    def _min_suffix(self, params: BatchCompressionParams) -> int:
        from ..compression.compression_config import CompressionMethod

        return 32 if params.compression_method == CompressionMethod.SNAPKV else 1  # SnapKV's window sits in the suffix

    def _has_prefix_hit(self, seq: Sequence) -> bool:
        params = self._batch_params
        if params is None or not self.prefix_cache.entries:
            return False
        method, chunk = self._scoring_key(params)
        return self.prefix_cache.lookup(seq.prompt_token_ids, method, chunk, self._min_suffix(params)) is not None

    @torch.inference_mode()
    def run_prefill_with_prefix(self, a: PrefillBatchArguments, batch_mapping: torch.Tensor, entry: PrefixEntry) -> torch.Tensor:
        """The request's row receives a copy of the prefix's K/V pages (every layer, every head), its lengths start at
        P, and the suffix runs as the remaining chunk(s) of a chunked prefill whose scoring state starts from the
        prefix's stash."""
        cache = self.kv_manager.paged_cache
        P, PS = entry.length, cache.page_size
        npg = P // PS
        row = int(batch_mapping[0])
        L, H = cache.num_layers, cache.H_kv
        src = cache.page_table[:, entry.row, :, :npg].long()  # [L, H, npg] physical pages
        dst = cache.page_table[:, row, :, :npg].long()
        kv = cache.kv_cache.view(2, L, cache.n_pages, PS, cache.head_dim)
        li = torch.arange(L, device=src.device)[:, None, None].expand_as(src)
        kv[:, li, dst] = kv[:, li, src]
        cache.bh_seq_lens[:, row] = P
        total = int(a.context_lens[0])
        state = ChunkedPrefillState(total, self.num_kv_heads, attention_modules(self.model))
        for layer, t in entry.mass.items():
            state.mass[layer] = torch.empty((total, self.num_kv_heads), dtype=t.dtype, device=t.device)
            state.mass[layer][:P] = t[:P]
        for layer, t in entry.pre.items():
            state.pre[layer] = torch.empty((total, self.num_kv_heads), dtype=t.dtype, device=t.device)
            state.pre[layer][:P] = t[:P]
        entry.hits += 1
        return self.run_prefill_chunked(a, batch_mapping, start=P, state=state)

    def _join_store_stream(self) -> None:
        if self.store_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.store_stream)

    def _check_prefill_health(self) -> None:
        """Once per prefill wave, after the store stream has been joined and before the per-head lengths are trusted:
        the selection kernels' sticky error word (a bounded look-back wait that expired, cvllm_select_status).  The
        page reclamation that follows synchronises anyway (it reads the lengths on the host)."""
        if not self.on_gpu:
            return
        from ..compression.common import select_status

        if select_status() != 0:
            raise RuntimeError("KV selection failed: a slice of the ordered write timed out waiting for the slices "
                               "before it (cvllm_select_status); the retained sets of this prefill are incomplete")

    def _probe_in_launch_merge(self) -> None:
        """Once, at start-up: ONE synthetic decode-attention launch that fills the chip the way the engine's launches do
        (256 workgroups, in-launch split merge), then the status word.  A device whose CUs this process does not have to
        itself at that moment - a CU mask, another process on the same GPU - fails HERE, where nothing is lost: the
        process then uses the two-kernel merge from the first token on.  (Contention that begins later is still caught
        by `_check_decode_health`.)"""
        from ..attention import sparse_decode_kernel as dk

        if os.environ.get("CVLLM_MERGE_PROBE", "1") == "0":  # tests of the later check (`_check_decode_health`) skip it
            return
        HKV, D, PS, P = self.num_kv_heads, int(self.kv_manager.head_dim), 128, 8
        hq = int(self.config.hf_config.num_attention_heads)
        dt, dev = self.kv_manager.model_dtype, self.device
        kc = torch.zeros(HKV * P * PS, D, dtype=dt, device=dev)
        q = torch.zeros(1, hq, D, dtype=dt, device=dev)
        pt = torch.arange(HKV * P, dtype=torch.int32, device=dev).view(1, HKV, P)
        lens = torch.full((1, HKV), P * PS, dtype=torch.int32, device=dev)
        bm = torch.zeros(1, dtype=torch.int32, device=dev)
        dk.head_sparse_decode_attention(q, kc, kc, lens, pt, bm, HKV, PS, key_split=max(1, 256 // HKV))
        if dk.merge_status(dev) != 0:
            dk.set_merge_mode("two-kernel")
            import warnings

            warnings.warn("decode attention: the in-launch split merge timed out in the start-up probe (the GPU's CUs are "
                          "not this process's alone); using the two-kernel merge")

    def _check_decode_health(self) -> None:
        """After every decode loop (the loop has just copied its tokens to the host, so the device is idle): the
        in-launch split merge's error word of every decode workspace.  A bounded wait that expired means the launch's
        workgroups were not co-resident (CU mask, shared GPU): the process switches to the two-kernel merge for good,
        drops the graphs that captured the in-launch form, and raises - the tokens of this call are invalid."""
        if not self.on_gpu:
            return
        from ..attention.sparse_decode_kernel import merge_status, set_merge_mode

        if merge_status(self.device) != 0:
            set_merge_mode("two-kernel")
            self.captured_graphs.clear()
            raise RuntimeError("decode attention: an in-launch split merge timed out (cvllm_decode_merge_status): the "
                               "launch's workgroups were not co-resident.  Tokens generated by this call are invalid; "
                               "the process now uses the two-kernel merge")

    # ------------------------------------------------------------------------------------------------ generate
    @torch.inference_mode()
    def generate(self, all_sequences: List[Sequence], batch_compression_params: Optional[BatchCompressionParams] = None):
        params = batch_compression_params if batch_compression_params is not None else BatchCompressionParams()
        self._batch_params = params  # the scheduler asks `_has_prefix_hit` through the KV manager
        sched = Scheduler(all_sequences, self.kv_manager, use_tqdm=bool(self.config.show_progress_bar))
        self.last_scheduler = sched
        batch = DecodeBatchArguments()
        pending_out: List[Tuple[torch.Tensor, List[int]]] = []  # (tokens on device, seq ids) not yet written back
        try:
            while not sched.is_finished():
                seqs = sched.get_prefill_batch()
                if seqs:
                    ids = [s.seq_id for s in seqs]
                    sched.add_running_sequence_ids(ids, update_status=True)
                    args = build_prefill_args(seqs, params, self.num_kv_heads, self.PHI, self.device)
                    max_ctx = args.max_new_tokens + args.context_lens.to(torch.int64)
                    ok, rows = self.kv_manager.allocate_sequences(ids, max_ctx.tolist())
                    if not ok:
                        raise RuntimeError("failed to allocate pages for sequences")
                    temps = torch.tensor([s.sampling_params.temperature for s in seqs], dtype=torch.float32,
                                         device=self.device)
                    hit = None
                    if len(seqs) == 1 and self.prefix_cache.entries:
                        method, chunk = self._scoring_key(params)
                        hit = self.prefix_cache.lookup(seqs[0].prompt_token_ids, method, chunk, self._min_suffix(params))
                    if hit is not None:
                        logits = self.run_prefill_with_prefix(args, rows, hit)
                    elif len(seqs) == 1 and seqs[0].prompt_len > self.max_batched_tokens:
                        logits = self.run_prefill_chunked(args, rows)
                    else:
                        logits = self.run_prefill(args, rows)
                    tokens = self.sampler(logits, temps)
                    pending_out.append((tokens, ids))
                    # H2: the per-head lengths are written on the store stream
                    self._join_store_stream()
                    if args.do_compression:
                        self._check_prefill_health()
                    self.kv_manager.reclaim_pages(ids, args.max_new_tokens.tolist())
                    occupancy = int((len(batch) + len(ids)) * 0.66) if sched.any_pending_sequences() else -1
                    batch.update(rows, tokens, args.context_lens.to(torch.int64), max_ctx, args.seq_ids, temps, occupancy)
                    if sched.can_prefill_another_batch():
                        continue
                elif len(batch) == 0:
                    raise RuntimeError("a pending prompt does not fit the KV cache even with nothing else running")
                else:
                    # nothing fits right now: decode until at least one running sequence finishes and frees its pages
                    batch.desired_batch_occupancy = len(batch) - 1
                self._join_store_stream()
                out, batch = self.run_decode_loop(batch, pending_out)
                self._check_decode_health()
                pending_out = []
                finished = sched.get_finished_sequence_ids_from_unfinished(batch.seq_ids.tolist() if len(batch) else [])
                sched.record_finished_sequence_ids(finished, update_status=True)
                self.kv_manager.free_sequences(finished)
                sched.update_sequences(out.output_tokens.tolist(), out.output_seq_ids.tolist())
        finally:
            sched.close()
        return all_sequences

    # ------------------------------------------------------------------------------------------------ decode loop
    @torch.inference_mode()
    def run_decode_loop(self, batch: DecodeBatchArguments, pending_out=None):
        """Decode until the batch is empty or has shrunk to `desired_batch_occupancy`.  `positions`, `max_ctx_lens` and
        `seq_ids` of the batch are host tensors; `token_ids`, `batch_mapping`, `temps` live on the device."""
        eos = int(self.config.eos)
        tok_chunks: List[torch.Tensor] = []
        id_chunks: List[List[int]] = []
        for toks, ids in (pending_out or []):
            tok_chunks.append(toks)
            id_chunks.append(list(ids))
        pos_dev = batch.positions.to(self.device)
        while True:
            running = batch.positions < batch.max_ctx_lens  # host
            if eos >= 0:
                running &= batch.token_ids.cpu() != eos  # one round trip per step only when an EOS id is configured
            if not bool(running.all()):
                keep = running.nonzero().flatten()
                batch.positions, batch.max_ctx_lens = batch.positions[keep], batch.max_ctx_lens[keep]
                batch.seq_ids = batch.seq_ids[keep]
                kd = keep.to(self.device)
                batch.token_ids, batch.batch_mapping = batch.token_ids[kd], batch.batch_mapping[kd]
                batch.temps, pos_dev = batch.temps[kd], pos_dev[kd]
            n = len(batch)
            if n == 0 or n <= batch.desired_batch_occupancy:
                batch.num_stashed_batches = n
                break
            hint = int(batch.positions.max()) + 1
            if self.enforce_eager:
                # the same context buckets as the graphs: the split plan (hence the fp32 summation order) of a sequence
                # then does not depend on which other sequences share its batch
                set_context(is_prefill=False, do_compression=False, batch_mapping=batch.batch_mapping,
                            decode_len_hint=self._bucket(n, hint)[1])
                logits = self.model.compute_logits(self.model(batch.token_ids, pos_dev))
            else:
                logits = self.run_graph_decode(batch.token_ids, pos_dev, batch.batch_mapping, hint)
            batch.token_ids = self.sampler(logits, batch.temps)
            tok_chunks.append(batch.token_ids)
            id_chunks.append(batch.seq_ids.tolist())
            batch.positions = batch.positions + 1
            pos_dev = pos_dev + 1
        reset_context()
        if tok_chunks:
            out = DecodeBatchOutput(torch.cat(tok_chunks).cpu(), torch.tensor([i for c in id_chunks for i in c],
                                                                              dtype=torch.int64))
        else:
            out = DecodeBatchOutput(torch.empty(0, dtype=torch.int64), torch.empty(0, dtype=torch.int64))
        return out, batch

    # ------------------------------------------------------------------------------------------------ graphs
    def _bucket(self, n: int, hint: int) -> Tuple[int, int]:
        bs = 1
        while bs < n:
            bs <<= 1
        ctx = next((c for c in CONTEXT_BUCKETS if c >= hint), CONTEXT_BUCKETS[-1])
        return bs, min(ctx, max(self.max_model_len, CONTEXT_BUCKETS[0]))

    def capture_graph(self, bs: int, ctx_bucket: int) -> _GraphSlot:
        """Capture one decode step for `bs` rows.  Every row maps to RESERVED_BATCH during the warm-up and the capture:
        the kernels neither append nor read anything for such rows, so no cache rows are needed."""
        dev = self.device
        slot = _GraphSlot()
        slot.input_ids = torch.zeros(bs, dtype=torch.int64, device=dev)
        slot.positions = torch.zeros(bs, dtype=torch.int64, device=dev)
        slot.batch_mapping = torch.full((bs,), RESERVED_BATCH, dtype=torch.int32, device=dev)
        set_context(is_prefill=False, do_compression=False, batch_mapping=slot.batch_mapping, decode_len_hint=ctx_bucket)
        side = self._capture_stream
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):  # warm-up on the capture stream: lazy allocations (per-stream workspaces) happen here
                self.model.compute_logits(self.model(slot.input_ids, slot.positions))
            side.synchronize()
            slot.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(slot.graph, stream=side):
                slot.logits = self.model.compute_logits(self.model(slot.input_ids, slot.positions))
        torch.cuda.current_stream(dev).wait_stream(side)
        self.captured_graphs[(bs, ctx_bucket)] = slot
        return slot

    def run_graph_decode(self, input_ids, positions, batch_mapping, hint: int) -> torch.Tensor:
        n = input_ids.shape[0]
        key = self._bucket(n, hint)
        slot = self.captured_graphs.get(key) or self.capture_graph(*key)
        slot.input_ids[:n] = input_ids
        slot.positions[:n] = positions
        slot.batch_mapping.fill_(RESERVED_BATCH)
        slot.batch_mapping[:n] = batch_mapping
        slot.graph.replay()
        return slot.logits[:n]
