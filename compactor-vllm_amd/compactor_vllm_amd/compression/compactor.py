"""Compactor scoring: sketched leverage scores (pre-RoPE) + chunked non-causal attention mass
(post-RoPE), on MI355X HIP kernels.

Mirror of the reference module `compactor_vllm/compression/compactor.py`: `CompactorCompression`
(:16-59), `split_into_chunks` (:62-110), `approximate_leverage_scores` (:113-221) and
`non_causal_attn_scores` (:489-599) keep their names, argument order and defaults.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from .. import _lib
from ..utils.helpers import maybe_execute_in_stream
from .common import BaseCompressionMethod


class CompactorCompression(BaseCompressionMethod):
    chunk_size: int = 128

    @staticmethod
    def pre_rope_scoring(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, context) -> Optional[torch.Tensor]:
        cc = context.compression_context
        return maybe_execute_in_stream(
            approximate_leverage_scores,
            k,
            cc.context_lens,
            cc.PHI,
            normalize=True,
            chunk_size=cc.compression_chunk_size,
            STORE_STREAM=context.STORE_STREAM,
        )

    @staticmethod
    def post_rope_scoring(q, k, v, pre_rope_scores: torch.Tensor, context) -> Optional[torch.Tensor]:
        cc = context.compression_context
        if getattr(context, "chunk", None) is not None:
            return maybe_execute_in_stream(_chunked_post_rope, q, k, v, pre_rope_scores, context,
                                           STORE_STREAM=context.STORE_STREAM)
        # The reference runs this on the main stream while `pre_rope_scores` is still being produced on
        # STORE_STREAM with no dependency edge (hazard H1).  Here it runs on STORE_STREAM like every
        # other piece of the scoring -> select -> compaction chain: in order behind the pre-RoPE scores (no hazard,
        # no wait on the main stream) and under the prefill attention instead of in front of it.
        return maybe_execute_in_stream(
            non_causal_attn_scores,
            q,
            k,
            v,
            context.cu_seqlens_q,
            context.max_seqlen_q,
            chunk_size=CompactorCompression.chunk_size,
            sm_scale=1.0,
            normalize=True,
            accum_scores=pre_rope_scores,
            context_lens=cc.context_lens,
            protected_first_tokens=cc.protected_first_tokens,
            protected_last_tokens=cc.protected_last_tokens,
            accum_blending=0.5,
            STORE_STREAM=context.STORE_STREAM,
        )


def _chunked_post_rope(q, k, v, pre_chunk, context):
    """Chunked prefill (utils/chunked.py): stash this chunk's leverage scores and raw attention mass; on the last chunk
    finish exactly like the one-shot call - z-score of the mass over the WHOLE sequence + 0.5 * leverage, protected
    tokens <- +inf - and return the [total_len, HKV] scores (None before the last chunk: nothing is evicted yet)."""
    ch, cc = context.chunk, context.compression_context
    st = ch.state
    li = st.next_layer()
    mass = non_causal_attn_scores(q, k, v, context.cu_seqlens_q, context.max_seqlen_q,
                                  chunk_size=CompactorCompression.chunk_size, sm_scale=1.0, normalize=False)
    pre_all, mass_all = st.buffers(li, pre_chunk, q.device)
    mass_all[ch.start : ch.start + ch.length] = mass
    pre_all[ch.start : ch.start + ch.length] = pre_chunk
    if not ch.is_last:
        return None
    cu = _cu_from_lens([ch.total_len], q.device)
    rng = _protected_ranges([ch.total_len], cc.protected_first_tokens, cc.protected_last_tokens, ch.total_len)
    prot = torch.tensor(rng, dtype=torch.int32).to(q.device, non_blocking=True) if rng else None
    out = mass_all  # the state is not needed after the last chunk: normalise in place
    zscore_segments_(out, cu, pre_all, 0.5, prot)
    return out


def split_into_chunks(xs, chunk_size):
    """(coalesced_chunks, chunks) exactly as the reference (:62-110): per sequence n // cs full chunks then
    the n % cs tail as its own chunk; `coalesced` merges the full chunks of one sequence."""
    coalesced_chunks, chunks = [], []
    for n in xs:
        nchunks = n // chunk_size
        prologue = nchunks * chunk_size
        epilogue = n - prologue
        if prologue > 0:
            coalesced_chunks.append(prologue)
            chunks.extend([chunk_size] * nchunks)
        if epilogue > 0:
            coalesced_chunks.append(epilogue)
            chunks.append(epilogue)
    return coalesced_chunks, chunks


def _cu_from_lens(lens, device) -> torch.Tensor:
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + int(n))
    return torch.tensor(cu, dtype=torch.int32).to(device, non_blocking=True)


def zscore_segments_(x: torch.Tensor, cu: torch.Tensor, accum: torch.Tensor | None = None, blend: float = 0.0,
                     prot_ranges: torch.Tensor | None = None) -> torch.Tensor:
    """In-place segmented z-score over (rows x H) (biased variance, no eps; reference :224-269), optional
    `+ blend * accum`, then rows in prot_ranges [n,2] <- +inf."""
    assert x.is_contiguous() and x.ndim == 2
    n_seg = cu.numel() - 1
    L = _lib.lib()
    ws_bytes = L.cvllm_zscore_workspace_bytes(n_seg)
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=x.device)
    st = L.cvllm_zscore_segments(
        x.data_ptr(), _lib.score_dtype_code(x.dtype), cu.data_ptr(), n_seg, x.shape[1],
        _lib.ptr(accum), 0 if accum is None else _lib.score_dtype_code(accum.dtype), float(blend),
        _lib.ptr(prot_ranges), 0 if prot_ranges is None else int(prot_ranges.shape[0]), x.shape[0],
        ws.data_ptr(), ws_bytes, _lib.stream(),
    )
    _lib.check(st, "cvllm_zscore_segments")
    return x


def approximate_leverage_scores(
    key_states: torch.Tensor,  # [N, H, D]
    context_lens: List[int],  # [B]
    PHI: torch.Tensor,  # [D, k]
    regularizer: float = 5e-3,
    normalize: bool = False,
    chunk_size: int = 512,
) -> torch.Tensor:  # returns [N, H]
    """score_i = x_i^T (Xc^T Xc + reg I)^-1 x_i per (head, chunk), X = K_h PHI, rows centred inside the chunk;
    optional z-score per chunk.  Returned in key_states.dtype like the reference.  The reference evaluates the
    same closed form through a batched SVD of the 16-bit Gram matrix (:173-210); the HIP kernel keeps X and the
    Gram matrix in fp32 and uses a Cholesky factorisation, so it is at least as accurate (tolerances: SURVEY P3)."""
    _lib.require_cuda(key_states, PHI)
    N, H, D = key_states.shape
    assert key_states.stride(-1) == 1
    assert PHI.shape[0] == D and PHI.is_contiguous()
    context_lens = [int(x) for x in context_lens]
    assert sum(context_lens) == N, "context_lens must sum to the number of packed tokens"
    if chunk_size > 0:
        _, chunks_lens = split_into_chunks(context_lens, chunk_size)
    else:
        chunks_lens = context_lens
    cu = _cu_from_lens(chunks_lens, key_states.device)
    phi = PHI if PHI.dtype == key_states.dtype else PHI.to(key_states.dtype)
    scores = torch.empty((N, H), dtype=torch.float32, device=key_states.device)
    L = _lib.lib()
    kdim = phi.shape[1]
    longest = max(chunks_lens) if chunks_lens else 0
    ws, ws_bytes = None, 0
    if longest > 512:  # chunks that do not fit the fused kernel's LDS image: X goes through a workspace
        ws_bytes = L.cvllm_leverage_workspace_bytes(N, H, kdim)
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=key_states.device)
    st = L.cvllm_leverage_scores(
        key_states.data_ptr(), key_states.stride(0), key_states.stride(1), phi.data_ptr(), scores.data_ptr(),
        cu.data_ptr(), len(chunks_lens), N, H, D, kdim, float(regularizer), _lib.dtype_code(key_states.dtype),
        int(longest), _lib.ptr(ws), ws_bytes, _lib.stream(),
    )
    _lib.check(st, "cvllm_leverage_scores")
    out = scores.to(key_states.dtype)  # the reference's scores live in the model dtype (:197-210)
    if normalize:
        zscore_segments_(out, cu)
    return out


def _protected_ranges(context_lens, first, last, total_rows: int):
    """The reference's plain python slices `out[start:start+first]`, `out[start+L-last:start+L]` (:591-598)
    resolved to explicit [lo, hi) row ranges, negative indices and overshoot included (quirk Q9)."""
    ranges = []
    start = 0
    for f, l, L in zip(first, last, context_lens):
        for sl in (slice(start, start + f), slice(start + L - l, start + L)):
            lo, hi, _ = sl.indices(total_rows)
            if hi > lo:
                ranges.append((lo, hi))
        start += L
    return ranges


def non_causal_attn_scores(
    q: torch.Tensor,  # [N, HQ, D]
    k: torch.Tensor,  # [N, HKV, D]
    v: torch.Tensor,  # [N, HKV, D]  (unused, as in the reference)
    cu_seqlens_qk: torch.Tensor,  # [B + 1]
    max_seqlen_qk: int,
    chunk_size: int,
    sm_scale: float = None,
    normalize: bool = True,
    context_lens: Optional[List[int]] = None,
    protected_first_tokens: Optional[List[int]] = None,
    protected_last_tokens: Optional[List[int]] = None,
    *,
    accum_scores: torch.Tensor = None,  # [N, HKV]
    accum_blending: float = None,
) -> torch.Tensor:
    """Per (sequence, `chunk_size`-token chunk, kv-head): column sums of the row softmax of all chunk queries
    (G heads) over all chunk keys (non-causal) + the reference's padded-row term; z-score per sequence;
    `+ accum_blending * accum_scores`; protected tokens <- +inf.  fp32 [N, HKV]."""
    assert q.ndim == 3 and k.ndim == 3
    assert q.shape[0] == k.shape[0] and q.shape[-1] == k.shape[-1]
    N, HQ, D = q.shape
    HKV = k.shape[1]
    assert HQ % HKV == 0, "Number of query heads must divide number of KV heads"
    assert (D & (D - 1)) == 0, "D must be a power of two"
    assert q.stride(-1) == 1 and k.stride(-1) == 1, "last dim must be contiguous"
    assert q.stride(1) == D
    _lib.require_cuda(q, k)
    B = cu_seqlens_qk.numel() - 1
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)
    cu = _lib.i32(cu_seqlens_qk.to(q.device))
    out = torch.empty((N, HKV), dtype=torch.float32, device=q.device)
    st = _lib.lib().cvllm_chunk_attn_mass(
        q.data_ptr(), k.data_ptr(), q.stride(0), k.stride(0), k.stride(1), out.data_ptr(), cu.data_ptr(), B, N,
        int(max_seqlen_qk), HQ, HKV, D, int(chunk_size), float(sm_scale), _lib.dtype_code(q.dtype), _lib.stream(),
    )
    _lib.check(st, "cvllm_chunk_attn_mass")
    prot = None
    if protected_first_tokens is not None or protected_last_tokens is not None:
        rng = _protected_ranges(context_lens, protected_first_tokens, protected_last_tokens, N)
        if rng:
            prot = torch.tensor(rng, dtype=torch.int32).to(q.device, non_blocking=True)
    blend = 0.0
    acc = None
    if accum_scores is not None:
        acc = accum_scores if accum_scores.is_contiguous() else accum_scores.contiguous()
        blend = 1.0 if accum_blending is None else float(accum_blending)
    if normalize:
        zscore_segments_(out, cu, acc, blend, prot)
    else:
        if acc is not None:
            out += acc.float() * blend
        if prot is not None:
            zscore_segments_(out, cu[:1], None, 0.0, prot)  # zero segments: only the +inf fill runs
    return out
