"""CPU oracle for the compactor-vllm hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain-torch, fp32, CPU restatement of the reference algorithms on the path named
by BASELINE.json's north_star.  It is the checker, never the product:
only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it.
The product package (compactor-vllm_amd/) must never import anything from oracle/.

Parity status: PINNED.  Every function below was checked in the build container against the
upstream reference itself, executed unmodified under the Triton CPU interpreter
(tests/golden/gen_fixtures.py + tests/golden/_ref_loader.py), and the resulting input/output
vectors are committed under tests/golden/*.npz; tests/test_oracle_golden.py re-checks the
oracle against those vectors on every run (CPU, no reference needed).

Citations are relative to /root/reference/src/compactor_vllm/ ("cv/").
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch

F32 = torch.float32
NEG_INF = float("-inf")
POS_INF = float("inf")


# --------------------------------------------------------------------------------------------
# paged-cache addressing  (cv/kv_cache/page_table.py:93-109, SURVEY App. A "Cache addressing")
# --------------------------------------------------------------------------------------------
def cache_rows(page_table_bh: torch.Tensor, L: int, page_size: int) -> torch.Tensor:
    """Physical cache rows of logical positions 0..L-1 of one (batch row, kv head).

    row(i) = page_table[bt, h, i // PS] * PS + i % PS   (cv/attention/sparse_decode_kernel.py:332-340)
    """
    i = torch.arange(L, dtype=torch.int64)
    return page_table_bh.to(torch.int64)[i // page_size] * page_size + (i % page_size)


# --------------------------------------------------------------------------------------------
# a1  prefill attention  (cv/attention/sparse_varlen_kernel.py:11-197, 277-519)
# --------------------------------------------------------------------------------------------
def prefill_attention(
    q: torch.Tensor,  # [N, HQ, D]
    k: torch.Tensor,  # [N, HKV, D]   (may be a strided view)
    v: torch.Tensor,  # [N, HKV, D]
    k_cache: torch.Tensor,  # [CACHE, D]
    v_cache: torch.Tensor,
    seq_lens_bh: torch.Tensor,  # [B, HKV] lengths BEFORE this step
    page_table: torch.Tensor,  # [Bmax+1, HKV, P]
    batch_mapping: torch.Tensor,  # [B]
    cu_seqlens_q: torch.Tensor,  # [B+1]
    HKV: int,
    PAGE_SIZE: int,
    sm_scale: Optional[float] = None,
) -> torch.Tensor:
    """softmax_causal(q [K_cache_prefix || k]^T * scale) [V_cache_prefix || v], per sequence and kv
    head, GQA.  Cached prefix rows are all visible; appended token t sees appended tokens 0..t.
    fp32 everywhere; output rounded once to q.dtype (reference: fp32 accumulate, `.to(q.dtype)`
    at :511)."""
    N, HQ, D = q.shape
    G = HQ // HKV
    scale = 1.0 / math.sqrt(D) if sm_scale is None else float(sm_scale)
    out = torch.zeros((N, HQ, D), dtype=F32)
    cu = cu_seqlens_q.tolist()
    B = len(cu) - 1
    for b in range(B):
        s, e = cu[b], cu[b + 1]
        La = e - s
        if La <= 0:
            continue
        bt = int(batch_mapping[b])
        for g in range(HKV):
            Lc = int(seq_lens_bh[b, g])
            rows = cache_rows(page_table[bt, g], Lc, PAGE_SIZE)
            K = torch.cat([k_cache[rows].to(F32), k[s:e, g].to(F32)], 0)  # [Lc+La, D]
            V = torch.cat([v_cache[rows].to(F32), v[s:e, g].to(F32)], 0)
            t = torch.arange(La)
            j = torch.arange(Lc + La)
            vis = (j[None, :] < Lc) | ((j[None, :] - Lc) <= t[:, None])  # [La, Lc+La]
            for hq in range(g * G, (g + 1) * G):
                logits = (q[s:e, hq].to(F32) @ K.T) * scale
                logits = logits.masked_fill(~vis, NEG_INF)
                p = torch.softmax(logits, dim=-1)
                out[s:e, hq] = p @ V
    return out.to(q.dtype)


# --------------------------------------------------------------------------------------------
# a2  decode attention  (cv/attention/sparse_decode_kernel.py:10-165, 246-435)
# --------------------------------------------------------------------------------------------
def decode_attention(
    q: torch.Tensor,  # [B, HQ, D]
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    seq_lens_bh: torch.Tensor,  # [B, HKV] lengths INCLUDING the current token
    page_table: torch.Tensor,
    batch_mapping: torch.Tensor,
    HKV: int,
    PAGE_SIZE: int,
    sm_scale: Optional[float] = None,
) -> torch.Tensor:
    """One query token per sequence attends to its whole paged cache.  Split-K in the reference
    (:302-306, :391-435) is an exact LSE merge, so the oracle is the unsplit softmax.
    Rows with L == 0 are ZERO here (reference leaves them uninitialised, :281-283; quirk Q6)."""
    B, HQ, D = q.shape
    G = HQ // HKV
    scale = 1.0 / math.sqrt(D) if sm_scale is None else float(sm_scale)
    out = torch.zeros((B, HQ, D), dtype=F32)
    for b in range(B):
        bt = int(batch_mapping[b])
        for g in range(HKV):
            L = int(seq_lens_bh[b, g])
            if L == 0:
                continue
            rows = cache_rows(page_table[bt, g], L, PAGE_SIZE)
            K = k_cache[rows].to(F32)
            V = v_cache[rows].to(F32)
            logits = (q[b, g * G : (g + 1) * G].to(F32) @ K.T) * scale
            out[b, g * G : (g + 1) * G] = torch.softmax(logits, -1) @ V
    return out.to(q.dtype)


# --------------------------------------------------------------------------------------------
# a3 / a4  cache writes  (cv/kv_cache/store_kv_cache.py:251-371, 374-466)
# --------------------------------------------------------------------------------------------
def store_all_kv(
    new_keys, new_values, cu_seqlens_k, k_cache, v_cache, page_table, bh_lens, batch_mapping, PAGE_SIZE
):
    """Row (t,h) of sequence b -> logical position bh_lens[b,h]+t; then bh_lens += len_b (:371).
    Mutates k_cache, v_cache, bh_lens in place."""
    cu = cu_seqlens_k.tolist()
    B = len(cu) - 1
    HKV = new_keys.shape[1]
    for b in range(B):
        s, e = cu[b], cu[b + 1]
        n = e - s
        if n <= 0:
            continue
        bt = int(batch_mapping[b])
        for h in range(HKV):
            L0 = int(bh_lens[b, h])
            pos = L0 + torch.arange(n, dtype=torch.int64)
            rows = page_table[bt, h].to(torch.int64)[pos // PAGE_SIZE] * PAGE_SIZE + pos % PAGE_SIZE
            k_cache[rows] = new_keys[s:e, h]
            v_cache[rows] = new_values[s:e, h]
            bh_lens[b, h] = L0 + n


def decode_store_kv(key, value, batch_mapping, bh_lens, page_table, k_cache, v_cache, PAGE_SIZE, reserved_batch=0):
    """Append one row per (b,h) at position bh_lens[b,h]; bh_lens += 1; rows whose batch_mapping
    equals the reserved batch row are skipped entirely (:395-397)."""
    B, HKV, D = key.shape
    for b in range(B):
        bt = int(batch_mapping[b])
        if bt == reserved_batch:
            continue
        for h in range(HKV):
            L = int(bh_lens[b, h])
            row = int(page_table[bt, h, L // PAGE_SIZE]) * PAGE_SIZE + L % PAGE_SIZE
            k_cache[row] = key[b, h]
            v_cache[row] = value[b, h]
            bh_lens[b, h] = L + 1


# --------------------------------------------------------------------------------------------
# a6  segmented z-score  (cv/compression/compactor.py:224-269)
# --------------------------------------------------------------------------------------------
def zscore_segments(x: torch.Tensor, cu: Sequence[int]) -> torch.Tensor:
    """Per segment [cu[i], cu[i+1]) over (rows x H): mean, biased variance = max(E[x^2]-mean^2, 0),
    (x-mean)/sqrt(var) with NO epsilon (:258-260); math in fp32, result stored back in x.dtype."""
    out = x.clone()
    for i in range(len(cu) - 1):
        s, e = int(cu[i]), int(cu[i + 1])
        if e <= s:
            continue
        seg = x[s:e].to(F32)
        cnt = float(seg.numel())
        mean = seg.sum() / cnt
        var = torch.clamp((seg * seg).sum() / cnt - mean * mean, min=0.0)
        out[s:e] = ((seg - mean) * (1.0 / torch.sqrt(var))).to(x.dtype)
    return out


# --------------------------------------------------------------------------------------------
# a5  Compactor pre-RoPE leverage scores  (cv/compression/compactor.py:62-221)
# --------------------------------------------------------------------------------------------
def split_into_chunks(xs: Sequence[int], chunk_size: int) -> List[int]:
    """Chunk lengths per sequence: n//cs full chunks then the n%cs tail as its own chunk (:99-110)."""
    chunks: List[int] = []
    for n in xs:
        chunks.extend([chunk_size] * (n // chunk_size))
        if n % chunk_size:
            chunks.append(n % chunk_size)
    return chunks


def leverage_scores(
    key_states: torch.Tensor,  # [N, H, D] pre-RoPE
    context_lens: Sequence[int],
    PHI: torch.Tensor,  # [D, k]
    regularizer: float = 5e-3,
    normalize: bool = False,
    chunk_size: int = 512,
    out_dtype: Optional[torch.dtype] = None,
) -> torch.Tensor:
    """score_i = x_i^T (Xc^T Xc + reg*I)^-1 x_i per (head, chunk), X = K_h PHI, rows centred inside
    the chunk; optional z-score per CHUNK over (rows x H).  The reference evaluates the same form
    through an SVD of the Gram matrix (:173-210: G=V S V^T, score = ||x V S^-1/2||^2) with model-
    dtype roundings of X, G, V S^-1/2 and the scores; this oracle evaluates it in fp32 and rounds
    the result once to `out_dtype` (default key_states.dtype), which is what the reference
    returns.  Agreement with the reference: <=1e-4 for fp32 inputs, ~0.1 abs after z-scoring for
    bf16 inputs (the reference's own bf16 noise)."""
    N, H, D = key_states.shape
    out_dtype = key_states.dtype if out_dtype is None else out_dtype
    chunks = split_into_chunks(context_lens, chunk_size) if chunk_size > 0 else list(context_lens)
    X = torch.matmul(key_states.to(F32).transpose(0, 1), PHI.to(F32))  # [H, N, k]
    kdim = X.shape[-1]
    scores = torch.empty((N, H), dtype=F32)
    start = 0
    eye = torch.eye(kdim, dtype=F32)
    for L in chunks:
        if L <= 0:
            continue
        Xc = X[:, start : start + L]
        Xc = Xc - Xc.mean(dim=-2, keepdim=True)
        Gm = Xc.transpose(-1, -2) @ Xc + regularizer * eye  # [H,k,k]
        sol = torch.linalg.solve(Gm, Xc.transpose(-1, -2))  # [H,k,L]
        sc = (Xc * sol.transpose(-1, -2)).sum(-1).clamp_min(0.0)  # [H, L]
        scores[start : start + L] = sc.T
        start += L
    scores = scores.to(out_dtype)
    if normalize:
        cu = [0]
        for L in chunks:
            cu.append(cu[-1] + L)
        scores = zscore_segments(scores, cu)
    return scores


# --------------------------------------------------------------------------------------------
# a7  Compactor post-RoPE chunked non-causal attention mass  (cv/compression/compactor.py:338-599)
# --------------------------------------------------------------------------------------------
def chunk_attn_mass(q, k, cu_seqlens, chunk_size: int, sm_scale: float, q_tile: int = 64) -> torch.Tensor:
    """mass[j,g] = sum over chunk queries i and the G query heads of softmax_row(q_i k_j * scale)
    over ALL keys of the chunk (non-causal), plus the padded-row term: every padding row of the
    last 64-row q-tile of a chunk adds 1/chunk_size to every key of the chunk (:385, :477-479),
    i.e. + G*(ceil(M/64)*64 - M)/chunk_size.  fp32 [N, HKV]."""
    N, HQ, D = q.shape
    HKV = k.shape[1]
    G = HQ // HKV
    out = torch.zeros((N, HKV), dtype=F32)
    cu = [int(x) for x in cu_seqlens]
    for b in range(len(cu) - 1):
        for cs in range(cu[b], cu[b + 1], chunk_size):
            ce = min(cs + chunk_size, cu[b + 1])
            M = ce - cs
            pad_rows = (-(-M // q_tile)) * q_tile - M
            for g in range(HKV):
                qq = q[cs:ce, g * G : (g + 1) * G].to(F32).reshape(M * G, D)
                kk = k[cs:ce, g].to(F32)
                p = torch.softmax((qq @ kk.T) * sm_scale, dim=-1)
                out[cs:ce, g] = p.sum(0) + G * pad_rows / float(chunk_size)
    return out


def fill_protected(out: torch.Tensor, context_lens, first, last) -> None:
    """Protected tokens <- +inf with the reference's plain python slices (:591-598), including
    their behaviour when L < first / L < last (quirk Q9: the slice spills / goes negative)."""
    start = 0
    for f, l, L in zip(first, last, context_lens):
        out[start : start + f] = POS_INF
        out[start + L - l : start + L] = POS_INF  # python slice semantics, negatives included
        start += L


def compactor_post_scores(
    q, k, cu_seqlens, context_lens, pre_scores, protected_first, protected_last,
    chunk_size: int = 128, sm_scale: float = 1.0, blending: float = 0.5,
) -> torch.Tensor:
    """non_causal_attn_scores as called by CompactorCompression.post_rope_scoring (:36-59):
    zscore_per_sequence(mass) + 0.5 * pre_scores, protected tokens <- +inf."""
    mass = chunk_attn_mass(q, k, cu_seqlens, chunk_size, sm_scale)
    out = zscore_segments(mass, [int(x) for x in cu_seqlens])
    if pre_scores is not None:
        out = out + pre_scores.to(F32) * blending
    if protected_first is not None or protected_last is not None:
        fill_protected(out, context_lens, protected_first, protected_last)
    return out


# --------------------------------------------------------------------------------------------
# a8  SnapKV query-aware scores  (cv/compression/snapkv.py:39-448)
# --------------------------------------------------------------------------------------------
def snapkv_scores(q, k, cu_seqlens_q, cu_seqlens_k, w, sm_scale: Optional[float] = None,
                  pool: int = 5, pool_tile: int = 128, normalize: bool = False) -> torch.Tensor:
    """rows = last w queries x G heads; keys = [k_beg, k_end-w); s_j = sum_rows softmax_row(q k / sqrt D)
    over those keys only; trailing `pool`-tap mean clipped at `pool_tile` boundaries measured from the
    sequence start (:253-262; the reference tile is its autotuned BLOCK_K, pinned to 128 here,
    SURVEY P3); last w keys <- +inf (:267-276).  Sequences with L <= w are left untouched by the
    reference (uninitialised memory, :203-205); this oracle defines them as all +inf.
    `w` is an int or one window per sequence (:351-357).  normalize=True: every sequence's scored rows [k_beg, k_end-w)
    x all heads are z-scored with the biased variance and eps 1e-12 INSIDE the square root (:279-329)."""
    Nq, HQ, D = q.shape
    Nk, HKV, _ = k.shape
    G = HQ // HKV
    scale = 1.0 / math.sqrt(D) if sm_scale is None else float(sm_scale)
    out = torch.full((Nk, HKV), POS_INF, dtype=F32)
    cq = [int(x) for x in cu_seqlens_q]
    ck = [int(x) for x in cu_seqlens_k]
    w_all = [int(w)] * (len(ck) - 1) if isinstance(w, int) else [int(x) for x in w]
    for b in range(len(ck) - 1):
        w = w_all[b]
        kb, ke = ck[b], ck[b + 1]
        qe = cq[b + 1]
        keff = ke - w
        if w <= 0:  # no window rows: an empty sum for every key (what the reference's loops leave; not in a fixture)
            out[kb:ke] = 0.0
            continue
        if keff <= kb:
            continue
        for g in range(HKV):
            qq = q[qe - w : qe, g * G : (g + 1) * G].to(F32).reshape(w * G, D)
            kk = k[kb:keff, g].to(F32)
            p = torch.softmax((qq @ kk.T) * scale, dim=-1)
            s = p.sum(0)  # [keff-kb]
            n = s.numel()
            i = torch.arange(n)
            lo = torch.maximum((i // pool_tile) * pool_tile, i - (pool - 1))
            # exact small-window sums: loop over taps
            acc = torch.zeros(n, dtype=F32)
            for t in range(pool):
                idx = i - t
                ok = idx >= lo
                acc = acc + torch.where(ok, s[idx.clamp_min(0)], torch.zeros((), dtype=F32))
            out[kb:keff, g] = acc / (i - lo + 1).to(F32)
        if normalize:
            blk = out[kb:keff]
            mean = blk.mean()
            var = ((blk * blk).mean() - mean * mean).clamp_min(0.0)
            out[kb:keff] = (blk - mean) / torch.sqrt(var + 1e-12)
    return out


# --------------------------------------------------------------------------------------------
# a9 + a10  ranking, retained sets, page padding, compaction
#   (cv/compression/common.py:171-243; cv/kv_cache/store_kv_cache.py:9-248)
# --------------------------------------------------------------------------------------------
def retain_count(ratio: float, L: int, first: int, last: int, HKV: int) -> int:
    """retain_b = max(round(ratio*(L-first-last)*HKV), 1), python banker's round (cv/utils/arguments.py:109-121)."""
    return max(int(round(ratio * (L - first - last) * HKV)), 1)


def rank_indices(scores: torch.Tensor, cu_seqlens_k, max_k_len: int, top_k: int, H: int,
                 padding: float = NEG_INF) -> torch.Tensor:
    """scores_to_retain_indices with the CANONICAL tie rule (score desc, flat index asc): int64
    [B, min(top_k, max_k_len*H)] global flat indices token*H+head; shorter sequences are padded with
    `padding` entries exactly like the reference (:233-243), so their tails index past the sequence."""
    cu = [int(x) for x in cu_seqlens_k]
    B = len(cu) - 1
    keff = min(top_k, max_k_len * H)
    out = torch.empty((B, keff), dtype=torch.int64)
    for b in range(B):
        pad = torch.full((max_k_len, H), padding, dtype=F32)
        pad[: cu[b + 1] - cu[b]] = scores[cu[b] : cu[b + 1]].to(F32)
        flat = pad.reshape(-1)
        order = torch.sort(flat, descending=True, stable=True).indices  # stable => ascending index on ties
        out[b] = order[:keff] + cu[b] * H
    return out


def retained_sets(
    scores: torch.Tensor,  # [N, H] (any float dtype; compared as fp32)
    cu_seqlens_k,
    retain,  # [B] ints
    bh_lens0: torch.Tensor,  # [B, H] lengths before the store
    batch_mapping,
    PAGE_SIZE: int,
    pad_to_page: bool = True,
    reserved_batch: int = 0,
):
    """Semantics of extract_and_store_top_kv (P1): per sequence rank all (t,h) jointly, keep ranks
    < retain_b; then every head whose new length is not a page multiple appends its own next-ranked
    tokens until the page is full, its tokens are exhausted, or written >= ctx_len_b - L
    (store_kv_cache.py:205-248).  Returns (kept, bh_lens_new): kept[b][h] = list of LOCAL token
    indices in rank order; bh_lens_new [B,H].  retain_b is clamped to L_b*H (the reference would
    read padding entries beyond that; documented deviation)."""
    cu = [int(x) for x in cu_seqlens_k]
    B = len(cu) - 1
    H = scores.shape[1]
    new_lens = bh_lens0.clone()
    kept = [[[] for _ in range(H)] for _ in range(B)]
    for b in range(B):
        Lb = cu[b + 1] - cu[b]
        if Lb <= 0 or int(batch_mapping[b]) == reserved_batch:
            continue
        r = max(0, min(int(retain[b]), Lb * H))  # r == 0: nothing scattered, the pad step still runs
        flat = scores[cu[b] : cu[b + 1]].to(F32).reshape(-1)
        order = torch.sort(flat, descending=True, stable=True).indices.tolist()
        per_head_rank = [[] for _ in range(H)]
        for idx in order:
            per_head_rank[idx % H].append(idx // H)
        cnt = [0] * H
        for idx in order[:r]:
            cnt[idx % H] += 1
        for h in range(H):
            L = int(bh_lens0[b, h]) + cnt[h]
            take = cnt[h]
            if pad_to_page and L % PAGE_SIZE != 0:
                need = PAGE_SIZE - L % PAGE_SIZE
                max_additional = Lb - L
                avail = Lb - cnt[h]
                take += max(0, min(need, max_additional, avail))
            kept[b][h] = per_head_rank[h][:take]
            new_lens[b, h] = int(bh_lens0[b, h]) + take
    return kept, new_lens


def compact_store(new_keys, new_vals, kept, cu_seqlens_k, bh_lens0, page_table, batch_mapping,
                  k_cache, v_cache, PAGE_SIZE: int):
    """Write the retained rows of every (b,h) at logical slots bh_lens0[b,h].. in the order given.
    (Slot order is not part of the contract: reference is atomics-ordered, its own test compares
    multisets, tests/test_store_kv.py:163-173.)"""
    cu = [int(x) for x in cu_seqlens_k]
    for b, per_b in enumerate(kept):
        bt = int(batch_mapping[b])
        for h, toks in enumerate(per_b):
            L0 = int(bh_lens0[b, h])
            for j, t in enumerate(toks):
                pos = L0 + j
                row = int(page_table[bt, h, pos // PAGE_SIZE]) * PAGE_SIZE + pos % PAGE_SIZE
                k_cache[row] = new_keys[cu[b] + t, h]
                v_cache[row] = new_vals[cu[b] + t, h]


def ranked_store(new_keys, new_vals, indices_topk, num_tokens_to_retain, page_table, batch_mapping,
                 bh_lens, k_cache, v_cache, PAGE_SIZE, pad_to_page=True, cu_seqlens_k=None,
                 reserved_batch=0):
    """prefill_store_topk_kv restated (store_kv_cache.py:81-248) for a GIVEN rank list
    indices_topk[B, MAX_SEL] of GLOBAL flat indices: first retain_b ranks scattered, then the
    per-head pad scan.  Deterministic slot order = rank order within the head.  Mutates caches and
    bh_lens; returns kept[b][h] (GLOBAL token indices)."""
    B, MAX_SEL = indices_topk.shape
    H = new_keys.shape[1]
    kept = [[[] for _ in range(H)] for _ in range(B)]
    for b in range(B):
        bt = int(batch_mapping[b])
        kt = int(num_tokens_to_retain[b])
        if bt == reserved_batch:
            continue
        idx = indices_topk[b].tolist()  # kt == 0: the scatter kernel returns (:39-40) but the pad kernel runs
        for r in range(min(kt, MAX_SEL)):
            kept[b][idx[r] % H].append(idx[r] // H)
        if pad_to_page:
            ctx = int(cu_seqlens_k[b + 1]) - int(cu_seqlens_k[b])
            for h in range(H):
                L = int(bh_lens[b, h]) + len(kept[b][h])
                if L % PAGE_SIZE == 0:
                    continue
                need = PAGE_SIZE - L % PAGE_SIZE
                max_additional = ctx - L
                written = 0
                r = kt
                while written < need and r < MAX_SEL and written < max_additional:
                    if idx[r] % H == h:
                        kept[b][h].append(idx[r] // H)
                        written += 1
                    r += 1
        for h in range(H):
            L0 = int(bh_lens[b, h])
            for j, t in enumerate(kept[b][h]):
                pos = L0 + j
                row = int(page_table[bt, h, pos // PAGE_SIZE]) * PAGE_SIZE + pos % PAGE_SIZE
                k_cache[row] = new_keys[t, h]
                v_cache[row] = new_vals[t, h]
            bh_lens[b, h] = L0 + len(kept[b][h])
    return kept


# --------------------------------------------------------------------------------------------
# f-2  producer step: qkv split + per-head q/k RMSNorm (Qwen3) + RoPE
#   (cv/models/llama3.py:96-110, cv/models/qwen3.py:88-102, cv/layers/layernorm.py:15-25,
#    cv/layers/rotary_embedding.py:8-17, 28-80)
# --------------------------------------------------------------------------------------------
def rope_cos_sin_cache(head_size: int, max_position: int, base: float, rope_scaling=None) -> torch.Tensor:
    """[max_position, head_size] fp32: cos(D/2) | sin(D/2) per position, llama3 frequency scaling included (:28-68)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_size, 2, dtype=torch.float) / head_size))
    if rope_scaling is not None:
        rope_type, factor, low_f, high_f, old_len = rope_scaling
        assert rope_type == "llama3"
        low_wl, high_wl = old_len / low_f, old_len / high_f
        wavelen = 2 * math.pi / inv_freq
        scaled = torch.where(wavelen > low_wl, inv_freq / factor, inv_freq)
        smooth = (old_len / wavelen - low_f) / (high_f - low_f)
        smoothed = (1 - smooth) * scaled / factor + smooth * scaled
        medium = ~(wavelen < high_wl) * ~(wavelen > low_wl)
        inv_freq = torch.where(medium, smoothed, scaled)
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1)


def head_rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """RMSNorm over the last dim (:15-25): fp32 x * rsqrt(mean(x^2) + eps), rounded to x.dtype, times the weight."""
    xf = x.to(F32)
    var = xf.pow(2).mean(dim=-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return xf.to(x.dtype) * weight


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """rotary_embedding.py:8-17: halves (x1, x2) -> (x1 cos - x2 sin, x2 cos + x1 sin) in fp32, back to x.dtype."""
    x1, x2 = torch.chunk(x.to(F32), 2, dim=-1)
    return torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), dim=-1).to(x.dtype)


def qkv_producer(qkv: torch.Tensor, positions: torch.Tensor, cos_sin: torch.Tensor, HQ: int, HKV: int, D: int,
                 q_norm_w: Optional[torch.Tensor] = None, k_norm_w: Optional[torch.Tensor] = None, eps: float = 1e-6):
    """(q_rot [N,HQ,D], k_rot [N,HKV,D], v [N,HKV,D], k_pre [N,HKV,D]) exactly as the model code produces them in front
    of `Attention.forward`; k_pre is what `apply_prerope_compression` receives (normed for Qwen3, raw for Llama)."""
    N = qkv.shape[0]
    q = qkv[:, : HQ * D].reshape(N, HQ, D)
    k = qkv[:, HQ * D : (HQ + HKV) * D].reshape(N, HKV, D)
    v = qkv[:, (HQ + HKV) * D :].reshape(N, HKV, D)
    if q_norm_w is not None:
        q, k = head_rms_norm(q, q_norm_w, eps), head_rms_norm(k, k_norm_w, eps)
    cs = cos_sin[positions.long()].unsqueeze(1)  # [N, 1, D]
    cos, sin = cs.chunk(2, dim=-1)
    return apply_rope(q, cos, sin), apply_rope(k, cos, sin), v, k


# --------------------------------------------------------------------------------------------
# f-4  split-KV decode across devices: per-shard (out, lse) and the LSE merge
#   (cv/attention/sparse_decode_kernel.py:246-388 stage 1 of ONE split, :391-435 stage 2)
# --------------------------------------------------------------------------------------------
def decode_attention_lse(q, k_cache, v_cache, seq_lens_bh, page_table, batch_mapping, HKV: int, PAGE_SIZE: int,
                         sm_scale: Optional[float] = None):
    """(out [B,HQ,D] q.dtype, lse [B,HQ] fp32): attention over the rows this shard holds and the natural-log LSE of
    its scaled logits (-inf, and zero output, where it holds no row) - what stage 1 hands to stage 2 (:355-372)."""
    B, HQ, D = q.shape
    G = HQ // HKV
    scale = 1.0 / math.sqrt(D) if sm_scale is None else float(sm_scale)
    out = torch.zeros((B, HQ, D), dtype=F32)
    lse = torch.full((B, HQ), NEG_INF, dtype=F32)
    for b in range(B):
        bt = int(batch_mapping[b])
        for g in range(HKV):
            L = int(seq_lens_bh[b, g])
            if L == 0:
                continue
            rows = cache_rows(page_table[bt, g], L, PAGE_SIZE)
            logits = (q[b, g * G : (g + 1) * G].to(F32) @ k_cache[rows].to(F32).T) * scale
            out[b, g * G : (g + 1) * G] = torch.softmax(logits, -1) @ v_cache[rows].to(F32)
            lse[b, g * G : (g + 1) * G] = torch.logsumexp(logits, -1)
    return out.to(q.dtype), lse


def merge_shards(out_all: torch.Tensor, lse_all: torch.Tensor) -> torch.Tensor:
    """out_all [W,B,HQ,D], lse_all [W,B,HQ] -> sum_r exp(lse_r - max) out_r / sum_r exp(lse_r - max) (:391-435);
    rows no shard has anything for come out zero."""
    mx = lse_all.max(dim=0, keepdim=True).values
    w = torch.where(torch.isfinite(lse_all), torch.exp(lse_all - torch.where(torch.isfinite(mx), mx, torch.zeros_like(mx))),
                    torch.zeros_like(lse_all))
    den = w.sum(0)
    num = (w[..., None] * out_all.to(F32)).sum(0)
    return torch.where(den[..., None] > 0, num / den[..., None].clamp_min(1e-38), torch.zeros_like(num)).to(out_all.dtype)
