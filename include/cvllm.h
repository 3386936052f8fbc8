/*
 * cvllm.h — C ABI of the MI355X-native hot path of compactor-vllm (libcvllm_hip.so).
 *
 * One entry point per row of SURVEY.md §8(a).  Every function is `extern "C"`, takes raw DEVICE
 * pointers, plain ints / strides (in ELEMENTS) and the HIP stream to enqueue on, never
 * synchronises the device, allocates nothing, and returns an int status (0 = ok, negative =
 * which precondition failed; see cvllm_error_string).  Entry points whose grid depends only on
 * their integer arguments (decode attention, decode store) are HIP-graph-capture safe.
 *
 * The reference is pure Python/Triton (no FFI of its own); the "reference interface" each
 * function replaces is therefore the Python wrapper + @triton.jit kernel cited next to it
 * (paths relative to /root/reference/src/compactor_vllm/).  The Python host side that reproduces
 * those wrappers' signatures lives in compactor-vllm_amd/compactor_vllm_amd/ and binds this
 * library with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * dtype codes: 0 = float16, 1 = bfloat16 (K/V/Q element type).  Scores are float32 unless noted.
 * All index tensors are int32 and contiguous.  Head dim D in {64,128,256}; GQA group
 * G = HQ/HKV in {1,2,4,8}.
 */
#ifndef CVLLM_H
#define CVLLM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cvllm_stream_t; /* hipStream_t */

#define CVLLM_F16 0
#define CVLLM_BF16 1

#define CVLLM_OK 0
#define CVLLM_ERR_ARG (-1)         /* null pointer / non-positive size */
#define CVLLM_ERR_SHAPE (-2)       /* unsupported D, G, page size or dtype */
#define CVLLM_ERR_WORKSPACE (-3)   /* workspace too small */
#define CVLLM_ERR_LAUNCH (-4)      /* hipLaunch error (hipGetLastError) */

int cvllm_version(void);
const char* cvllm_error_string(int status);

/* ---- a2: decode attention ---------------------------------------------------------------
 * replaces attention/sparse_decode_kernel.py:10-165 head_sparse_decode_attention,
 *          :246-388 _varkv_stage1_groupM, :391-435 _varkv_stage2_reduce.
 * q[B,HQ,D] contiguous; k_cache/v_cache [cache_rows,D]; seq_lens_bh[B,HKV] (lengths INCLUDING the
 * current token); page_table[*,HKV,n_logical_pages_max]; batch_mapping[B]; out[B,HQ,D].
 * n_splits is the number of key splits the kernel uses internally (>=1).  With n_splits > 1 `workspace`
 * (cvllm_decode_workspace_bytes) is required and MUST BE ZERO when first handed to the library; every completed
 * call leaves it zero again.  When B*HKV*n_splits fits the device's CUs the splits are merged inside the one launch
 * (workgroups exchange partials through self-validating mailbox words in the workspace); otherwise, or with
 * CVLLM_DECODE_MERGE=two-kernel / cvllm_decode_set_merge_mode(1), fp32 partials are merged by a second kernel.  THE
 * IN-LAUNCH FORM ASSUMES EVERY WORKGROUP OF THE GRID IS CO-RESIDENT; a plain launch does not guarantee that (CU masks, a
 * shared GPU, a long kernel on another stream), so every wait is bounded (0.5 s), a timed-out slice is written as NaN
 * and the sticky error word is raised: callers poll cvllm_decode_merge_status (the engine does after every decode
 * loop and then falls back to the two-kernel path).  One workspace serves one
 * stream at a time.  Rows with L==0 produce zeros (reference: uninitialised, quirk Q6).
 * Any n_logical_pages_max is accepted (page ids are windowed through registers 512 at a time).              */
size_t cvllm_decode_workspace_bytes(int B, int HQ, int D, int n_splits);
/* health check (synchronises `stream`): 0 fine, 1 = an in-launch merge timed out since the workspace was zeroed
 * (that call's outputs are NaN; re-zero the workspace), negative = error                                     */
int cvllm_decode_merge_status(const void* workspace, cvllm_stream_t stream);
/* process-wide choice of the split merge for grids that fit the chip (host state only): 0 = environment
 * (CVLLM_DECODE_MERGE=in-launch|two-kernel; default in-launch), 1 = two-kernel, 2 = in-launch; returns the previous
 * value.  Not to be flipped between the capture and the replay of a graph that must keep its mode (the mode is baked
 * into the captured launch).                                                                                   */
int cvllm_decode_set_merge_mode(int mode);
int cvllm_decode_attn(const void* q, const void* k_cache, const void* v_cache, void* out,
                      const int32_t* seq_lens_bh, const int32_t* page_table,
                      const int32_t* batch_mapping, void* workspace, size_t workspace_bytes,
                      int B, int HQ, int HKV, int D, int page_size, int n_logical_pages_max,
                      float sm_scale, int n_splits, int dtype, cvllm_stream_t stream);

/* ---- f-4: one shard of a cross-device split-KV decode ----------------------------------------------------
 * A single very long sequence's rows are partitioned over W devices (each holds a slice of every (layer, head) in its
 * own paged cache).  Every device calls cvllm_decode_attn_lse on its slice: out[B,HQ,D] as cvllm_decode_attn plus
 * lse_out[B,HQ] fp32 = natural-log LSE of the scaled logits over the rows it saw (-inf when it saw none); the
 * (out, lse) pairs are all-gathered (RCCL over xGMI: W * B*HQ*(2D+4) bytes per layer) and merged with the LSE rule of
 * attention/sparse_decode_kernel.py:391-435 by cvllm_decode_merge_shards: out_all[W,B,HQ,D], lse_all[W,B,HQ].  */
int cvllm_decode_attn_lse(const void* q, const void* k_cache, const void* v_cache, void* out, float* lse_out,
                          const int32_t* seq_lens_bh, const int32_t* page_table,
                          const int32_t* batch_mapping, void* workspace, size_t workspace_bytes, int B,
                          int HQ, int HKV, int D, int page_size, int n_logical_pages_max, float sm_scale,
                          int n_splits, int dtype, cvllm_stream_t stream);
int cvllm_decode_merge_shards(const void* out_all, const float* lse_all, void* out, int n_shards, int B, int HQ,
                              int D, int dtype, cvllm_stream_t stream);

/* Fused decode step of the boundary orchestrator: replaces the decode branch of layers/attention.py:127-160
 * (index_select of the lengths, decode_store_kv, head_sparse_decode_attention, index_copy_ back) in one call.
 * bh_seq_lens is the LAYER's full [Bmax+1,HKV] table (indexed by batch_mapping[b]), updated in place;
 * key/value [B,HKV,D] are the new token's rows.  Rows with batch_mapping[b]==reserved_batch are not
 * stored and attend to nothing (zeros).  HIP-graph capture safe.                                 */
int cvllm_decode_append_attn(const void* q, const void* key, const void* value, int64_t sk_b,
                             int64_t sk_h, int64_t sv_b, int64_t sv_h, void* k_cache, void* v_cache,
                             void* out, int32_t* bh_seq_lens, const int32_t* page_table,
                             const int32_t* batch_mapping, void* workspace, size_t workspace_bytes,
                             int B, int HQ, int HKV, int D, int page_size, int n_logical_pages_max,
                             float sm_scale, int n_splits, int reserved_batch, int dtype,
                             cvllm_stream_t stream);

/* replaces attention/sparse_decode_kernel.py:169-192 num_splits_heuristic (host, same results) */
int cvllm_num_splits(int total_mblocks, int max_seq_len, int num_sms, int max_splits);

/* ---- a4: decode cache append --------------------------------------------------------------
 * replaces kv_cache/store_kv_cache.py:419-466 decode_store_kv + :374-416 kernel.
 * key/value [B,HKV,D] with element strides (s_b,s_h), last dim contiguous.  Appends one row per
 * (b,h) at bh_lens[b,h] and increments it; rows with batch_mapping[b]==reserved_batch skipped. */
int cvllm_store_decode_kv(const void* key, const void* value, int64_t sk_b, int64_t sk_h,
                          int64_t sv_b, int64_t sv_h, const int32_t* batch_mapping,
                          int32_t* bh_lens, const int32_t* page_table, void* k_cache,
                          void* v_cache, int B, int HKV, int D, int page_size,
                          int n_logical_pages_max, int reserved_batch, int dtype,
                          cvllm_stream_t stream);

/* ---- a3: uncompressed prefill cache write -------------------------------------------------
 * replaces kv_cache/store_kv_cache.py:322-371 prefill_store_all_kv + :251-319 kernel.
 * new_keys/new_values [N,HKV,D] with element strides (s_n,s_h).  Row (t,h) of sequence b goes
 * to logical position bh_lens[b,h]+t; afterwards bh_lens[b,:] += len_b (done on device).   */
int cvllm_store_all_kv(const void* new_keys, const void* new_values, int64_t sk_n, int64_t sk_h,
                       int64_t sv_n, int64_t sv_h, const int32_t* cu_seqlens_k,
                       const int32_t* batch_mapping, int32_t* bh_lens,
                       const int32_t* page_table, void* k_cache, void* v_cache, int B,
                       int total_tokens, int HKV, int D, int page_size,
                       int n_logical_pages_max, int dtype, cvllm_stream_t stream);

/* ---- a1: prefill attention ------------------------------------------------------------------
 * replaces attention/sparse_varlen_kernel.py:11-197 causal_sparse_varlen_with_cache + :277-519.
 * q[N,HQ,D] (stride sq_n, heads contiguous), k/v [N,HKV,D] with strides (s_n,s_h);
 * seq_lens_bh[B,HKV] = cached prefix lengths BEFORE this step; out[N,HQ,D] contiguous.       */
int cvllm_prefill_attn(const void* q, const void* k, const void* v, int64_t sq_n, int64_t sk_n,
                       int64_t sk_h, int64_t sv_n, int64_t sv_h, const void* k_cache,
                       const void* v_cache, void* out, const int32_t* seq_lens_bh,
                       const int32_t* page_table, const int32_t* batch_mapping,
                       const int32_t* cu_seqlens_q, int B, int total_tokens, int max_seqlen_q,
                       int HQ, int HKV, int D, int page_size, int n_logical_pages_max,
                       float sm_scale, int dtype, cvllm_stream_t stream);

/* ---- a6: segmented z-score -------------------------------------------------------------------
 * replaces compression/compactor.py:224-269 _zscore_per_batch_epilogue_no_window.
 * x[N,H] in place; score_dtype 0/1 = f16/bf16, 2 = f32; cu[n_segments+1] int32 row offsets.
 * Optional fused epilogue (compactor.py:586-598): x = z(x) + blend*accum (accum_dtype coded as
 * score_dtype, accum may be NULL), then the rows of prot_ranges (int32 [n_ranges,2] = [lo,hi)
 * row ranges, may be NULL) <- +inf.  The python-slice semantics of the reference's fills
 * (quirk Q9) are resolved to explicit ranges by the host wrapper.  Long segments are reduced by
 * several workgroups through per-tile partials in `workspace` (fixed fold order: deterministic). */
size_t cvllm_zscore_workspace_bytes(int n_segments);
int cvllm_zscore_segments(void* x, int score_dtype, const int32_t* cu, int n_segments, int H,
                          const void* accum, int accum_dtype, float blend,
                          const int32_t* prot_ranges, int n_ranges, int total_rows,
                          void* workspace, size_t workspace_bytes, cvllm_stream_t stream);
/* windowed form: replaces compression/snapkv.py:279-329 _zscore_per_batch_epilogue.  Segment s is rows
 * [cu[s], cu[s+1] - trim_s) with trim_s = trim_b[s] (int32 [n_segments]) or `trim` when trim_b is NULL; empty
 * segments are left alone; the variance gets `eps` added inside the square root (1e-12 upstream).            */
int cvllm_zscore_windowed(void* x, int score_dtype, const int32_t* cu, const int32_t* trim_b, int trim,
                          int n_segments, int H, float eps, int total_rows, void* workspace,
                          size_t workspace_bytes, cvllm_stream_t stream);

/* ---- a7: Compactor post-RoPE chunked non-causal attention mass -------------------------------
 * replaces compression/compactor.py:338-486 _non_causal_attn_kernel (+ wrapper :489-580).
 * mass[N,HKV] f32 (overwritten): column sums of row-softmax(q k^T * sm_scale) inside each
 * chunk of `chunk_size` tokens of one sequence, plus the reference's padded-row term.        */
int cvllm_chunk_attn_mass(const void* q, const void* k, int64_t sq_n, int64_t sk_n, int64_t sk_h,
                          float* mass, const int32_t* cu_seqlens, int B, int total_tokens,
                          int max_seqlen, int HQ, int HKV, int D, int chunk_size, float sm_scale,
                          int dtype, cvllm_stream_t stream);

/* ---- a5: Compactor pre-RoPE leverage scores ----------------------------------------------------
 * replaces compression/compactor.py:113-221 approximate_leverage_scores (matmul + SVD path).
 * key_states[N,HKV,D] (strides s_n,s_h), PHI[D,k] row-major same dtype, chunk_cu[n_chunks+1]
 * int32 row offsets of the chunks (host-built exactly like split_into_chunks :62-110),
 * scores[N,HKV] f32 out: x_i^T (Xc^T Xc + reg I)^-1 x_i, X = K_h PHI centred per chunk, evaluated
 * by Cholesky in fp32 (the reference takes an SVD of the 16-bit Gram matrix; same closed form).
 * sketch_dim must be 48 (LLMConfig.leverage_sketch_size).  max_chunk_rows = the longest chunk (a host value:
 * the chunk list is built on the host); when 0 < max_chunk_rows <= 512 one kernel keeps the sketch X in LDS and
 * needs no workspace, otherwise X goes through `workspace` in fp32 (two kernels); equal up to fp32 rounding.       */
size_t cvllm_leverage_workspace_bytes(int total_tokens, int HKV, int sketch_dim);
int cvllm_leverage_scores(const void* key_states, int64_t s_n, int64_t s_h, const void* phi,
                          float* scores, const int32_t* chunk_cu, int n_chunks, int total_tokens,
                          int HKV, int D, int sketch_dim, float regularizer, int dtype,
                          int max_chunk_rows, void* workspace, size_t workspace_bytes,
                          cvllm_stream_t stream);

/* ---- a8: SnapKV window scores -------------------------------------------------------------------
 * replaces compression/snapkv.py:332-448 query_aware_key_scores + :39-157 + :160-276.
 * scores[Nk,HKV] f32 out; rows = last w queries x G heads (w*G <= 256); keys [0, L-w);
 * `pool`-tap trailing mean clipped at the start of the key's `pool_tile`-wide block counted from the sequence start:
 * pool_tile in {32, 64, 128} is the reference's autotuned BLOCK_K of _scores_from_logits_kernel (snapkv.py:160-168,
 * 253-262), which changes the pooled values; 0 = 128.  Last w keys <- +inf; sequences with L <= w are all +inf
 * (reference: uninitialised).
 * No fp32 logits buffer: two passes over K with per-tile (max,sum) partials in `workspace`.     */
size_t cvllm_snapkv_workspace_bytes(int B, int HKV, int w, int max_seqlen_k);
int cvllm_snapkv_scores(const void* q, const void* k, int64_t sq_n, int64_t sk_n, int64_t sk_h,
                        float* scores, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                        int B, int HQ, int HKV, int D, int w, float sm_scale, int pool, int pool_tile,
                        int max_seqlen_k, int dtype, void* workspace, size_t workspace_bytes,
                        cvllm_stream_t stream);
/* the same with one window per sequence (snapkv.py:351-357: `w` may be a [B] int32 tensor): window_b[b] <= w_max,
 * w_max * G <= 256; a sequence with window 0 is left untouched.                                              */
int cvllm_snapkv_scores_wb(const void* q, const void* k, int64_t sq_n, int64_t sk_n, int64_t sk_h,
                           float* scores, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                           const int32_t* window_b, int B, int HQ, int HKV, int D, int w_max, float sm_scale,
                           int pool, int pool_tile, int max_seqlen_k, int dtype, void* workspace,
                           size_t workspace_bytes, cvllm_stream_t stream);

/* ---- a9: joint top-k selection ------------------------------------------------------------------
 * replaces compression/common.py:171-243 scores_to_retain_indices (torch.topk full sort) and the
 * rank-consuming half of kv_cache/store_kv_cache.py:9-78,178-248.
 * scores[N,H] f32; per sequence keep the retain[b] largest (score desc, flat index asc), then
 * pad every head up to a page boundary with its next-best tokens (pad_to_page).  Outputs:
 * kept_idx[B,H,max_seqlen] int32 = LOCAL token indices of the retained rows of every (b,h) in
 * ascending token order (first new_lens-bh_lens0 entries valid), new_lens[B,H] = bh_lens0 + kept
 * count (bh_lens0 itself untouched).  Sequences with batch_mapping[b]==reserved_batch keep
 * nothing.  retain[b] is clamped to L_b*H.                                                      */
size_t cvllm_select_workspace_bytes(int B, int H, int max_seqlen);
int cvllm_select_topk(const float* scores, const int32_t* cu_seqlens_k, const int32_t* retain,
                      const int32_t* bh_lens0, const int32_t* batch_mapping, int32_t* kept_idx,
                      int32_t* new_lens, int B, int H, int max_seqlen, int page_size,
                      int pad_to_page, int reserved_batch, void* workspace,
                      size_t workspace_bytes, cvllm_stream_t stream);
/* health check (synchronises `stream`): 0 fine; 1 = since the last check a slice workgroup of the per-head ordered
 * write (sequences >= 8192 tokens) gave up waiting 0.5 s for the counts of the slices before it - the library never
 * traps or spins on: that call's kept lists are incomplete (every index still a valid token: the consumers clamp), the
 * sticky per-device word is cleared by this call; negative = error.  Call it once per prefill, after the store stream
 * has been joined (compactor_vllm_amd.compression.common.select_status).                                           */
int cvllm_select_status(cvllm_stream_t stream);

/* ---- a10: compaction -----------------------------------------------------------------------------
 * replaces kv_cache/store_kv_cache.py:81-175 prefill_store_topk_kv (scatter + pad kernels).
 * Copies the kept rows of every (b,h) to logical slots bh_lens0[b,h].. in TOKEN order
 * (deterministic; the reference is atomics-ordered and its test compares multisets).
 * kept_idx / new_lens are the outputs of cvllm_select_topk.                                   */
int cvllm_compact_store(const void* new_keys, const void* new_vals, int64_t sk_n, int64_t sk_h,
                        int64_t sv_n, int64_t sv_h, const int32_t* kept_idx,
                        const int32_t* new_lens, const int32_t* cu_seqlens_k,
                        const int32_t* bh_lens0, const int32_t* page_table,
                        const int32_t* batch_mapping, void* k_cache, void* v_cache, int B, int H,
                        int D, int max_seqlen, int page_size, int n_logical_pages_max, int dtype,
                        cvllm_stream_t stream);

/* ranked-list variant with the reference's own argument list (indices_topk int32 [B,MAX_SEL]
 * GLOBAL flat indices): store_kv_cache.py:81-248.  bh_lens updated in place.                 */
int cvllm_store_topk_ranked(const void* new_keys, const void* new_vals, int64_t sk_n, int64_t sk_h,
                            int64_t sv_n, int64_t sv_h, const int32_t* indices_topk,
                            const int32_t* num_tokens_to_retain, const int32_t* page_table,
                            const int32_t* batch_mapping, int32_t* bh_lens, void* k_cache,
                            void* v_cache, const int32_t* cu_seqlens_k, int B, int H, int D,
                            int max_sel, int page_size, int n_logical_pages_max,
                            int pad_to_page, int reserved_batch, int dtype,
                            cvllm_stream_t stream);

/* full ranking for API parity with scores_to_retain_indices: out int64 [B, k_eff] global flat
 * indices, (score desc, flat index asc); padded entries of shorter sequences follow in index
 * order exactly as a stable sort of the reference's -inf padded matrix would give.          */
size_t cvllm_rank_workspace_bytes(int B, int H, int max_seqlen);
int cvllm_rank_indices(const float* scores, const int32_t* cu_seqlens_k, int64_t* out, int B,
                       int H, int max_seqlen, int k_eff, void* workspace,
                       size_t workspace_bytes, cvllm_stream_t stream);

/* ---- f-3: compaction in place (last chunk of a chunked prefill) -------------------------------------------
 * The chunks of a long prompt went into the cache uncompressed (attention/sparse_varlen_kernel.py:362-401 attends to
 * [cached prefix || chunk]); after the last chunk the selection of cvllm_select_topk runs over the whole sequence and
 * the kept rows of every (b,h) are moved, inside the cache, from logical rows src_base[b,h] + kept_idx[b,h,j] to
 * dst_base[b,h] + j (token order; dst_base <= src_base).  Same final cache bytes and lengths as cvllm_compact_store
 * fed with the packed keys / values of the whole sequence.  kept_idx [B,H,max_seqlen], new_lens [B,H] (= dst_base +
 * count) are the outputs of cvllm_select_topk called with bh_lens0 = dst_base.                               */
int cvllm_compact_cache_inplace(const int32_t* kept_idx, const int32_t* new_lens, const int32_t* dst_base,
                                const int32_t* src_base, const int32_t* page_table,
                                const int32_t* batch_mapping, void* k_cache, void* v_cache, int B, int H,
                                int D, int max_seqlen, int page_size, int n_logical_pages_max, int dtype,
                                cvllm_stream_t stream);

/* ---- f-2: fused producer step in front of the attention boundary -------------------------------------
 * replaces models/llama3.py:96-110 / qwen3.py:88-102 (qkv.split + views), layers/layernorm.py:15-25 (per-head q/k
 * RMSNorm of Qwen3), layers/rotary_embedding.py:8-17,69-80 (RoPE), and - when k_cache is given (no compression) -
 * kv_cache/store_kv_cache.py:251-371 (prefill_store_all_kv), in ONE pass over the fused projection output.
 * qkv[N,(HQ+2*HKV)*D] with token stride s_n (elements): q heads, then k heads, then v heads; positions[N] int64;
 * cos_sin[max_pos,D] fp32 = cos(D/2) | sin(D/2) per position (the reference's cos_sin_cache); q_norm_w / k_norm_w [D]
 * model dtype or both NULL; q_out[N,HQ,D] / k_out[N,HKV,D] with token strides so_q / so_k (they may be two views of one
 * [N,HQ+HKV,D] buffer); k_pre_out[N,HKV,D] contiguous or NULL: the NORMED pre-RoPE keys (only with norm weights - without
 * them the pre-RoPE keys are the projection itself).  k_cache != NULL: rotated K rows and V rows are also appended to
 * the paged cache at bh_lens[b,h] + t (cu_seqlens[B+1], batch_mapping[B], bh_lens[B,HKV] updated += len_b afterwards).
 * D in {64,128}.  Without norm weights the outputs equal the eager fp32 evaluation bit for bit.                 */
int cvllm_qkv_rope_producer(const void* qkv, int64_t s_n, const int64_t* positions, const float* cos_sin,
                            const void* q_norm_w, const void* k_norm_w, float eps, void* q_out,
                            int64_t so_q, void* k_out, int64_t so_k, void* k_pre_out, void* k_cache,
                            void* v_cache, const int32_t* cu_seqlens, const int32_t* batch_mapping,
                            int32_t* bh_lens, const int32_t* page_table, int B, int page_size,
                            int n_logical_pages_max, int N, int HQ, int HKV, int D, int max_pos, int dtype,
                            cvllm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CVLLM_H */
