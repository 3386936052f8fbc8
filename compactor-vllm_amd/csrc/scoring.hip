// a5 / a6 / a7 / a8 — KV scoring kernels for gfx950.
//
// Replaces cv/compression/compactor.py: approximate_leverage_scores (:113-221, torch.matmul + batched SVD),
// _zscore_per_batch_epilogue_no_window (:224-269), _non_causal_attn_kernel (:338-486) and
// cv/compression/snapkv.py: _lse_and_store_logits_kernel (:39-157) + _scores_from_logits_kernel (:160-276).
//
// All of them are tiny next to the prefill attention they overlap with; the MFMA kernels reuse the register
// fragment convention of prefill_attn.hip (32x32x16: lane = (row/col r = lane&31, k-half h = lane>>5)).
#include "common.h"

namespace cvllm {

template <typename T>
__device__ __forceinline__ f32x16 mfma32s(s16x8 a, s16x8 b, f32x16 c);
template <>
__device__ __forceinline__ f32x16 mfma32s<F16>(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mfma32s<BF16>(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}

// -DCVLLM_SC_TS (debug builds of tools/dbg only): thread 0 of every workgroup records s_memrealtime (100 MHz, one
// clock for all CUs) at phase boundaries.  Not compiled into libcvllm_hip.so.
#ifdef CVLLM_SC_TS
__device__ unsigned long long g_sc_rt[1024 * 16];
#define SC_RT(i)                                                                                                   \
  do {                                                                                                             \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_sc_rt[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime();  \
  } while (0)
#else
#define SC_RT(i) \
  do {           \
  } while (0)
#endif
__device__ __forceinline__ uint32_t ktile_off(int row, int ch) { return row * 256 + 16 * (ch ^ (row & 15)); }

// =====================================================================================================
// a6: segmented z-score (+ optional blend with an accumulated score tensor), then protected-row fill
// =====================================================================================================
template <int DT>  // 0 f16, 1 bf16, 2 f32
__device__ __forceinline__ float ld_score(const void* p, size_t i) {
  if (DT == 2) return reinterpret_cast<const float*>(p)[i];
  if (DT == 0) return from16<F16>(reinterpret_cast<const uint16_t*>(p)[i]);
  return from16<BF16>(reinterpret_cast<const uint16_t*>(p)[i]);
}
template <int DT>
__device__ __forceinline__ void st_score(void* p, size_t i, float v) {
  if (DT == 2) reinterpret_cast<float*>(p)[i] = v;
  else if (DT == 0) reinterpret_cast<uint16_t*>(p)[i] = to16<F16>(v);
  else reinterpret_cast<uint16_t*>(p)[i] = to16<BF16>(v);
}

constexpr int ZS_T = 1024;
constexpr int ZS_MAXTILES = 32;
// Two kernels, both grid (segment, tile): (1) per-tile partial sum / sum of squares, (2) every tile folds the
// partials of its segment IN TILE ORDER (deterministic: no float atomics), then normalises its own slice.
template <int DT>
__global__ __launch_bounds__(ZS_T) void zscore_partial_kernel(const void* __restrict__ x, const int* __restrict__ cu,
                                                              int H, int T, float* __restrict__ part,
                                                              const int* __restrict__ trim_b, int trim) {
  __shared__ float s_a[ZS_T / 64], s_b[ZS_T / 64];
  const int seg = blockIdx.x, tile = blockIdx.y;
  const int tr = trim_b ? trim_b[seg] : trim;  // rows left out at the end of the segment (windowed form)
  const size_t beg = (size_t)cu[seg] * H, end = (size_t)max(cu[seg + 1] - tr, 0) * H;
  const size_t n = end > beg ? end - beg : 0;
  const size_t per = (n + T - 1) / T;
  const size_t lo = beg + per * tile, hi = min(lo + per, end);
  const int tid = threadIdx.x;
  float sum = 0.f, sq = 0.f;
  for (size_t i = lo + tid; i < hi; i += ZS_T) {
    const float v = ld_score<DT>(x, i);
    sum += v;
    sq += v * v;
  }
  sum = wave_reduce_sum(sum);
  sq = wave_reduce_sum(sq);
  if ((tid & 63) == 0) {
    s_a[tid >> 6] = sum;
    s_b[tid >> 6] = sq;
  }
  __syncthreads();
  if (tid == 0) {
    sum = 0.f;
    sq = 0.f;
#pragma unroll
    for (int w = 0; w < ZS_T / 64; ++w) {
      sum += s_a[w];
      sq += s_b[w];
    }
    part[((size_t)seg * T + tile) * 2] = sum;
    part[((size_t)seg * T + tile) * 2 + 1] = sq;
  }
}

template <int DT, int ADT>
__global__ __launch_bounds__(ZS_T) void zscore_apply_kernel(void* __restrict__ x, const int* __restrict__ cu, int H,
                                                            int T, const float* __restrict__ part,
                                                            const void* __restrict__ accum, float blend,
                                                            const int* __restrict__ trim_b, int trim, float eps) {
  const int seg = blockIdx.x, tile = blockIdx.y;
  const int tr = trim_b ? trim_b[seg] : trim;
  const size_t beg = (size_t)cu[seg] * H, end = (size_t)max(cu[seg + 1] - tr, 0) * H;
  if (end <= beg) return;
  float sum = 0.f, sq = 0.f;
  for (int t = 0; t < T; ++t) {
    sum += part[((size_t)seg * T + t) * 2];
    sq += part[((size_t)seg * T + t) * 2 + 1];
  }
  const float cnt = (float)(end - beg);
  const float mean = sum / cnt;
  const float var = fmaxf(sq / cnt - mean * mean, 0.f);  // biased, clamped, NO epsilon (compactor.py:258-260)
  const float invstd = 1.0f / sqrtf(var + eps);  // eps = 0 for the Compactor form (compactor.py:258-260)
  const size_t per = (end - beg + T - 1) / T;
  const size_t lo = beg + per * tile, hi = min(lo + per, end);
  for (size_t i = lo + threadIdx.x; i < hi; i += ZS_T) {
    float v = (ld_score<DT>(x, i) - mean) * invstd;
    if (accum) v += blend * ld_score<ADT>(accum, i);
    st_score<DT>(x, i, v);
  }
}

template <int DT>
__global__ void fill_inf_kernel(void* __restrict__ x, const int* __restrict__ ranges, int H, int total_rows) {
  const int lo = max(ranges[2 * blockIdx.x], 0), hi = min(ranges[2 * blockIdx.x + 1], total_rows);
  for (size_t i = (size_t)lo * H + threadIdx.x; i < (size_t)hi * H; i += blockDim.x) st_score<DT>(x, i, INFINITY);
}

// =====================================================================================================
// a7: chunked non-causal attention mass (Compactor post-RoPE score)
//   per (sequence, 128-token chunk, kv-head): rows = tokens x G heads, keys = the chunk's tokens.
//   pass 1 (S^T = K Q^T, query on the lane): exact row max / row sum of every query over all chunk keys.
//   pass 2 (S = Q K^T, key on the lane): p = exp2(s*c - lse2[q]) summed over queries -> per-key mass.
//   Same fragments for both passes, only the operand order swaps.
// =====================================================================================================
constexpr int CM_CHUNK = 128;
constexpr int CM_SMEM = CM_CHUNK * 256 + 4 * 32 * 256 + 4 * 32 * 4;
#ifndef CM_OCC
#define CM_OCC 2
#endif
template <typename T, int D, int G>
// min 2 waves per SIMD: left alone hipcc spends 272 registers (1 wave per SIMD = ONE workgroup per CU; measured 180 us)
__global__ __launch_bounds__(256, CM_OCC) void chunk_mass_kernel(const uint16_t* __restrict__ q,
                                                         const uint16_t* __restrict__ k, int64_t sq_n, int64_t sk_n,
                                                         int64_t sk_h, float* __restrict__ mass,
                                                         const int* __restrict__ cu, int B, int HKV, int nchunk_max,
                                                         float scale_log2e, float pad_unit) {
  constexpr int KS = D / 16;
  constexpr int CH = D / 8;
  constexpr int QB = G;  // 32-row query blocks per wave: rows = 128*G over 4 waves
  // (Round 2, not kept: 16-row Q tiles = 49.7 KB for a third workgroup per CU - the kernel needs ~230 registers, 168 are
  // left at three waves per SIMD: 118-150 spilled, 244 us; the same structure at two workgroups per CU: 110 us.)
  // dynamic LDS (CM_SMEM = 66,048 B, over the static limit): K tile | one 32-row Q tile per wave | row LSEs; the
  // per-wave column sums reuse the Q tiles after the loop
  extern __shared__ __attribute__((aligned(16))) char cm_smem[];
  char* s_k = cm_smem;                                                        // [CM_CHUNK * 256]
  char* s_q = cm_smem + CM_CHUNK * 256;                                       // [4][32 * 256]
  float(*s_lse)[32] = reinterpret_cast<float(*)[32]>(cm_smem + CM_CHUNK * 256 + 4 * 32 * 256);  // [4][32]
  float(*s_mass)[CM_CHUNK] = reinterpret_cast<float(*)[CM_CHUNK]>(s_q);       // [4][CM_CHUNK]

  SC_RT(0);
  const int bid = blockIdx.x;
  const int g = bid % HKV;
  const int c = (bid / HKV) % nchunk_max;
  const int b = bid / (HKV * nchunk_max);
  const int s0 = cu[b], Lb = cu[b + 1] - s0;
  const int t0 = c * CM_CHUNK;
  if (t0 >= Lb) return;
  const int M = min(CM_CHUNK, Lb - t0);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;

  // K tile -> LDS (rows >= M zero).  The first query block's rows are requested right behind the K tile, before the
  // tile is staged: asked for after the barrier they cost every workgroup a second exposed memory latency (query
  // block 0 took 6.3 us against 2.9-4.6 for the others, profiles/r02_scoring_phase_stamps.txt)
  constexpr int KPT = CM_CHUNK * CH / 256;
  uint4 kv[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int e = tid + 256 * j, row = e / CH, ch = e % CH;
    kv[j] = make_uint4(0, 0, 0, 0);
    if (row < M) kv[j] = *reinterpret_cast<const uint4*>(k + (size_t)(s0 + t0 + row) * sk_n + (size_t)g * sk_h + ch * 8);
  }
  float colsum[4] = {0.f, 0.f, 0.f, 0.f};  // key = kb*32 + r, this lane's half of the query rows
  // Q rows of a 32-row query block (row in [0, 128*G): head = row / 128, token = row % 128) are fetched as WHOLE rows
  // (a wave instruction = 64 / CH rows of D*2 contiguous bytes) and turned into MFMA fragments through a wave-private
  // LDS tile with the K tile's swizzle; the NEXT block's rows are in flight while a block is processed.  (Loading
  // the fragments straight from global memory is 32-byte pieces of rows 8 KB apart: 117 us for 336 MB.)
  constexpr int RPI = 64 / CH;   // rows per load instruction
  constexpr int NI = 32 / RPI;   // load instructions per 32-row block
  uint4 qn[NI];  // (two blocks in flight, block loop unrolled in pairs: 228-254 registers spilled - not kept; four chunks
                 // per workgroup with the next chunk's K tile / first Q block requested ahead: 110-149 spilled - not kept)
  char* s_qw = s_q + wave * (32 * 256);
  const int lrow = lane / CH, lch = lane % CH;
  auto load_q = [&](int qb) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int row = (wave * QB + qb) * 32 + RPI * j + lrow;
      const int head = row / CM_CHUNK, tok = row % CM_CHUNK;
      qn[j] = tok < M ? *reinterpret_cast<const uint4*>(q + (size_t)(s0 + t0 + tok) * sq_n + (size_t)(g * G + head) * D + 8 * lch)
                      : make_uint4(0, 0, 0, 0);
    }
  };
  load_q(0);
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int e = tid + 256 * j;
    *reinterpret_cast<uint4*>(s_k + ktile_off(e / CH, e % CH)) = kv[j];
  }
  __syncthreads();
  SC_RT(1);
  // K fragments (A operand of pass 1, B operand of pass 2: the same layout) come from LDS in batches of one 32-key
  // block (8 reads), the NEXT block's batch issued before the current block's 8 MFMAs: read one fragment ahead of
  // every MFMA (what hipcc schedules by itself) the loop ran at LDS latency - 64 reads of ~ 120 cycles per query
  // block; all 32 fragments resident (128 registers) spilled.  Step st = 4 * pass + kb uses ring slot st & 1.
  s16x8 kf[2][KS];
  auto load_kf = [&](int slot, int kb) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < KS; ++s) kf[slot][s] = *reinterpret_cast<const s16x8*>(s_k + ktile_off(kb * 32 + r, 2 * s + h));
  };

#pragma unroll 1
  for (int qb = 0; qb < QB; ++qb) {
    if (qb < 4) SC_RT(2 + qb);
    const int row = (wave * QB + qb) * 32 + r;
    const int tok = row % CM_CHUNK;
    const bool valid_q = tok < M;
#pragma unroll
    for (int j = 0; j < NI; ++j) *reinterpret_cast<uint4*>(s_qw + ktile_off(RPI * j + lrow, lch)) = qn[j];
    if (qb + 1 < QB) load_q(qb + 1);
    load_kf(0, 0);
    s16x8 qf[KS];  // the wave reads back only what it wrote itself: LDS operations of one wave execute in order
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const s16x8*>(s_qw + ktile_off(r, 2 * s + h));
    // pass 1: queries on lanes; row max / row sum folded over the four 32-key blocks (online: one block of logits
    // live at a time).  (Issuing a step's MFMAs before the previous step's VALU work, with two accumulators, measured
    // 109 us against 101: the register pressure costs more than the overlap gives at two waves per SIMD.)
    float mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      load_kf((kb + 1) & 1, (kb + 1) & 3);  // steps 0-3: slot kb & 1 now, kb + 1 (or pass 2's block 0) next
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = mfma32s<T>(kf[kb & 1][s], qf[s], acc);
      if (M < CM_CHUNK) {  // workgroup-uniform: only a sequence's last chunk has keys to mask
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int kk = kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          acc[i] = kk < M ? acc[i] : -INFINITY;
        }
      }
      // max over the RAW logits (scale > 0 commutes with max), then exp2(s c - m c): one FMA per logit
      float bm = fmaxf(acc[0], acc[1]);
#pragma unroll
      for (int i = 2; i < 16; ++i) bm = fmaxf(bm, acc[i]);
      bm = fmaxf(bm, __shfl_xor(bm, 32, 64)) * scale_log2e;  // key 0 of the chunk is always valid: finite from block 0 on
      const float mn = fmaxf(mx, bm);
      float bs = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) bs += __builtin_amdgcn_exp2f(fmaf(acc[i], scale_log2e, -mn));
      sum = sum * __builtin_amdgcn_exp2f(mx - mn) + bs;
      mx = mn;
    }
    sum += __shfl_xor(sum, 32, 64);
    // lse in the exp2 domain; invalid query rows get +inf so that they contribute p = 0 in pass 2
    const float lse2 = valid_q ? mx + __builtin_amdgcn_logf(sum) : INFINITY;  // v_log_f32 = log2
    // the block's 32 row values through LDS (wave-private): pass 2 needs the lse of query rows 8 j + 4 h + 0..3, four
    // consecutive floats per accumulator quad - 4 ds_read_b128 per step instead of 16 ds_bpermute
    if (h == 0) s_lse[wave][r] = lse2;

    // pass 2: keys on lanes, query rows in the accumulator registers
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      f32x16 a2;
#pragma unroll
      for (int i = 0; i < 16; ++i) a2[i] = 0.f;
      if (kb < 3) load_kf((kb + 1) & 1, kb + 1);  // steps 4-6 fetch the next block; block 0 is re-read per query block
#pragma unroll
      for (int s = 0; s < KS; ++s) a2 = mfma32s<T>(qf[s], kf[kb & 1][s], a2);
      float cs = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // registers 4 j .. 4 j + 3 hold query rows 8 j + 4 h + 0..3
        const float4 l4 = *reinterpret_cast<const float4*>(&s_lse[wave][8 * j + 4 * h]);
        cs += __builtin_amdgcn_exp2f(fmaf(a2[4 * j], scale_log2e, -l4.x));
        cs += __builtin_amdgcn_exp2f(fmaf(a2[4 * j + 1], scale_log2e, -l4.y));
        cs += __builtin_amdgcn_exp2f(fmaf(a2[4 * j + 2], scale_log2e, -l4.z));
        cs += __builtin_amdgcn_exp2f(fmaf(a2[4 * j + 3], scale_log2e, -l4.w));
      }
      colsum[kb] += cs;
    }
  }
  SC_RT(6);
  __syncthreads();  // every wave is done with its Q tile: the tiles become s_mass
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const float v = colsum[kb] + __shfl_xor(colsum[kb], 32, 64);
    if (h == 0) s_mass[wave][kb * 32 + r] = v;
  }
  __syncthreads();
  if (tid < M) {
    // + the reference's padded-row term: every padding row of a 64-row query tile adds 1/chunk per key (:385,:477)
    const float pad_rows = (float)(((M + 63) / 64) * 64 - M) * (float)G;
    mass[(size_t)(s0 + t0 + tid) * HKV + g] =
        s_mass[0][tid] + s_mass[1][tid] + s_mass[2][tid] + s_mass[3][tid] + pad_rows * pad_unit;
  }
  SC_RT(7);
}

// =====================================================================================================
// a5: leverage scores.  K1: X = K_h PHI (MFMA, fp32 out).  K2: per (chunk, head): centre, Gram + reg*I,
// Cholesky G = L L^T, score_i = xc_i^T G^-1 xc_i = ||L^-1 xc_i||^2  (all fp32).
// =====================================================================================================
constexpr int LV_KMAX = 64;  // sketch columns padded to two 32-wide MFMA blocks
template <typename T, int D>
__global__ __launch_bounds__(256) void sketch_kernel(const uint16_t* __restrict__ key, int64_t s_n, int64_t s_h,
                                                     const uint16_t* __restrict__ phi, float* __restrict__ X, int N,
                                                     int HKV, int kdim) {
  constexpr int KS = D / 16;
  __shared__ uint16_t s_phiT[LV_KMAX][D + 8];  // PHI^T, zero padded to 64 columns
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  for (int e = tid; e < LV_KMAX * D; e += 256) {
    const int col = e / D, d = e % D;
    s_phiT[col][d] = col < kdim ? phi[(size_t)d * kdim + col] : (uint16_t)0;
  }
  __syncthreads();
  s16x8 pf[2][KS];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int s = 0; s < KS; ++s)
      pf[cb][s] = *reinterpret_cast<const s16x8*>(&s_phiT[cb * 32 + r][16 * s + 8 * h]);

  const long nblk = ((long)N + 31) / 32;
  for (long blk = (long)blockIdx.x * 4 + wave; blk < nblk * HKV; blk += (long)gridDim.x * 4) {
    const int hh = (int)(blk % HKV);
    const long n0 = (blk / HKV) * 32;
    const long n = n0 + r;
    const bool valid = n < N;
    f32x16 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
    const uint16_t* kp = key + (size_t)(valid ? n : 0) * s_n + (size_t)hh * s_h + 8 * h;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const uint4 a = valid ? *reinterpret_cast<const uint4*>(kp + 16 * s) : make_uint4(0, 0, 0, 0);
      acc[0] = mfma32s<T>(__builtin_bit_cast(s16x8, a), pf[0][s], acc[0]);
      acc[1] = mfma32s<T>(__builtin_bit_cast(s16x8, a), pf[1][s], acc[1]);
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int col = cb * 32 + r;
      if (col < kdim) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const long nn = n0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (nn < N) X[((size_t)hh * N + nn) * kdim + col] = acc[cb][i];
        }
      }
    }
  }
}

constexpr int LV_KD = 48;   // supported sketch dim (LLMConfig.leverage_sketch_size default, engine_config.py)
constexpr int LV_LD = 49;   // padded leading dimension in LDS
__global__ __launch_bounds__(256) void leverage_solve_kernel(const float* __restrict__ X, float* __restrict__ scores,
                                                             const int* __restrict__ chunk_cu, int N, int HKV,
                                                             float reg) {
  __shared__ float s_mean[LV_KD];
  __shared__ float s_G[LV_KD * LV_LD];    // Gram -> Cholesky factor L (lower)
  __shared__ float s_Li[LV_KD];           // 1 / L[k][k]
  __shared__ __attribute__((aligned(16))) float s_tile[64 * LV_LD];  // 64 centred rows; later L repacked [48][48]
  const int cidx = blockIdx.x / HKV, hh = blockIdx.x % HKV;
  const int beg = chunk_cu[cidx], end = chunk_cu[cidx + 1];
  const int L = end - beg;
  if (L <= 0) return;
  const int tid = threadIdx.x;
  const float* Xh = X + ((size_t)hh * N + beg) * LV_KD;

  // Two passes over the chunk in 64-row tiles (column sums, then the centred Gram).  A tile is 64 x 48 floats =
  // 12 coalesced loads per thread; the next tile's loads are in flight while the current one is consumed from LDS
  // (the first version's column means were 102 dependent strided global loads per thread).
  constexpr int TPT = 64 * LV_KD / 256;  // tile elements per thread
  float pre[TPT];
  auto tile_load = [&](int i0) {
#pragma unroll
    for (int j = 0; j < TPT; ++j) {
      const int e = tid + 256 * j, rr = e / LV_KD;
      pre[j] = (i0 + rr) < L ? Xh[(size_t)i0 * LV_KD + e] : 0.f;
    }
  };
  // ---- pass A: column means
  {
    const int col = tid % LV_KD, sl = tid / LV_KD;  // 5 row slices for tid < 240
    float sum = 0.f;
    tile_load(0);
    for (int i0 = 0; i0 < L; i0 += 64) {
#pragma unroll
      for (int j = 0; j < TPT; ++j) {
        const int e = tid + 256 * j;
        s_tile[(e / LV_KD) * LV_LD + e % LV_KD] = pre[j];
      }
      __syncthreads();
      if (i0 + 64 < L) tile_load(i0 + 64);
      if (tid < 5 * LV_KD)
        for (int rr = sl; rr < 64; rr += 5) sum += s_tile[rr * LV_LD + col];  // rows past L are zero
      __syncthreads();
    }
    s_tile[tid] = (tid < 5 * LV_KD) ? sum : 0.f;
    __syncthreads();
    if (tid < LV_KD) {
      float t = 0.f;
      for (int j = 0; j < 5; ++j) t += s_tile[j * LV_KD + tid];
      s_mean[tid] = t / (float)L;
    }
    __syncthreads();
  }
  // ---- pass B: Gram.  Thread (ta, tb) owns the 3 x 3 block G[3ta.., 3tb..] (16 x 16 threads cover 48 x 48): 6 LDS
  // reads feed 9 FMAs per row (one entry per thread index needed 18).  In a wave the a-operands are 4 broadcast groups
  // and the b-operands 16 addresses 3 floats apart: conflict free.
  static_assert(LV_KD == 48, "3 x 3 blocks on a 16 x 16 thread grid");
  const int ta = tid >> 4, tb = tid & 15;
  float gacc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) gacc[i][j] = 0.f;
  tile_load(0);
  for (int i0 = 0; i0 < L; i0 += 64) {
#pragma unroll
    for (int j = 0; j < TPT; ++j) {
      const int e = tid + 256 * j, rr = e / LV_KD, cc = e % LV_KD;
      s_tile[rr * LV_LD + cc] = (i0 + rr) < L ? pre[j] - s_mean[cc] : 0.f;
    }
    __syncthreads();
    if (i0 + 64 < L) tile_load(i0 + 64);
#pragma unroll 8
    for (int rr = 0; rr < 64; ++rr) {
      const float* row = s_tile + rr * LV_LD;
      const float a0 = row[3 * ta], a1 = row[3 * ta + 1], a2 = row[3 * ta + 2];
      const float b0 = row[3 * tb], b1 = row[3 * tb + 1], b2 = row[3 * tb + 2];
      gacc[0][0] = fmaf(a0, b0, gacc[0][0]); gacc[0][1] = fmaf(a0, b1, gacc[0][1]); gacc[0][2] = fmaf(a0, b2, gacc[0][2]);
      gacc[1][0] = fmaf(a1, b0, gacc[1][0]); gacc[1][1] = fmaf(a1, b1, gacc[1][1]); gacc[1][2] = fmaf(a1, b2, gacc[1][2]);
      gacc[2][0] = fmaf(a2, b0, gacc[2][0]); gacc[2][1] = fmaf(a2, b1, gacc[2][1]); gacc[2][2] = fmaf(a2, b2, gacc[2][2]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ar = 3 * ta + i, bc = 3 * tb + j;
      s_G[ar * LV_LD + bc] = gacc[i][j] + (ar == bc ? reg : 0.f);
    }
  __syncthreads();
  // Cholesky G = L L^T by ONE wave, entirely in registers: lane i owns row i (48 floats, compile-time indexed by
  // full unrolling); column j of L is broadcast to the other lanes with v_readlane.  ~2.3K instructions, no barrier,
  // no LDS round trip per step (a barrier-per-column workgroup version took ~10x longer).
  if (tid < 64) {
    const int lane = tid;
    const int rown = lane < LV_KD ? lane : LV_KD - 1;
    float a[LV_KD];
#pragma unroll
    for (int cc = 0; cc < LV_KD; ++cc) a[cc] = s_G[rown * LV_LD + cc];
#pragma unroll
    for (int j = 0; j < LV_KD; ++j) {
      const float d = sqrtf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[j]), j)));
      a[j] = (lane == j) ? d : a[j] / d;  // rows i > j: L[i][j]; rows i < j hold don't-care values
#pragma unroll
      for (int kk = j + 1; kk < LV_KD; ++kk) {
        const float lkj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[j]), kk));
        a[kk] = fmaf(-a[j], lkj, a[kk]);  // A[i][k] -= L[i][j] L[k][j]   (meaningful for i >= k)
      }
    }
    if (lane < LV_KD) {
#pragma unroll
      for (int cc = 0; cc < LV_KD; ++cc) s_G[lane * LV_LD + cc] = a[cc];  // lower triangle incl. diagonal = L
    }
  }
  __syncthreads();
  // L (strictly lower part, zero elsewhere) repacked with a 16-byte aligned leading dimension so that the forward
  // substitution reads four entries of a row per ds_read_b128 (every thread reads the same address: broadcast)
  float* s_L = s_tile;  // [48][48], the tile buffer is free now
  for (int e = tid; e < LV_KD * LV_KD; e += 256) {
    const int rr = e / LV_KD, cc = e % LV_KD;
    s_L[e] = cc < rr ? s_G[rr * LV_LD + cc] : 0.f;
  }
  if (tid < LV_KD) s_Li[tid] = 1.0f / s_G[tid * LV_LD + tid];  // reciprocal diagonal
  __syncthreads();
  // score_i = || L^-1 xc_i ||^2 by forward substitution, one row per thread, everything at compile-time offsets
  for (int i = tid; i < L; i += 256) {
    float y[LV_KD];
#pragma unroll
    for (int cc = 0; cc < LV_KD; ++cc) y[cc] = Xh[(size_t)i * LV_KD + cc] - s_mean[cc];
    float sc = 0.f;
#pragma unroll
    for (int kk = 0; kk < LV_KD; ++kk) {
      float t = y[kk];
#pragma unroll
      for (int m4 = 0; m4 < (kk + 3) / 4; ++m4) {  // entries m >= kk of row kk are zero: no masking needed
        const float4 l4 = *reinterpret_cast<const float4*>(s_L + kk * LV_KD + 4 * m4);
        t = fmaf(-l4.x, y[4 * m4], t);
        if (4 * m4 + 1 < kk) t = fmaf(-l4.y, y[4 * m4 + 1], t);
        if (4 * m4 + 2 < kk) t = fmaf(-l4.z, y[4 * m4 + 2], t);
        if (4 * m4 + 3 < kk) t = fmaf(-l4.w, y[4 * m4 + 3], t);
      }
      t *= s_Li[kk];
      y[kk] = t;
      sc = fmaf(t, t, sc);
      // keep hipcc from hoisting the L loads of all 48 steps to the top (381 registers = one workgroup per CU)
      __builtin_amdgcn_sched_barrier(0);
    }
    scores[(size_t)(beg + i) * HKV + hh] = fmaxf(sc, 0.f);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// a5 in ONE kernel for chunks of up to LV_FMAX rows (the engine's chunk_size = 512): sketch X = K_h PHI by MFMA straight
// into LDS (512 x 48 fp32 = 98 KB), then column means, centred Gram, in-register Cholesky and the forward substitution
// all read X from LDS.  HBM traffic = the chunk's keys once + its scores (the two-kernel path writes X in fp32 and
// re-reads it three times: 2.5 x the algorithmic bytes).  Arithmetic and summation orders are those of sketch_kernel +
// leverage_solve_kernel up to the Gram matrix, which is summed in two halves here (equal to fp32 rounding).
constexpr int LV_FMAX = 512;
constexpr int LV_FT = 512;  // threads: 8 waves (the single-wave Cholesky is the serial part; everything else scales)
template <typename T, int D>
__global__ __launch_bounds__(LV_FT) void leverage_fused_kernel(const uint16_t* __restrict__ key, int64_t s_n, int64_t s_h,
                                                             const uint16_t* __restrict__ phi,
                                                             float* __restrict__ scores,
                                                             const int* __restrict__ chunk_cu, int HKV, int kdim,
                                                             float reg) {
  constexpr int KS = D / 16;
  extern __shared__ __attribute__((aligned(16))) char lv_smem[];
  float* s_X = reinterpret_cast<float*>(lv_smem);                 // [LV_FMAX][LV_LD]
  float* s_G = s_X + LV_FMAX * LV_LD;                             // [48][LV_LD]
  float* s_L = s_G + LV_KD * LV_LD;                               // [48][48]
  float* s_mean = s_L + LV_KD * LV_KD;                            // [48]
  float* s_Li = s_mean + LV_KD;                                   // [48]
  float* s_part = s_Li + LV_KD;                                   // [256]
  float* s_G2 = s_part + 256;                                     // [48][LV_LD] Gram of the second half of the rows
  uint16_t(*s_phiT)[D + 8] = reinterpret_cast<uint16_t(*)[D + 8]>(lv_smem);  // aliases s_X until the fragments are loaded

  const int cidx = blockIdx.x / HKV, hh = blockIdx.x % HKV;
  const int beg = chunk_cu[cidx], end = chunk_cu[cidx + 1];
  const int L = end - beg;
  if (L <= 0) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;

  // ---- sketch: PHI^T fragments (as sketch_kernel), then 32-row blocks of the chunk, one per wave at a time
  for (int e = tid; e < LV_KMAX * D; e += LV_FT) {
    const int col = e / D, d = e % D;
    s_phiT[col][d] = col < kdim ? phi[(size_t)d * kdim + col] : (uint16_t)0;
  }
  __syncthreads();
  s16x8 pf[2][KS];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int s = 0; s < KS; ++s)
      pf[cb][s] = *reinterpret_cast<const s16x8*>(&s_phiT[cb * 32 + r][16 * s + 8 * h]);
  __syncthreads();  // s_phiT is dead from here: its bytes become s_X
  for (int n0 = wave * 32; n0 < L; n0 += (LV_FT / 64) * 32) {
    const int n = n0 + r;
    const bool valid = n < L;
    f32x16 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
    const uint16_t* kp = key + (size_t)(beg + (valid ? n : 0)) * s_n + (size_t)hh * s_h + 8 * h;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const uint4 a = valid ? *reinterpret_cast<const uint4*>(kp + 16 * s) : make_uint4(0, 0, 0, 0);
      acc[0] = mfma32s<T>(__builtin_bit_cast(s16x8, a), pf[0][s], acc[0]);
      acc[1] = mfma32s<T>(__builtin_bit_cast(s16x8, a), pf[1][s], acc[1]);
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int col = cb * 32 + r;
      if (col < LV_KD) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int nn = n0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (nn < LV_FMAX) s_X[nn * LV_LD + col] = nn < L ? acc[cb][i] : 0.f;  // rows past L: zeros (as the tile loads)
        }
      }
    }
  }
  // rows of the last partial 64-row tile that no wave touched must read as zero in the passes below
  {
    const int Lr = (L + 31) / 32 * 32, L64 = (L + 63) / 64 * 64;
    for (int e = tid; e < (L64 - Lr) * LV_KD; e += LV_FT) s_X[(Lr + e / LV_KD) * LV_LD + e % LV_KD] = 0.f;
  }
  __syncthreads();
  // ---- column means: thread (col, slice) sums rows slice, slice+5, ... of every 64-row tile, tile after tile
  {
    const int col = tid % LV_KD, sl = tid / LV_KD;  // 5 row slices for tid < 240
    float sum = 0.f;
    if (tid < 5 * LV_KD)
      for (int i0 = 0; i0 < L; i0 += 64)
        for (int rr = sl; rr < 64; rr += 5) sum += s_X[(i0 + rr) * LV_LD + col];
    if (tid < 256) s_part[tid] = (tid < 5 * LV_KD) ? sum : 0.f;
    __syncthreads();
    if (tid < LV_KD) {
      float t = 0.f;
      for (int j = 0; j < 5; ++j) t += s_part[j * LV_KD + tid];
      s_mean[tid] = t / (float)L;
    }
    __syncthreads();
  }
  // ---- centre in place (rows past L stay zero), then the Gram with 3 x 3 register blocks
  {
    const int L64 = (L + 63) / 64 * 64;
    for (int e = tid; e < L64 * LV_KD; e += LV_FT) {
      const int rr = e / LV_KD, cc = e % LV_KD;
      s_X[rr * LV_LD + cc] = rr < L ? s_X[rr * LV_LD + cc] - s_mean[cc] : 0.f;
    }
    __syncthreads();
    // two halves of the thread block take the two halves of the 64-row tiles; their Gram matrices are added in LDS
    const int half = tid >> 8, ta = (tid & 255) >> 4, tb = tid & 15;
    const int nt = L64 / 64, r_lo = half == 0 ? 0 : (nt + 1) / 2 * 64, r_hi = half == 0 ? (nt + 1) / 2 * 64 : L64;
    float gacc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) gacc[i][j] = 0.f;
#pragma unroll 8
    for (int rr = r_lo; rr < r_hi; ++rr) {
      const float* row = s_X + rr * LV_LD;
      const float a0 = row[3 * ta], a1 = row[3 * ta + 1], a2 = row[3 * ta + 2];
      const float b0 = row[3 * tb], b1 = row[3 * tb + 1], b2 = row[3 * tb + 2];
      gacc[0][0] = fmaf(a0, b0, gacc[0][0]); gacc[0][1] = fmaf(a0, b1, gacc[0][1]); gacc[0][2] = fmaf(a0, b2, gacc[0][2]);
      gacc[1][0] = fmaf(a1, b0, gacc[1][0]); gacc[1][1] = fmaf(a1, b1, gacc[1][1]); gacc[1][2] = fmaf(a1, b2, gacc[1][2]);
      gacc[2][0] = fmaf(a2, b0, gacc[2][0]); gacc[2][1] = fmaf(a2, b1, gacc[2][1]); gacc[2][2] = fmaf(a2, b2, gacc[2][2]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int ar = 3 * ta + i, bc = 3 * tb + j;
        (half ? s_G2 : s_G)[ar * LV_LD + bc] = gacc[i][j] + ((ar == bc && !half) ? reg : 0.f);
      }
    __syncthreads();
    for (int e = tid; e < LV_KD * LV_LD; e += LV_FT) s_G[e] += s_G2[e];
    __syncthreads();
  }
  // ---- Cholesky by one wave in registers (see leverage_solve_kernel)
  if (tid < 64) {
    const int rown = lane < LV_KD ? lane : LV_KD - 1;
    float a[LV_KD];
#pragma unroll
    for (int cc = 0; cc < LV_KD; ++cc) a[cc] = s_G[rown * LV_LD + cc];
#pragma unroll
    for (int j = 0; j < LV_KD; ++j) {
      const float d = sqrtf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[j]), j)));
      a[j] = (lane == j) ? d : a[j] / d;
#pragma unroll
      for (int kk = j + 1; kk < LV_KD; ++kk) {
        const float lkj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[j]), kk));
        a[kk] = fmaf(-a[j], lkj, a[kk]);
      }
    }
    if (lane < LV_KD) {
#pragma unroll
      for (int cc = 0; cc < LV_KD; ++cc) s_G[lane * LV_LD + cc] = a[cc];
    }
  }
  __syncthreads();
  for (int e = tid; e < LV_KD * LV_KD; e += LV_FT) {
    const int rr = e / LV_KD, cc = e % LV_KD;
    s_L[e] = cc < rr ? s_G[rr * LV_LD + cc] : 0.f;
  }
  if (tid < LV_KD) s_Li[tid] = 1.0f / s_G[tid * LV_LD + tid];
  __syncthreads();
  // ---- score_i = || L^-1 xc_i ||^2, one row per thread, the centred row read from LDS
  for (int i = tid; i < L; i += LV_FT) {
    float y[LV_KD];
#pragma unroll
    for (int cc = 0; cc < LV_KD; ++cc) y[cc] = s_X[i * LV_LD + cc];
    float sc = 0.f;
#pragma unroll
    for (int kk = 0; kk < LV_KD; ++kk) {
      float t = y[kk];
#pragma unroll
      for (int m4 = 0; m4 < (kk + 3) / 4; ++m4) {
        const float4 l4 = *reinterpret_cast<const float4*>(s_L + kk * LV_KD + 4 * m4);
        t = fmaf(-l4.x, y[4 * m4], t);
        if (4 * m4 + 1 < kk) t = fmaf(-l4.y, y[4 * m4 + 1], t);
        if (4 * m4 + 2 < kk) t = fmaf(-l4.z, y[4 * m4 + 2], t);
        if (4 * m4 + 3 < kk) t = fmaf(-l4.w, y[4 * m4 + 3], t);
      }
      t *= s_Li[kk];
      y[kk] = t;
      sc = fmaf(t, t, sc);
      __builtin_amdgcn_sched_barrier(0);
    }
    scores[(size_t)(beg + i) * HKV + hh] = fmaxf(sc, 0.f);
  }
}
// ---------------------------------------------------------------------------------------------------------------------
// a5 in one kernel with the sketch held in REGISTERS (default for chunks <= 512 rows).  leverage_fused_kernel above
// keeps X (512 x 48 fp32 = 98 KB) in LDS, so one workgroup owns a CU, and its Gram and its per-row forward
// substitution run on the VALU (LDS-read bound / 1152 dependent FMAs per row: ~ 12 + 50 us at 32 K x 8).  Here a
// wave owns 128 rows of the chunk and keeps their sketch in its MFMA accumulators (4 blocks x 32 registers), the
// chunk's keys are read once, and every O(rows x 48 x 48) step runs on the matrix pipe in fp32:
//   1. X = K PHI (MFMA 32x32x16) for the wave's four 32-row blocks, the next block's key rows in flight;
//      column sums of X from the accumulators -> mu; X - mu in place (rows past the chunk: zero);
//   2. per block: accumulators -> wave-private LDS tile, Gram of the tile accumulated by fp32 MFMAs
//      (v_mfma_f32_16x16x4_f32: exact fp32 products, six 16 x 16 blocks of the symmetric 48 x 48 matrix); the waves'
//      partial Grams are summed in a fixed order;
//   3. one wave: Cholesky G = L L^T in registers (one v_rsq per column instead of sqrt + division), then W = L^-1
//      (lane c solves L w = e_c);
//   4. per block: tile again, Y = X W^T by fp32 MFMAs (the zero blocks of the triangular W skipped),
//      score_i = || y_i ||^2 = x_i^T G^-1 x_i.
// A workgroup is 4 waves, 45 KB of LDS and < 256 registers: two share a CU.  Sums are in a fixed order: bit-reproducible.
constexpr int LV2_T = 256;
constexpr int LV2_W = LV2_T / 64;     // waves = wave tiles
constexpr int LV2_NB = LV_FMAX / (32 * LV2_W);  // 32-row blocks per wave (4)
constexpr int LV2_TILE = 32 * LV_LD;  // floats per wave tile (1568 >= 6 * 256: also holds a wave's partial Gram)
constexpr int LV2_LP = 52;  // row pitch of L in LDS: 16-byte rows whose float4 stores by 16 lanes fall into distinct banks
constexpr size_t LV2_SMEM =
    (size_t)(LV2_W * LV2_TILE + LV_KD * LV_LD + LV_KD * LV2_LP + LV_KD + LV2_W * LV_KMAX + LV_KMAX) * sizeof(float);
template <typename T, int D>
__global__ __launch_bounds__(LV2_T, 2) void leverage_fused2_kernel(const uint16_t* __restrict__ key, int64_t s_n,
                                                                  int64_t s_h, const uint16_t* __restrict__ phi,
                                                                  float* __restrict__ scores,
                                                                  const int* __restrict__ chunk_cu, int HKV, int kdim,
                                                                  float reg) {
  constexpr int KS = D / 16;
  static_assert(D <= 128 && LV_KD == 48, "tile / block layout");
  extern __shared__ __attribute__((aligned(16))) char lv_smem[];
  float* s_tiles = reinterpret_cast<float*>(lv_smem);   // [LV2_W][32][LV_LD]
  float* s_G = s_tiles + LV2_W * LV2_TILE;              // [48][LV_LD]  Gram, then W = L^-1
  float* s_L = s_G + LV_KD * LV_LD;                     // [48][LV2_LP] L, strictly lower part (zero elsewhere)
  float* s_Li = s_L + LV_KD * LV2_LP;                   // [48]         1 / L[k][k]
  float* s_part = s_Li + LV_KD;                         // [LV2_W][64]  per-wave column sums
  float* s_mu = s_part + LV2_W * LV_KMAX;               // [64]
  uint16_t(*s_phiT)[D + 8] = reinterpret_cast<uint16_t(*)[D + 8]>(lv_smem);  // aliases the tiles until pf is loaded

  const int cidx = blockIdx.x / HKV, hh = blockIdx.x % HKV;
  const int beg = chunk_cu[cidx], end = chunk_cu[cidx + 1];
  const int L = end - beg;
  if (L <= 0) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const uint16_t* kh = key + (size_t)beg * s_n + (size_t)hh * s_h;
  const int w0 = wave * (32 * LV2_NB);  // first row of this wave

  // key fragments of a block (A operand: lane = row r, dims 16 s + 8 h ..); rows past the chunk read as zero.  ALL of
  // the wave's blocks are requested at the start, behind the PHI elements (loads return in order): with one block
  // fetched after the previous block's MFMAs the sketch phase was four exposed memory latencies (7 us of the kernel's
  // 53, profiles/r02_scoring_phase_stamps.txt)
  uint4 kr[LV2_NB][KS];
  auto load_block = [&](int b) __attribute__((always_inline)) {
    const int n = w0 + 32 * b + r;
    const bool valid = n < L;
    const uint16_t* kp = kh + (size_t)(valid ? n : 0) * s_n + 8 * h;
#pragma unroll
    for (int s = 0; s < KS; ++s) kr[b][s] = valid ? *reinterpret_cast<const uint4*>(kp + 16 * s) : make_uint4(0, 0, 0, 0);
  };
  SC_RT(0);
  // ---- PHI^T fragments (B operand of the sketch), as in sketch_kernel
  constexpr int PV = LV_KMAX * D / LV2_T;
  uint16_t pv[PV];
#pragma unroll
  for (int i = 0; i < PV; ++i) {
    const int e = tid + LV2_T * i, col = e / D, d = e % D;
    pv[i] = col < kdim ? phi[(size_t)d * kdim + col] : (uint16_t)0;
  }
#pragma unroll
  for (int b = 0; b < LV2_NB - 1; ++b) load_block(b);  // the last block follows the first block's MFMAs (registers)
#pragma unroll
  for (int i = 0; i < PV; ++i) {
    const int e = tid + LV2_T * i;
    s_phiT[e / D][e % D] = pv[i];
  }
  __syncthreads();
  SC_RT(1);
  f32x16 x[LV2_NB][2];  // the wave's sketch: x[b][cb][i] = X[row w0 + 32 b + (i&3) + 8 (i>>2) + 4 h][col 32 cb + r]
  {
    s16x8 pf[2][KS];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int s = 0; s < KS; ++s) pf[cb][s] = *reinterpret_cast<const s16x8*>(&s_phiT[cb * 32 + r][16 * s + 8 * h]);
    // ---- 1. sketch of the wave's blocks
#pragma unroll
    for (int b = 0; b < LV2_NB; ++b) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[b][cb][i] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        x[b][0] = mfma32s<T>(__builtin_bit_cast(s16x8, kr[b][s]), pf[0][s], x[b][0]);
        x[b][1] = mfma32s<T>(__builtin_bit_cast(s16x8, kr[b][s]), pf[1][s], x[b][1]);
      }
      if (b == 0) load_block(LV2_NB - 1);
    }
  }
  // column means: lane sums its rows, + the partner half, waves combined through LDS in a fixed order
  {
    float c0 = 0.f, c1 = 0.f;
#pragma unroll
    for (int b = 0; b < LV2_NB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) c0 += x[b][0][i], c1 += x[b][1][i];
    c0 += __shfl_xor(c0, 32, 64);
    c1 += __shfl_xor(c1, 32, 64);
    if (h == 0) s_part[wave * LV_KMAX + r] = c0, s_part[wave * LV_KMAX + 32 + r] = c1;
  SC_RT(2);
    __syncthreads();  // also: every wave has its PHI fragments, s_phiT may become the tiles
    if (tid < LV_KMAX) {
      float t = 0.f;
      for (int w = 0; w < LV2_W; ++w) t += s_part[w * LV_KMAX + tid];
      s_mu[tid] = t / (float)L;
    }
    __syncthreads();
  SC_RT(3);
    const float mu0 = s_mu[r], mu1 = s_mu[32 + r];
#pragma unroll
    for (int b = 0; b < LV2_NB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool in = w0 + 32 * b + (i & 3) + 8 * (i >> 2) + 4 * h < L;
        x[b][0][i] = in ? x[b][0][i] - mu0 : 0.f;
        x[b][1][i] = in ? x[b][1][i] - mu1 : 0.f;
      }
  }
  float* tile = s_tiles + wave * LV2_TILE;
  auto put_tile = [&](int b) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int rr = (i & 3) + 8 * (i >> 2) + 4 * h;
      tile[rr * LV_LD + r] = x[b][0][i];
      if (r < LV_KD - 32) tile[rr * LV_LD + 32 + r] = x[b][1][i];
    }
  };
  const int kq = lane >> 4, cl = lane & 15;  // fp32 MFMA 16x16x4: k index / row-or-column index of a lane's operand

  // ---- 2. Gram: six 16 x 16 blocks (ja <= jb) per wave, fp32 MFMA over the tile's rows.  (Round 2, not kept: Gram
  // straight from the accumulators - they already have the layout of both operands of v_mfma_f32_32x32x16_bf16 when
  // k runs over rows - with the fp32 values split into bf16 terms: hi + lo (72 MFMAs) cut this phase from 9.9 to 5.8 us
  // but leverage scores near 1 came out 1e-3 off; hi + mid + lo with six products (144 MFMAs) is fp32-exact to the
  // strict test and gains 1.2 us - not worth a second arithmetic path.)
  {
    f32x4 g[6];
#pragma unroll
    for (int bI = 0; bI < 6; ++bI) g[bI] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < LV2_NB; ++b) {
      if (w0 + 32 * b < L) {  // wave-uniform: blocks past the chunk are all zero
        put_tile(b);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const float* row = tile + (4 * ks + kq) * LV_LD + cl;
          const float f0 = row[0], f1 = row[16], f2 = row[32];
          g[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f0, f0, g[0], 0, 0, 0);
          g[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f0, f1, g[1], 0, 0, 0);
          g[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(f0, f2, g[2], 0, 0, 0);
          g[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(f1, f1, g[3], 0, 0, 0);
          g[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(f1, f2, g[4], 0, 0, 0);
          g[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(f2, f2, g[5], 0, 0, 0);
        }
      }
    }
    // partial blocks into the wave's own tile: [block][row 16][col 16]; C[row = 4 (lane / 16) + i][col = lane % 16]
#pragma unroll
    for (int bI = 0; bI < 6; ++bI)
#pragma unroll
      for (int i = 0; i < 4; ++i) tile[bI * 256 + (4 * kq + i) * 16 + cl] = g[bI][i];
    __syncthreads();
    for (int e = tid; e < LV_KD * LV_KD; e += LV2_T) {
      const int a = e / LV_KD, b = e % LV_KD;
      const int lo = a < b ? a : b, hi = a < b ? b : a;  // block (lo / 16, hi / 16) holds G[lo][hi]
      const int ja = lo >> 4, jb = hi >> 4;
      const int bI = ja == 0 ? jb : (ja == 1 ? 2 + jb : 5);
      float t = 0.f;
      for (int w = 0; w < LV2_W; ++w) t += s_tiles[w * LV2_TILE + bI * 256 + (lo & 15) * 16 + (hi & 15)];
      s_G[a * LV_LD + b] = t + (a == b ? reg : 0.f);
    }
    __syncthreads();
  SC_RT(4);
  }
  // ---- 3. one wave: Cholesky in registers (lane i owns row i; column j of L is broadcast with v_readlane), W = L^-1
  if (tid < 64) {
    const int rown = lane < LV_KD ? lane : LV_KD - 1;
    {
      float a[LV_KD];
      float dinv_mine = 0.f;
#pragma unroll
      for (int cc = 0; cc < LV_KD; ++cc) a[cc] = s_G[rown * LV_LD + cc];
#pragma clang loop unroll(full)
      for (int j = 0; j < LV_KD; ++j) {
        const float piv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[j]), j));
        const float dinv = __builtin_amdgcn_rsqf(piv);  // 1 / L[j][j]
        a[j] *= dinv;                                    // lane j: sqrt(piv) = L[j][j]; lanes i > j: L[i][j]
        dinv_mine = (lane == j) ? dinv : dinv_mine;
        // A[i][k] -= L[i][j] L[k][j] (meaningful for i >= k), eight columns at a time: eight v_readlane into eight
        // SGPRs, then eight FMAs.  Left to itself hipcc funnels every broadcast through ONE SGPR - readlane, two wait
        // states, FMA, 1 176 times: 7 of the 15.7 us this phase took (profiles/r02_scoring_phase_stamps.txt)
#pragma clang loop unroll(full)
        for (int k0 = j + 1; k0 < LV_KD; k0 += 8) {
          float lk[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (k0 + u < LV_KD)
              lk[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[j]), k0 + u));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (k0 + u < LV_KD) a[k0 + u] = fmaf(-a[j], lk[u], a[k0 + u]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (lane < LV_KD) {
#pragma unroll
        for (int c4 = 0; c4 < LV_KD / 4; ++c4)  // (scalar stores at a pitch of 48 floats: 16-way bank conflicts)
          *reinterpret_cast<float4*>(s_L + lane * LV2_LP + 4 * c4) =
              make_float4(4 * c4 < lane ? a[4 * c4] : 0.f, 4 * c4 + 1 < lane ? a[4 * c4 + 1] : 0.f,
                          4 * c4 + 2 < lane ? a[4 * c4 + 2] : 0.f, 4 * c4 + 3 < lane ? a[4 * c4 + 3] : 0.f);
        s_Li[lane] = dinv_mine;
      }
    }
    // W = L^-1, column c by lane c: forward substitution of e_c; L is read as broadcast float4s (same wave wrote it)
    {
      float w[LV_KD];
#pragma unroll
      for (int kk = 0; kk < LV_KD; ++kk) {
        float t = (kk == rown) ? 1.f : 0.f;
#pragma unroll
        for (int m4 = 0; m4 < (kk + 3) / 4; ++m4) {  // entries m >= kk of row kk are zero: no masking needed
          const float4 l4 = *reinterpret_cast<const float4*>(s_L + kk * LV2_LP + 4 * m4);
          t = fmaf(-l4.x, w[4 * m4], t);
          if (4 * m4 + 1 < kk) t = fmaf(-l4.y, w[4 * m4 + 1], t);
          if (4 * m4 + 2 < kk) t = fmaf(-l4.z, w[4 * m4 + 2], t);
          if (4 * m4 + 3 < kk) t = fmaf(-l4.w, w[4 * m4 + 3], t);
        }
        w[kk] = t * s_Li[kk];
        // anchor: w is only stored after the loop, so hipcc otherwise sinks the whole FMA chain below the 294 float4
        // loads of the 48 steps and spills every loaded value.  (Requesting row kk + 1 before row kk's chain: this
        // phase 13.2 -> 11.3 us, but 28 registers spilled around it - the sketch stays in 128 accumulators - and the
        // solve phase 11.5 -> 17.5 us: not kept.)
        asm volatile("" : "+v"(w[kk]));
        __builtin_amdgcn_sched_barrier(0);
      }
      if (lane < LV_KD) {
#pragma unroll
        for (int kk = 0; kk < LV_KD; ++kk) s_G[kk * LV_LD + lane] = w[kk];  // W[kk][c]; zero above the diagonal
      }
    }
  }
  __syncthreads();
  SC_RT(5);
  // ---- 4. Y = X W^T by fp32 MFMAs, score_i = sum_j Y[i][j]^2.  B operand: B[k = c][col = j] = W[16 jb + j][c]; the
  // blocks of W right of the diagonal block are zero: column block jb needs k-steps 0 .. 4 (jb + 1) - 1 only.
  {
    float wB[3][12];
#pragma unroll
    for (int jb = 0; jb < 3; ++jb)
#pragma unroll
      for (int ks = 0; ks < 4 * (jb + 1); ++ks) wB[jb][ks] = s_G[(16 * jb + cl) * LV_LD + 4 * ks + kq];
#pragma unroll
    for (int b = 0; b < LV2_NB; ++b) {
      const int n0 = w0 + 32 * b;
      if (n0 < L) {  // wave-uniform
        put_tile(b);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          f32x4 y[3];
#pragma unroll
          for (int jb = 0; jb < 3; ++jb) y[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 12; ++ks) {
            const float a = tile[(16 * nb + cl) * LV_LD + 4 * ks + kq];  // A[row = n][k = c] = Xc[n][c]
#pragma unroll
            for (int jb = 0; jb < 3; ++jb)
              if (ks < 4 * (jb + 1)) y[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wB[jb][ks], y[jb], 0, 0, 0);
          }
          // lane (kq, cl) holds Y[n = 16 nb + 4 kq + v][j = 16 jb + cl]: squares summed over jb, then over 16 lanes
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            float q = y[0][v] * y[0][v];
            q = fmaf(y[1][v], y[1][v], q);
            q = fmaf(y[2][v], y[2][v], q);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o, 64);
            const int n = n0 + 16 * nb + 4 * kq + v;
            if (cl == v && n < L) scores[(size_t)(beg + n) * HKV + hh] = fmaxf(q, 0.f);
          }
        }
      }
    }
  }
  SC_RT(6);
}

constexpr size_t LV_FUSED_SMEM =
    (size_t)(LV_FMAX * LV_LD + 2 * LV_KD * LV_LD + LV_KD * LV_KD + 2 * LV_KD + 256) * sizeof(float);

// =====================================================================================================
// a8: SnapKV.  rows = last w queries x G heads of a sequence; keys = [0, L-w).
//   K1 (grid b, g, group of SK_TPW key tiles of 128): (max, sum) of every window row over the group's keys.
//   K2 (same grid): recompute the group's logits with the key on the lane, sum exp2(s - lse) over rows -> raw score
//       of every key, then the trailing 5-tap mean clipped at the tile start (the reference's BLOCK_K, pinned to 128).
//   Between them snapkv_lse_kernel folds the groups' partials into the exact lse of every row.
// Where a pass's ~24 us go at 1 x 32 K x 8, w = 32 (profiles/r02_snapkv_phase_stamps.txt: s_memrealtime of all 512
// workgroups, and A/B builds): the whole grid is resident at once and asks for Q + three K tiles per workgroup at its
// start (48 MB outstanding), so the FIRST tile is usable only after ~7 us whatever the issue order; the 83 MB of a pass
// (K 67 MB + Q 16 MB re-read per workgroup) stream in ~12.5 us (a build without the MFMA / exp work: 13.4 us); the four
// tiles of a workgroup then compute back to back at 1.8-3.5 us per tile and the slowest workgroup exits 21 us in.
// Deeper prefetch (1 / 2 / 3 tiles), K fragments in batches, half the VALU work per logit in pass 1 and one barrier per
// tile instead of three each left the kernel time unchanged (23.4-24 us): the pass is a burst of memory latency
// followed by a dependent per-workgroup compute chain, not a bandwidth or an issue bound.  Kept from those
// experiments because they are simpler or cheaper: per-block max with the scale folded into the exp argument and the
// key mask only on a sequence's last tile (3.5 VALU per logit instead of 7.4), row state carried over the group's
// tiles (four times fewer partials), the K tile double-buffered in LDS (one barrier per tile), pass 2's column sums
// parked in LDS and pooled / stored once per workgroup.
// =====================================================================================================
__device__ __forceinline__ float sk_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
constexpr int SK_TILE = 128;
constexpr int SK_MAXQB = 8;  // 32-row query blocks (w*G <= 256)
constexpr int SK_TPW = 4;    // consecutive key tiles per workgroup (Q fragments loaded once)
constexpr int SK_TILE_BYTES = SK_TILE * 256;
constexpr size_t SK_SMEM = 2 * SK_TILE_BYTES + 4 * SK_TPW * SK_TILE * sizeof(float) + SK_MAXQB * 32 * sizeof(float);

// NJQ = 32-row query blocks per wave (1: w*G <= 128, 2: up to 256)
template <typename T, int D, int G, int NJQ, bool PASS2>
__global__ __launch_bounds__(256, 2) void snapkv_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                        int64_t sq_n, int64_t sk_n, int64_t sk_h,
                                                        float* __restrict__ scores, float* __restrict__ part,
                                                        const float* __restrict__ lse,
                                                        const int* __restrict__ cu_q, const int* __restrict__ cu_k,
                                                        int B, int HKV, int w_max, const int* __restrict__ w_b,
                                                        int ntile_max, float scale_log2e,
                                                        int pool, int pool_tile) {
  constexpr int KS = D / 16;
  constexpr int CH = D / 8;
  constexpr int KPT = SK_TILE * CH / 256;  // 16-byte chunks of a K tile per thread
  constexpr int PF = NJQ == 1 ? 2 : 1;     // K tiles in flight in registers behind the one being staged
  extern __shared__ __attribute__((aligned(16))) char sk_smem[];
  char* s_k = sk_smem;                                                       // [2][SK_TILE_BYTES]
  float* s_col = reinterpret_cast<float*>(sk_smem + 2 * SK_TILE_BYTES);      // [4 waves][SK_TPW * SK_TILE]
  float* s_lse = s_col + 4 * SK_TPW * SK_TILE;                               // [SK_MAXQB * 32]

  SC_RT(PASS2 ? 8 : 0);
  const int ngrp_max = (ntile_max + SK_TPW - 1) / SK_TPW;
  const int bid = blockIdx.x;
  const int g = bid % HKV;
  const int grp = (bid / HKV) % ngrp_max;
  const int b = bid / (HKV * ngrp_max);
  const int kb0 = cu_k[b], Lk = cu_k[b + 1] - kb0;
  const int qend = cu_q[b + 1];
  const int w = min(w_b ? w_b[b] : w_max, w_max);  // this sequence's window (snapkv.py:351-357: int or [B] tensor)
  const int keff = Lk - w;  // keys that are scored
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  if (w <= 0) {  // no window rows: every key's score is an empty sum (what the reference's loops leave: 0)
    if (PASS2 && grp == 0)
      for (int i = tid; i < Lk; i += 256) scores[(size_t)(kb0 + i) * HKV + g] = 0.f;
    return;
  }
  if (keff <= 0) {  // L <= w: every key is "recent" -> +inf (the reference leaves these rows uninitialised)
    if (PASS2 && grp == 0)
      for (int i = tid; i < Lk; i += 256) scores[(size_t)(kb0 + i) * HKV + g] = INFINITY;
    return;
  }
  const int ntile = (keff + SK_TILE - 1) / SK_TILE;
  const int tile0 = grp * SK_TPW, tile1 = min(ntile, tile0 + SK_TPW);
  if (tile0 >= ntile) return;
  const int rows_b = w * G;
  const int nqb = (rows_b + 31) / 32;

  uint4 kreg[PF][KPT];  // register ring; every index is a constant once the tile loop is unrolled
  auto gload = [&](int slot, int tile) {
    const int t0 = tile * SK_TILE, M = min(SK_TILE, keff - t0);
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int e = tid + 256 * j, row = e / CH, ch = e % CH;
      kreg[slot][j] = make_uint4(0, 0, 0, 0);
      if (row < M)
        kreg[slot][j] = *reinterpret_cast<const uint4*>(k + (size_t)(kb0 + t0 + row) * sk_n + (size_t)g * sk_h + ch * 8);
    }
  };
  auto lstore = [&](int slot, int buf) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int e = tid + 256 * j;
      *reinterpret_cast<uint4*>(s_k + buf * SK_TILE_BYTES + ktile_off(e / CH, e % CH)) = kreg[slot][j];
    }
  };
  // this wave's query blocks (qb = wave, wave + 4): fragments stay in registers for all tiles of the workgroup.
  // Issued BEFORE the K tiles: everything a workgroup asks for at its start arrives in about issue order while the whole
  // grid's first requests (48 MB) queue at the memory - with Q behind two K tiles the first MFMA waited 6.7 us
  s16x8 qf[NJQ][KS];
#pragma unroll
  for (int jq = 0; jq < NJQ; ++jq) {
    const int row = (wave + 4 * jq) * 32 + r;  // window row -> (query offset, local head): row = qoff*G + hq_local
    const bool valid_q = row < rows_b;
    const int qoff = row / G, hql = row % G;
    const uint16_t* qp = q + (size_t)(qend - w + (valid_q ? qoff : 0)) * sq_n + (size_t)(g * G + hql) * D + 8 * h;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      uint4 t = valid_q ? *reinterpret_cast<const uint4*>(qp + 16 * s) : make_uint4(0, 0, 0, 0);
      qf[jq][s] = __builtin_bit_cast(s16x8, t);
    }
  }
  if (PASS2) {
    // exact lse of every window row: reduced over the groups ONCE per (b, g) by snapkv_lse_kernel (also ahead of the K tiles:
    // loads return in order)
    const float* lp = lse + ((size_t)b * HKV + g) * (SK_MAXQB * 32);
    for (int row = tid; row < nqb * 32; row += 256) s_lse[row] = row < rows_b ? lp[row] : INFINITY;
  }
  SC_RT(PASS2 ? 9 : 1);
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (tile0 + p < tile1) gload(p, tile0 + p);
  lstore(0, 0);
  if (tile0 + PF < tile1) gload(0, tile0 + PF);
  __syncthreads();
  SC_RT(PASS2 ? 10 : 2);

  // pass 1: (max, sum) of this lane's half of the keys, per query block, carried over the group's tiles
  float mx[NJQ], sum[NJQ];
#pragma unroll
  for (int jq = 0; jq < NJQ; ++jq) mx[jq] = -INFINITY, sum[jq] = 0.f;
  // pass 2: -lse of the 16 query rows this lane's accumulators hold (rows past the window: -inf -> weight 0)
  float nlse[NJQ][16];
  if (PASS2) {
#pragma unroll
    for (int jq = 0; jq < NJQ; ++jq)
#pragma unroll
      for (int i = 0; i < 16; ++i) nlse[jq][i] = -s_lse[(wave + 4 * jq) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h];
  }

#pragma unroll
  for (int ti = 0; ti < SK_TPW; ++ti) {
    const int tile = tile0 + ti;
    if (tile >= tile1) break;  // workgroup-uniform
    const int t0 = tile * SK_TILE, M = min(SK_TILE, keff - t0);
    const bool full = M == SK_TILE;  // only a sequence's last tile needs the key mask
    const char* kt = s_k + (ti & 1) * SK_TILE_BYTES;
    // K fragments (A operand of pass 1, B operand of pass 2: the same layout) come from LDS in batches of one 32-key
    // block, the NEXT block's batch issued before the current block's MFMAs
    s16x8 kf[2][KS];
    auto load_kf = [&](int slot, int kb) {
#pragma unroll
      for (int s = 0; s < KS; ++s)
        kf[slot][s] = *reinterpret_cast<const s16x8*>(kt + ktile_off(kb * 32 + r, 2 * s + h));
    };
    float colsum[4] = {0.f, 0.f, 0.f, 0.f};
    load_kf(0, 0);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (kb < 3) load_kf((kb + 1) & 1, kb + 1);
#pragma unroll
      for (int jq = 0; jq < NJQ; ++jq) {
        if (wave + 4 * jq >= nqb) continue;  // wave-uniform
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        if (!PASS2) {
#pragma unroll
          for (int s = 0; s < KS; ++s) acc = mfma32s<T>(kf[kb & 1][s], qf[jq][s], acc);
          // acc[i]: key kb*32 + (i & 3) + 8 (i >> 2) + 4 h, query row (wave + 4 jq) * 32 + r.  Block max on the raw
          // logits (scale > 0), the scale folded into the exp argument; the block's (max, sum) then joins the running
          // pair - nothing per element depends on the blocks before
          if (!full) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h < M ? acc[i] : -INFINITY;
          }
          float m0 = sk_max3(acc[0], acc[1], acc[2]), m1 = sk_max3(acc[3], acc[4], acc[5]);
          m0 = sk_max3(m0, acc[6], acc[7]), m1 = sk_max3(m1, acc[8], acc[9]);
          m0 = sk_max3(m0, acc[10], acc[11]), m1 = sk_max3(m1, acc[12], acc[13]);
          const float mb = sk_max3(m0, m1, fmaxf(acc[14], acc[15])) * scale_log2e;
          const float mbs = (!full && mb == -INFINITY) ? 0.f : mb;  // a lane whose 16 keys are all past the end
          float bs0 = 0.f, bs1 = 0.f;
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            bs0 += __builtin_amdgcn_exp2f(fmaf(acc[i], scale_log2e, -mbs));
            bs1 += __builtin_amdgcn_exp2f(fmaf(acc[i + 1], scale_log2e, -mbs));
          }
          const float mn = fmaxf(mx[jq], mb);
          const float mns = (!full && mn == -INFINITY) ? 0.f : mn;
          sum[jq] = sum[jq] * __builtin_amdgcn_exp2f(mx[jq] - mns) + (bs0 + bs1) * __builtin_amdgcn_exp2f(mb - mns);
          mx[jq] = mn;
        } else {
#pragma unroll
          for (int s = 0; s < KS; ++s) acc = mfma32s<T>(qf[jq][s], kf[kb & 1][s], acc);
          // acc[i]: query row (wave + 4 jq) * 32 + (i & 3) + 8 (i >> 2) + 4 h, key kb*32 + r
          float c0 = 0.f, c1 = 0.f;
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            c0 += __builtin_amdgcn_exp2f(fmaf(acc[i], scale_log2e, nlse[jq][i]));
            c1 += __builtin_amdgcn_exp2f(fmaf(acc[i + 1], scale_log2e, nlse[jq][i + 1]));
          }
          colsum[kb] += c0 + c1;
        }
      }
    }
    if (PASS2) {
      // both halves of the query rows (lanes l, l ^ 32), then parked per wave until the workgroup's epilogue
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(colsum[kb]), __float_as_uint(colsum[kb]), false, false);
        if (h == 0) s_col[(wave * SK_TPW + ti) * SK_TILE + kb * 32 + r] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
    }
    // the next tile goes to the other LDS buffer (its readers passed the barrier that closed the previous iteration);
    // its register slot is then free for the tile after the ones in flight
    if (tile + 1 < tile1) {
      lstore((ti + 1) % PF, (ti + 1) & 1);
      if (tile + 1 + PF < tile1) gload((ti + 1) % PF, tile + 1 + PF);
    }
    __syncthreads();
    SC_RT((PASS2 ? 11 : 3) + ti);
  }

  if (!PASS2) {
    // the two key halves of a row (lanes l, l ^ 32) merged in a fixed order; one (max, sum) per row and GROUP
#pragma unroll
    for (int jq = 0; jq < NJQ; ++jq) {
      const int row = (wave + 4 * jq) * 32 + r;
      const auto sm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx[jq]), __float_as_uint(mx[jq]), false, false);
      const auto ss = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum[jq]), __float_as_uint(sum[jq]), false, false);
      const float m_lo = __uint_as_float(sm[0]), m_hi = __uint_as_float(sm[1]);
      const float mn = fmaxf(m_lo, m_hi);  // finite: key 0 of the group's first tile is valid and sits in the low half
      const float tot = __uint_as_float(ss[0]) * __builtin_amdgcn_exp2f(m_lo - mn) +
                        __uint_as_float(ss[1]) * __builtin_amdgcn_exp2f(m_hi - mn);
      if (h == 0 && row < rows_b) {
        float* pp = part + ((((size_t)b * HKV + g) * ngrp_max + grp) * (SK_MAXQB * 32) + row) * 2;
        pp[0] = mn;
        pp[1] = tot;
      }
    }
  } else {
    // epilogue, once per workgroup: the four waves' column sums, then the causal avg-pool, kernel `pool`, clipped
    // at the start of the key's pool_tile-wide block counted from the sequence start (snapkv.py:253-262: the reference's
    // autotuned BLOCK_K in {32, 64, 128}; pool_tile divides SK_TILE, so a block never straddles a workgroup's tiles)
    const int nk = min(keff - tile0 * SK_TILE, (tile1 - tile0) * SK_TILE);
    constexpr int WS = SK_TPW * SK_TILE;
    for (int i = tid; i < nk; i += 256) s_col[i] = s_col[i] + s_col[WS + i] + s_col[2 * WS + i] + s_col[3 * WS + i];
    __syncthreads();
    for (int i = tid; i < nk; i += 256) {
      const int j = i & (pool_tile - 1);
      const int lo = max(0, j - (pool - 1));
      float t = 0.f;
      for (int jj = lo; jj <= j; ++jj) t += s_col[i - j + jj];
      scores[(size_t)(kb0 + tile0 * SK_TILE + i) * HKV + g] = t / (float)(j - lo + 1);
    }
    if (tile0 == 0)  // last w keys <- +inf (snapkv.py:267-276)
      for (int i = tid; i < w; i += 256) scores[(size_t)(kb0 + keff + i) * HKV + g] = INFINITY;
  }
  SC_RT(PASS2 ? 15 : 7);
}

// lse[b, g, row] = log2-domain log-sum-exp of window row `row` over all scored keys, from the per-group (max, sum)
// partials of pass 1.  One WAVE per (b, g, row): lane l folds groups l, l + 64, ... (independent loads, fixed order),
// then the 64 lane states are merged by a fixed shuffle tree - reproducible.  (The first version gave a row to one
// thread per 16-tile slice, 32 workgroups in all: 24 us of mostly exposed load latency at 256 tiles.)
constexpr int SK_LSE_T = 256;  // 4 rows per workgroup
__global__ __launch_bounds__(SK_LSE_T) void snapkv_lse_kernel(const float* __restrict__ part, float* __restrict__ lse,
                                                              const int* __restrict__ cu_k, int HKV, int w_max,
                                                              const int* __restrict__ w_b, int G, int ntile_max) {
  constexpr int ROWS = SK_MAXQB * 32;
  const int lane = threadIdx.x & 63;
  const int gr = blockIdx.x * (SK_LSE_T / 64) + (threadIdx.x >> 6);  // global row index: (b * HKV + g) * ROWS + row
  const int bg = gr / ROWS, row = gr % ROWS, b = bg / HKV;
  const int w = min(w_b ? w_b[b] : w_max, w_max);
  const int keff = cu_k[b + 1] - cu_k[b] - w;
  const int ntile = keff > 0 ? ((keff + SK_TILE - 1) / SK_TILE + SK_TPW - 1) / SK_TPW : 0;  // groups of SK_TPW tiles
  const int ngrp_max = (ntile_max + SK_TPW - 1) / SK_TPW;
  if (row >= w * G) return;  // wave-uniform
  const float2* pp = reinterpret_cast<const float2*>(part) + (size_t)bg * ngrp_max * ROWS + row;
  float m = -INFINITY, ssum = 0.f;
  for (int t0 = lane; t0 < ntile; t0 += 4 * 64) {
    float2 ms[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      ms[u] = t0 + 64 * u < ntile ? pp[(size_t)(t0 + 64 * u) * ROWS] : make_float2(-INFINITY, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float mn = fmaxf(m, ms[u].x);
      const float mn_safe = mn == -INFINITY ? 0.f : mn;
      ssum = ssum * __builtin_amdgcn_exp2f(m - mn_safe) + ms[u].y * __builtin_amdgcn_exp2f(ms[u].x - mn_safe);
      m = mn;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float mo = __shfl_xor(m, o, 64), so = __shfl_xor(ssum, o, 64);
    const float mn = fmaxf(m, mo);
    const float mn_safe = mn == -INFINITY ? 0.f : mn;
    // the partner computes the same two products in the other order; float addition commutes: identical in both lanes
    const float a = ssum * __builtin_amdgcn_exp2f(m - mn_safe), c2 = so * __builtin_amdgcn_exp2f(mo - mn_safe);
    ssum = (lane & o) ? c2 + a : a + c2;
    m = mn;
  }
  if (lane == 0) lse[(size_t)bg * ROWS + row] = m + __builtin_amdgcn_logf(ssum);
}

}  // namespace cvllm

using namespace cvllm;

// ---------------------------------------------------------------------------------------------------------
extern "C" size_t cvllm_zscore_workspace_bytes(int n_segments) {
  return n_segments > 0 ? (size_t)n_segments * ZS_MAXTILES * 2 * sizeof(float) : 0;
}

static int zscore_impl(void* x, int score_dtype, const int32_t* cu, int n_segments, int H, const void* accum,
                       int accum_dtype, float blend, const int32_t* prot_ranges, int n_ranges, int total_rows,
                       void* workspace, size_t workspace_bytes, const int32_t* trim_b, int trim, float eps,
                       cvllm_stream_t stream) {
  if (!x || (n_segments > 0 && !cu) || H <= 0 || n_segments < 0 || n_ranges < 0) return CVLLM_ERR_ARG;
  if (score_dtype < 0 || score_dtype > 2 || (accum && (accum_dtype < 0 || accum_dtype > 2))) return CVLLM_ERR_SHAPE;
  if (n_ranges > 0 && !prot_ranges) return CVLLM_ERR_ARG;
  if (n_segments > 0 && (!workspace || workspace_bytes < cvllm_zscore_workspace_bytes(n_segments)))
    return CVLLM_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (n_segments > 0) {
    // tiles per segment from the AVERAGE segment size (host ints only): ~4096 elements per tile
    long avg = ((long)total_rows / n_segments) * H;
    int T = (int)((avg + 4095) / 4096);
    T = T < 1 ? 1 : (T > ZS_MAXTILES ? ZS_MAXTILES : T);
    float* part = (float*)workspace;
    dim3 grid(n_segments, T);
    if (score_dtype == 0) hipLaunchKernelGGL((zscore_partial_kernel<0>), grid, dim3(ZS_T), 0, st, x, cu, H, T, part, trim_b, trim);
    else if (score_dtype == 1) hipLaunchKernelGGL((zscore_partial_kernel<1>), grid, dim3(ZS_T), 0, st, x, cu, H, T, part, trim_b, trim);
    else hipLaunchKernelGGL((zscore_partial_kernel<2>), grid, dim3(ZS_T), 0, st, x, cu, H, T, part, trim_b, trim);
#define ZS(DT, ADT)                                                                                                  \
  hipLaunchKernelGGL((zscore_apply_kernel<DT, ADT>), grid, dim3(ZS_T), 0, st, x, cu, H, T, part, accum, blend, trim_b, \
                     trim, eps)
    const int adt = accum ? accum_dtype : 2;
    switch (score_dtype * 3 + adt) {
      case 0: ZS(0, 0); break;
      case 1: ZS(0, 1); break;
      case 2: ZS(0, 2); break;
      case 3: ZS(1, 0); break;
      case 4: ZS(1, 1); break;
      case 5: ZS(1, 2); break;
      case 6: ZS(2, 0); break;
      case 7: ZS(2, 1); break;
      default: ZS(2, 2); break;
    }
#undef ZS
  }
  if (n_ranges > 0) {
    if (score_dtype == 0) hipLaunchKernelGGL((fill_inf_kernel<0>), dim3(n_ranges), dim3(256), 0, st, x, prot_ranges, H, total_rows);
    else if (score_dtype == 1) hipLaunchKernelGGL((fill_inf_kernel<1>), dim3(n_ranges), dim3(256), 0, st, x, prot_ranges, H, total_rows);
    else hipLaunchKernelGGL((fill_inf_kernel<2>), dim3(n_ranges), dim3(256), 0, st, x, prot_ranges, H, total_rows);
  }
  return check_launch();
}

extern "C" int cvllm_zscore_segments(void* x, int score_dtype, const int32_t* cu, int n_segments, int H,
                                     const void* accum, int accum_dtype, float blend, const int32_t* prot_ranges,
                                     int n_ranges, int total_rows, void* workspace, size_t workspace_bytes,
                                     cvllm_stream_t stream) {
  return zscore_impl(x, score_dtype, cu, n_segments, H, accum, accum_dtype, blend, prot_ranges, n_ranges, total_rows,
                     workspace, workspace_bytes, nullptr, 0, 0.f, stream);
}

extern "C" int cvllm_zscore_windowed(void* x, int score_dtype, const int32_t* cu, const int32_t* trim_b, int trim,
                                     int n_segments, int H, float eps, int total_rows, void* workspace,
                                     size_t workspace_bytes, cvllm_stream_t stream) {
  if (trim < 0 || eps < 0.f) return CVLLM_ERR_ARG;
  return zscore_impl(x, score_dtype, cu, n_segments, H, nullptr, 2, 0.f, nullptr, 0, total_rows, workspace,
                     workspace_bytes, trim_b, trim, eps, stream);
}

// ---------------------------------------------------------------------------------------------------------
template <typename T, int D>
static int chunk_mass_g(int G, const void* q, const void* k, int64_t sq_n, int64_t sk_n, int64_t sk_h, float* mass,
                        const int* cu, int B, int HKV, int nchunk, float scale, hipStream_t st) {
  const float c = scale * 1.4426950408889634f, pu = 1.0f / (float)CM_CHUNK;
  dim3 grid(B * nchunk * HKV), block(256);
#define CM(G_)                                                                                                       \
  {                                                                                                                    \
    auto kern = chunk_mass_kernel<T, D, G_>;                                                                           \
    set_dyn_lds_once(kern, CM_SMEM);                                                                                   \
    hipLaunchKernelGGL(kern, grid, block, CM_SMEM, st, (const uint16_t*)q, (const uint16_t*)k, sq_n, sk_n, sk_h, mass, \
                       cu, B, HKV, nchunk, c, pu);                                                                     \
  }
  switch (G) {
    case 1: CM(1); break;
    case 2: CM(2); break;
    case 4: CM(4); break;
    case 8: CM(8); break;
    default: return CVLLM_ERR_SHAPE;
  }
#undef CM
  return check_launch();
}

extern "C" int cvllm_chunk_attn_mass(const void* q, const void* k, int64_t sq_n, int64_t sk_n, int64_t sk_h,
                                     float* mass, const int32_t* cu_seqlens, int B, int total_tokens, int max_seqlen,
                                     int HQ, int HKV, int D, int chunk_size, float sm_scale, int dtype,
                                     cvllm_stream_t stream) {
  if (!q || !k || !mass || !cu_seqlens) return CVLLM_ERR_ARG;
  if (B <= 0 || HQ <= 0 || HKV <= 0 || total_tokens < 0) return CVLLM_ERR_ARG;
  if (HQ % HKV != 0 || chunk_size != CM_CHUNK) return CVLLM_ERR_SHAPE;  // reference hard-codes chunk 128 (:17)
  if ((sq_n % 8) || (sk_n % 8) || (sk_h % 8)) return CVLLM_ERR_SHAPE;
  if (total_tokens == 0 || max_seqlen <= 0) return CVLLM_OK;
  const int G = HQ / HKV;
  const int nchunk = (max_seqlen + CM_CHUNK - 1) / CM_CHUNK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVLLM_F16 && D == 128) return chunk_mass_g<F16, 128>(G, q, k, sq_n, sk_n, sk_h, mass, cu_seqlens, B, HKV, nchunk, sm_scale, st);
  if (dtype == CVLLM_F16 && D == 64) return chunk_mass_g<F16, 64>(G, q, k, sq_n, sk_n, sk_h, mass, cu_seqlens, B, HKV, nchunk, sm_scale, st);
  if (dtype == CVLLM_BF16 && D == 128) return chunk_mass_g<BF16, 128>(G, q, k, sq_n, sk_n, sk_h, mass, cu_seqlens, B, HKV, nchunk, sm_scale, st);
  if (dtype == CVLLM_BF16 && D == 64) return chunk_mass_g<BF16, 64>(G, q, k, sq_n, sk_n, sk_h, mass, cu_seqlens, B, HKV, nchunk, sm_scale, st);
  return CVLLM_ERR_SHAPE;
}

// ---------------------------------------------------------------------------------------------------------
extern "C" size_t cvllm_leverage_workspace_bytes(int total_tokens, int HKV, int sketch_dim) {
  if (total_tokens <= 0 || HKV <= 0 || sketch_dim <= 0) return 0;
  return (size_t)total_tokens * HKV * sketch_dim * sizeof(float);
}

extern "C" int cvllm_leverage_scores(const void* key_states, int64_t s_n, int64_t s_h, const void* phi, float* scores,
                                     const int32_t* chunk_cu, int n_chunks, int total_tokens, int HKV, int D,
                                     int sketch_dim, float regularizer, int dtype, int max_chunk_rows, void* workspace,
                                     size_t workspace_bytes, cvllm_stream_t stream) {
  if (!key_states || !phi || !scores || !chunk_cu) return CVLLM_ERR_ARG;
  if (n_chunks <= 0 || total_tokens <= 0 || HKV <= 0) return CVLLM_ERR_ARG;
  if (sketch_dim != LV_KD) return CVLLM_ERR_SHAPE;
  if ((s_n % 8) || (s_h % 8)) return CVLLM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (max_chunk_rows > 0 && max_chunk_rows <= LV_FMAX) {
    // one launch, no workspace: the recomputing kernel (two workgroups per CU), or with CVLLM_LEVERAGE=resident the
    // kernel that keeps the chunk's sketch in LDS (A/B measurements and the cross-check test)
    const char* lv_env = getenv("CVLLM_LEVERAGE");
    const bool resident = lv_env && lv_env[0] == 'r';
#define LF(T_, D_)                                                                                                     \
  {                                                                                                                    \
    static_assert((size_t)LV_KMAX * (D_ + 8) * 2 <= (size_t)LV_FMAX * LV_LD * 4, "PHI^T must fit inside the X image"); \
    static_assert((size_t)LV_KMAX * (D_ + 8) * 2 <= (size_t)LV2_W * LV2_TILE * 4, "PHI^T must fit inside the tiles");      \
    if (resident) {                                                                                                    \
      auto kern = leverage_fused_kernel<T_, D_>;                                                                       \
      set_dyn_lds_once(kern, (int)LV_FUSED_SMEM);                                                                      \
      hipLaunchKernelGGL(kern, dim3(n_chunks* HKV), dim3(LV_FT), LV_FUSED_SMEM, st, (const uint16_t*)key_states, s_n,   \
                         s_h, (const uint16_t*)phi, scores, chunk_cu, HKV, sketch_dim, regularizer);                   \
    } else {                                                                                                           \
      auto kern = leverage_fused2_kernel<T_, D_>;                                                                      \
      set_dyn_lds_once(kern, (int)LV2_SMEM);                                                                           \
      hipLaunchKernelGGL(kern, dim3(n_chunks* HKV), dim3(LV2_T), LV2_SMEM, st, (const uint16_t*)key_states, s_n, s_h,   \
                         (const uint16_t*)phi, scores, chunk_cu, HKV, sketch_dim, regularizer);                        \
    }                                                                                                                  \
  }
    if (dtype == CVLLM_F16 && D == 128) LF(F16, 128)
    else if (dtype == CVLLM_F16 && D == 64) LF(F16, 64)
    else if (dtype == CVLLM_BF16 && D == 128) LF(BF16, 128)
    else if (dtype == CVLLM_BF16 && D == 64) LF(BF16, 64)
    else return CVLLM_ERR_SHAPE;
#undef LF
    return check_launch();
  }
  if (!workspace || workspace_bytes < cvllm_leverage_workspace_bytes(total_tokens, HKV, sketch_dim))
    return CVLLM_ERR_WORKSPACE;
  float* X = (float*)workspace;
  long wave_tiles = (((long)total_tokens + 31) / 32) * HKV;
  int blocks = (int)((wave_tiles + 3) / 4);
  if (blocks > 1024) blocks = 1024;
#define SK(T_, D_)                                                                                                  \
  hipLaunchKernelGGL((sketch_kernel<T_, D_>), dim3(blocks), dim3(256), 0, st, (const uint16_t*)key_states, s_n, s_h, \
                     (const uint16_t*)phi, X, total_tokens, HKV, sketch_dim)
  if (dtype == CVLLM_F16 && D == 128) SK(F16, 128);
  else if (dtype == CVLLM_F16 && D == 64) SK(F16, 64);
  else if (dtype == CVLLM_BF16 && D == 128) SK(BF16, 128);
  else if (dtype == CVLLM_BF16 && D == 64) SK(BF16, 64);
  else return CVLLM_ERR_SHAPE;
#undef SK
  hipLaunchKernelGGL(leverage_solve_kernel, dim3(n_chunks * HKV), dim3(256), 0, st, X, scores, chunk_cu, total_tokens,
                     HKV, regularizer);
  return check_launch();
}

// ---------------------------------------------------------------------------------------------------------
extern "C" size_t cvllm_snapkv_workspace_bytes(int B, int HKV, int w, int max_seqlen_k) {
  if (B <= 0 || HKV <= 0 || max_seqlen_k <= 0) return 0;
  (void)w;
  const size_t ntile = ((size_t)max_seqlen_k + SK_TILE - 1) / SK_TILE;
  // per-tile (max, sum) partials of pass 1 + the reduced lse per window row
  return (size_t)B * HKV * (ntile * 2 + 1) * (SK_MAXQB * 32) * sizeof(float);
}

template <typename T, int D>
static int snapkv_g(int G, const void* q, const void* k, int64_t sq_n, int64_t sk_n, int64_t sk_h, float* scores,
                    float* part, const int* cu_q, const int* cu_k, int B, int HKV, int w, const int* w_b, int ntile,
                    float scale, int pool, int pool_tile, hipStream_t st) {
  const float c = scale * 1.4426950408889634f;
  dim3 grid(B * ((ntile + SK_TPW - 1) / SK_TPW) * HKV), block(256);
  float* lse = part + (size_t)B * HKV * ntile * (SK_MAXQB * 32) * 2;
  const bool two = w * G > 128;  // query blocks per wave
#define SNAP2(G_, NJQ_)                                                                                               \
  {                                                                                                                   \
    auto k1 = snapkv_kernel<T, D, G_, NJQ_, false>;                                                                   \
    auto k2 = snapkv_kernel<T, D, G_, NJQ_, true>;                                                                    \
    set_dyn_lds_once(k1, (int)SK_SMEM);                                                                               \
    set_dyn_lds_once(k2, (int)SK_SMEM);                                                                               \
    hipLaunchKernelGGL(k1, grid, block, SK_SMEM, st, (const uint16_t*)q, (const uint16_t*)k, sq_n, sk_n, sk_h,        \
                       scores, part, (const float*)lse, cu_q, cu_k, B, HKV, w, w_b, ntile, c, pool, pool_tile);       \
    hipLaunchKernelGGL(snapkv_lse_kernel, dim3(B* HKV*(SK_MAXQB * 32) / (SK_LSE_T / 64)), dim3(SK_LSE_T), 0, st,      \
                       (const float*)part, lse, cu_k, HKV, w, w_b, G_, ntile);                                        \
    hipLaunchKernelGGL(k2, grid, block, SK_SMEM, st, (const uint16_t*)q, (const uint16_t*)k, sq_n, sk_n, sk_h,        \
                       scores, part, (const float*)lse, cu_q, cu_k, B, HKV, w, w_b, ntile, c, pool, pool_tile);       \
  }
#define SNAP(G_)                                                                                                      \
  {                                                                                                                   \
    if (two) SNAP2(G_, 2) else SNAP2(G_, 1)                                                                           \
  }
  switch (G) {
    case 1: SNAP(1); break;
    case 2: SNAP(2); break;
    case 4: SNAP(4); break;
    case 8: SNAP(8); break;
    default: return CVLLM_ERR_SHAPE;
  }
#undef SNAP2
#undef SNAP
  return check_launch();
}

extern "C" int cvllm_snapkv_scores_wb(const void* q, const void* k, int64_t sq_n, int64_t sk_n, int64_t sk_h,
                                      float* scores, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                      const int32_t* window_b, int B, int HQ, int HKV, int D, int w, float sm_scale,
                                      int pool, int pool_tile, int max_seqlen_k, int dtype, void* workspace,
                                      size_t workspace_bytes, cvllm_stream_t stream) {
  if (!q || !k || !scores || !cu_seqlens_q || !cu_seqlens_k) return CVLLM_ERR_ARG;
  if (B <= 0 || HQ <= 0 || HKV <= 0 || max_seqlen_k <= 0 || pool <= 0) return CVLLM_ERR_ARG;
  if (pool_tile == 0) pool_tile = SK_TILE;
  if (pool_tile != 32 && pool_tile != 64 && pool_tile != 128) return CVLLM_ERR_SHAPE;  // the reference's three configs
  if (HQ % HKV != 0) return CVLLM_ERR_SHAPE;
  const int G = HQ / HKV;
  if (w <= 0 || w * G > SK_MAXQB * 32) return CVLLM_ERR_SHAPE;
  if ((sq_n % 8) || (sk_n % 8) || (sk_h % 8)) return CVLLM_ERR_SHAPE;
  if (!workspace || workspace_bytes < cvllm_snapkv_workspace_bytes(B, HKV, w, max_seqlen_k)) return CVLLM_ERR_WORKSPACE;
  const int ntile = (max_seqlen_k + SK_TILE - 1) / SK_TILE;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
#define SNAPD(T_, D_)                                                                                              \
  return snapkv_g<T_, D_>(G, q, k, sq_n, sk_n, sk_h, scores, part, cu_seqlens_q, cu_seqlens_k, B, HKV, w, window_b, \
                          ntile, sm_scale, pool, pool_tile, st)
  if (dtype == CVLLM_F16 && D == 128) SNAPD(F16, 128);
  if (dtype == CVLLM_F16 && D == 64) SNAPD(F16, 64);
  if (dtype == CVLLM_BF16 && D == 128) SNAPD(BF16, 128);
  if (dtype == CVLLM_BF16 && D == 64) SNAPD(BF16, 64);
#undef SNAPD
  return CVLLM_ERR_SHAPE;
}

extern "C" int cvllm_snapkv_scores(const void* q, const void* k, int64_t sq_n, int64_t sk_n, int64_t sk_h,
                                   float* scores, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, int B,
                                   int HQ, int HKV, int D, int w, float sm_scale, int pool, int pool_tile,
                                   int max_seqlen_k, int dtype, void* workspace, size_t workspace_bytes,
                                   cvllm_stream_t stream) {
  return cvllm_snapkv_scores_wb(q, k, sq_n, sk_n, sk_h, scores, cu_seqlens_q, cu_seqlens_k, nullptr, B, HQ, HKV, D, w,
                                sm_scale, pool, pool_tile, max_seqlen_k, dtype, workspace, workspace_bytes, stream);
}

#ifdef CVLLM_SC_TS
extern "C" void cvllm_debug_scoring_stamps(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(cvllm::g_sc_rt), sizeof(unsigned long long) * 1024 * 16);
}
#endif
