// a3 / a4 / a10 — paged KV-cache write kernels for gfx950 (pure HBM-bound row copies).
//
// Replaces cv/kv_cache/store_kv_cache.py: _prefill_store_all_kv_kernel (:251-319),
// _decode_store_kv_kernel (:374-416), _prefill_store_topk_kv_kernel (:9-78) and
// _prefill_store_topk_pad_kernel (:178-248).
//
// Every kernel moves whole rows of D 16-bit elements: D/8 lanes each move 16 B, so a row is one
// contiguous 2*D-byte read and one contiguous write; 64/(D/8) rows per wave-instruction.
#include "common.h"

namespace cvllm {

// ---- a4: decode append ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void decode_store_kernel(
    const uint16_t* __restrict__ key, const uint16_t* __restrict__ value, int64_t sk_b, int64_t sk_h,
    int64_t sv_b, int64_t sv_h, const int* __restrict__ bmap, int* __restrict__ bh_lens,
    const int* __restrict__ page_table, uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, int BH, int HKV,
    int PS, int NLP, int reserved, int lens_by_row) {
  constexpr int LPR = D / 8;
  const int row = (blockIdx.x * 256 + threadIdx.x) / LPR;  // (b,h) pair
  const int dl = threadIdx.x % LPR;
  if (row >= BH) return;
  const int b = row / HKV, h = row % HKV;
  const int bt = bmap[b];
  if (bt == reserved) return;  // padding row of a captured decode batch (store_kv_cache.py:395-397)
  const int lrow = lens_by_row ? bt * HKV + h : row;  // fused decode: the layer's full table, true batch row
  const int L = bh_lens[lrow];
  const int pg = page_table[((size_t)bt * HKV + h) * NLP + L / PS];
  const size_t dst = ((size_t)pg * PS + L % PS) * D + dl * 8;
  const uint4 kv = *reinterpret_cast<const uint4*>(key + b * sk_b + h * sk_h + dl * 8);
  const uint4 vv = *reinterpret_cast<const uint4*>(value + b * sv_b + h * sv_h + dl * 8);
  *reinterpret_cast<uint4*>(kc + dst) = kv;
  *reinterpret_cast<uint4*>(vc + dst) = vv;
  // every lane of the row has read L before any lane of the same wave stores it (same instruction stream)
  if (dl == 0) bh_lens[lrow] = L + 1;
}

// ---- a3: store every new row ------------------------------------------------------------------------
__device__ __forceinline__ int find_seq(const int* __restrict__ cu, int B, int n) {
  int lo = 0, hi = B - 1;  // largest b with cu[b] <= n
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (cu[mid] <= n) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <int D>
__global__ __launch_bounds__(256) void store_all_kernel(
    const uint16_t* __restrict__ key, const uint16_t* __restrict__ value, int64_t sk_n, int64_t sk_h,
    int64_t sv_n, int64_t sv_h, const int* __restrict__ cu, const int* __restrict__ bmap,
    const int* __restrict__ bh_lens, const int* __restrict__ page_table, uint16_t* __restrict__ kc,
    uint16_t* __restrict__ vc, int B, int N, int HKV, int PS, int NLP) {
  constexpr int LPR = D / 8;
  constexpr int RPB = 256 / LPR;  // rows per block iteration
  const int dl = threadIdx.x % LPR;
  const int rl = threadIdx.x / LPR;
  const long total = (long)N * HKV;
  for (long r = (long)blockIdx.x * RPB + rl; r < total; r += (long)gridDim.x * RPB) {
    const int n = (int)(r / HKV), h = (int)(r % HKV);
    const int b = find_seq(cu, B, n);
    const int t = n - cu[b];
    const int pos = bh_lens[b * HKV + h] + t;
    const int pg = page_table[((size_t)bmap[b] * HKV + h) * NLP + pos / PS];
    const size_t dst = ((size_t)pg * PS + pos % PS) * D + dl * 8;
    const uint4 kv = *reinterpret_cast<const uint4*>(key + (size_t)n * sk_n + h * sk_h + dl * 8);
    const uint4 vv = *reinterpret_cast<const uint4*>(value + (size_t)n * sv_n + h * sv_h + dl * 8);
    *reinterpret_cast<uint4*>(kc + dst) = kv;
    *reinterpret_cast<uint4*>(vc + dst) = vv;
  }
}

__global__ void add_seq_lens_kernel(const int* __restrict__ cu, int* __restrict__ bh_lens, int B, int HKV) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * HKV) return;
  const int b = i / HKV;
  bh_lens[i] += cu[b + 1] - cu[b];  // store_kv_cache.py:371
}

// ---- a10: compaction from per-(b,h) kept-token lists (token order) -------------------------------------
// kept_idx[B,H,max_seqlen] local token indices; count = new_lens - bh_lens0.
template <int D>
__global__ __launch_bounds__(256) void compact_store_kernel(
    const uint16_t* __restrict__ key, const uint16_t* __restrict__ value, int64_t sk_n, int64_t sk_h,
    int64_t sv_n, int64_t sv_h, const int* __restrict__ kept_idx, const int* __restrict__ new_lens,
    const int* __restrict__ cu, const int* __restrict__ bh_lens0, const int* __restrict__ page_table,
    const int* __restrict__ bmap, uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, int H, int max_seqlen,
    int PS, int NLP, int tiles_per_bh) {
  constexpr int LPR = D / 8;
  constexpr int RPB = 256 / LPR;
  const int bh = blockIdx.x / tiles_per_bh;
  const int tile = blockIdx.x % tiles_per_bh;
  const int b = bh / H, h = bh % H;
  const int L0 = bh_lens0[bh];
  const int cnt = new_lens[bh] - L0;
  const int dl = threadIdx.x % LPR;
  const int rl = threadIdx.x / LPR;
  const int* list = kept_idx + (size_t)bh * max_seqlen;
  const int* pt = page_table + ((size_t)bmap[b] * H + h) * NLP;
  const int n0 = cu[b];
  const int Lb1 = cu[b + 1] - n0 - 1;
  for (int j = tile * RPB + rl; j < cnt; j += tiles_per_bh * RPB) {
    // clamped into the sequence: a kept list left incomplete by a failed selection (cvllm_select_status) must never
    // turn into an out-of-range row
    const int n = n0 + max(0, min(list[j], Lb1));
    const int pos = L0 + j;
    const int pg = pt[pos / PS];
    const size_t dst = ((size_t)pg * PS + pos % PS) * D + dl * 8;
    const uint4 kv = *reinterpret_cast<const uint4*>(key + (size_t)n * sk_n + h * sk_h + dl * 8);
    const uint4 vv = *reinterpret_cast<const uint4*>(value + (size_t)n * sv_n + h * sv_h + dl * 8);
    *reinterpret_cast<uint4*>(kc + dst) = kv;
    *reinterpret_cast<uint4*>(vc + dst) = vv;
  }
}

// ---- f-3: compaction IN PLACE, for the last chunk of a chunked prefill -----------------------------------------------
// The chunks of a long prompt were written to the cache uncompressed (rows src_base + t for token t); selection then
// runs once over the whole sequence and the kept rows of every (b,h) move down to slots dst_base.. in token order.
// With an ascending kept list, slot j receives the row of token t_j >= j, so (dst_base <= src_base) a row never moves
// up.  One workgroup owns a (b,h) and walks its list in tiles of TILE rows: all rows of a tile are loaded into registers,
// the workgroup synchronises, then they are stored.  Tile i writes slots below dst_base + TILE*(i+1), and every source
// of a later tile lies at or above src_base + TILE*(i+1) - so later tiles can never read a slot an earlier tile wrote;
// inside a tile the barrier separates all reads from all writes.  The next tile's loads are issued before the current
// tile's stores (they cannot collide for the same reason), which keeps two tiles of traffic in flight.
template <int D>
__global__ __launch_bounds__(256) void compact_inplace_kernel(
    const int* __restrict__ kept_idx, const int* __restrict__ new_lens, const int* __restrict__ dst_base,
    const int* __restrict__ src_base, const int* __restrict__ page_table, const int* __restrict__ bmap,
    uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, int H, int max_seqlen, int PS, int NLP) {
  constexpr int LPR = D / 8;          // lanes per row
  constexpr int RPP = 256 / LPR;      // rows per pass of the workgroup
  constexpr int PASSES = 8;           // rows per thread and tile
  constexpr int TILE = RPP * PASSES;  // 128 rows at D = 128
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh % H;
  const int d0 = dst_base[bh], s0 = src_base[bh];
  const int cnt = new_lens[bh] - d0;
  const int tok_max = NLP * PS - 1 - s0;  // last row the page table can address
  const int dl = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  const int* list = kept_idx + (size_t)bh * max_seqlen;
  const int* pt = page_table + ((size_t)bmap[b] * H + h) * NLP;
  auto row_addr = [&](int pos) { return ((size_t)pt[pos / PS] * PS + pos % PS) * D + dl * 8; };
  uint4 kb[2][PASSES], vb[2][PASSES];
  auto load_tile = [&](int t0, uint4(&kk)[PASSES], uint4(&vv)[PASSES]) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int j = t0 + p * RPP + rl;
      if (j < cnt) {
        const size_t a = row_addr(s0 + max(0, min(list[j], tok_max)));  // clamped: see compact_store_kernel
        kk[p] = *reinterpret_cast<const uint4*>(kc + a);
        vv[p] = *reinterpret_cast<const uint4*>(vc + a);
      }
    }
  };
  auto store_tile = [&](int t0, const uint4(&kk)[PASSES], const uint4(&vv)[PASSES]) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int j = t0 + p * RPP + rl;
      if (j < cnt) {
        const size_t a = row_addr(d0 + j);
        *reinterpret_cast<uint4*>(kc + a) = kk[p];
        *reinterpret_cast<uint4*>(vc + a) = vv[p];
      }
    }
  };
  if (cnt <= 0) return;
  load_tile(0, kb[0], vb[0]);
  for (int t0 = 0; t0 < cnt; t0 += 2 * TILE) {
    if (t0 + TILE < cnt) load_tile(t0 + TILE, kb[1], vb[1]);
    __syncthreads();  // every row of tile t0 is in registers (the barrier also drains each wave's loads)
    store_tile(t0, kb[0], vb[0]);
    if (t0 + TILE >= cnt) break;
    if (t0 + 2 * TILE < cnt) load_tile(t0 + 2 * TILE, kb[0], vb[0]);
    __syncthreads();
    store_tile(t0 + TILE, kb[1], vb[1]);
  }
}

// ---- ranked-list variant (reference argument list): store_kv_cache.py:81-248 --------------------------
// One workgroup per sequence walks the rank list in order; the slot of rank r inside its head is the
// number of earlier ranks of the same head (deterministic replacement of the reference's atomics).
// Phase 1: ranks [0, retain) are all stored.  Phase 2 (pad): a head whose length is not a page multiple
// accepts further ranks of its own until full / exhausted / written >= ctx_len - L (reference :216-220).
constexpr int RK_MAXH = 64;
template <int D>
__global__ __launch_bounds__(256) void store_ranked_kernel(
    const uint16_t* __restrict__ key, const uint16_t* __restrict__ value, int64_t sk_n, int64_t sk_h,
    int64_t sv_n, int64_t sv_h, const int* __restrict__ indices, const int* __restrict__ retain,
    const int* __restrict__ page_table, const int* __restrict__ bmap, int* __restrict__ bh_lens,
    uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, const int* __restrict__ cu, int H, int max_sel, int PS,
    int NLP, int pad, int reserved) {
  constexpr int LPR = D / 8;
  __shared__ int s_base[RK_MAXH];   // running length of every head
  __shared__ int s_quota[RK_MAXH];  // remaining pad quota (phase 2)
  __shared__ int s_slot[256];
  __shared__ int s_tcnt[RK_MAXH];
  __shared__ int s_wcnt[4];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int bt = bmap[b];
  const int kt = retain[b];
  if (bt == reserved) return;  // kt == 0: nothing scattered, the pad phase still runs (reference :150-175)
  const int* idx = indices + (size_t)b * max_sel;
  if (tid < H) s_base[tid] = bh_lens[b * H + tid];
  __syncthreads();
  const int kmax = min(kt, max_sel);
  const int ctx = pad ? cu[b + 1] - cu[b] : 0;
  for (int phase = 0; phase < (pad ? 2 : 1); ++phase) {
    const int r_beg = phase == 0 ? 0 : kmax;
    const int r_end = phase == 0 ? kmax : max_sel;
    if (phase == 1) {
      if (tid < H) {
        const int L = s_base[tid];
        const int mod = L % PS;
        s_quota[tid] = mod == 0 ? 0 : min(PS - mod, max(ctx - L, 0));
      }
      __syncthreads();
    }
    for (int r0 = r_beg; r0 < r_end; r0 += 256) {
      if (phase == 1) {  // uniform early exit once every quota is used up
        int left = 0;
        for (int hh = 0; hh < H; ++hh) left += s_quota[hh];
        if (left == 0) break;
      }
      const int r = r0 + tid;
      int sel = -1, head = 0;
      if (r < r_end) {
        sel = idx[r];
        head = sel % H;
      }
      if (tid < H) s_tcnt[tid] = 0;
      __syncthreads();
      // rank of this entry among the tile's entries of the same head (serial over H heads, ballot based)
      int my = -1;
      for (int hh = 0; hh < H; ++hh) {
        const bool mine = sel >= 0 && head == hh;
        const unsigned long long bal = __ballot(mine);
        const int wave = tid >> 6, lane = tid & 63;
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wcnt[wave] = __popcll(bal);  // per-wave count (4 waves)
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += s_wcnt[w];
        const int tot = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        if (mine) my = woff + before;
        if (tid == 0) s_tcnt[hh] = tot;
        __syncthreads();
      }
      bool take = sel >= 0;
      if (phase == 1 && take) take = my < s_quota[head];
      if (take) s_slot[tid] = s_base[head] + my; else s_slot[tid] = -1;
      __syncthreads();
      if (tid < H) {
        const int used = phase == 0 ? s_tcnt[tid] : min(s_tcnt[tid], s_quota[tid]);
        s_base[tid] += used;
        if (phase == 1) s_quota[tid] -= used;
      }
      // copy the accepted rows of this tile: LPR lanes per row
      for (int e = tid / LPR; e < 256; e += 256 / LPR) {
        const int pos = s_slot[e];
        if (pos < 0) continue;
        const int er = r0 + e;
        const int s2 = idx[er];
        const int tok = s2 / H, hd = s2 % H;
        const int dl = tid % LPR;
        const int pg = page_table[((size_t)bt * H + hd) * NLP + pos / PS];
        const size_t dst = ((size_t)pg * PS + pos % PS) * D + dl * 8;
        *reinterpret_cast<uint4*>(kc + dst) =
            *reinterpret_cast<const uint4*>(key + (size_t)tok * sk_n + hd * sk_h + dl * 8);
        *reinterpret_cast<uint4*>(vc + dst) =
            *reinterpret_cast<const uint4*>(value + (size_t)tok * sv_n + hd * sv_h + dl * 8);
      }
      __syncthreads();
    }
  }
  __syncthreads();
  if (tid < H) bh_lens[b * H + tid] = s_base[tid];
}

}  // namespace cvllm

using namespace cvllm;

#define DISPATCH_D(D_, CALL)            \
  switch (D_) {                         \
    case 32: { constexpr int DD = 32; CALL; break; }   \
    case 64: { constexpr int DD = 64; CALL; break; }   \
    case 128: { constexpr int DD = 128; CALL; break; } \
    case 256: { constexpr int DD = 256; CALL; break; } \
    default: return CVLLM_ERR_SHAPE;    \
  }

int cvllm::store_decode_kv_impl(const void* key, const void* value, int64_t sk_b, int64_t sk_h, int64_t sv_b,
                                int64_t sv_h, const int32_t* batch_mapping, int32_t* bh_lens,
                                const int32_t* page_table, void* k_cache, void* v_cache, int B, int HKV, int D,
                                int page_size, int n_logical_pages_max, int reserved_batch, int dtype,
                                int lens_by_row, cvllm_stream_t stream) {
  if (!key || !value || !batch_mapping || !bh_lens || !page_table || !k_cache || !v_cache) return CVLLM_ERR_ARG;
  if (B <= 0 || HKV <= 0 || page_size <= 0 || n_logical_pages_max <= 0) return CVLLM_ERR_ARG;
  if (dtype != CVLLM_F16 && dtype != CVLLM_BF16) return CVLLM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int BH = B * HKV;
  DISPATCH_D(D, {
    const int rows_per_block = 256 / (DD / 8);
    hipLaunchKernelGGL((decode_store_kernel<DD>), dim3((BH + rows_per_block - 1) / rows_per_block), dim3(256), 0,
                       st, (const uint16_t*)key, (const uint16_t*)value, sk_b, sk_h, sv_b, sv_h, batch_mapping,
                       bh_lens, page_table, (uint16_t*)k_cache, (uint16_t*)v_cache, BH, HKV, page_size,
                       n_logical_pages_max, reserved_batch, lens_by_row);
  });
  return check_launch();
}

extern "C" int cvllm_store_decode_kv(const void* key, const void* value, int64_t sk_b, int64_t sk_h,
                                     int64_t sv_b, int64_t sv_h, const int32_t* batch_mapping, int32_t* bh_lens,
                                     const int32_t* page_table, void* k_cache, void* v_cache, int B, int HKV,
                                     int D, int page_size, int n_logical_pages_max, int reserved_batch,
                                     int dtype, cvllm_stream_t stream) {
  return cvllm::store_decode_kv_impl(key, value, sk_b, sk_h, sv_b, sv_h, batch_mapping, bh_lens, page_table, k_cache,
                                     v_cache, B, HKV, D, page_size, n_logical_pages_max, reserved_batch, dtype, 0,
                                     stream);
}

extern "C" int cvllm_store_all_kv(const void* new_keys, const void* new_values, int64_t sk_n, int64_t sk_h,
                                  int64_t sv_n, int64_t sv_h, const int32_t* cu_seqlens_k,
                                  const int32_t* batch_mapping, int32_t* bh_lens, const int32_t* page_table,
                                  void* k_cache, void* v_cache, int B, int total_tokens, int HKV, int D,
                                  int page_size, int n_logical_pages_max, int dtype, cvllm_stream_t stream) {
  if (!new_keys || !new_values || !cu_seqlens_k || !batch_mapping || !bh_lens || !page_table || !k_cache ||
      !v_cache)
    return CVLLM_ERR_ARG;
  if (B <= 0 || HKV <= 0 || page_size <= 0 || n_logical_pages_max <= 0 || total_tokens < 0) return CVLLM_ERR_ARG;
  if (dtype != CVLLM_F16 && dtype != CVLLM_BF16) return CVLLM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (total_tokens > 0) {
    DISPATCH_D(D, {
      const int rpb = 256 / (DD / 8);
      long blocks = ((long)total_tokens * HKV + rpb - 1) / rpb;
      if (blocks > 4096) blocks = 4096;  // grid-stride beyond 16 blocks per CU
      hipLaunchKernelGGL((store_all_kernel<DD>), dim3((int)blocks), dim3(256), 0, st, (const uint16_t*)new_keys,
                         (const uint16_t*)new_values, sk_n, sk_h, sv_n, sv_h, cu_seqlens_k, batch_mapping,
                         bh_lens, page_table, (uint16_t*)k_cache, (uint16_t*)v_cache, B, total_tokens, HKV,
                         page_size, n_logical_pages_max);
    });
  }
  hipLaunchKernelGGL(add_seq_lens_kernel, dim3((B * HKV + 255) / 256), dim3(256), 0, st, cu_seqlens_k, bh_lens, B,
                     HKV);
  return check_launch();
}

extern "C" int cvllm_compact_store(const void* new_keys, const void* new_vals, int64_t sk_n, int64_t sk_h,
                                   int64_t sv_n, int64_t sv_h, const int32_t* kept_idx, const int32_t* new_lens,
                                   const int32_t* cu_seqlens_k, const int32_t* bh_lens0,
                                   const int32_t* page_table, const int32_t* batch_mapping, void* k_cache,
                                   void* v_cache, int B, int H, int D, int max_seqlen, int page_size,
                                   int n_logical_pages_max, int dtype, cvllm_stream_t stream) {
  if (!new_keys || !new_vals || !kept_idx || !new_lens || !cu_seqlens_k || !bh_lens0 || !page_table ||
      !batch_mapping || !k_cache || !v_cache)
    return CVLLM_ERR_ARG;
  if (B <= 0 || H <= 0 || max_seqlen <= 0 || page_size <= 0 || n_logical_pages_max <= 0) return CVLLM_ERR_ARG;
  if (dtype != CVLLM_F16 && dtype != CVLLM_BF16) return CVLLM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_D(D, {
    const int rpb = 256 / (DD / 8);
    int tiles = (max_seqlen + rpb * 8 - 1) / (rpb * 8);  // ~8 row-groups per block
    if (tiles < 1) tiles = 1;
    if (tiles > 256) tiles = 256;
    hipLaunchKernelGGL((compact_store_kernel<DD>), dim3(B * H * tiles), dim3(256), 0, st,
                       (const uint16_t*)new_keys, (const uint16_t*)new_vals, sk_n, sk_h, sv_n, sv_h, kept_idx,
                       new_lens, cu_seqlens_k, bh_lens0, page_table, batch_mapping, (uint16_t*)k_cache,
                       (uint16_t*)v_cache, H, max_seqlen, page_size, n_logical_pages_max, tiles);
  });
  return check_launch();
}

extern "C" int cvllm_compact_cache_inplace(const int32_t* kept_idx, const int32_t* new_lens, const int32_t* dst_base,
                                          const int32_t* src_base, const int32_t* page_table,
                                          const int32_t* batch_mapping, void* k_cache, void* v_cache, int B, int H,
                                          int D, int max_seqlen, int page_size, int n_logical_pages_max, int dtype,
                                          cvllm_stream_t stream) {
  if (!kept_idx || !new_lens || !dst_base || !src_base || !page_table || !batch_mapping || !k_cache || !v_cache)
    return CVLLM_ERR_ARG;
  if (B <= 0 || H <= 0 || max_seqlen <= 0 || page_size <= 0 || n_logical_pages_max <= 0) return CVLLM_ERR_ARG;
  if (dtype != CVLLM_F16 && dtype != CVLLM_BF16) return CVLLM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_D(D, {
    hipLaunchKernelGGL((compact_inplace_kernel<DD>), dim3(B * H), dim3(256), 0, st, kept_idx, new_lens, dst_base,
                       src_base, page_table, batch_mapping, (uint16_t*)k_cache, (uint16_t*)v_cache, H, max_seqlen,
                       page_size, n_logical_pages_max);
  });
  return check_launch();
}

extern "C" int cvllm_store_topk_ranked(const void* new_keys, const void* new_vals, int64_t sk_n, int64_t sk_h,
                                       int64_t sv_n, int64_t sv_h, const int32_t* indices_topk,
                                       const int32_t* num_tokens_to_retain, const int32_t* page_table,
                                       const int32_t* batch_mapping, int32_t* bh_lens, void* k_cache,
                                       void* v_cache, const int32_t* cu_seqlens_k, int B, int H, int D,
                                       int max_sel, int page_size, int n_logical_pages_max, int pad_to_page,
                                       int reserved_batch, int dtype, cvllm_stream_t stream) {
  if (!new_keys || !new_vals || !indices_topk || !num_tokens_to_retain || !page_table || !batch_mapping ||
      !bh_lens || !k_cache || !v_cache)
    return CVLLM_ERR_ARG;
  if (pad_to_page && !cu_seqlens_k) return CVLLM_ERR_ARG;
  if (B <= 0 || H <= 0 || H > RK_MAXH || max_sel <= 0 || page_size <= 0) return CVLLM_ERR_ARG;
  if (dtype != CVLLM_F16 && dtype != CVLLM_BF16) return CVLLM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_D(D, {
    hipLaunchKernelGGL((store_ranked_kernel<DD>), dim3(B), dim3(256), 0, st, (const uint16_t*)new_keys,
                       (const uint16_t*)new_vals, sk_n, sk_h, sv_n, sv_h, indices_topk, num_tokens_to_retain,
                       page_table, batch_mapping, bh_lens, (uint16_t*)k_cache, (uint16_t*)v_cache, cu_seqlens_k, H,
                       max_sel, page_size, n_logical_pages_max, pad_to_page, reserved_batch);
  });
  return check_launch();
}
