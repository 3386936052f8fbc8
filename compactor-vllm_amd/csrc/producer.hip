// f-2 — the producer step in front of the attention boundary, in one pass over the fused qkv projection (gfx950).
//
// Replaces, for one layer of a packed batch, the chain of the reference's model code
//   qkv.split -> views                               (cv/models/llama3.py:96-100, qwen3.py:88-91)
//   q_norm / k_norm per head (Qwen3)                 (cv/layers/layernorm.py:15-25; qwen3.py:90-91)
//   rotary_emb(positions, q, k)                      (cv/layers/rotary_embedding.py:8-17, 69-80)
//   [no compression] prefill_store_all_kv            (cv/kv_cache/store_kv_cache.py:251-371)
// which is 3-4 full passes over q / k / v in HBM, by ONE bandwidth-bound kernel: every (token, head) row of q and k is
// read once from the projection output, normalised (optional), rotated and written to its destination(s); v is only
// touched when the rows also go straight into the paged cache.
//
// Numerics follow the reference op by op: RMSNorm in fp32 (x * rsqrt(mean(x^2) + eps)), rounded to the model dtype,
// then times the weight with one more rounding; RoPE in fp32 with separately rounded products
// (y1 = x1*cos - x2*sin, y2 = x2*cos + x1*sin; no FMA contraction) and one final rounding.  Without the norm the
// outputs are bit-identical to the eager torch evaluation; with it they can differ by the summation order of mean(x^2).
//
// Layout: 8 lanes (D = 128) own one row: lane c holds elements [8c, 8c+8) of the first half and the same of the second
// half (the two members of every rotation pair), i.e. two 16-byte loads and two 16-byte stores per lane.
#include "common.h"

namespace cvllm {

template <typename T>
__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  float2 a = unpack2<T>(u.x), b = unpack2<T>(u.y), c = unpack2<T>(u.z), d = unpack2<T>(u.w);
  f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y; f[4] = c.x; f[5] = c.y; f[6] = d.x; f[7] = d.y;
}
template <typename T>
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack2<T>(f[0], f[1]), pack2<T>(f[2], f[3]), pack2<T>(f[4], f[5]), pack2<T>(f[6], f[7]));
}
template <typename T>
__device__ __forceinline__ float round16(float x) {
  return from16<T>(to16<T>(x));
}

__device__ __forceinline__ int seq_of_token(const int* __restrict__ cu, int B, int n) {
  int lo = 0, hi = B;  // cu[lo] <= n < cu[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cu[mid] <= n) lo = mid; else hi = mid;
  }
  return lo;
}

template <typename T, int D>
__global__ __launch_bounds__(256) void qkv_producer_kernel(
    const uint16_t* __restrict__ qkv, int64_t s_n, const int64_t* __restrict__ positions,
    const float* __restrict__ cos_sin, const uint16_t* __restrict__ qw, const uint16_t* __restrict__ kw, float eps,
    uint16_t* __restrict__ q_out, int64_t so_q, uint16_t* __restrict__ k_out, int64_t so_k,
    uint16_t* __restrict__ k_pre, uint16_t* __restrict__ kc, uint16_t* __restrict__ vc,
    const int* __restrict__ cu, const int* __restrict__ bmap, const int* __restrict__ bh_lens,
    const int* __restrict__ page_table, int B, int PS, int NLP, int N, int HQ, int HKV, int max_pos) {
  constexpr int LPR = D / 16;  // lanes per row
  constexpr int HALF = D / 2;
  const bool to_cache = kc != nullptr;
  const int slots = HQ + HKV + (to_cache ? HKV : 0);
  const int c = threadIdx.x % LPR;
  const long rows = (long)N * slots;
  for (long r = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR; r < rows; r += (long)gridDim.x * (256 / LPR)) {
    const int n = (int)(r / slots), j = (int)(r % slots);
    const uint16_t* src = qkv + (size_t)n * s_n + (size_t)j * D;  // q heads, then k heads, then v heads
    const uint4 u1 = *reinterpret_cast<const uint4*>(src + 8 * c);
    const uint4 u2 = *reinterpret_cast<const uint4*>(src + HALF + 8 * c);
    int b = 0, pos_c = 0, pg = 0, hh = 0;
    if (to_cache && j >= HQ) {  // k or v row: where it lands in the paged cache (cache write of store_kv_cache.py:251-319)
      hh = (j - HQ) % HKV;
      b = seq_of_token(cu, B, n);
      pos_c = bh_lens[b * HKV + hh] + (n - cu[b]);
      pg = page_table[((size_t)bmap[b] * HKV + hh) * NLP + pos_c / PS];
    }
    if (j >= HQ + HKV) {  // v: copied as is
      const size_t dst = ((size_t)pg * PS + pos_c % PS) * D;
      *reinterpret_cast<uint4*>(vc + dst + 8 * c) = u1;
      *reinterpret_cast<uint4*>(vc + dst + HALF + 8 * c) = u2;
      continue;
    }
    float x1[8], x2[8];
    unpack8<T>(u1, x1);
    unpack8<T>(u2, x2);
    const bool is_q = j < HQ;
    const uint16_t* w = is_q ? qw : kw;
    if (w != nullptr) {  // per-head RMSNorm (layernorm.py:15-25)
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) ss = fmaf(x1[i], x1[i], ss);
#pragma unroll
      for (int i = 0; i < 8; ++i) ss = fmaf(x2[i], x2[i], ss);
      ss += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0xB1, 0xf, 0xf, true));
      ss += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0x4E, 0xf, 0xf, true));
      if (LPR == 8)
        ss += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0x141, 0xf, 0xf, true));
      const float inv = 1.0f / sqrtf(ss / (float)D + eps);  // correctly rounded sqrt and divide (hipcc default)
      float w1[8], w2[8];
      unpack8<T>(*reinterpret_cast<const uint4*>(w + 8 * c), w1);
      unpack8<T>(*reinterpret_cast<const uint4*>(w + HALF + 8 * c), w2);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        x1[i] = round16<T>(round16<T>(x1[i] * inv) * w1[i]);
        x2[i] = round16<T>(round16<T>(x2[i] * inv) * w2[i]);
      }
      if (!is_q && k_pre != nullptr) {  // the normed PRE-RoPE key: input of Compactor's pre-RoPE scoring (qwen3.py:92-94)
        uint16_t* d = k_pre + ((size_t)n * HKV + (j - HQ)) * D;
        *reinterpret_cast<uint4*>(d + 8 * c) = pack8<T>(x1);
        *reinterpret_cast<uint4*>(d + HALF + 8 * c) = pack8<T>(x2);
      }
    }
    // RoPE (rotary_embedding.py:8-17): fp32, products rounded separately like the eager torch ops
    long p = positions[n];
    p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
    const float* cs = cos_sin + (size_t)p * D;
    const float4 c0 = *reinterpret_cast<const float4*>(cs + 8 * c), c1 = *reinterpret_cast<const float4*>(cs + 8 * c + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(cs + HALF + 8 * c);
    const float4 s1 = *reinterpret_cast<const float4*>(cs + HALF + 8 * c + 4);
    const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    float y1[8], y2[8];
    {
#pragma clang fp contract(off)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float a = x1[i] * cc[i], bb = x2[i] * sn[i], d = x2[i] * cc[i], e = x1[i] * sn[i];
        y1[i] = a - bb;
        y2[i] = d + e;
      }
    }
    const uint4 o1 = pack8<T>(y1), o2 = pack8<T>(y2);
    uint16_t* dst = is_q ? q_out + (size_t)n * so_q + (size_t)j * D : k_out + (size_t)n * so_k + (size_t)(j - HQ) * D;
    *reinterpret_cast<uint4*>(dst + 8 * c) = o1;
    *reinterpret_cast<uint4*>(dst + HALF + 8 * c) = o2;
    if (to_cache && !is_q) {
      const size_t cd = ((size_t)pg * PS + pos_c % PS) * D;
      *reinterpret_cast<uint4*>(kc + cd + 8 * c) = o1;
      *reinterpret_cast<uint4*>(kc + cd + HALF + 8 * c) = o2;
    }
  }
}

__global__ void producer_add_lens_kernel(const int* __restrict__ cu, int* __restrict__ bh_lens, int B, int HKV) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * HKV) bh_lens[i] += cu[i / HKV + 1] - cu[i / HKV];
}

}  // namespace cvllm

using namespace cvllm;

extern "C" int cvllm_qkv_rope_producer(const void* qkv, int64_t s_n, const int64_t* positions, const float* cos_sin,
                                       const void* q_norm_w, const void* k_norm_w, float eps, void* q_out, int64_t so_q,
                                       void* k_out, int64_t so_k, void* k_pre_out, void* k_cache, void* v_cache,
                                       const int32_t* cu_seqlens, const int32_t* batch_mapping, int32_t* bh_lens,
                                       const int32_t* page_table, int B, int page_size, int n_logical_pages_max, int N,
                                       int HQ, int HKV, int D, int max_pos, int dtype, cvllm_stream_t stream) {
  if (!qkv || !positions || !cos_sin || !q_out || !k_out) return CVLLM_ERR_ARG;
  if (N <= 0 || HQ < 0 || HKV < 0 || HQ + HKV <= 0 || max_pos <= 0) return CVLLM_ERR_ARG;  // HQ or HKV may be 0
  if ((s_n % 8) || (so_q % 8) || (so_k % 8)) return CVLLM_ERR_SHAPE;
  if ((q_norm_w == nullptr) != (k_norm_w == nullptr)) return CVLLM_ERR_ARG;
  if (k_pre_out && !k_norm_w) return CVLLM_ERR_ARG;  // without a norm the pre-RoPE key is the projection itself
  const bool to_cache = k_cache != nullptr;
  if (to_cache && (!v_cache || !cu_seqlens || !batch_mapping || !bh_lens || !page_table || B <= 0 || page_size <= 0 ||
                   n_logical_pages_max <= 0))
    return CVLLM_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long rows = (long)N * (HQ + HKV + (to_cache ? HKV : 0));
#define PR(T_, D_)                                                                                                    \
  {                                                                                                                   \
    long blocks = (rows + (256 / (D_ / 16)) - 1) / (256 / (D_ / 16));                                                 \
    if (blocks > 8192) blocks = 8192;                                                                                 \
    hipLaunchKernelGGL((qkv_producer_kernel<T_, D_>), dim3((int)blocks), dim3(256), 0, st, (const uint16_t*)qkv, s_n, \
                       positions, cos_sin, (const uint16_t*)q_norm_w, (const uint16_t*)k_norm_w, eps,                \
                       (uint16_t*)q_out, so_q, (uint16_t*)k_out, so_k, (uint16_t*)k_pre_out, (uint16_t*)k_cache,      \
                       (uint16_t*)v_cache, cu_seqlens, batch_mapping, bh_lens, page_table, B, page_size,             \
                       n_logical_pages_max, N, HQ, HKV, max_pos);                                                     \
  }
  if (dtype == CVLLM_F16 && D == 128) PR(F16, 128)
  else if (dtype == CVLLM_F16 && D == 64) PR(F16, 64)
  else if (dtype == CVLLM_BF16 && D == 128) PR(BF16, 128)
  else if (dtype == CVLLM_BF16 && D == 64) PR(BF16, 64)
  else return CVLLM_ERR_SHAPE;
#undef PR
  if (to_cache)
    hipLaunchKernelGGL(producer_add_lens_kernel, dim3((B * HKV + 255) / 256), dim3(256), 0, st, cu_seqlens, bh_lens, B,
                       HKV);
  return check_launch();
}
