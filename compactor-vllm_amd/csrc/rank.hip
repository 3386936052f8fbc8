// Full ranking for API parity with scores_to_retain_indices (cv/compression/common.py:171-243): the reference
// pads every sequence to max_len with -inf and takes torch.topk(k = max_len*H, sorted) = a full descending sort.
// Not on the fast path (extract_and_store_top_kv uses select.hip); kept so callers of the public function get
// the same int64 [B, k_eff] tensor.  Canonical tie rule: (score desc, flat index asc) = a STABLE sort of the
// order-inverted keys; rocPRIM's segmented radix sort is stable.  This is the one place a ROCm library
// primitive (rocprim, header-only) is used instead of a hand-written kernel.
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "common.h"

namespace cvllm {

__device__ __forceinline__ uint32_t rank_key(float x) {
  uint32_t u = __float_as_uint(x);
  if (u == 0x80000000u) u = 0u;                               // -0.0 == +0.0
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0u;             // NaN first (torch.topk treats NaN as largest)
  const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~asc;                                                // ascending sort of this == descending score
}

__global__ void rank_prep_kernel(const float* __restrict__ scores, const int* __restrict__ cu, uint32_t* keys,
                                 int* vals, int* offs, int B, int H, int row) {
  const int b = blockIdx.y;
  const int n = (cu[b + 1] - cu[b]) * H;
  const float* src = scores + (size_t)cu[b] * H;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < row; i += gridDim.x * blockDim.x) {
    keys[(size_t)b * row + i] = i < n ? rank_key(src[i]) : rank_key(-INFINITY);
    vals[(size_t)b * row + i] = i;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    offs[b] = b * row;
    if (b == B - 1) offs[B] = B * row;
  }
}

__global__ void rank_post_kernel(const int* __restrict__ vals, const int* __restrict__ cu, int64_t* out, int H,
                                 int row, int k_eff) {
  const int b = blockIdx.y;
  const int64_t base = (int64_t)cu[b] * H;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < k_eff; i += gridDim.x * blockDim.x)
    out[(size_t)b * k_eff + i] = base + vals[(size_t)b * row + i];
}

static size_t rank_temp_bytes(size_t total, int B) {
  size_t temp = 0;
  (void)rocprim::segmented_radix_sort_pairs<rocprim::default_config, uint32_t*, uint32_t*, int*, int*, int*>(
      nullptr, temp, nullptr, nullptr, nullptr, nullptr, (unsigned)total, (unsigned)B, nullptr, nullptr, 0, 32,
      (hipStream_t)0, false);
  return temp;
}
static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace cvllm

using namespace cvllm;

extern "C" size_t cvllm_rank_workspace_bytes(int B, int H, int max_seqlen) {
  if (B <= 0 || H <= 0 || max_seqlen <= 0) return 0;
  const size_t total = (size_t)B * max_seqlen * H;
  return 4 * align256(total * 4) + align256((size_t)(B + 1) * 4) + align256(rank_temp_bytes(total, B));
}

extern "C" int cvllm_rank_indices(const float* scores, const int32_t* cu_seqlens_k, int64_t* out, int B, int H,
                                  int max_seqlen, int k_eff, void* workspace, size_t workspace_bytes,
                                  cvllm_stream_t stream) {
  if (!scores || !cu_seqlens_k || !out) return CVLLM_ERR_ARG;
  if (B <= 0 || H <= 0 || max_seqlen <= 0 || k_eff < 0 || k_eff > max_seqlen * H) return CVLLM_ERR_ARG;
  if (!workspace || workspace_bytes < cvllm_rank_workspace_bytes(B, H, max_seqlen)) return CVLLM_ERR_WORKSPACE;
  if (k_eff == 0) return CVLLM_OK;
  hipStream_t st = (hipStream_t)stream;
  const int row = max_seqlen * H;
  const size_t total = (size_t)B * row;
  char* p = (char*)workspace;
  uint32_t* keys_in = (uint32_t*)p;  p += align256(total * 4);
  uint32_t* keys_out = (uint32_t*)p; p += align256(total * 4);
  int* vals_in = (int*)p;            p += align256(total * 4);
  int* vals_out = (int*)p;           p += align256(total * 4);
  int* offs = (int*)p;               p += align256((size_t)(B + 1) * 4);
  size_t temp = rank_temp_bytes(total, B);
  int gx = (row + 255) / 256;
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(rank_prep_kernel, dim3(gx, B), dim3(256), 0, st, scores, cu_seqlens_k, keys_in, vals_in, offs, B,
                     H, row);
  hipError_t e = rocprim::segmented_radix_sort_pairs(p, temp, keys_in, keys_out, vals_in, vals_out, (unsigned)total,
                                                     (unsigned)B, offs, offs + 1, 0, 32, st, false);
  if (e != hipSuccess) return CVLLM_ERR_LAUNCH;
  int gk = (k_eff + 255) / 256;
  if (gk > 1024) gk = 1024;
  hipLaunchKernelGGL(rank_post_kernel, dim3(gk, B), dim3(256), 0, st, vals_out, cu_seqlens_k, out, H, row, k_eff);
  return check_launch();
}
