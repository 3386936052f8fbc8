// Shared device helpers for libcvllm_hip (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cvllm.h"

namespace cvllm {

typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf162_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;

struct F16 {
  typedef _Float16 elem;
  static constexpr int code = CVLLM_F16;
};
struct BF16 {
  typedef __bf16 elem;
  static constexpr int code = CVLLM_BF16;
};

// ---- 16-bit pair (one 32-bit word) -> two fp32 ------------------------------------------------
template <typename T>
__device__ __forceinline__ float2 unpack2(uint32_t w);
template <>
__device__ __forceinline__ float2 unpack2<F16>(uint32_t w) {
  half2_t h = __builtin_bit_cast(half2_t, w);
  return make_float2((float)h[0], (float)h[1]);
}
template <>
__device__ __forceinline__ float2 unpack2<BF16>(uint32_t w) {
  return make_float2(__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u));
}

// ---- fp32 dot-accumulate of a 16-bit pair: acc + a.x*b.x + a.y*b.y (v_dot2c_f32_{f16,bf16}) ----
template <typename T>
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc);
template <>
__device__ __forceinline__ float dot2<F16>(uint32_t a, uint32_t b, float acc) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a), __builtin_bit_cast(half2_t, b), acc, false);
}
template <>
__device__ __forceinline__ float dot2<BF16>(uint32_t a, uint32_t b, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf162_t, a), __builtin_bit_cast(bf162_t, b), acc,
                                         false);
}

// ---- fp32 -> 16-bit, round to nearest even --------------------------------------------------------
template <typename T>
__device__ __forceinline__ uint16_t to16(float x);
template <>
__device__ __forceinline__ uint16_t to16<F16>(float x) {
  _Float16 h = (_Float16)x;
  return __builtin_bit_cast(uint16_t, h);
}
template <>
__device__ __forceinline__ uint16_t to16<BF16>(float x) {
  __bf16 h = (__bf16)x;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(uint16_t, h);
}
// two fp32 -> one packed 16-bit pair, RNE (built as a vector so that hipcc selects ONE v_cvt_pk_{bf16,f16}_f32
// instead of two conversions and an OR)
template <typename T>
__device__ __forceinline__ uint32_t pack2(float lo, float hi);
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
template <>
__device__ __forceinline__ uint32_t pack2<F16>(float lo, float hi) {
  const f32x2_t f = {lo, hi};  // ONE vector conversion: selected as v_cvt_pk_f16_f32 whatever the SLP vectoriser does
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, half2_t));
}
template <>
__device__ __forceinline__ uint32_t pack2<BF16>(float lo, float hi) {
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf162_t));
}
template <typename T>
__device__ __forceinline__ float from16(uint16_t b);
template <>
__device__ __forceinline__ float from16<F16>(uint16_t b) {
  return (float)__builtin_bit_cast(_Float16, b);
}
template <>
__device__ __forceinline__ float from16<BF16>(uint16_t b) {
  return __uint_as_float((uint32_t)b << 16);
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// store_kv.hip: decode append with optional indexing of bh_lens by the TRUE batch row (fused decode step)
int store_decode_kv_impl(const void* key, const void* value, int64_t sk_b, int64_t sk_h, int64_t sv_b, int64_t sv_h,
                         const int32_t* batch_mapping, int32_t* bh_lens, const int32_t* page_table, void* k_cache,
                         void* v_cache, int B, int HKV, int D, int page_size, int n_logical_pages_max,
                         int reserved_batch, int dtype, int lens_by_row, cvllm_stream_t stream);

// dynamic-LDS limit of a kernel, set once per (kernel instantiation, device): the attribute lives on the device's
// code object, so a second device in the same process needs its own call
template <typename K>
static void set_dyn_lds_once(K kern, int bytes) {
  static bool done[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !done[dev]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (dev >= 0 && dev < 64) done[dev] = true;
  }
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? CVLLM_OK : CVLLM_ERR_LAUNCH; }

}  // namespace cvllm
