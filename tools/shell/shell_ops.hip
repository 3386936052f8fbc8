// Glue kernels of the benchmark's random-weight model shell (RMSNorm, RoPE, SiLU*mul).  NOT part of the
// compactor hot path and not part of libcvllm_hip.so: bench.py needs a Llama/Qwen3-shaped producer of
// q/k/v around the attention boundary to measure tokens/s; these keep that producer from
// being dominated by unfused elementwise passes.  bf16 in/out, fp32 math.
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ uint16_t f2bf(float x) {
  __bf16 h = (__bf16)x;
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// h[n,:] += delta[n,:] (if delta), y[n,:] = rmsnorm(h[n,:]) * w ; one 256-thread block per row, C % 8 == 0
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(uint16_t* __restrict__ h, const uint16_t* __restrict__ delta,
                                                          const uint16_t* __restrict__ w, uint16_t* __restrict__ y,
                                                          int C, float eps) {
  __shared__ float s_part[4];
  const size_t row = blockIdx.x;
  uint16_t* hr = h + row * C;
  const uint16_t* dr = delta ? delta + row * C : nullptr;
  float ss = 0.f;
  for (int c = threadIdx.x * 8; c < C; c += 256 * 8) {
    uint4 hv = *reinterpret_cast<const uint4*>(hr + c);
    uint16_t* hp = reinterpret_cast<uint16_t*>(&hv);
    if (dr) {
      uint4 dv = *reinterpret_cast<const uint4*>(dr + c);
      const uint16_t* dp = reinterpret_cast<const uint16_t*>(&dv);
#pragma unroll
      for (int j = 0; j < 8; ++j) hp[j] = f2bf(bf2f(hp[j]) + bf2f(dp[j]));
      *reinterpret_cast<uint4*>(hr + c) = hv;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = bf2f(hp[j]);
      ss += f * f;
    }
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float inv = rsqrtf((s_part[0] + s_part[1] + s_part[2] + s_part[3]) / (float)C + eps);
  for (int c = threadIdx.x * 8; c < C; c += 256 * 8) {
    uint4 hv = *reinterpret_cast<const uint4*>(hr + c);
    uint4 wv = *reinterpret_cast<const uint4*>(w + c);
    const uint16_t* hp = reinterpret_cast<const uint16_t*>(&hv);
    const uint16_t* wp = reinterpret_cast<const uint16_t*>(&wv);
    uint4 ov;
    uint16_t* op = reinterpret_cast<uint16_t*>(&ov);
#pragma unroll
    for (int j = 0; j < 8; ++j) op[j] = f2bf(bf2f(hp[j]) * inv * bf2f(wp[j]));
    *reinterpret_cast<uint4*>(y + row * C + c) = ov;
  }
}

// neox RoPE (+ optional per-head RMSNorm before it, Qwen3 q/k-norm): src rows [N, H, 128] with token stride
// s_n -> dst contiguous [N, H, 128].  one wave per (token, head): lane l handles dims l and l+64.
__global__ __launch_bounds__(256) void rope_kernel(const uint16_t* __restrict__ src, int64_t s_n,
                                                   uint16_t* __restrict__ dst, const int64_t* __restrict__ pos,
                                                   const float* __restrict__ cs, const uint16_t* __restrict__ nwq,
                                                   const uint16_t* __restrict__ nwk, int N, int H, int HQ,
                                                   float eps) {
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (long)N * H) return;
  const int n = (int)(wid / H), hh = (int)(wid % H), l = threadIdx.x & 63;
  const uint16_t* sp = src + (size_t)n * s_n + (size_t)hh * 128;
  float a = bf2f(sp[l]), b = bf2f(sp[l + 64]);
  const uint16_t* nw = hh < HQ ? nwq : nwk;  // heads [0,HQ) are queries, [HQ,H) keys (q/k-norm weights differ)
  if (nw) {
    const float inv = rsqrtf(wave_sum(a * a + b * b) / 128.f + eps);
    a = bf2f(f2bf(a * inv * bf2f(nw[l])));
    b = bf2f(f2bf(b * inv * bf2f(nw[l + 64])));
  }
  const float* t = cs + (size_t)pos[n] * 128;  // [cos(64) | sin(64)]
  const float c = t[l], s = t[64 + l];
  uint16_t* dp = dst + ((size_t)n * H + hh) * 128;
  dp[l] = f2bf(a * c - b * s);
  dp[l + 64] = f2bf(b * c + a * s);
}

// RoPE without q/k-norm, vectorised: a thread rotates 8 + 8 elements (dims 8c.. and 64 + 8c..) of one (token, head)
// with 16-byte loads / stores (the wave-per-head kernel above moves 2-byte elements: 0.40 ms per 32 K-token layer where
// the bytes need 0.13 ms)
__global__ __launch_bounds__(256) void rope_vec_kernel(const uint16_t* __restrict__ src, int64_t s_n,
                                                       uint16_t* __restrict__ dst, const int64_t* __restrict__ pos,
                                                       const float* __restrict__ cs, long total, int H) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;  // (n, hh, c) with c in [0, 8)
  if (e >= total) return;
  const int c = (int)(e & 7);
  const long nh = e >> 3;
  const int hh = (int)(nh % H);
  const long n = nh / H;
  const uint16_t* sp = src + (size_t)n * s_n + (size_t)hh * 128 + c * 8;
  const uint4 av = *reinterpret_cast<const uint4*>(sp), bv = *reinterpret_cast<const uint4*>(sp + 64);
  const float* t = cs + (size_t)pos[n] * 128 + c * 8;  // [cos(64) | sin(64)]
  const float4 c0 = *reinterpret_cast<const float4*>(t), c1 = *reinterpret_cast<const float4*>(t + 4);
  const float4 s0 = *reinterpret_cast<const float4*>(t + 64), s1 = *reinterpret_cast<const float4*>(t + 68);
  const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  const float ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  const uint16_t* ap = reinterpret_cast<const uint16_t*>(&av);
  const uint16_t* bp = reinterpret_cast<const uint16_t*>(&bv);
  uint4 oa, ob;
  uint16_t* oap = reinterpret_cast<uint16_t*>(&oa);
  uint16_t* obp = reinterpret_cast<uint16_t*>(&ob);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float a = bf2f(ap[j]), b = bf2f(bp[j]);
    oap[j] = f2bf(a * cc[j] - b * ss[j]);
    obp[j] = f2bf(b * cc[j] + a * ss[j]);
  }
  uint16_t* dp = dst + ((size_t)n * H + hh) * 128 + c * 8;
  *reinterpret_cast<uint4*>(dp) = oa;
  *reinterpret_cast<uint4*>(dp + 64) = ob;
}

// out[n, i] = silu(gu[n, i]) * gu[n, I + i]
__global__ __launch_bounds__(256) void silu_mul_kernel(const uint16_t* __restrict__ gu, uint16_t* __restrict__ out,
                                                       long total, int I) {
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 8; e < total; e += (long)gridDim.x * 256 * 8) {
    const long n = e / I;
    const int i = (int)(e % I);
    uint4 gv = *reinterpret_cast<const uint4*>(gu + n * 2 * I + i);
    uint4 uv = *reinterpret_cast<const uint4*>(gu + n * 2 * I + I + i);
    const uint16_t* gp = reinterpret_cast<const uint16_t*>(&gv);
    const uint16_t* up = reinterpret_cast<const uint16_t*>(&uv);
    uint4 ov;
    uint16_t* op = reinterpret_cast<uint16_t*>(&ov);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float g = bf2f(gp[j]);
      op[j] = f2bf(g / (1.f + __expf(-g)) * bf2f(up[j]));
    }
    *reinterpret_cast<uint4*>(out + n * I + i) = ov;
  }
}

// y[b, n] = sum_k W[n, k] * x[b, k]   (bf16 in/out, fp32 accumulate), B <= 4 rows of x: the decode-step GEMV.
// Bandwidth bound on W (read once per step -> non-temporal loads).  A wave owns RPW rows at a time with all of
// their loads in flight; x sits in LDS (read as 16-byte chunks, lane-consecutive = conflict free).
typedef __attribute__((ext_vector_type(2))) __bf16 bf162_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float dot2bf(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf162_t, a), __builtin_bit_cast(bf162_t, b), c, false);
}
template <int B, int RPW>
__global__ __launch_bounds__(256) void gemv_kernel(const uint16_t* __restrict__ W, const uint16_t* __restrict__ x,
                                                   uint16_t* __restrict__ y, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) uint16_t s_x[];  // [B][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid * 8; i < B * K; i += 256 * 8) *reinterpret_cast<uint4*>(s_x + i) = *reinterpret_cast<const uint4*>(x + i);
  __syncthreads();
  const int nchunk = K / 8;  // 16-byte chunks per row
  for (int row0 = (blockIdx.x * 4 + wave) * RPW; row0 < N; row0 += gridDim.x * 4 * RPW) {
    float acc[RPW][B];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) acc[r][b] = 0.f;
    for (int c = lane; c < nchunk; c += 64) {
      u32x4_t w[RPW];
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, N - 1);
        w[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(W + (size_t)row * K + c * 8));
      }
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const uint4 xv = *reinterpret_cast<const uint4*>(s_x + b * K + c * 8);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          float a = acc[r][b];
          a = dot2bf(w[r].x, xv.x, a);
          a = dot2bf(w[r].y, xv.y, a);
          a = dot2bf(w[r].z, xv.z, a);
          a = dot2bf(w[r].w, xv.w, a);
          acc[r][b] = a;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const float t = wave_sum(acc[r][b]);
        if (lane == 0 && row0 + r < N) y[(size_t)b * N + row0 + r] = f2bf(t);
      }
  }
}

// Decode-step (one token) GEMV with the producer fused into the x load:
//   MODE 1: x = rmsnorm(h + delta) * nw   (block 0 also writes h_out = h + delta, the residual stream; h_out must
//           not alias h: other blocks are still reading h)
//   MODE 2: x = silu(gu[0:K]) * gu[K:2K]
// Every block recomputes the K-element prologue (K <= 14336: a few hundred cycles) instead of a separate launch.
template <int MODE, int RPW>
__global__ __launch_bounds__(256) void gemv_fused_kernel(const uint16_t* __restrict__ W, const uint16_t* __restrict__ h,
                                                         uint16_t* __restrict__ h_out,
                                                         const uint16_t* __restrict__ delta,
                                                         const uint16_t* __restrict__ nw,
                                                         const uint16_t* __restrict__ gu, uint16_t* __restrict__ y,
                                                         int N, int K, float eps) {
  extern __shared__ __attribute__((aligned(16))) uint16_t s_x[];  // [K]
  __shared__ float s_part[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (MODE == 1) {
    float ss = 0.f;
    for (int i = tid * 8; i < K; i += 256 * 8) {
      uint4 hv = *reinterpret_cast<const uint4*>(h + i);
      uint16_t* hp = reinterpret_cast<uint16_t*>(&hv);
      if (delta) {
        uint4 dv = *reinterpret_cast<const uint4*>(delta + i);
        const uint16_t* dp = reinterpret_cast<const uint16_t*>(&dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) hp[j] = f2bf(bf2f(hp[j]) + bf2f(dp[j]));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = bf2f(hp[j]);
        ss += f * f;
      }
      *reinterpret_cast<uint4*>(s_x + i) = hv;
    }
    ss = wave_sum(ss);
    if (lane == 0) s_part[wave] = ss;
    __syncthreads();
    const float inv = rsqrtf((s_part[0] + s_part[1] + s_part[2] + s_part[3]) / (float)K + eps);
    for (int i = tid * 8; i < K; i += 256 * 8) {
      uint4 hv = *reinterpret_cast<const uint4*>(s_x + i);
      uint4 wv = *reinterpret_cast<const uint4*>(nw + i);
      if (blockIdx.x == 0) *reinterpret_cast<uint4*>(h_out + i) = hv;  // residual stream, written once
      uint16_t* hp = reinterpret_cast<uint16_t*>(&hv);
      const uint16_t* wp = reinterpret_cast<const uint16_t*>(&wv);
#pragma unroll
      for (int j = 0; j < 8; ++j) hp[j] = f2bf(bf2f(hp[j]) * inv * bf2f(wp[j]));
      *reinterpret_cast<uint4*>(s_x + i) = hv;
    }
  } else {
    for (int i = tid * 8; i < K; i += 256 * 8) {
      uint4 gv = *reinterpret_cast<const uint4*>(gu + i);
      uint4 uv = *reinterpret_cast<const uint4*>(gu + K + i);
      uint16_t* gp = reinterpret_cast<uint16_t*>(&gv);
      const uint16_t* up = reinterpret_cast<const uint16_t*>(&uv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = bf2f(gp[j]);
        gp[j] = f2bf(g / (1.f + __expf(-g)) * bf2f(up[j]));
      }
      *reinterpret_cast<uint4*>(s_x + i) = gv;
    }
  }
  __syncthreads();
  const int nchunk = K / 8;
  for (int row0 = (blockIdx.x * 4 + wave) * RPW; row0 < N; row0 += gridDim.x * 4 * RPW) {
    float acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r] = 0.f;
    for (int c = lane; c < nchunk; c += 64) {
      u32x4_t w[RPW];
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, N - 1);
        w[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(W + (size_t)row * K + c * 8));
      }
      const uint4 xv = *reinterpret_cast<const uint4*>(s_x + c * 8);
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        float a = acc[r];
        a = dot2bf(w[r].x, xv.x, a);
        a = dot2bf(w[r].y, xv.y, a);
        a = dot2bf(w[r].z, xv.z, a);
        a = dot2bf(w[r].w, xv.w, a);
        acc[r] = a;
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const float t = wave_sum(acc[r]);
      if (lane == 0 && row0 + r < N) y[row0 + r] = f2bf(t);
    }
  }
}


// Pipelined decode GEMV (one x row): a wave owns RPW weight rows and walks K in batches of UN 16-byte chunks per
// lane, with batch b+1's loads issued before batch b's dot products, and batch 0 issued BEFORE the x prologue
// (norm / SiLU / copy into LDS), whose latency then hides behind the first HBM round trip.  One row group per wave
// (grid = N / (4 RPW)): N = 4096 gives 512 workgroups = 8 waves per CU with 8 loads of 1 KiB in flight each.
//   MODE 0: x as given   MODE 1: x = rmsnorm(h + delta) * nw (block 0 writes h_out)   MODE 2: x = silu(gu[:K]) * gu[K:]
//   ROPE (RPW == 2): the wave's two rows are (head*128 + d, head*128 + d + 64), i.e. one rotary pair; rows below
//   rope_rows (the q and k heads of a fused qkv projection) are rotated by cs[pos] before the store - the RoPE launch
//   of the decode step disappears.  Values are rounded to bf16 before the rotation, like the separate kernel sees them.
template <int MODE, int RPW, int UN, bool ROPE = false>
__global__ __launch_bounds__(256) void gemv2_kernel(const uint16_t* __restrict__ W, const uint16_t* __restrict__ h,
                                                    uint16_t* __restrict__ h_out, const uint16_t* __restrict__ delta,
                                                    const uint16_t* __restrict__ nw, const uint16_t* __restrict__ gu,
                                                    uint16_t* __restrict__ y, int N, int K, float eps,
                                                    const int64_t* __restrict__ pos = nullptr,
                                                    const float* __restrict__ cs = nullptr, int rope_rows = 0) {
  static_assert(!ROPE || RPW == 2, "a rotary pair per wave");
  extern __shared__ __attribute__((aligned(16))) uint16_t s_x[];  // [K]
  __shared__ float s_part[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wid = blockIdx.x * 4 + wave;
  const int row0 = ROPE ? (wid >> 6) * 128 + (wid & 63) : wid * RPW;
  constexpr int RSTEP = ROPE ? 64 : 1;  // distance between the wave's rows
  const int nb = K / (8 * 64 * UN);  // batches; K % (512 UN) == 0 checked on the host
  const uint16_t* wrow[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) wrow[r] = W + (size_t)min(row0 + r * RSTEP, N - 1) * K + lane * 8;
  u32x4_t cur[RPW][UN], nxt[RPW][UN];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int u = 0; u < UN; ++u)
      cur[r][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wrow[r] + u * 512));

  if (MODE == 1) {
    float ss = 0.f;
    for (int i = tid * 8; i < K; i += 256 * 8) {
      uint4 hv = *reinterpret_cast<const uint4*>(h + i);
      uint16_t* hp = reinterpret_cast<uint16_t*>(&hv);
      if (delta) {
        uint4 dv = *reinterpret_cast<const uint4*>(delta + i);
        const uint16_t* dp = reinterpret_cast<const uint16_t*>(&dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) hp[j] = f2bf(bf2f(hp[j]) + bf2f(dp[j]));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = bf2f(hp[j]);
        ss += f * f;
      }
      *reinterpret_cast<uint4*>(s_x + i) = hv;
    }
    ss = wave_sum(ss);
    if (lane == 0) s_part[wave] = ss;
    __syncthreads();
    const float inv = rsqrtf((s_part[0] + s_part[1] + s_part[2] + s_part[3]) / (float)K + eps);
    for (int i = tid * 8; i < K; i += 256 * 8) {
      uint4 hv = *reinterpret_cast<const uint4*>(s_x + i);
      uint4 wv = *reinterpret_cast<const uint4*>(nw + i);
      if (blockIdx.x == 0) *reinterpret_cast<uint4*>(h_out + i) = hv;  // residual stream, written once
      uint16_t* hp = reinterpret_cast<uint16_t*>(&hv);
      const uint16_t* wp = reinterpret_cast<const uint16_t*>(&wv);
#pragma unroll
      for (int j = 0; j < 8; ++j) hp[j] = f2bf(bf2f(hp[j]) * inv * bf2f(wp[j]));
      *reinterpret_cast<uint4*>(s_x + i) = hv;
    }
  } else if (MODE == 2) {
    for (int i = tid * 8; i < K; i += 256 * 8) {
      uint4 gv = *reinterpret_cast<const uint4*>(gu + i);
      uint4 uv = *reinterpret_cast<const uint4*>(gu + K + i);
      uint16_t* gp = reinterpret_cast<uint16_t*>(&gv);
      const uint16_t* up = reinterpret_cast<const uint16_t*>(&uv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = bf2f(gp[j]);
        gp[j] = f2bf(g / (1.f + __expf(-g)) * bf2f(up[j]));
      }
      *reinterpret_cast<uint4*>(s_x + i) = gv;
    }
  } else {
    for (int i = tid * 8; i < K; i += 256 * 8) *reinterpret_cast<uint4*>(s_x + i) = *reinterpret_cast<const uint4*>(h + i);
  }
  __syncthreads();

  float acc[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) acc[r] = 0.f;
  for (int b = 0; b < nb; ++b) {
    if (b + 1 < nb) {
#pragma unroll
      for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int u = 0; u < UN; ++u)
          nxt[r][u] = __builtin_nontemporal_load(
              reinterpret_cast<const u32x4_t*>(wrow[r] + (size_t)(b + 1) * UN * 512 + u * 512));
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const uint4 xv = *reinterpret_cast<const uint4*>(s_x + (b * UN + u) * 512 + lane * 8);
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        float a = acc[r];
        a = dot2bf(cur[r][u].x, xv.x, a);
        a = dot2bf(cur[r][u].y, xv.y, a);
        a = dot2bf(cur[r][u].z, xv.z, a);
        a = dot2bf(cur[r][u].w, xv.w, a);
        acc[r] = a;
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int u = 0; u < UN; ++u) cur[r][u] = nxt[r][u];
  }
  if (ROPE) {
    const float a = bf2f(f2bf(wave_sum(acc[0]))), b = bf2f(f2bf(wave_sum(acc[RPW - 1])));
    if (lane == 0 && row0 + 64 < N) {
      if (row0 < rope_rows) {
        const float* t = cs + (size_t)pos[0] * 128;  // [cos(64) | sin(64)]
        const int d = row0 & 63;
        const float c = t[d], sn = t[64 + d];
        y[row0] = f2bf(a * c - b * sn);
        y[row0 + 64] = f2bf(b * c + a * sn);
      } else {
        y[row0] = f2bf(a);
        y[row0 + 64] = f2bf(b);
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const float t = wave_sum(acc[r]);
    if (lane == 0 && row0 + r < N) y[row0 + r] = f2bf(t);
  }
}

constexpr int G2_RPW = 2, G2_UN = 4;
static bool gemv2_ok(int N, int K) { return K % (512 * G2_UN) == 0 && (size_t)K * 2 <= 64 * 1024 && N <= 8192 * 4 * G2_RPW; }
template <int MODE>
static void gemv2_launch(const void* W, const void* h, void* h_out, const void* delta, const void* nw, const void* gu,
                         void* y, int N, int K, float eps, void* stream) {
  const int blocks = (N + 4 * G2_RPW - 1) / (4 * G2_RPW);
  hipLaunchKernelGGL((gemv2_kernel<MODE, G2_RPW, G2_UN>), dim3(blocks), dim3(256), (size_t)K * 2, (hipStream_t)stream,
                     (const uint16_t*)W, (const uint16_t*)h, (uint16_t*)h_out, (const uint16_t*)delta,
                     (const uint16_t*)nw, (const uint16_t*)gu, (uint16_t*)y, N, K, eps);
}

extern "C" int shell_gemv_norm(const void* W, const void* h, void* h_out, const void* delta, const void* nw, void* y,
                               int N, int K, float eps, void* stream) {
  if (h == h_out) return -1;
  if (K % 8 || (size_t)K * 2 > 64 * 1024) return -1;
  if (gemv2_ok(N, K)) {
    gemv2_launch<1>(W, h, h_out, delta, nw, nullptr, y, N, K, eps, stream);
    return 0;
  }
  int blocks = (N + 15) / 16;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL((gemv_fused_kernel<1, 4>), dim3(blocks), dim3(256), (size_t)K * 2, (hipStream_t)stream,
                     (const uint16_t*)W, (const uint16_t*)h, (uint16_t*)h_out, (const uint16_t*)delta,
                     (const uint16_t*)nw, nullptr, (uint16_t*)y, N, K, eps);
  return 0;
}
// qkv projection of one decode token with RMSNorm on the way in and RoPE on the way out (no q/k-norm models only)
extern "C" int shell_gemv_norm_rope(const void* W, const void* h, void* h_out, const void* delta, const void* nw,
                                    void* y, int N, int K, float eps, const void* pos, const void* cs, int rope_rows,
                                    void* stream) {
  if (h == h_out || !gemv2_ok(N, K) || N % 128 || rope_rows % 128) return -1;
  const int blocks = N / (4 * G2_RPW);
  hipLaunchKernelGGL((gemv2_kernel<1, G2_RPW, G2_UN, true>), dim3(blocks), dim3(256), (size_t)K * 2,
                     (hipStream_t)stream, (const uint16_t*)W, (const uint16_t*)h, (uint16_t*)h_out,
                     (const uint16_t*)delta, (const uint16_t*)nw, nullptr, (uint16_t*)y, N, K, eps,
                     (const int64_t*)pos, (const float*)cs, rope_rows);
  return 0;
}
extern "C" int shell_gemv_silu(const void* W, const void* gu, void* y, int N, int K, void* stream) {
  if (K % 8 || (size_t)K * 2 > 64 * 1024) return -1;
  if (gemv2_ok(N, K)) {
    gemv2_launch<2>(W, nullptr, nullptr, nullptr, nullptr, gu, y, N, K, 0.f, stream);
    return 0;
  }
  int blocks = (N + 15) / 16;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL((gemv_fused_kernel<2, 4>), dim3(blocks), dim3(256), (size_t)K * 2, (hipStream_t)stream,
                     (const uint16_t*)W, nullptr, nullptr, nullptr, nullptr, (const uint16_t*)gu, (uint16_t*)y, N, K,
                     0.f);
  return 0;
}

extern "C" int shell_gemv(const void* W, const void* x, void* y, int B, int N, int K, void* stream) {
  if (K % 8 || B < 1 || B > 4) return -1;
  const size_t smem = (size_t)B * K * 2;
  if (smem > 64 * 1024) return -1;
  if (B == 1 && gemv2_ok(N, K)) {
    gemv2_launch<0>(W, x, nullptr, nullptr, nullptr, nullptr, y, N, K, 0.f, stream);
    return 0;
  }
  constexpr int RPW = 4;
  int blocks = (N + 4 * RPW - 1) / (4 * RPW);
  if (blocks > 2048) blocks = 2048;
#define GV(B_) hipLaunchKernelGGL((gemv_kernel<B_, RPW>), dim3(blocks), dim3(256), smem, (hipStream_t)stream, \
                                  (const uint16_t*)W, (const uint16_t*)x, (uint16_t*)y, N, K)
  switch (B) {
    case 1: GV(1); break;
    case 2: GV(2); break;
    case 3: GV(3); break;
    default: GV(4); break;
  }
#undef GV
  return 0;
}

extern "C" void shell_add_rmsnorm(void* h, const void* delta, const void* w, void* y, int N, int C, float eps,
                                  void* stream) {
  if (N > 0)
    hipLaunchKernelGGL(add_rmsnorm_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (uint16_t*)h,
                       (const uint16_t*)delta, (const uint16_t*)w, (uint16_t*)y, C, eps);
}
extern "C" void shell_rope(const void* src, int64_t s_n, void* dst, const void* pos, const void* cs, const void* nwq,
                           const void* nwk, int N, int H, int HQ, float eps, void* stream) {
  const long waves = (long)N * H;
  if (waves > 0 && !nwq && !nwk && (s_n % 8) == 0 && N >= 64) {  // no q/k-norm, many tokens: 16-byte lanes
    const long total = waves * 8;
    hipLaunchKernelGGL(rope_vec_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)src, s_n, (uint16_t*)dst, (const int64_t*)pos, (const float*)cs, total, H);
    return;
  }
  if (waves > 0)
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)src, s_n, (uint16_t*)dst, (const int64_t*)pos, (const float*)cs,
                       (const uint16_t*)nwq, (const uint16_t*)nwk, N, H, HQ, eps);
}
extern "C" void shell_silu_mul(const void* gu, void* out, long N, int I, void* stream) {
  const long total = N * I;
  long blocks = (total / 8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (total > 0)
    hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)gu, (uint16_t*)out, total, I);
}
