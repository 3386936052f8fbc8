// a2 — paged, head-sparse, split-K decode attention for gfx950.
//
// Replaces cv/attention/sparse_decode_kernel.py:246-388 (_varkv_stage1_groupM) and :391-435
// (_varkv_stage2_reduce).  HBM-bandwidth bound: every K and V row of every (batch, kv-head) is read
// exactly once; the G query heads of the kv-head share the rows.
//
// Mapping (wave64):
//   * one workgroup (4 waves) per (b, kv-head, split); the split is a contiguous range of logical rows.
//   * a lane loads 16 B (8 x 16-bit) of a row, so D/8 lanes cover a row and one global_load_dwordx4
//     wave-instruction covers 64/(D/8) consecutive rows = 1 KiB contiguous inside a page: fully coalesced.
//   * a "unit" = 4 such loads of K + 4 of V per lane (8 KiB per wave); units are double-buffered in
//     registers so the next unit's 8 loads are in flight while the current one is reduced.
//   * q.k by v_dot2c_f32_{f16,bf16} on the lane's 8 dims, DPP all-reduce over the D/8 lanes of the row.
//   * every lane group keeps its own online-softmax state (m, l, acc[G][8]); groups are merged once at
//     the end (shuffles), waves through LDS, splits by the stage-2 kernel with the LSE rule.
// No MFMA: M = G <= 8 rows would waste the tile and the kernel is bandwidth bound.
#include "common.h"

namespace cvllm {

constexpr int DEC_NW = 4;        // waves per workgroup
constexpr int DEC_NL = 4;        // row-loads per unit per lane (K and V each)
constexpr int DEC_PGCACHE = 512; // page ids cached in LDS per split

template <int LPR>
__device__ __forceinline__ float group_allreduce_sum(float v) {
  // all-reduce over the LPR consecutive lanes that share one row (LPR = 8, 16 or 32)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // xor 1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // xor 2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  if (LPR >= 16)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  if (LPR >= 32) v += __shfl_xor(v, 16, 64);
  return v;
}

template <typename T, int D, int G, bool DIRECT>
__global__ __launch_bounds__(DEC_NW * 64) void decode_stage1_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc, const uint16_t* __restrict__ vc,
    uint16_t* __restrict__ out, float* __restrict__ part_o, float* __restrict__ part_lse,
    const int* __restrict__ seq_lens, const int* __restrict__ page_table, const int* __restrict__ bmap,
    int HKV, int PS, int NLP, int S, float scale) {
  constexpr int LPR = D / 8;          // lanes per row
  constexpr int RPL = 64 / LPR;       // rows per wave-load
  constexpr int UR = RPL * DEC_NL;    // rows per unit
  constexpr int ROUND = UR * DEC_NW;  // rows per workgroup round
  constexpr int HQ_G = G;

  __shared__ int s_pg[DEC_PGCACHE];
  __shared__ float s_m[DEC_NW][G];
  __shared__ float s_l[DEC_NW][G];
  __shared__ float s_acc[DEC_NW][G][D];

  const int bid = blockIdx.x;
  const int s = bid % S;
  const int bh = bid / S;
  const int h = bh % HKV;
  const int b = bh / HKV;
  const int HQ = HKV * HQ_G;
  const int L = seq_lens[bh];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c = lane / LPR;
  const int dl = lane % LPR;

  int per = (L + S - 1) / S;
  per = (per + ROUND - 1) / ROUND * ROUND;
  const int start = s * per;
  const int end = min(start + per, L);

  if (start >= end) {  // empty split (covers L == 0)
    if (DIRECT) {
      for (int i = tid; i < G * D; i += DEC_NW * 64) out[((size_t)b * HQ + h * G) * D + i] = 0;
    } else if (tid < G) {
      part_lse[(size_t)(b * S + s) * HQ + h * G + tid] = -INFINITY;
    }
    return;
  }

  // page ids of this split -> LDS
  const int bt = bmap[b];
  const int* pt = page_table + ((size_t)bt * HKV + h) * NLP;
  const int lp0 = start / PS;
  const int nlp = (end - 1) / PS - lp0 + 1;
  for (int i = tid; i < nlp && i < DEC_PGCACHE; i += DEC_NW * 64) s_pg[i] = pt[lp0 + i];

  // q fragment: the lane's 8 dims of each of the G query heads (kept in the model dtype for dot2)
  uint4 qf[G];
#pragma unroll
  for (int g = 0; g < G; ++g)
    qf[g] = *reinterpret_cast<const uint4*>(q + ((size_t)b * HQ + h * G + g) * D + dl * 8);

  float m[G], l[G], acc[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    m[g] = -INFINITY;
    l[g] = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[g][d] = 0.f;
  }
  __syncthreads();

  const int nunits = (end - start + UR - 1) / UR;

  auto issue = [&](int uu, uint4(&kk)[DEC_NL], uint4(&vv)[DEC_NL]) {
    const int row0 = start + uu * UR;  // multiple of UR; PS % UR == 0 so the unit lies inside one page
    const int lpi = row0 / PS - lp0;
    const int pg = lpi < DEC_PGCACHE ? s_pg[lpi] : pt[lp0 + lpi];
    const size_t base = ((size_t)pg * PS + (row0 % PS) + c) * D + dl * 8;
#pragma unroll
    for (int i = 0; i < DEC_NL; ++i) {
      // rows past `end` are still inside the (allocated) page: load unconditionally, mask later
      kk[i] = *reinterpret_cast<const uint4*>(kc + base + (size_t)i * RPL * D);
      vv[i] = *reinterpret_cast<const uint4*>(vc + base + (size_t)i * RPL * D);
    }
  };

  auto compute = [&](int uu, const uint4(&kk)[DEC_NL], const uint4(&vv)[DEC_NL]) {
    const int rbase = start + uu * UR + c;
    float sc[DEC_NL][G];
#pragma unroll
    for (int i = 0; i < DEC_NL; ++i) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float a = dot2<T>(kk[i].x, qf[g].x, 0.f);
        a = dot2<T>(kk[i].y, qf[g].y, a);
        a = dot2<T>(kk[i].z, qf[g].z, a);
        a = dot2<T>(kk[i].w, qf[g].w, a);
        sc[i][g] = a;
      }
    }
#pragma unroll
    for (int i = 0; i < DEC_NL; ++i) {
      const bool valid = (rbase + i * RPL) < end;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float r = group_allreduce_sum<LPR>(sc[i][g]);
        sc[i][g] = valid ? r * scale : -INFINITY;
      }
    }
    float vf[DEC_NL][8];
#pragma unroll
    for (int i = 0; i < DEC_NL; ++i) {
      const bool valid = (rbase + i * RPL) < end;
      float2 t0 = unpack2<T>(vv[i].x), t1 = unpack2<T>(vv[i].y), t2 = unpack2<T>(vv[i].z), t3 = unpack2<T>(vv[i].w);
      vf[i][0] = valid ? t0.x : 0.f; vf[i][1] = valid ? t0.y : 0.f;
      vf[i][2] = valid ? t1.x : 0.f; vf[i][3] = valid ? t1.y : 0.f;
      vf[i][4] = valid ? t2.x : 0.f; vf[i][5] = valid ? t2.y : 0.f;
      vf[i][6] = valid ? t3.x : 0.f; vf[i][7] = valid ? t3.y : 0.f;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float mx = m[g];
#pragma unroll
      for (int i = 0; i < DEC_NL; ++i) mx = fmaxf(mx, sc[i][g]);
      const float mxs = (mx == -INFINITY) ? 0.f : mx;
      const float alpha = __expf(m[g] - mxs);
      float p[DEC_NL];
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < DEC_NL; ++i) {
        p[i] = __expf(sc[i][g] - mxs);
        ps += p[i];
      }
      l[g] = l[g] * alpha + ps;
      m[g] = mx;
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        float a = acc[g][d] * alpha;
#pragma unroll
        for (int i = 0; i < DEC_NL; ++i) a = fmaf(p[i], vf[i][d], a);
        acc[g][d] = a;
      }
    }
  };

  {
    uint4 ka[DEC_NL], va[DEC_NL], kb[DEC_NL], vb[DEC_NL];
    int u = wave;
    if (u < nunits) issue(u, ka, va);
    while (u < nunits) {
      int un = u + DEC_NW;
      if (un < nunits) issue(un, kb, vb);
      compute(u, ka, va);
      u = un;
      if (u >= nunits) break;
      un = u + DEC_NW;
      if (un < nunits) issue(un, ka, va);
      compute(u, kb, vb);
      u = un;
    }
  }

  // merge the RPL lane groups of the wave (lanes that differ in `c`)
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float mo = __shfl_xor(m[g], off, 64);
      const float lo = __shfl_xor(l[g], off, 64);
      const float mx = fmaxf(m[g], mo);
      const float mxs = (mx == -INFINITY) ? 0.f : mx;
      const float a0 = __expf(m[g] - mxs), a1 = __expf(mo - mxs);
      l[g] = l[g] * a0 + lo * a1;
      m[g] = mx;
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const float ao = __shfl_xor(acc[g][d], off, 64);
        acc[g][d] = acc[g][d] * a0 + ao * a1;
      }
    }
  }
  if (c == 0) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (dl == 0) {
        s_m[wave][g] = m[g];
        s_l[wave][g] = l[g];
      }
#pragma unroll
      for (int d = 0; d < 8; ++d) s_acc[wave][g][dl * 8 + d] = acc[g][d];
    }
  }
  __syncthreads();

  // merge waves; one thread per (g, d)
  for (int idx = tid; idx < G * D; idx += DEC_NW * 64) {
    const int g = idx / D, d = idx % D;
    float M = s_m[0][g];
#pragma unroll
    for (int w = 1; w < DEC_NW; ++w) M = fmaxf(M, s_m[w][g]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < DEC_NW; ++w) {
      const float a = __expf(s_m[w][g] - M);  // M is finite: wave 0 always owns a valid row
      num += a * s_acc[w][g][d];
      den += a * s_l[w][g];
    }
    const float o = num / den;
    if (DIRECT) {
      out[((size_t)b * HQ + h * G + g) * D + d] = to16<T>(o);
    } else {
      part_o[((size_t)(b * S + s) * HQ + h * G + g) * D + d] = o;
      if (d == 0) part_lse[(size_t)(b * S + s) * HQ + h * G + g] = M + __logf(den);
    }
  }
}

// stage 2: LSE-weighted merge of the S partials of one (b, query head)   (reference :391-435)
template <typename T, int D>
__global__ __launch_bounds__(64) void decode_stage2_kernel(const float* __restrict__ part_o,
                                                          const float* __restrict__ part_lse,
                                                          uint16_t* __restrict__ out, int HQ, int S) {
  constexpr int VPT = D / 64;  // values per thread
  const int bhq = blockIdx.x;
  const int b = bhq / HQ, hq = bhq % HQ;
  const int lane = threadIdx.x;
  float M = -INFINITY;
  for (int s = lane; s < S; s += 64) M = fmaxf(M, part_lse[(size_t)(b * S + s) * HQ + hq]);
  M = wave_reduce_max(M);
  float acc[VPT];
#pragma unroll
  for (int j = 0; j < VPT; ++j) acc[j] = 0.f;
  float den = 0.f;
  if (M != -INFINITY) {
    for (int s = 0; s < S; ++s) {
      const float lse = part_lse[(size_t)(b * S + s) * HQ + hq];
      if (lse == -INFINITY) continue;  // empty split: its partial row was never written
      const float w = __expf(lse - M);
      const float* po = part_o + ((size_t)(b * S + s) * HQ + hq) * D + lane * VPT;
#pragma unroll
      for (int j = 0; j < VPT; ++j) acc[j] += w * po[j];
      den += w;
    }
  }
  const float inv = den > 0.f ? 1.f / den : 0.f;
  uint16_t* o = out + ((size_t)b * HQ + hq) * D + lane * VPT;
#pragma unroll
  for (int j = 0; j < VPT; ++j) o[j] = to16<T>(acc[j] * inv);
}

template <typename T, int D, int G>
static int launch_decode(const void* q, const void* kc, const void* vc, void* out, const int* seq_lens,
                         const int* page_table, const int* bmap, float* ws, int B, int HKV, int PS, int NLP,
                         float scale, int S, hipStream_t st) {
  const int HQ = HKV * G;
  float* part_o = ws;
  float* part_lse = ws + (size_t)B * S * HQ * D;
  dim3 grid(B * HKV * S), block(DEC_NW * 64);
  if (S == 1) {
    hipLaunchKernelGGL((decode_stage1_kernel<T, D, G, true>), grid, block, 0, st, (const uint16_t*)q,
                       (const uint16_t*)kc, (const uint16_t*)vc, (uint16_t*)out, part_o, part_lse, seq_lens,
                       page_table, bmap, HKV, PS, NLP, S, scale);
    return check_launch();
  }
  hipLaunchKernelGGL((decode_stage1_kernel<T, D, G, false>), grid, block, 0, st, (const uint16_t*)q,
                     (const uint16_t*)kc, (const uint16_t*)vc, (uint16_t*)out, part_o, part_lse, seq_lens,
                     page_table, bmap, HKV, PS, NLP, S, scale);
  hipLaunchKernelGGL((decode_stage2_kernel<T, D>), dim3(B * HQ), dim3(64), 0, st, part_o, part_lse,
                     (uint16_t*)out, HQ, S);
  return check_launch();
}

template <typename T, int D>
static int dispatch_g(int G, const void* q, const void* kc, const void* vc, void* out, const int* seq_lens,
                      const int* page_table, const int* bmap, float* ws, int B, int HKV, int PS, int NLP,
                      float scale, int S, hipStream_t st) {
  switch (G) {
    case 1: return launch_decode<T, D, 1>(q, kc, vc, out, seq_lens, page_table, bmap, ws, B, HKV, PS, NLP, scale, S, st);
    case 2: return launch_decode<T, D, 2>(q, kc, vc, out, seq_lens, page_table, bmap, ws, B, HKV, PS, NLP, scale, S, st);
    case 4: return launch_decode<T, D, 4>(q, kc, vc, out, seq_lens, page_table, bmap, ws, B, HKV, PS, NLP, scale, S, st);
    case 8: return launch_decode<T, D, 8>(q, kc, vc, out, seq_lens, page_table, bmap, ws, B, HKV, PS, NLP, scale, S, st);
    default: return CVLLM_ERR_SHAPE;
  }
}

template <typename T>
static int dispatch_d(int D, int G, const void* q, const void* kc, const void* vc, void* out,
                      const int* seq_lens, const int* page_table, const int* bmap, float* ws, int B, int HKV,
                      int PS, int NLP, float scale, int S, hipStream_t st) {
  switch (D) {
    case 64: return dispatch_g<T, 64>(G, q, kc, vc, out, seq_lens, page_table, bmap, ws, B, HKV, PS, NLP, scale, S, st);
    case 128: return dispatch_g<T, 128>(G, q, kc, vc, out, seq_lens, page_table, bmap, ws, B, HKV, PS, NLP, scale, S, st);
    case 256: return dispatch_g<T, 256>(G, q, kc, vc, out, seq_lens, page_table, bmap, ws, B, HKV, PS, NLP, scale, S, st);
    default: return CVLLM_ERR_SHAPE;
  }
}

}  // namespace cvllm

using namespace cvllm;

extern "C" size_t cvllm_decode_workspace_bytes(int B, int HQ, int D, int n_splits) {
  if (B <= 0 || HQ <= 0 || D <= 0 || n_splits <= 0) return 0;
  return ((size_t)B * n_splits * HQ * D + (size_t)B * n_splits * HQ) * sizeof(float);
}

extern "C" int cvllm_decode_attn(const void* q, const void* k_cache, const void* v_cache, void* out,
                                 const int32_t* seq_lens_bh, const int32_t* page_table,
                                 const int32_t* batch_mapping, void* workspace, size_t workspace_bytes, int B,
                                 int HQ, int HKV, int D, int page_size, int n_logical_pages_max,
                                 float sm_scale, int n_splits, int dtype, cvllm_stream_t stream) {
  if (!q || !k_cache || !v_cache || !out || !seq_lens_bh || !page_table || !batch_mapping) return CVLLM_ERR_ARG;
  if (B <= 0 || HQ <= 0 || HKV <= 0 || n_splits <= 0 || n_logical_pages_max <= 0) return CVLLM_ERR_ARG;
  if (HQ % HKV != 0) return CVLLM_ERR_SHAPE;
  // a unit of (64/(D/8))*4 rows must not straddle a page; the reference requires PAGE_SIZE % 32 == 0 (:80)
  if (page_size <= 0 || page_size % 32 != 0) return CVLLM_ERR_SHAPE;
  if (n_splits > 1) {
    if (!workspace || workspace_bytes < cvllm_decode_workspace_bytes(B, HQ, D, n_splits)) return CVLLM_ERR_WORKSPACE;
  }
  const int G = HQ / HKV;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVLLM_F16)
    return dispatch_d<F16>(D, G, q, k_cache, v_cache, out, seq_lens_bh, page_table, batch_mapping,
                           (float*)workspace, B, HKV, page_size, n_logical_pages_max, sm_scale, n_splits, st);
  if (dtype == CVLLM_BF16)
    return dispatch_d<BF16>(D, G, q, k_cache, v_cache, out, seq_lens_bh, page_table, batch_mapping,
                            (float*)workspace, B, HKV, page_size, n_logical_pages_max, sm_scale, n_splits, st);
  return CVLLM_ERR_SHAPE;
}

// Host restatement of num_splits_heuristic (cv/attention/sparse_decode_kernel.py:169-192).
extern "C" int cvllm_num_splits(int total_mblocks, int max_seq_len, int num_sms, int max_splits) {
  if ((double)total_mblocks >= 0.8 * (double)num_sms || max_seq_len <= 1024) return 1;
  const int lim = max_splits < num_sms ? max_splits : num_sms;
  auto eff_of = [&](int s) {
    const double n_waves = (double)((long long)total_mblocks * s) / (double)num_sms;
    if (!(n_waves > 0)) return 0.0;
    double c = (double)(long long)n_waves;
    if (c < n_waves) c += 1.0;
    return n_waves / c;
  };
  int n = 0;
  double max_eff = 0.0;
  for (int s = 1; s <= lim; ++s) {
    if ((double)max_seq_len / (double)s <= 512.0) break;
    const double e = eff_of(s);
    if (e > max_eff) max_eff = e;
    n = s;
  }
  const double thr = 0.75 * max_eff;
  for (int s = 1; s <= n; ++s)
    if (eff_of(s) >= thr) return s;
  return 1;
}
