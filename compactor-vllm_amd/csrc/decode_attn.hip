// a2 — paged, head-sparse, split-K decode attention for gfx950.
//
// Replaces cv/attention/sparse_decode_kernel.py:246-388 (_varkv_stage1_groupM) and :391-435
// (_varkv_stage2_reduce).  HBM-bandwidth bound: every K and V row of every (batch, kv-head) is read
// exactly once; the G query heads of the kv-head share the rows.
//
// Two kernels live here.  decode_fused_kernel (below, the default for D <= 128) streams K/V through per-wave LDS rings
// filled by LDS-DMA and can append the new token's row on the way; decode_stage1_kernel (first in the file) is the
// register double-buffered version kept for D = 256.  Common mapping (wave64):
//   * one workgroup (4 waves) per (b, kv-head, split); the split is a contiguous range of logical rows.
//   * a lane loads 16 B (8 x 16-bit) of a row, so D/8 lanes cover a row and one wave-instruction covers
//     64/(D/8) consecutive rows = 1 KiB contiguous inside a page: fully coalesced.
//   * a "unit" = 4 such loads of K + 4 of V per lane (8 KiB per wave).
//   * q.k by v_dot2c_f32_{f16,bf16} on the lane's 8 dims, DPP all-reduce over the D/8 lanes of the row.
//   * every lane group keeps its own online-softmax state (m, l, acc[G][8]); the states of a split are merged once
//     at the end through LDS, splits by the stage-2 kernel with the LSE rule.
// No MFMA: M = G <= 8 rows would waste the tile and the kernel is bandwidth bound.
#include "common.h"

namespace cvllm {

constexpr int DEC_NW = 4;        // waves per workgroup
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

template <typename T, int D, int G, bool DIRECT, int DEC_NL, int MINW>
__global__ __launch_bounds__(DEC_NW * 64, MINW) void decode_stage1_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc, const uint16_t* __restrict__ vc,
    uint16_t* __restrict__ out, float* __restrict__ part_o, float* __restrict__ part_lse,
    const int* __restrict__ seq_lens, const int* __restrict__ page_table, const int* __restrict__ bmap,
    int HKV, int PS, int NLP, int S, float scale, int lens_by_row) {
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
  // lens_by_row: seq_lens is the layer's full [Bmax+1, HKV] table indexed by the TRUE batch row (fused decode)
  const int L = seq_lens[lens_by_row ? bmap[b] * HKV + h : bh];
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

// ---------------------------------------------------------------------------------------------------
// The decode kernel (default): LDS-ring streaming + in-launch split merge + optional fused cache append.
//
// * A single CU sustains only ~24 GB/s of HBM loads, so the chip rate needs every CU busy with >= ~64 KiB in
//   flight for the whole launch.  Registers cannot hold that, LDS can: each wave owns a private ring of R unit
//   slots in LDS and fills it with global_load_lds_dwordx4 (LDS-DMA, non-temporal: lane l's 16 B land at
//   slot + 16*l, the same coalesced row image), keeps R-1 units in flight behind a COUNTED s_waitcnt vmcnt,
//   and pulls the landed unit into registers with ds_read_b128.  Rings are wave-private: no barrier in the
//   loop.  The ds_reads and waits are inline asm because hipcc otherwise drains vmcnt(0) before any LDS read
//   while an LDS-DMA is pending.  Page ids of the split live in VGPRs (lane i holds page i) and are fetched
//   with v_readlane: the loop has no other memory instruction.
// * Splits of one (b, kv-head) are merged by a second, tiny kernel.  (An in-launch last-arriver merge was built
//   and measured: the agent-scope release + acquire fences and the single merger's dependent L2 reads cost
//   ~17 us against 4.6 us + one launch boundary for the separate kernel, so the seam is cut — CDNA guide 5.6.)
// * Fused append (key_new != NULL): the new token's K/V row of (b,h) belongs at position L_old.  Every
//   workgroup uses L = L_old + 1; the workgroup whose split owns row L_old substitutes the row from registers
//   and writes it to the cache; the merge kernel (stream-ordered after every stage-1 workgroup has read L_old)
//   publishes L_old + 1.  This replaces decode_store_kv + index_select/index_copy_ of the reference's decode
//   branch (cv/layers/attention.py:127-160) with zero extra launches.
template <int N>
struct VmWait;
#define CVLLM_VMWAIT(N)                                                             \
  template <>                                                                       \
  struct VmWait<N> {                                                                \
    static __device__ __forceinline__ void wait() { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); } \
  };
CVLLM_VMWAIT(0) CVLLM_VMWAIT(8) CVLLM_VMWAIT(12) CVLLM_VMWAIT(16) CVLLM_VMWAIT(24) CVLLM_VMWAIT(32)

// -DCVLLM_DEC_TS (tools/decode_ts.py builds a separate debug library with it): wave 0 of workgroup 0 records s_memtime
// at the phase boundaries of decode_fused_kernel.  Not compiled into libcvllm_hip.so.
#ifdef CVLLM_DEC_TS
__device__ unsigned long long g_dec_ts[16];
__device__ unsigned long long g_dec_rt[1024 * 16];  // [workgroup][stamp]: s_memrealtime (100 MHz, one clock for all CUs)
#define DEC_TS(i)                                                                    \
  do {                                                                               \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_dec_ts[i] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define DEC_RT(i)                                                                                      \
  do {                                                                                                 \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_dec_rt[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define DEC_TS(i) \
  do {            \
  } while (0)
#define DEC_RT(i) \
  do {            \
  } while (0)
#endif

// wave-wide all-reduces without LDS round trips: DPP inside the rows of 16 lanes, v_readlane across the four rows
__device__ __forceinline__ float wave_allreduce_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true)));
  const int i = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ float wave_allreduce_sum_fixed(float v) {  // the same tree on every call: reproducible bits
  v = group_allreduce_sum<16>(v);
  const int i = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
  return (r0 + r1) + (r2 + r3);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0), i.e. it would make every wave
// wait for its in-flight write-through (sc1) mailbox stores - about a microsecond each time in the merge tail.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int DEC_PGREGS = 8;       // page ids cached in registers: 8 x 64 pages
constexpr int DEC_MAX_SPLITS = 256;

template <typename T, int D, int G, int NW, int NL, int R>
__global__ __launch_bounds__(NW * 64) void decode_fused_kernel(
    // argument order = order of first use: the leading 14 dwords are preloaded into SGPRs at wave launch
    // (-mllvm -amdgpu-kernarg-preload-count=16, build.py), the rest is fetched in ONE batch at the top (below)
    const uint16_t* __restrict__ q, const int* __restrict__ bmap, int* __restrict__ seq_lens,
    const int* __restrict__ page_table, const uint16_t* __restrict__ key_new, const uint16_t* __restrict__ val_new,
    uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, int HKV, int PS, int NLP, int S, float scale,
    int lens_by_row, int reserved, int64_t sk_b, int64_t sk_h, int64_t sv_b, int64_t sv_h,
    uint16_t* __restrict__ out, float* __restrict__ part_o, float* __restrict__ part_lse,
    unsigned* __restrict__ merge_ws, unsigned* __restrict__ err_word, int mode, float* __restrict__ lse_out) {
  // hipcc fetches kernel arguments lazily, one dependent s_load + wait in front of each first use (four round trips
  // in this kernel's prologue); naming them all here puts every fetch into one clause.
  asm volatile("" ::"s"(vc), "s"(HKV), "s"(PS), "s"(NLP), "s"(S), "s"(scale), "s"(lens_by_row), "s"(reserved));
  asm volatile("" ::"s"(sk_b), "s"(sk_h), "s"(sv_b), "s"(sv_h), "s"(out), "s"(part_o), "s"(part_lse), "s"(merge_ws),
               "s"(err_word), "s"(mode), "s"(lse_out));
  DEC_TS(0);
  DEC_RT(0);
  constexpr int LPR = D / 8;
  constexpr int RPL = 64 / LPR;
  constexpr int UR = RPL * NL;
  constexpr int UNIT_BYTES = 2 * NL * 1024;  // K loads then V loads
  constexpr int P = R - 1;                   // units in flight ahead of the one being reduced
  constexpr int NT_AUX = 2;                  // non-temporal: K/V are read once per step
  static_assert(2 * NL * P <= 60, "vmcnt is a 6-bit counter");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [NW][R][UNIT_BYTES] rings; after the loop each wave's ring doubles as its merge staging area

  const int bid = blockIdx.x;
  const int s = bid % S;
  const int bh = bid / S;
  const int h = bh % HKV;
  const int b = bh / HKV;
  const int HQ = HKV * G;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c = lane / LPR;
  const int dl = lane % LPR;

  const int bt = bmap[b];  // first link of the metadata chain: issued before anything else
  // q and the new K/V row depend on (b, h) only: issued first, in flight together with the batch_mapping -> length /
  // page-table chain below instead of after it (one dependent memory round trip less in front of the first K/V byte;
  // the fixed ~6 us of this kernel is that chain).  Every split loads the new row; its owner uses it.
  uint4 qf[G];
#pragma unroll
  for (int g = 0; g < G; ++g)
    qf[g] = *reinterpret_cast<const uint4*>(q + ((size_t)b * HQ + h * G + g) * D + dl * 8);
  // unconditional loads (a "load or zero" select would need the data, i.e. a wait, right here): without an appended
  // row the pointers fall back to q and the values are never used
  const uint16_t* q0 = q + ((size_t)b * HQ + h * G) * D + dl * 8;
  const uint16_t* knp = key_new != nullptr ? key_new + b * sk_b + h * sk_h + dl * 8 : q0;
  const uint16_t* vnp = key_new != nullptr ? val_new + b * sv_b + h * sv_h + dl * 8 : q0;
  asm volatile("" : "+v"(knp), "+v"(vnp));  // opaque: hipcc otherwise folds the fallback into "wait for q, then copy"
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) const u32x4_t gptr_t;  // global, not flat: flat loads also count in lgkmcnt
  const u32x4_t kn4 = *(gptr_t*)(uintptr_t)knp, vn4 = *(gptr_t*)(uintptr_t)vnp;
  uint4 knew = make_uint4(kn4[0], kn4[1], kn4[2], kn4[3]);
  uint4 vnew = make_uint4(vn4[0], vn4[1], vn4[2], vn4[3]);
  const int lidx = lens_by_row ? bt * HKV + h : bh;
  DEC_TS(1);
  const int L_old = seq_lens[lidx];  // issued before the page-id loads, consumed after them
  const bool append = key_new != nullptr && bt != reserved;
  // Page ids: when the whole row of the page table fits the register cache (NLP <= 512 pages = 64K rows at
  // PS 128) it is loaded by ABSOLUTE logical page, which depends on batch_mapping only - so it is in flight
  // together with the length load instead of behind it (one dependent HBM/L2 round trip less before the
  // first K/V byte).  Longer tables fall back to loading just the split's window once L is known.
  const bool pg_abs = NLP <= 64 * DEC_PGREGS;  // uniform
  const int* pt = page_table + ((size_t)bt * HKV + h) * NLP;
  int pgreg[DEC_PGREGS];
  if (pg_abs) {
#pragma unroll
    for (int j = 0; j < DEC_PGREGS; ++j) {
      const int i = lane + 64 * j;
      pgreg[j] = i < NLP ? pt[i] : 0;
    }
  }
  const int L = (key_new != nullptr && bt == reserved) ? 0 : L_old + (append ? 1 : 0);
  DEC_TS(2);

  // rows are dealt to the S splits in whole units (UR rows, the granule of one wave's load): split s gets units
  // [s*U/S, (s+1)*U/S), so every split is within one unit of every other and no workgroup idles
  const int U = (L + UR - 1) / UR;
  const int start = (int)(((long long)s * U) / S) * UR;
  const int end = min((int)(((long long)(s + 1) * U) / S) * UR, L);
  const bool empty = start >= end;
  const bool owns_new = append && !empty && L_old >= start && L_old < end;

  float* po = part_o + ((size_t)(b * S + s) * HQ + h * G) * D;
  float* pl = part_lse + (size_t)(b * S + s) * HQ + h * G;

  // what this split hands to the merge: an UN-normalised partial (M, den, num[]) of the thread's OPT output dims
  constexpr int NE = G * D;  // outputs of one (b, kv-head)
  constexpr int OPT = (NE >= NW * 64) ? NE / (NW * 64) : 1;
  static_assert(NE % OPT == 0 && D % OPT == 0, "output pass tiling");
  const bool owner = tid * OPT < NE;  // this thread owns OPT consecutive dims of ONE head
  const int g_t = (tid * OPT) / D, d0 = (tid * OPT) % D;
  float pM = -INFINITY, pden = 0.f, pnum[OPT];
#pragma unroll
  for (int o = 0; o < OPT; ++o) pnum[o] = 0.f;

  if (!empty) {
    const int lp0 = start / PS;
    int pg_win = 0;  // first page of the register window, relative to lp0 (window reloads: splits longer than 512 pages)
    if (!pg_abs) {
#pragma unroll
      for (int j = 0; j < DEC_PGREGS; ++j) {
        const int i = lane + 64 * j;
        pgreg[j] = lp0 + i < NLP ? pt[lp0 + i] : 0;
      }
    }
    const int pg_bias = pg_abs ? lp0 : 0;  // page_of() takes a page index relative to the split
    // Retire every ordinary load before the first LDS-DMA and hide the registers' origin from hipcc: it would
    // otherwise wait vmcnt(0) at each later use of q / page ids, draining the ring every iteration.
#pragma unroll
    for (int j = 0; j < DEC_PGREGS; ++j) asm volatile("" : "+v"(pgreg[j]));
#pragma unroll
    for (int g = 0; g < G; ++g) asm volatile("" : "+v"(qf[g].x), "+v"(qf[g].y), "+v"(qf[g].z), "+v"(qf[g].w));
    asm volatile("" : "+v"(knew.x), "+v"(knew.y), "+v"(knew.z), "+v"(knew.w));
    asm volatile("" : "+v"(vnew.x), "+v"(vnew.y), "+v"(vnew.z), "+v"(vnew.w));

    DEC_TS(3);
    DEC_RT(1);
    float m[G], l[G], acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      m[g] = -INFINITY;
      l[g] = 0.f;
#pragma unroll
      for (int d = 0; d < 8; ++d) acc[g][d] = 0.f;
    }

    const int nunits = (end - start + UR - 1) / UR;
    const int njw = nunits > wave ? (nunits - wave + NW - 1) / NW : 0;  // units owned by this wave
    char* ring = smem + wave * (R * UNIT_BYTES);
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
    const uint32_t lane_off = lane * 16;

    auto page_of = [&](int lpi_rel) {  // lpi_rel inside the register window
      const int lpi = lpi_rel + pg_bias - pg_win;
      int pg = __builtin_amdgcn_readlane(pgreg[0], lpi & 63);
#pragma unroll
      for (int j = 1; j < DEC_PGREGS; ++j)
        if ((lpi >> 6) == j) pg = __builtin_amdgcn_readlane(pgreg[j], lpi & 63);
      return pg;
    };

    auto issue = [&](int j) {  // wave-local unit j -> ring slot j % R
      const int row0 = start + (wave + j * NW) * UR;
      const int lpi_rel = row0 / PS - lp0;
      if (!pg_abs && lpi_rel - pg_win >= 64 * DEC_PGREGS) {
        // rare (a split longer than 512 pages = 64 K rows at PS 128): the wave walks its rows in order, so the window
        // only ever moves forward.  The reload is an ordinary load: it drains the ring once (hipcc's vmcnt(0) at the
        // asm below, INSIDE this block, so the loop's other iterations keep their counted waits).
        pg_win = lpi_rel;
#pragma unroll
        for (int jj = 0; jj < DEC_PGREGS; ++jj) {
          const int i = lp0 + pg_win + lane + 64 * jj;
          pgreg[jj] = i < NLP ? pt[i] : 0;
        }
#pragma unroll
        for (int jj = 0; jj < DEC_PGREGS; ++jj) asm volatile("" : "+v"(pgreg[jj]));
      }
      const int pg = page_of(lpi_rel);
      const size_t base = ((size_t)pg * PS + (row0 % PS) + c) * D + dl * 8;
      char* slot = ring + (j % R) * UNIT_BYTES;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const uint16_t* kp = kc + base + (size_t)i * RPL * D;
        const uint16_t* vp = vc + base + (size_t)i * RPL * D;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)kp,
                                         (__attribute__((address_space(3))) void*)(slot + i * 1024), 16, 0, NT_AUX);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)vp,
                                         (__attribute__((address_space(3))) void*)(slot + (NL + i) * 1024), 16, 0,
                                         NT_AUX);
      }
    };

#pragma unroll
    for (int j = 0; j < P; ++j)
      if (j < njw) issue(j);
    DEC_TS(4);
    DEC_RT(2);

    for (int j = 0; j < njw; ++j) {
      if (j + P < njw) {
        issue(j + P);
        VmWait<2 * NL * P>::wait();
      } else {
        VmWait<0>::wait();  // tail: drain
      }
      uint4 kk[NL], vv[NL];
      {
        const uint32_t a = ring_lds + (j % R) * UNIT_BYTES + lane_off;
        static_assert(NL == 4, "unit = 4 K loads + 4 V loads");
        asm volatile(
            "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\t"
            "ds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\t"
            "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(kk[0]), "=&v"(kk[1]), "=&v"(kk[2]), "=&v"(kk[3]), "=&v"(vv[0]), "=&v"(vv[1]), "=&v"(vv[2]),
              "=&v"(vv[3])
            : "v"(a)
            : "memory");
      }
      const int rbase = start + (wave + j * NW) * UR + c;
      if (owns_new) {  // the appended row comes from registers (its cache slot still holds stale bytes)
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          const bool is_new = (rbase + i * RPL) == L_old;
          kk[i].x = is_new ? knew.x : kk[i].x; kk[i].y = is_new ? knew.y : kk[i].y;
          kk[i].z = is_new ? knew.z : kk[i].z; kk[i].w = is_new ? knew.w : kk[i].w;
          vv[i].x = is_new ? vnew.x : vv[i].x; vv[i].y = is_new ? vnew.y : vv[i].y;
          vv[i].z = is_new ? vnew.z : vv[i].z; vv[i].w = is_new ? vnew.w : vv[i].w;
        }
      }
      float sc[NL][G];
#pragma unroll
      for (int i = 0; i < NL; ++i) {
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
      for (int i = 0; i < NL; ++i) {
        const bool valid = (rbase + i * RPL) < end;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float r = group_allreduce_sum<LPR>(sc[i][g]);
          sc[i][g] = valid ? r * scale : -INFINITY;
        }
      }
      float vf[NL][8];
#pragma unroll
      for (int i = 0; i < NL; ++i) {
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
        for (int i = 0; i < NL; ++i) mx = fmaxf(mx, sc[i][g]);
        const float mxs = (mx == -INFINITY) ? 0.f : mx;
        const float alpha = __expf(m[g] - mxs);
        float p[NL];
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          p[i] = __expf(sc[i][g] - mxs);
          ps += p[i];
        }
        l[g] = l[g] * alpha + ps;
        m[g] = mx;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          float a = acc[g][d] * alpha;
#pragma unroll
          for (int i = 0; i < NL; ++i) a = fmaf(p[i], vf[i][d], a);
          acc[g][d] = a;
        }
      }
    }

    DEC_TS(5);
    DEC_RT(3);
    if (owns_new) {  // write the appended row into the paged cache (one 16-lane group owns it)
      const int u = (L_old - start) / UR;  // unit of the new row
      if ((u % NW) == wave && c == ((L_old - start) % UR) % RPL) {
        // the owning wave issued this unit last, so its page is inside the wave's register window
        const int pg = page_of(L_old / PS - lp0);
        const size_t dst = ((size_t)pg * PS + L_old % PS) * D + dl * 8;
        *reinterpret_cast<uint4*>(kc + dst) = knew;
        *reinterpret_cast<uint4*>(vc + dst) = vnew;
      }
    }

    // ---- merge of the NW * RPL partial softmax states of this split ---------------------------------------------
    // Every 16-lane row group (RPL per wave) holds its own (m, l, acc).  They are NOT combined by cross-lane
    // shuffles (two rounds of 10 dependent ds_bpermute per head: 2,600 cycles measured with s_memtime, more than the
    // whole streaming phase of a 16-row split) but staged once into the wave's own, now idle, ring region and reduced
    // by the output pass below, which needed LDS and a barrier for the cross-wave merge anyway.
    constexpr int NPART = NW * RPL;
    constexpr int ST_ML = RPL * G * D * 4;  // byte offset of the (m, l) pairs inside a wave's staging area
    static_assert(ST_ML + RPL * G * 8 <= R * UNIT_BYTES - 512, "staging area must fit the wave's ring (minus scratch)");
    {
      char* st = ring;  // all of this wave's DMA has landed and been read (tail iterations drained vmcnt / lgkmcnt)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float4* dst = reinterpret_cast<float4*>(st + ((c * G + g) * D + dl * 8) * 4);
        dst[0] = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
        dst[1] = make_float4(acc[g][4], acc[g][5], acc[g][6], acc[g][7]);
        if (dl == 0) *reinterpret_cast<float2*>(st + ST_ML + (c * G + g) * 8) = make_float2(m[g], l[g]);
      }
    }
    DEC_TS(6);
    __syncthreads();
    DEC_TS(7);
    DEC_RT(4);
    // output pass: a thread owns OPT consecutive dims of ONE head, so the NPART weights exp(m_p - M) are computed once
    // per thread (wave-uniform head -> broadcast LDS reads)
    if (owner) {
      float2 ml[NPART];
      float M = -INFINITY;
#pragma unroll
      for (int p_ = 0; p_ < NPART; ++p_) {
        const char* st = smem + (p_ / RPL) * (R * UNIT_BYTES);
        ml[p_] = *reinterpret_cast<const float2*>(st + ST_ML + ((p_ % RPL) * G + g_t) * 8);
        M = fmaxf(M, ml[p_].x);  // finite: group 0 of wave 0 always owns a valid row
      }
      float den = 0.f;
#pragma unroll
      for (int p_ = 0; p_ < NPART; ++p_) {
        const char* st = smem + (p_ / RPL) * (R * UNIT_BYTES);
        const float a = __expf(ml[p_].x - M);
        den = fmaf(a, ml[p_].y, den);
        const float* ap = reinterpret_cast<const float*>(st + (((p_ % RPL) * G + g_t) * D + d0) * 4);
#pragma unroll
        for (int o = 0; o < OPT; ++o) pnum[o] = fmaf(a, ap[o], pnum[o]);
      }
      pM = M;
      pden = den;
    }
  }

  if (S == 1) {
    // the only workgroup of (b,h): normalise and store; it has consumed L_old -> publish the new length here
    if (owner) {
      const float inv = pden > 0.f ? 1.f / pden : 0.f;  // empty (L == 0, RESERVED rows): zeros
#pragma unroll
      for (int o = 0; o < OPT; ++o) out[((size_t)b * HQ + h * G + g_t) * D + d0 + o] = to16<T>(pnum[o] * inv);
      // natural-log LSE of the scaled logits this call saw (-inf for an empty row): what a cross-device merge needs
      if (lse_out != nullptr && d0 == 0) lse_out[(size_t)b * HQ + h * G + g_t] = pden > 0.f ? pM + __logf(pden) : -INFINITY;
    }
    if (append && tid == 0) seq_lens[lidx] = L_old + 1;
  } else if (mode != 2) {
    // two-kernel path: normalised fp32 partial + lse for decode_stage2_kernel (which also publishes the length)
    if (!empty) {
      if (owner) {
        const float inv = 1.f / pden;
#pragma unroll
        for (int o = 0; o < OPT; ++o) po[g_t * D + d0 + o] = pnum[o] * inv;
        if (d0 == 0) pl[g_t] = pM + __logf(pden);
      }
    } else if (tid < G) {
      pl[tid] = -INFINITY;
    }
  } else {
    // ---- in-launch merge of the S splits of (b, kv-head) ----------------------------------------------------------
    // All S workgroups of a (b, kv-head) are co-resident (the host only selects this mode when the whole grid fits the
    // chip's CUs).  Workgroup s_m merges output elements [s_m*EPW, (s_m+1)*EPW) of the NE = G*D outputs: every split
    // sends it that slice of its un-normalised numerator and the (M, den) of the heads the slice touches, and it
    // combines them with the LSE rule (reference :391-435) in a FIXED order (bit-reproducible).
    // Hand-off (CDNA guide, Guideline 16 form R2 "the data is the flag"): every word is written exactly once per
    // launch with an agent-scope (sc1, write-through) store, carries its own validity (fp32 bits inverted, 0 = not
    // yet written), is polled with agent-scope (L1-bypassing) loads, and is reset to 0 by its ONE reader once read,
    // so the workspace is all zeros again when the launch ends: no fences, no counters, no epoch.
    typedef __attribute__((address_space(1))) unsigned gu32;
    auto st_w = [](unsigned* p, unsigned v) {
      __hip_atomic_store((gu32*)(uintptr_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto ld_w = [](const unsigned* p) {
      return __hip_atomic_load((gu32*)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto enc = [](float x) {
      const unsigned u = ~__float_as_uint(x);
      return u ? u : 1u;  // 0 is "not written"; the one pattern that maps to it (a NaN) becomes another NaN
    };
    auto dec = [](unsigned u) { return __uint_as_float(~u); };

    int EPW = (NE + S - 1) / S;
    EPW += EPW & 1;
    const int NM = (NE + EPW - 1) / EPW;      // workgroups that own a slice
    const int NHB = (EPW + D - 1) / D + 1;    // heads one slice can touch
    const int MLSZ = NHB * S * 2;
    const int AREA = MLSZ + S * EPW;          // dwords of one merger's private mailbox: [NHB][S][2] (M, den) | [S][EPW]
    unsigned* grp = merge_ws + (size_t)bh * S * AREA;
    const float rEPW = 1.0f / (float)EPW;
    auto div_epw = [&](int e) {  // e / EPW for e < 2^16 without the integer-division sequence
      int qd = (int)((float)e * rEPW);
      qd -= (qd * EPW > e) ? 1 : 0;
      qd += ((qd + 1) * EPW <= e) ? 1 : 0;
      return qd;
    };

    constexpr int SCR = NW * R * UNIT_BYTES - 512;  // tail of wave NW-1's ring: never part of a staging area
    float* sMl = reinterpret_cast<float*>(smem + SCR);          // [G][2] this split's (M, den) per head
    int* sFail = reinterpret_cast<int*>(smem + SCR + 256);
    if (owner && d0 == 0) {
      sMl[2 * g_t] = pM;
      sMl[2 * g_t + 1] = pden;
    }
    if (tid == 0) *sFail = 0;
#ifdef CVLLM_DEC_WITHHOLD  // tools/dbg test build only: split 1 never sends its numerators (and the waits are 2 ms)
    if (owner && s != 1) {
#else
    if (owner) {
#endif
#pragma unroll
      for (int o = 0; o < OPT; ++o) {
        const int e = tid * OPT + o;
        const int sm = div_epw(e);
        st_w(grp + (size_t)sm * AREA + MLSZ + s * EPW + (e - sm * EPW), enc(pnum[o]));
      }
    }
    DEC_RT(5);
    lds_barrier();  // sMl visible; every read of the staging areas is done -> LDS below is free
    for (int sm = tid; sm < NM; sm += NW * 64) {
      const int e0 = sm * EPW, e1 = min(NE, e0 + EPW);
      const int glo = e0 / D, ghi = (e1 - 1) / D;
      for (int g = glo; g <= ghi; ++g) {
        unsigned* p = grp + (size_t)sm * AREA + ((g - glo) * S + s) * 2;
        st_w(p, enc(sMl[2 * g]));
        st_w(p + 1, enc(sMl[2 * g + 1]));
      }
    }

    DEC_RT(6);
    if (s < NM) {
      constexpr int MLCAP = 544;  // nh * S <= G + 2 S (1 + 1/D) <= 528
      float* sM = reinterpret_cast<float*>(smem);
      float* sDen = sM + MLCAP;
      float* sMx = sDen + MLCAP;  // [nh <= 16] max over the splits
      float* sV = sMx + 16;       // [S][EPW] (S*EPW <= NE + 512) numerators
      const int e0 = s * EPW, e1 = min(NE, e0 + EPW);
      const int glo = e0 / D, nh = (e1 - 1) / D - glo + 1;
      const int nml = nh * S * 2;
      const int nel = e1 - e0;  // elements of this slice that exist (< EPW only in the last slice)
      unsigned* my = grp + (size_t)s * AREA;
      // item mapping without divisions: a thread walks a [S][EPW] grid in steps of (SXW, ELW), ELW a power of two
      int ELW = 2;
      while (ELW < EPW && ELW < NW * 64) ELW <<= 1;
      const int lg = 31 - __builtin_clz(ELW);
      const int SXW = (NW * 64) >> lg;
      const int el_l = tid & (ELW - 1), sx_l = tid >> lg;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
      for (;;) {  // poll: every pass re-reads this thread's words; a word is valid once it is non-zero
        bool ok = true;
        for (int j = tid; j < nml; j += NW * 64) {
          const unsigned u = ld_w(my + j);
          ok &= u != 0u;
          ((j & 1) ? sDen : sM)[j >> 1] = dec(u);
        }
        for (int el0 = 0; el0 < nel; el0 += ELW)
          for (int sx0 = 0; sx0 < S; sx0 += SXW) {
            const int el = el0 + el_l, sx = sx0 + sx_l;
            if (el < nel && sx < S) {
              const unsigned u = ld_w(my + MLSZ + sx * EPW + el);
              ok &= u != 0u;
              sV[sx * EPW + el] = dec(u);
            }
          }
        if (__all(ok)) break;
#ifdef CVLLM_DEC_WITHHOLD
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000ull) {  // 2 ms
#else
        if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) {  // 0.5 s: a sibling split never arrived
#endif
          if (lane == 0) {
            *sFail = 1;
            __hip_atomic_fetch_or((gu32*)(uintptr_t)err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
      DEC_RT(7);
      // hand the mailbox back zeroed (every word this thread polled) NOW: the write-through stores then complete under
      // the merge arithmetic below instead of holding up the end of the kernel (the barriers below are LDS-only)
      for (int j = tid; j < nml; j += NW * 64) st_w(my + j, 0u);
      for (int el0 = 0; el0 < nel; el0 += ELW)
        for (int sx0 = 0; sx0 < S; sx0 += SXW) {
          const int el = el0 + el_l, sx = sx0 + sx_l;
          if (el < nel && sx < S) st_w(my + MLSZ + sx * EPW + el, 0u);
        }
      lds_barrier();
      DEC_RT(10);
      // LSE rule (reference :391-435): w_s = exp(M_s - max_s M_s); empty splits carry M = -inf and weigh nothing.
      // Everything below avoids serial LDS chains (one wave per SIMD: ~130 cycles per dependent LDS read): maxima and
      // the denominator by wave reductions that every wave does for itself (no barrier), numerators by all threads.
      float* sP = sV + (NE + 2 * DEC_MAX_SPLITS);  // [nel][SXW] partial sums of the weighted numerators
      float* sDs = sMx;                            // [nh] sum_s w_s * den_s
      float* sMxv = sP + (NE + 2 * DEC_MAX_SPLITS);   // [nh] max_s M_s (for the optional LSE output)
      float part[2] = {0.f, 0.f};                  // nel <= 2 * ELW whenever ELW < 256; else one el per pass (below)
      for (int gi = 0; gi < nh; ++gi) {
        float mv = -INFINITY;
        for (int sx = lane; sx < S; sx += 64) mv = fmaxf(mv, sM[gi * S + sx]);
        const float mx = wave_allreduce_max(mv);  // wave-uniform
        float dv = 0.f;
        for (int sx = lane; sx < S; sx += 64) {
          const float mm = sM[gi * S + sx];
          dv += (mm == -INFINITY) ? 0.f : __expf(mm - mx) * sDen[gi * S + sx];
        }
        dv = wave_allreduce_sum_fixed(dv);  // fixed tree: bit-reproducible
        if (tid == 0) {
          sDs[gi] = dv;
          sMxv[gi] = mx;
        }
        int pi = 0;
        for (int el0 = 0; el0 < nel; el0 += ELW, ++pi) {
          const int el = el0 + el_l;
          float acc_p = 0.f;
          if (el < nel && (e0 + el) / D - glo == gi) {
            for (int sx = sx_l; sx < S; sx += SXW) {  // ascending splits: fixed order
              const float mm = sM[gi * S + sx];
              acc_p += (mm == -INFINITY) ? 0.f : __expf(mm - mx) * sV[sx * EPW + el];
            }
            if (ELW < NW * 64) part[pi & 1] = acc_p;
            else sP[el] = acc_p;  // SXW == 1: the partial is the sum
          }
        }
      }
      if (ELW < NW * 64) {
        int pi = 0;
        for (int el0 = 0; el0 < nel; el0 += ELW, ++pi) {
          const int el = el0 + el_l;
          if (el < nel) sP[el * SXW + sx_l] = part[pi & 1];
        }
      }
      lds_barrier();
      DEC_RT(12);
      const bool failed = *sFail != 0;
      if (SXW == 16) {
        // 16 partials per output element sit in 16 consecutive lanes: one LDS read + a row reduction (fixed tree)
        for (int t = tid; t < nel * 16; t += NW * 64) {
          const int el = t >> 4;
          const float a = group_allreduce_sum<16>(sP[t]);
          if ((t & 15) == 0) {
            const float ds = sDs[(e0 + el) / D - glo];
            float o = ds > 0.f ? a / ds : 0.f;  // every split empty (L == 0, RESERVED rows): zeros
            if (failed) o = __uint_as_float(0x7fc00000u);
            out[((size_t)b * HQ + h * G) * D + e0 + el] = to16<T>(o);
            if (lse_out != nullptr && (e0 + el) % D == 0)
              lse_out[(size_t)b * HQ + h * G + (e0 + el) / D] = ds > 0.f ? sMxv[(e0 + el) / D - glo] + __logf(ds) : -INFINITY;
          }
        }
      } else {
        for (int el = tid; el < nel; el += NW * 64) {
          float a = 0.f;
          for (int k = 0; k < SXW; ++k) a += sP[el * SXW + k];  // fixed order
          const float ds = sDs[(e0 + el) / D - glo];
          float o = ds > 0.f ? a / ds : 0.f;
          if (failed) o = __uint_as_float(0x7fc00000u);
          out[((size_t)b * HQ + h * G) * D + e0 + el] = to16<T>(o);
          if (lse_out != nullptr && (e0 + el) % D == 0)
            lse_out[(size_t)b * HQ + h * G + (e0 + el) / D] = ds > 0.f ? sMxv[(e0 + el) / D - glo] + __logf(ds) : -INFINITY;
        }
      }
      DEC_RT(8);
      // split 0's mailbox has been filled by ALL S splits, i.e. every one of them has read L_old: publish the new length
      if (s == 0 && append && tid == 0) seq_lens[lidx] = L_old + 1;
    }
  }
  DEC_TS(8);
  DEC_RT(9);
}
#ifdef CVLLM_DEC_TS
}  // namespace cvllm
extern "C" void cvllm_debug_read_decode_ts(unsigned long long* out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(cvllm::g_dec_ts), sizeof(unsigned long long) * 16);
}
extern "C" void cvllm_debug_read_decode_rt(unsigned long long* out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(cvllm::g_dec_rt), sizeof(unsigned long long) * 1024 * 16);
}
namespace cvllm {
#endif

template <int D, int G, int NW, int NL, int R>
constexpr size_t ring_smem_bytes() {
  return (size_t)NW * R * 2 * NL * 1024;
}

// stage 2: LSE-weighted merge of the S partials of one (b, query head)   (reference :391-435)
// 4 waves per (b, hq): every wave first reads all S lse values (<= 256), then waves take the splits
// round-robin with independent, unrolled loads; the four partial sums meet in LDS.
template <typename T, int D>
__global__ __launch_bounds__(256) void decode_stage2_kernel(float* __restrict__ part_o, float* __restrict__ part_lse,
                                                           uint16_t* __restrict__ out, int HQ, int S,
                                                           int* __restrict__ seq_lens, const int* __restrict__ bmap,
                                                           int G, int append, int lens_by_row, int reserved,
                                                           float* __restrict__ lse_out) {
  // The partials live in the same workspace the in-launch merge uses as zero-means-empty mailboxes, so this kernel
  // hands the workspace back the way it got it: every word it consumed is reset to zero.
  constexpr int VPT = D / 64;  // values per lane
  __shared__ float s_acc[4][D];
  const int bhq = blockIdx.x;
  const int b = bhq / HQ, hq = bhq % HQ;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (append && tid == 0 && (hq % G) == 0) {  // fused append: publish L_old + 1 (every stage-1 reader is done)
    const int bt = bmap[b];
    const int HKV = HQ / G, h = hq / G;
    if (bt != reserved) seq_lens[lens_by_row ? bt * HKV + h : b * HKV + h] += 1;
  }
  // The first batch of partial rows does not depend on the lse words: both sets of loads are issued together, so the
  // kernel pays ONE memory round trip before its arithmetic instead of two (it is a chain of dependent latencies: 32
  // workgroups, nothing to hide them behind).  Rows of splits >= S are clamped to split 0 and weighted 0.
  auto load_rows = [&](int s0, float (&v)[4][VPT]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = s0 + wave + 4 * i;
      const int sc = s < S ? s : 0;
      const float* po = part_o + ((size_t)(b * S + sc) * HQ + hq) * D + lane * VPT;
#pragma unroll
      for (int j = 0; j < VPT; ++j) v[i][j] = po[j];
    }
  };
  float v[4][VPT];
  load_rows(0, v);
  float lse[DEC_MAX_SPLITS / 64];
  float M = -INFINITY;
#pragma unroll
  for (int j = 0; j < DEC_MAX_SPLITS / 64; ++j) {
    const int s = lane + 64 * j;
    lse[j] = s < S ? part_lse[(size_t)(b * S + s) * HQ + hq] : -INFINITY;
    M = fmaxf(M, lse[j]);
  }
  M = wave_reduce_max(M);
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < DEC_MAX_SPLITS / 64; ++j) {
    lse[j] = (lse[j] == -INFINITY) ? 0.f : __expf(lse[j] - M);  // weight; empty splits (never written) -> 0
    den += lse[j];
  }
  den = wave_reduce_sum(den);
  float acc[VPT];
#pragma unroll
  for (int j = 0; j < VPT; ++j) acc[j] = 0.f;
  for (int s0 = 0; s0 < S; s0 += 16) {  // 4 independent partial rows per wave, the next batch in flight behind them
    float vn[4][VPT];
    if (s0 + 16 < S) load_rows(s0 + 16, vn);
    float w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = s0 + wave + 4 * i;
      w[i] = 0.f;
#pragma unroll
      for (int j = 0; j < DEC_MAX_SPLITS / 64; ++j)
        if ((s >> 6) == j) w[i] = __shfl(lse[j], s & 63, 64);
      if (s >= S) w[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < VPT; ++j) acc[j] += (w[i] != 0.f) ? w[i] * v[i][j] : 0.f;  // never 0 * garbage
      const int s = s0 + wave + 4 * i;
      if (s < S) {  // this wave is the only reader of split s's row: hand it back zeroed
        float* po = part_o + ((size_t)(b * S + s) * HQ + hq) * D + lane * VPT;
#pragma unroll
        for (int j = 0; j < VPT; ++j) po[j] = 0.f;
      }
    }
    if (s0 + 16 < S) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < VPT; ++j) v[i][j] = vn[i][j];
    }
  }
#pragma unroll
  for (int j = 0; j < VPT; ++j) s_acc[wave][lane * VPT + j] = acc[j];
  __syncthreads();  // also: every wave has read the lse words
  if (tid < D) {
    const float inv = den > 0.f ? 1.f / den : 0.f;
    const float v = (s_acc[0][tid] + s_acc[1][tid] + s_acc[2][tid] + s_acc[3][tid]) * inv;
    out[((size_t)b * HQ + hq) * D + tid] = to16<T>(v);
    if (lse_out != nullptr && tid == 0) lse_out[(size_t)b * HQ + hq] = den > 0.f ? M + __logf(den) : -INFINITY;
  }
  for (int s = tid; s < S; s += 256) part_lse[(size_t)(b * S + s) * HQ + hq] = 0.f;
}

// f-4: merge of W shard results of one decode step (a sequence's KV rows partitioned over W devices, each device ran
// decode attention over its rows and all-gathered (out, lse)); the LSE rule of the reference's stage 2 (:391-435):
// out = sum_r exp(lse_r - max) out_r / sum_r exp(lse_r - max).  One thread per 8 output values.
template <typename T>
__global__ __launch_bounds__(256) void decode_merge_shards_kernel(const uint16_t* __restrict__ o_all,
                                                                  const float* __restrict__ lse_all,
                                                                  uint16_t* __restrict__ out, int W, long rows, int D) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // (row, 8-value chunk)
  const int cpr = D / 8;
  if (i >= rows * cpr) return;
  const long row = i / cpr;
  const int c = (int)(i % cpr);
  float mx = -INFINITY;
  for (int r = 0; r < W; ++r) mx = fmaxf(mx, lse_all[(size_t)r * rows + row]);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, den = 0.f;
  for (int r = 0; r < W; ++r) {  // fixed order: reproducible
    const float l = lse_all[(size_t)r * rows + row];
    if (l == -INFINITY) continue;  // a shard with no rows for this head contributes nothing
    const float w = __expf(l - mx);
    den += w;
    const uint4 u = *reinterpret_cast<const uint4*>(o_all + ((size_t)r * rows + row) * D + 8 * c);
    const float2 a = unpack2<T>(u.x), b2 = unpack2<T>(u.y), c2 = unpack2<T>(u.z), d2 = unpack2<T>(u.w);
    acc[0] = fmaf(w, a.x, acc[0]); acc[1] = fmaf(w, a.y, acc[1]); acc[2] = fmaf(w, b2.x, acc[2]); acc[3] = fmaf(w, b2.y, acc[3]);
    acc[4] = fmaf(w, c2.x, acc[4]); acc[5] = fmaf(w, c2.y, acc[5]); acc[6] = fmaf(w, d2.x, acc[6]); acc[7] = fmaf(w, d2.y, acc[7]);
  }
  const float inv = den > 0.f ? 1.f / den : 0.f;
  const uint4 o = make_uint4(pack2<T>(acc[0] * inv, acc[1] * inv), pack2<T>(acc[2] * inv, acc[3] * inv),
                             pack2<T>(acc[4] * inv, acc[5] * inv), pack2<T>(acc[6] * inv, acc[7] * inv));
  *reinterpret_cast<uint4*>(out + (size_t)row * D + 8 * c) = o;
}

constexpr size_t DEC_WS_HEADER = 256;  // bytes: word 0 = in-launch merge error flag; the rest reserved (zero)

struct DecodeArgs {
  const void *q, *key_new, *val_new;
  void *kc, *vc, *out;
  int* seq_lens;
  const int *page_table, *bmap;
  char* ws;
  size_t ws_bytes;
  float* lse_out;  // optional [B, HQ]: natural-log LSE of every query head over the rows this call saw
  int64_t sk_b, sk_h, sv_b, sv_h;
  int B, HKV, PS, NLP, S, lens_by_row, reserved;
  float scale;
  hipStream_t st;
};

// dwords of the in-launch merge mailboxes for (B*HKV groups, S splits, NE = G*D outputs per group); mirrors the
// kernel's EPW / NHB / AREA arithmetic
static size_t merge_mailbox_dwords(int groups, int S, int NE, int D) {
  int EPW = (NE + S - 1) / S;
  EPW += EPW & 1;
  const int NHB = (EPW + D - 1) / D + 1;
  return (size_t)groups * S * ((size_t)NHB * S * 2 + (size_t)S * EPW);
}

// CUs of the current device (cached per device): the in-launch merge needs every workgroup of the grid resident
static int device_cus() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cus[dev] = n;
  }
  return cus[dev];
}

// Which split merge runs when the grid fits the chip (one ring workgroup per CU, all resident).  Default: inside the
// launch - chosen by the measurement the round-2 review asked for: rocprofv3 --kernel-trace of the C3 launch INSIDE the
// engine's HIP graph reads 23.68 us for the one kernel against 20.15 + 4.75 us for stage 1 + decode_stage2_kernel on the
// same box (profiles/r03_decode_merge_ab.txt; tokens/s +0.26 %).  The in-launch form assumes co-residency, which a plain
// launch cannot guarantee: every wait is bounded, a timeout raises the sticky error word, and the engine polls it after
// every decode loop and falls back to the two-kernel path for the rest of the process (core/model_runner.py).
// CVLLM_DECODE_MERGE=two-kernel (or cvllm_decode_set_merge_mode(1)) forces the two-kernel path.
static int g_merge_override = 0;  // cvllm_decode_set_merge_mode: 0 = environment / default, 1 = two-kernel, 2 = in-launch
static bool merge_in_launch_allowed() {
  if (g_merge_override) return g_merge_override == 2;
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CVLLM_DECODE_MERGE");
    v = (e && (e[0] == 't' || e[0] == '2')) ? 0 : 1;
  }
  return v == 1;
}

template <typename T, int D, int G>
static int launch_fused(const DecodeArgs& a) {
  constexpr int NW = 4, NL = 4, R = 4;
  const int HQ = a.HKV * G;
  constexpr size_t smem = ring_smem_bytes<D, G, NW, NL, R>();
  static_assert(smem <= 160 * 1024, "LDS budget");
  auto kern = decode_fused_kernel<T, D, G, NW, NL, R>;
  set_dyn_lds_once(kern, (int)smem);
  const int grid = a.B * a.HKV * a.S;
  // mode 0: one split, direct output.  mode 2: splits merged inside the launch (needs the whole grid co-resident: one
  // workgroup per CU).  mode 1: fp32 partials + decode_stage2_kernel (oversubscribed grids, or forced by the environment).
  int mode = 0;
  if (a.S > 1) {
    const bool fits = a.ws_bytes >= DEC_WS_HEADER + 4 * merge_mailbox_dwords(a.B * a.HKV, a.S, G * D, D);
    mode = (merge_in_launch_allowed() && grid <= device_cus() && fits) ? 2 : 1;
  }
  char* body = a.ws ? a.ws + DEC_WS_HEADER : nullptr;  // no workspace is needed (or read) with one split
  unsigned* err_word = reinterpret_cast<unsigned*>(a.ws);
  float* part_o = reinterpret_cast<float*>(body);
  float* part_lse = mode == 1 ? part_o + (size_t)a.B * a.S * HQ * D : nullptr;
  unsigned* mailboxes = reinterpret_cast<unsigned*>(body);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, a.st, (const uint16_t*)a.q, a.bmap, a.seq_lens,
                     a.page_table, (const uint16_t*)a.key_new, (const uint16_t*)a.val_new, (uint16_t*)a.kc,
                     (uint16_t*)a.vc, a.HKV, a.PS, a.NLP, a.S, a.scale, a.lens_by_row, a.reserved, a.sk_b, a.sk_h,
                     a.sv_b, a.sv_h, (uint16_t*)a.out, part_o, part_lse, mailboxes, err_word, mode, a.lse_out);
  if (mode == 1)
    hipLaunchKernelGGL((decode_stage2_kernel<T, D>), dim3(a.B * HQ), dim3(256), 0, a.st, part_o, part_lse,
                       (uint16_t*)a.out, HQ, a.S, a.seq_lens, a.bmap, G, a.key_new != nullptr ? 1 : 0, a.lens_by_row,
                       a.reserved, a.lse_out);
  return check_launch();
}

// D = 256 fallback: register double-buffered stage 1 + separate merge kernel (no fused append)
template <typename T, int D, int G>
static int launch_fallback(const DecodeArgs& a) {
  const int HQ = a.HKV * G;
  float* part_o = reinterpret_cast<float*>(a.ws + DEC_WS_HEADER);
  float* part_lse = part_o + (size_t)a.B * a.S * HQ * D;
  dim3 grid(a.B * a.HKV * a.S), block(DEC_NW * 64);
  if (a.S == 1) {
    hipLaunchKernelGGL((decode_stage1_kernel<T, D, G, true, 4, 2>), grid, block, 0, a.st, (const uint16_t*)a.q,
                       (const uint16_t*)a.kc, (const uint16_t*)a.vc, (uint16_t*)a.out, part_o, part_lse, a.seq_lens,
                       a.page_table, a.bmap, a.HKV, a.PS, a.NLP, a.S, a.scale, a.lens_by_row);
    return check_launch();
  }
  hipLaunchKernelGGL((decode_stage1_kernel<T, D, G, false, 4, 2>), grid, block, 0, a.st, (const uint16_t*)a.q,
                     (const uint16_t*)a.kc, (const uint16_t*)a.vc, (uint16_t*)a.out, part_o, part_lse, a.seq_lens,
                     a.page_table, a.bmap, a.HKV, a.PS, a.NLP, a.S, a.scale, a.lens_by_row);
  hipLaunchKernelGGL((decode_stage2_kernel<T, D>), dim3(a.B * HQ), dim3(256), 0, a.st, part_o, part_lse,
                     (uint16_t*)a.out, HQ, a.S, a.seq_lens, a.bmap, G, 0, a.lens_by_row, a.reserved, a.lse_out);
  return check_launch();
}

template <typename T, int D, int G>
static int launch_decode(const DecodeArgs& a) {
  if constexpr (D <= 128) return launch_fused<T, D, G>(a);
  else return launch_fallback<T, D, G>(a);
}

template <typename T, int D>
static int dispatch_g(int G, const DecodeArgs& a) {
  switch (G) {
    case 1: return launch_decode<T, D, 1>(a);
    case 2: return launch_decode<T, D, 2>(a);
    case 4: return launch_decode<T, D, 4>(a);
    case 8: return launch_decode<T, D, 8>(a);
    default: return CVLLM_ERR_SHAPE;
  }
}

template <typename T>
static int dispatch_d(int D, int G, const DecodeArgs& a) {
  switch (D) {
    case 64: return dispatch_g<T, 64>(G, a);
    case 128: return dispatch_g<T, 128>(G, a);
    case 256: return dispatch_g<T, 256>(G, a);
    default: return CVLLM_ERR_SHAPE;
  }
}

}  // namespace cvllm

using namespace cvllm;

extern "C" size_t cvllm_decode_workspace_bytes(int B, int HQ, int D, int n_splits) {
  if (B <= 0 || HQ <= 0 || D <= 0 || n_splits <= 0) return 0;
  // header | max(fp32 partials + lse of the two-kernel path, in-launch merge mailboxes for any G in {1,2,4,8})
  const size_t S = (size_t)n_splits;
  const size_t two_kernel = ((size_t)B * S * HQ * D + (size_t)B * S * HQ) * sizeof(float);
  const size_t mailboxes = (size_t)B * HQ * S * ((size_t)D + 8 * S + 8) * sizeof(unsigned);  // bound of merge_mailbox_dwords
  return DEC_WS_HEADER + (two_kernel > mailboxes ? two_kernel : mailboxes);
}

static int decode_common(DecodeArgs a, int HQ, int D, int dtype, size_t workspace_bytes) {
  if (!a.q || !a.kc || !a.vc || !a.out || !a.seq_lens || !a.page_table || !a.bmap) return CVLLM_ERR_ARG;
  if (a.B <= 0 || HQ <= 0 || a.HKV <= 0 || a.S <= 0 || a.NLP <= 0) return CVLLM_ERR_ARG;
  if (HQ % a.HKV != 0 || a.S > DEC_MAX_SPLITS) return CVLLM_ERR_SHAPE;
  // a unit of (64/(D/8))*4 rows must not straddle a page; the reference requires PAGE_SIZE % 32 == 0 (:80)
  if (a.PS <= 0 || a.PS % 32 != 0) return CVLLM_ERR_SHAPE;
  if (a.S > 1 && (!a.ws || workspace_bytes < cvllm_decode_workspace_bytes(a.B, HQ, D, a.S))) return CVLLM_ERR_WORKSPACE;
  a.ws_bytes = workspace_bytes;
  if (a.key_new && D > 128) return CVLLM_ERR_SHAPE;  // fused append exists in the ring kernel only
  if (a.lse_out && D > 128) return CVLLM_ERR_SHAPE;  // the LSE output too
  const int G = HQ / a.HKV;
  if (dtype == CVLLM_F16) return dispatch_d<F16>(D, G, a);
  if (dtype == CVLLM_BF16) return dispatch_d<BF16>(D, G, a);
  return CVLLM_ERR_SHAPE;
}

extern "C" int cvllm_decode_attn(const void* q, const void* k_cache, const void* v_cache, void* out,
                                 const int32_t* seq_lens_bh, const int32_t* page_table,
                                 const int32_t* batch_mapping, void* workspace, size_t workspace_bytes, int B,
                                 int HQ, int HKV, int D, int page_size, int n_logical_pages_max,
                                 float sm_scale, int n_splits, int dtype, cvllm_stream_t stream) {
  DecodeArgs a{};
  a.q = q; a.kc = (void*)k_cache; a.vc = (void*)v_cache; a.out = out;
  a.seq_lens = (int*)seq_lens_bh;  // read-only without an appended row
  a.page_table = page_table; a.bmap = batch_mapping; a.ws = (char*)workspace;
  a.B = B; a.HKV = HKV; a.PS = page_size; a.NLP = n_logical_pages_max; a.S = n_splits; a.lens_by_row = 0;
  a.reserved = -1; a.scale = sm_scale; a.st = (hipStream_t)stream;
  return decode_common(a, HQ, D, dtype, workspace_bytes);
}

// decode attention that also returns the natural-log LSE of every query head over the rows it saw: one shard of a
// cross-device split-KV decode; merged with cvllm_decode_merge_shards after the all-gather
extern "C" int cvllm_decode_attn_lse(const void* q, const void* k_cache, const void* v_cache, void* out, float* lse_out,
                                     const int32_t* seq_lens_bh, const int32_t* page_table,
                                     const int32_t* batch_mapping, void* workspace, size_t workspace_bytes, int B,
                                     int HQ, int HKV, int D, int page_size, int n_logical_pages_max, float sm_scale,
                                     int n_splits, int dtype, cvllm_stream_t stream) {
  if (!lse_out) return CVLLM_ERR_ARG;
  DecodeArgs a{};
  a.q = q; a.kc = (void*)k_cache; a.vc = (void*)v_cache; a.out = out; a.lse_out = lse_out;
  a.seq_lens = (int*)seq_lens_bh;
  a.page_table = page_table; a.bmap = batch_mapping; a.ws = (char*)workspace;
  a.B = B; a.HKV = HKV; a.PS = page_size; a.NLP = n_logical_pages_max; a.S = n_splits; a.lens_by_row = 0;
  a.reserved = -1; a.scale = sm_scale; a.st = (hipStream_t)stream;
  return decode_common(a, HQ, D, dtype, workspace_bytes);
}

extern "C" int cvllm_decode_merge_shards(const void* out_all, const float* lse_all, void* out, int n_shards, int B,
                                         int HQ, int D, int dtype, cvllm_stream_t stream) {
  if (!out_all || !lse_all || !out) return CVLLM_ERR_ARG;
  if (n_shards <= 0 || B <= 0 || HQ <= 0 || D <= 0 || D % 8) return CVLLM_ERR_ARG;
  const long rows = (long)B * HQ;
  const long n = rows * (D / 8);
  const int blocks = (int)((n + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVLLM_F16)
    hipLaunchKernelGGL(decode_merge_shards_kernel<F16>, dim3(blocks), dim3(256), 0, st, (const uint16_t*)out_all, lse_all,
                       (uint16_t*)out, n_shards, rows, D);
  else if (dtype == CVLLM_BF16)
    hipLaunchKernelGGL(decode_merge_shards_kernel<BF16>, dim3(blocks), dim3(256), 0, st, (const uint16_t*)out_all, lse_all,
                       (uint16_t*)out, n_shards, rows, D);
  else
    return CVLLM_ERR_SHAPE;
  return check_launch();
}

// Fused decode step of the boundary orchestrator (cv/layers/attention.py:127-160 decode branch): append the new
// K/V row of every (b,h) at bh_seq_lens[batch_mapping[b], h] (in place, RESERVED rows skipped), then attend.
// bh_seq_lens is the LAYER's full [Bmax+1, HKV] table: no index_select / index_copy round trip on the host side.
extern "C" int cvllm_decode_append_attn(const void* q, const void* key, const void* value, int64_t sk_b, int64_t sk_h,
                                        int64_t sv_b, int64_t sv_h, void* k_cache, void* v_cache, void* out,
                                        int32_t* bh_seq_lens, const int32_t* page_table,
                                        const int32_t* batch_mapping, void* workspace, size_t workspace_bytes, int B,
                                        int HQ, int HKV, int D, int page_size, int n_logical_pages_max, float sm_scale,
                                        int n_splits, int reserved_batch, int dtype, cvllm_stream_t stream) {
  if (!key || !value) return CVLLM_ERR_ARG;
  if ((sk_b % 8) || (sk_h % 8) || (sv_b % 8) || (sv_h % 8)) return CVLLM_ERR_SHAPE;
  DecodeArgs a{};
  a.q = q; a.key_new = key; a.val_new = value; a.kc = k_cache; a.vc = v_cache; a.out = out;
  a.seq_lens = bh_seq_lens; a.page_table = page_table; a.bmap = batch_mapping; a.ws = (char*)workspace;
  a.sk_b = sk_b; a.sk_h = sk_h; a.sv_b = sv_b; a.sv_h = sv_h;
  a.B = B; a.HKV = HKV; a.PS = page_size; a.NLP = n_logical_pages_max; a.S = n_splits; a.lens_by_row = 1;
  a.reserved = reserved_batch; a.scale = sm_scale; a.st = (hipStream_t)stream;
  return decode_common(a, HQ, D, dtype, workspace_bytes);
}

// Process-wide choice of the split merge for grids that fit the chip (host state, not a launch): 0 = what the
// environment says (CVLLM_DECODE_MERGE, default in-launch), 1 = two-kernel, 2 = in-launch.  Returns the previous value.
extern "C" int cvllm_decode_set_merge_mode(int mode) {
  const int prev = g_merge_override;
  if (mode >= 0 && mode <= 2) g_merge_override = mode;
  return prev;
}

// Status of the in-launch split merge of the calls that used `workspace` so far: 0 = fine, 1 = a merging workgroup
// gave up waiting for a sibling split (outputs of that call hold NaN; re-zero the workspace before reusing it).
// Synchronises the stream: a health check for tests and engines, not part of the data path.
extern "C" int cvllm_decode_merge_status(const void* workspace, cvllm_stream_t stream) {
  if (!workspace) return CVLLM_ERR_ARG;
  unsigned w = 0;
  if (hipMemcpyAsync(&w, workspace, sizeof(w), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
    return CVLLM_ERR_LAUNCH;
  return (int)(w & 1u);
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
