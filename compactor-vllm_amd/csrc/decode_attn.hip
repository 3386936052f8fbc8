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
#define DEC_TS(i)                                                                    \
  do {                                                                               \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_dec_ts[i] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define DEC_TS(i) \
  do {            \
  } while (0)
#endif

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
    uint16_t* __restrict__ out, float* __restrict__ part_o, float* __restrict__ part_lse) {
  // hipcc fetches kernel arguments lazily, one dependent s_load + wait in front of each first use (four round trips
  // in this kernel's prologue); naming them all here puts every fetch into one clause.
  asm volatile("" ::"s"(vc), "s"(HKV), "s"(PS), "s"(NLP), "s"(S), "s"(scale), "s"(lens_by_row), "s"(reserved));
  asm volatile("" ::"s"(sk_b), "s"(sk_h), "s"(sv_b), "s"(sv_h), "s"(out), "s"(part_o), "s"(part_lse));
  DEC_TS(0);
  constexpr int LPR = D / 8;
  constexpr int RPL = 64 / LPR;
  constexpr int UR = RPL * NL;
  constexpr int ROUND = UR * NW;
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

  int per = (L + S - 1) / S;
  per = (per + ROUND - 1) / ROUND * ROUND;
  const int start = s * per;
  const int end = min(start + per, L);
  const bool empty = start >= end;
  const bool owns_new = append && !empty && L_old >= start && L_old < end;

  float* po = part_o + ((size_t)(b * S + s) * HQ + h * G) * D;
  float* pl = part_lse + (size_t)(b * S + s) * HQ + h * G;

  if (!empty) {
    const int lp0 = start / PS;
    if (!pg_abs) {
      const int nlp = (end - 1) / PS - lp0 + 1;  // <= 64 * DEC_PGREGS (checked on the host)
#pragma unroll
      for (int j = 0; j < DEC_PGREGS; ++j) {
        const int i = lane + 64 * j;
        pgreg[j] = i < nlp ? pt[lp0 + i] : 0;
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

    auto page_of = [&](int lpi_rel) {
      const int lpi = lpi_rel + pg_bias;
      int pg = __builtin_amdgcn_readlane(pgreg[0], lpi & 63);
#pragma unroll
      for (int j = 1; j < DEC_PGREGS; ++j)
        if ((lpi >> 6) == j) pg = __builtin_amdgcn_readlane(pgreg[j], lpi & 63);
      return pg;
    };

    auto issue = [&](int j) {  // wave-local unit j -> ring slot j % R
      const int row0 = start + (wave + j * NW) * UR;
      const int pg = page_of(row0 / PS - lp0);
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
    if (owns_new) {  // write the appended row into the paged cache (one 16-lane group owns it)
      const int u = (L_old - start) / UR;  // unit of the new row
      if ((u % NW) == wave && c == ((L_old - start) % UR) % RPL) {
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
    static_assert(ST_ML + RPL * G * 8 <= R * UNIT_BYTES, "staging area must fit the wave's ring");
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
    // output pass: a thread owns OPT consecutive dims of ONE head, so the NPART weights exp(m_p - M) are computed once
    // per thread (wave-uniform head -> broadcast LDS reads)
    constexpr int OPT = (G * D >= NW * 64) ? (G * D) / (NW * 64) : 1;
    static_assert((G * D) % OPT == 0 && D % OPT == 0, "output pass tiling");
    if (tid * OPT < G * D) {
      const int g = (tid * OPT) / D, d0 = (tid * OPT) % D;
      float2 ml[NPART];
      float M = -INFINITY;
#pragma unroll
      for (int p_ = 0; p_ < NPART; ++p_) {
        const char* st = smem + (p_ / RPL) * (R * UNIT_BYTES);
        ml[p_] = *reinterpret_cast<const float2*>(st + ST_ML + ((p_ % RPL) * G + g) * 8);
        M = fmaxf(M, ml[p_].x);  // finite: group 0 of wave 0 always owns a valid row
      }
      float num[OPT], den = 0.f;
#pragma unroll
      for (int o = 0; o < OPT; ++o) num[o] = 0.f;
#pragma unroll
      for (int p_ = 0; p_ < NPART; ++p_) {
        const char* st = smem + (p_ / RPL) * (R * UNIT_BYTES);
        const float a = __expf(ml[p_].x - M);
        den = fmaf(a, ml[p_].y, den);
        const float* ap = reinterpret_cast<const float*>(st + (((p_ % RPL) * G + g) * D + d0) * 4);
#pragma unroll
        for (int o = 0; o < OPT; ++o) num[o] = fmaf(a, ap[o], num[o]);
      }
      const float inv = 1.f / den;
#pragma unroll
      for (int o = 0; o < OPT; ++o) {
        const float ov = num[o] * inv;
        if (S == 1) {
          out[((size_t)b * HQ + h * G + g) * D + d0 + o] = to16<T>(ov);
        } else {
          po[g * D + d0 + o] = ov;
        }
      }
      if (S != 1 && d0 == 0) pl[g] = M + __logf(den);
    }
  } else {  // empty split (covers L == 0 and RESERVED rows)
    if (S == 1) {
      for (int i = tid; i < G * D; i += NW * 64) out[((size_t)b * HQ + h * G) * D + i] = 0;
    } else if (tid < G) {
      pl[tid] = -INFINITY;
    }
  }

  // S == 1: this is the only workgroup of (b,h) and it has consumed L_old -> publish the new length here.
  // S > 1: the merge kernel (stream-ordered after every stage-1 workgroup) publishes it.
  if (S == 1 && append && tid == 0) seq_lens[lidx] = L_old + 1;
  DEC_TS(8);
}
#ifdef CVLLM_DEC_TS
}  // namespace cvllm
extern "C" void cvllm_debug_read_decode_ts(unsigned long long* out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(cvllm::g_dec_ts), sizeof(unsigned long long) * 16);
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
__global__ __launch_bounds__(256) void decode_stage2_kernel(const float* __restrict__ part_o,
                                                           const float* __restrict__ part_lse,
                                                           uint16_t* __restrict__ out, int HQ, int S,
                                                           int* __restrict__ seq_lens, const int* __restrict__ bmap,
                                                           int G, int append, int lens_by_row, int reserved) {
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
  for (int s0 = 0; s0 < S; s0 += 16) {  // 4 independent partial rows per wave in flight
    float w[4];
    float v[4][VPT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = s0 + wave + 4 * i;
      w[i] = 0.f;
#pragma unroll
      for (int j = 0; j < DEC_MAX_SPLITS / 64; ++j)
        if ((s >> 6) == j) w[i] = __shfl(lse[j], s & 63, 64);
      if (s >= S) w[i] = 0.f;
      const int sc = s < S ? s : 0;  // clamp: always a valid address; weight 0 discards it
      const float* po = part_o + ((size_t)(b * S + sc) * HQ + hq) * D + lane * VPT;
#pragma unroll
      for (int j = 0; j < VPT; ++j) v[i][j] = po[j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < VPT; ++j) acc[j] += (w[i] != 0.f) ? w[i] * v[i][j] : 0.f;  // never 0 * garbage
  }
#pragma unroll
  for (int j = 0; j < VPT; ++j) s_acc[wave][lane * VPT + j] = acc[j];
  __syncthreads();
  if (tid < D) {
    const float inv = den > 0.f ? 1.f / den : 0.f;
    const float v = (s_acc[0][tid] + s_acc[1][tid] + s_acc[2][tid] + s_acc[3][tid]) * inv;
    out[((size_t)b * HQ + hq) * D + tid] = to16<T>(v);
  }
}

static hipEvent_t g_evt_start = nullptr, g_evt_stop = nullptr;  // bench.py roofline leg
static int g_skip_stage2 = 0;                                        // bench.py roofline leg: stage-1 launches only

struct DecodeArgs {
  const void *q, *key_new, *val_new;
  void *kc, *vc, *out;
  int* seq_lens;
  const int *page_table, *bmap;
  float* ws;
  int64_t sk_b, sk_h, sv_b, sv_h;
  int B, HKV, PS, NLP, S, lens_by_row, reserved;
  float scale;
  hipStream_t st;
};



template <typename T, int D, int G>
static int launch_fused(const DecodeArgs& a) {
  constexpr int NW = 4, NL = 4, R = 4;
  const int HQ = a.HKV * G;
  float* part_o = a.ws;
  float* part_lse = part_o + (size_t)a.B * a.S * HQ * D;
  constexpr size_t smem = ring_smem_bytes<D, G, NW, NL, R>();
  static_assert(smem <= 160 * 1024, "LDS budget");
  auto kern = decode_fused_kernel<T, D, G, NW, NL, R>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  if (g_evt_start) (void)hipEventRecord(g_evt_start, a.st);
  hipLaunchKernelGGL(kern, dim3(a.B * a.HKV * a.S), dim3(NW * 64), smem, a.st, (const uint16_t*)a.q, a.bmap, a.seq_lens,
                     a.page_table, (const uint16_t*)a.key_new, (const uint16_t*)a.val_new, (uint16_t*)a.kc,
                     (uint16_t*)a.vc, a.HKV, a.PS, a.NLP, a.S, a.scale, a.lens_by_row, a.reserved, a.sk_b, a.sk_h,
                     a.sv_b, a.sv_h, (uint16_t*)a.out, part_o, part_lse);
  if (g_evt_stop) (void)hipEventRecord(g_evt_stop, a.st);
  if (a.S > 1 && !g_skip_stage2)
    hipLaunchKernelGGL((decode_stage2_kernel<T, D>), dim3(a.B * HQ), dim3(256), 0, a.st, part_o, part_lse,
                       (uint16_t*)a.out, HQ, a.S, a.seq_lens, a.bmap, G, a.key_new != nullptr ? 1 : 0, a.lens_by_row,
                       a.reserved);
  return check_launch();
}

// D = 256 fallback: register double-buffered stage 1 + separate merge kernel (no fused append)
template <typename T, int D, int G>
static int launch_fallback(const DecodeArgs& a) {
  const int HQ = a.HKV * G;
  float* part_o = a.ws;
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
                     (uint16_t*)a.out, HQ, a.S, a.seq_lens, a.bmap, G, 0, a.lens_by_row, a.reserved);
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
  return ((size_t)B * n_splits * HQ * D + (size_t)B * n_splits * HQ) * sizeof(float);  // fp32 partials | fp32 lse
}

static int decode_common(DecodeArgs a, int HQ, int D, int dtype, size_t workspace_bytes) {
  if (!a.q || !a.kc || !a.vc || !a.out || !a.seq_lens || !a.page_table || !a.bmap) return CVLLM_ERR_ARG;
  if (a.B <= 0 || HQ <= 0 || a.HKV <= 0 || a.S <= 0 || a.NLP <= 0) return CVLLM_ERR_ARG;
  if (HQ % a.HKV != 0 || a.S > DEC_MAX_SPLITS) return CVLLM_ERR_SHAPE;
  // a unit of (64/(D/8))*4 rows must not straddle a page; the reference requires PAGE_SIZE % 32 == 0 (:80)
  if (a.PS <= 0 || a.PS % 32 != 0) return CVLLM_ERR_SHAPE;
  // the ring kernel keeps page ids in 8 VGPRs (512 pages): the whole row, or else one split's window
  if (a.NLP > 64 * DEC_PGREGS && (a.NLP + a.S - 1) / a.S + 1 > 64 * DEC_PGREGS) return CVLLM_ERR_SHAPE;
  if (a.S > 1 && (!a.ws || workspace_bytes < cvllm_decode_workspace_bytes(a.B, HQ, D, a.S))) return CVLLM_ERR_WORKSPACE;
  if (a.key_new && D > 128) return CVLLM_ERR_SHAPE;  // fused append exists in the ring kernel only
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
  a.page_table = page_table; a.bmap = batch_mapping; a.ws = (float*)workspace;
  a.B = B; a.HKV = HKV; a.PS = page_size; a.NLP = n_logical_pages_max; a.S = n_splits; a.lens_by_row = 0;
  a.reserved = -1; a.scale = sm_scale; a.st = (hipStream_t)stream;
  return decode_common(a, HQ, D, dtype, workspace_bytes);
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
  a.seq_lens = bh_seq_lens; a.page_table = page_table; a.bmap = batch_mapping; a.ws = (float*)workspace;
  a.sk_b = sk_b; a.sk_h = sk_h; a.sv_b = sv_b; a.sv_h = sv_h;
  a.B = B; a.HKV = HKV; a.PS = page_size; a.NLP = n_logical_pages_max; a.S = n_splits; a.lens_by_row = 1;
  a.reserved = reserved_batch; a.scale = sm_scale; a.st = (hipStream_t)stream;
  return decode_common(a, HQ, D, dtype, workspace_bytes);
}

// bench.py: HIP events recorded immediately before / after the decode kernel launch on the launch stream
extern "C" void cvllm_debug_set_decode_events(void* start, void* stop) {
  g_evt_start = (hipEvent_t)start;
  g_evt_stop = (hipEvent_t)stop;
}

// bench.py: launch only decode_fused_kernel (no merge kernel; the output is then NOT an attention result) so that a
// run of back-to-back launches times that kernel alone
extern "C" void cvllm_debug_set_decode_stage2(int enabled) { g_skip_stage2 = enabled ? 0 : 1; }

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
