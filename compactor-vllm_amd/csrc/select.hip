// a9 — joint top-k selection + per-head page padding, without a global sort.
//
// Replaces cv/compression/common.py:171-243 (scores_to_retain_indices: pad + torch.topk full sort,
// python loop with host syncs) and the rank-consuming logic of cv/kv_cache/store_kv_cache.py:9-78,
// 178-248 (atomic scatter + serial per-head pad scan).
//
// Semantics: rank all (token, head) pairs of a sequence by (score desc, flat index asc);
// keep ranks < retain_b; every head whose new length is not a page multiple also keeps its own next-ranked
// tokens until the page is full / its tokens are exhausted / written >= ctx_len - L.  Because the joint
// order restricted to one head IS that head's order, the final kept set of head h is simply the top
//   t_h = c_h + pad_h     of head h's own scores,   c_h = #{kept pairs of head h in the joint top-retain}.
// So:  kernel 1 (one workgroup per sequence): exact radix select of the retain-th largest joint key,
//      count c_h, derive t_h;   kernel 2 (one workgroup per (sequence, head)): exact radix select of the
//      t_h-th largest key of the head, then an ordered compaction (block scan) of the kept token indices.
// Both are L2-resident passes over <= L*H floats; they run on the store stream under the prefill attention.
#include "common.h"

namespace cvllm {

// -DCVLLM_SEL_TS (debug builds of tools/dbg only): s_memrealtime stamps of the slice-histogram kernel, per launch slot
#ifdef CVLLM_SEL_TS
__device__ unsigned long long g_sel_rt[8 * 64 * 8];  // [pass slot][workgroup][stamp]
#define SEL_RT(slot, i)                                                                                        \
  do {                                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 64) g_sel_rt[((slot) * 64 + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define SEL_RT(slot, i) \
  do {                  \
  } while (0)
#endif

constexpr int SEL_T = 1024;  // threads per workgroup
constexpr int SEL_W = SEL_T / 64;
constexpr int SEL_MAXH = 64;

__device__ __forceinline__ uint32_t order_key(float x) {
  uint32_t u = __float_as_uint(x);
  if (u == 0x80000000u) u = 0u;             // -0.0 == +0.0 under float comparison (torch.topk / sort semantics)
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;  // NaN ranks above +inf, like torch.sort(descending)
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending uint order == ascending float order
}

// agent-scope load of a word other workgroups updated with atomics in this launch (bypasses the non-coherent caches)
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_agent(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int SEL_E = 4;  // elements per thread per iteration: 4 independent loads in flight (the loops are
                          // L2-latency bound with one), thread t owns the CONSECUTIVE indices base + 4t .. +3

// block-wide exclusive scan of a small per-thread count over NT threads (thread order = index order)
template <int NT>
__device__ __forceinline__ int block_excl_scan_cnt_t(int cnt, int* s_wsum, int& total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const int c = s_wsum[w];
    if (w < wave) woff += c;
    tot += c;
  }
  __syncthreads();
  total = tot;
  return woff + incl - cnt;
}
__device__ __forceinline__ int block_excl_scan_cnt(int cnt, int* s_wsum, int& total) {
  return block_excl_scan_cnt_t<SEL_T>(cnt, s_wsum, total);
}

// Exact selection of the r-th largest (1-based, 1 <= r <= n) ordered key among n strided floats: three radix
// passes over 12 + 12 + 8 key bits with a 4096-bin LDS histogram.  12-bit digits spread z-scored data over
// hundreds of bins, so plain LDS atomics see few same-address conflicts (an 8-bit first digit put a wave's 64
// keys into a handful of bins and serialised ~30-way; leader-aggregation rounds cost more than they saved).
// Returns the key value v and `quota` = how many elements equal to v belong to the top r
// (count(key > v) = r - quota).  All SEL_T threads must call; hist is LDS [4096], s_state LDS [2], s_wsum [SEL_W].
constexpr int SEL_BINS = 4096;
__device__ void radix_select_desc(const float* __restrict__ base, int stride, int n, int r, uint32_t* hist,
                                  uint32_t* s_state, int* s_wsum, uint32_t& v_out, int& quota_out) {
  const int tid = threadIdx.x;
  uint32_t prefix = 0;  // the key bits fixed so far (high bits)
  int remaining = r;
  int fixed_bits = 0;
#pragma unroll 1
  for (int pass = 0; pass < 3; ++pass) {
    const int bits = pass < 2 ? 12 : 8;
    const int shift = 32 - fixed_bits - bits;
    const uint32_t dmask = (1u << bits) - 1u;
    for (int i = tid; i < SEL_BINS; i += SEL_T) hist[i] = 0;
    __syncthreads();
    float nxt[SEL_E];  // software prefetch: the next iteration's loads are in flight while this one is binned
#pragma unroll
    for (int e = 0; e < SEL_E; ++e) {
      const int i = tid * SEL_E + e;
      nxt[e] = i < n ? base[(size_t)i * stride] : 0.f;
    }
    for (int i0 = 0; i0 < n; i0 += SEL_T * SEL_E) {
      uint32_t key[SEL_E];
      bool in[SEL_E];
#pragma unroll
      for (int e = 0; e < SEL_E; ++e) {
        in[e] = i0 + tid * SEL_E + e < n;
        key[e] = order_key(nxt[e]);
        const int i2 = i0 + SEL_T * SEL_E + tid * SEL_E + e;
        nxt[e] = i2 < n ? base[(size_t)i2 * stride] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < SEL_E; ++e) {
        const bool act = in[e] && (fixed_bits == 0 || (key[e] >> (32 - fixed_bits)) == prefix);
        if (act) atomicAdd(&hist[(key[e] >> shift) & dmask], 1u);
      }
    }
    __syncthreads();
    // walk the bins from the top: thread t owns bins top-4t .. top-4t-3 (top = 2^bits - 1); block scan of the sums
    {
      const int nb = 1 << bits;
      const int b0 = nb - 1 - 4 * tid;
      uint32_t c[4] = {0, 0, 0, 0};
      if (b0 >= 3) {
        c[0] = hist[b0]; c[1] = hist[b0 - 1]; c[2] = hist[b0 - 2]; c[3] = hist[b0 - 3];
      }
      const int mine = (int)(c[0] + c[1] + c[2] + c[3]);
      int tot;
      const int excl = block_excl_scan_cnt(mine, s_wsum, tot);  // elements in strictly larger bins
      if (excl < remaining && remaining <= excl + mine) {
        int acc = excl, d = b0;
        if (acc + (int)c[0] >= remaining) { d = b0; }
        else { acc += c[0]; if (acc + (int)c[1] >= remaining) { d = b0 - 1; }
        else { acc += c[1]; if (acc + (int)c[2] >= remaining) { d = b0 - 2; }
        else { acc += c[2]; d = b0 - 3; } } }
        s_state[0] = (uint32_t)d;
        s_state[1] = (uint32_t)(remaining - acc);  // remaining rank inside bin d
      }
    }
    __syncthreads();
    prefix = (prefix << bits) | s_state[0];
    remaining = (int)s_state[1];
    fixed_bits += bits;
    __syncthreads();
  }
  v_out = prefix;
  quota_out = remaining;
}

// kernel 1: one workgroup per sequence -> target[b,h] = t_h (rows of head h to keep), new_lens
__global__ __launch_bounds__(SEL_T) void select_joint_kernel(
    const float* __restrict__ scores, const int* __restrict__ cu, const int* __restrict__ retain,
    const int* __restrict__ bh_lens0, const int* __restrict__ bmap, int* __restrict__ target,
    int* __restrict__ new_lens, int H, int PS, int pad, int reserved) {
  __shared__ uint32_t hist[SEL_BINS];
  __shared__ uint32_t s_state[2];
  __shared__ int s_wsum[SEL_W];
  __shared__ int s_cnt[SEL_MAXH];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int n0 = cu[b];
  const int Lb = cu[b + 1] - n0;
  const bool skip = Lb <= 0 || bmap[b] == reserved;
  if (skip) {
    if (tid < H) {
      target[b * H + tid] = 0;
      new_lens[b * H + tid] = bh_lens0[b * H + tid];
    }
    return;
  }
  const int n = Lb * H;
  int r = retain[b];
  r = r < 0 ? 0 : (r > n ? n : r);
  if (tid < H) s_cnt[tid] = 0;
  __syncthreads();
  const float* base = scores + (size_t)n0 * H;
  if (r > 0) {
    uint32_t v;
    int quota;
    radix_select_desc(base, 1, n, r, hist, s_state, s_wsum, v, quota);
    // count kept pairs per head; ties at v are taken in ascending flat index (running tie counter)
    int ties_before = 0;
    // (SEL_T*SEL_E) % H == 0 (H a power of two): element e of thread t always belongs to head (4t+e) % H, so
    // the per-head counts accumulate in registers and hit LDS once at the end (per-element LDS atomics on H
    // addresses serialised ~32-way and dominated the kernel)
    const bool fixed_head = ((SEL_T * SEL_E) % H) == 0;
    int local[SEL_E];
#pragma unroll
    for (int e = 0; e < SEL_E; ++e) local[e] = 0;
    float nxt[SEL_E];
#pragma unroll
    for (int e = 0; e < SEL_E; ++e) nxt[e] = tid * SEL_E + e < n ? base[tid * SEL_E + e] : 0.f;
    for (int i0 = 0; i0 < n; i0 += SEL_T * SEL_E) {
      uint32_t key[SEL_E];
      bool in[SEL_E];
      int ntie = 0;
#pragma unroll
      for (int e = 0; e < SEL_E; ++e) {
        in[e] = i0 + tid * SEL_E + e < n;
        key[e] = order_key(nxt[e]);
        const int i2 = i0 + SEL_T * SEL_E + tid * SEL_E + e;
        nxt[e] = i2 < n ? base[i2] : 0.f;
        ntie += (in[e] && key[e] == v) ? 1 : 0;
      }
      int tot;
      int tr = ties_before + block_excl_scan_cnt(ntie, s_wsum, tot);
#pragma unroll
      for (int e = 0; e < SEL_E; ++e) {
        const bool tie = in[e] && key[e] == v;
        if (in[e] && (key[e] > v || (tie && tr < quota))) {
          if (fixed_head) ++local[e];
          else atomicAdd(&s_cnt[(i0 + tid * SEL_E + e) % H], 1);
        }
        tr += tie ? 1 : 0;
      }
      ties_before += tot;
    }
    if (fixed_head) {
#pragma unroll
      for (int e = 0; e < SEL_E; ++e)
        if (local[e]) atomicAdd(&s_cnt[(tid * SEL_E + e) % H], local[e]);
    }
    __syncthreads();
  }
  if (tid < H) {
    const int c = s_cnt[tid];
    const int L0 = bh_lens0[b * H + tid];
    const int L = L0 + c;
    int take = c;
    if (pad && (L % PS) != 0) {
      const int need = PS - L % PS;
      int extra = min(need, min(Lb - L, Lb - c));  // store_kv_cache.py:209-220
      take += extra > 0 ? extra : 0;
    }
    target[b * H + tid] = take;
    new_lens[b * H + tid] = L0 + take;
  }
}

// kernel 2: one workgroup per (sequence, head): top-t_h of the head's column -> ordered token list
__global__ __launch_bounds__(SEL_T) void select_head_kernel(const float* __restrict__ scores,
                                                            const int* __restrict__ cu,
                                                            const int* __restrict__ target,
                                                            int* __restrict__ kept_idx, int H, int max_seqlen) {
  __shared__ uint32_t hist[SEL_BINS];
  __shared__ uint32_t s_state[2];
  __shared__ int s_wsum[SEL_W];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int tid = threadIdx.x;
  const int n0 = cu[b];
  const int Lb = cu[b + 1] - n0;
  const int t = target[b * H + h];
  if (t <= 0 || Lb <= 0) return;
  const float* base = scores + (size_t)n0 * H + h;
  int* list = kept_idx + ((size_t)b * H + h) * max_seqlen;
  if (t >= Lb) {
    for (int i = tid; i < Lb; i += SEL_T) list[i] = i;
    return;
  }
  uint32_t v;
  int quota;
  radix_select_desc(base, H, Lb, t, hist, s_state, s_wsum, v, quota);
  int ties_before = 0, kept_before = 0;
  float nxt[SEL_E];
#pragma unroll
  for (int e = 0; e < SEL_E; ++e) nxt[e] = tid * SEL_E + e < Lb ? base[(size_t)(tid * SEL_E + e) * H] : 0.f;
  for (int i0 = 0; i0 < Lb; i0 += SEL_T * SEL_E) {
    uint32_t key[SEL_E];
    bool in[SEL_E];
    int ntie = 0;
#pragma unroll
    for (int e = 0; e < SEL_E; ++e) {
      in[e] = i0 + tid * SEL_E + e < Lb;
      key[e] = order_key(nxt[e]);
      const int i2 = i0 + SEL_T * SEL_E + tid * SEL_E + e;
      nxt[e] = i2 < Lb ? base[(size_t)i2 * H] : 0.f;
      ntie += (in[e] && key[e] == v) ? 1 : 0;
    }
    int tot_t, tot_k;
    int tr = ties_before + block_excl_scan_cnt(ntie, s_wsum, tot_t);
    bool keep[SEL_E];
    int nkeep = 0;
#pragma unroll
    for (int e = 0; e < SEL_E; ++e) {
      const bool tie = in[e] && key[e] == v;
      keep[e] = in[e] && (key[e] > v || (tie && tr < quota));
      tr += tie ? 1 : 0;
      nkeep += keep[e] ? 1 : 0;
    }
    int slot = kept_before + block_excl_scan_cnt(nkeep, s_wsum, tot_k);
#pragma unroll
    for (int e = 0; e < SEL_E; ++e)
      if (keep[e]) list[slot++] = i0 + tid * SEL_E + e;
    ties_before += tot_t;
    kept_before += tot_k;
  }
}

// -----------------------------------------------------------------------------------------------------------------
// Multi-workgroup joint selection for long sequences (L*H >= SJ_MIN keys).  The single-workgroup kernel above walks
// the L*H scores of a sequence three times on ONE CU (288 us at 32 K x 8: instruction-issue bound on that CU); here
// every radix pass is a histogram kernel over SJ_SLICE-key slices of the flat score array (private LDS histogram,
// non-zero bins added to a global 4096-bin histogram) followed by a one-workgroup scan that fixes the next digit.
// Then a slice kernel counts, per head, keys above / equal to the threshold, and a final kernel turns the counts into
// t_h.  Only when the threshold's ties are kept PARTIALLY (quota < #ties) does the final kernel walk the array once
// more to take the first `quota` ties in flat-index order (the canonical tie rule).  All sums are integer: the result
// is deterministic and identical to the single-workgroup path.
constexpr int SJ_T = 512;
constexpr int SJ_E = 4;
constexpr int SJ_SLICE = 8192;   // flat keys per slice workgroup (a multiple of SJ_T * SJ_E)
constexpr int SJ_MIN = 32768;    // below this many keys per sequence the single-workgroup kernel is used
struct SjState {
  uint32_t prefix;
  int remaining;
  int fixed_bits;
  int pad_;
};

template <int NT>
__device__ void radix_scan_step(uint32_t* __restrict__ g, SjState* __restrict__ stp, int r0, int pass,
                                uint32_t* s_state, int* s_wsum);
__device__ __forceinline__ bool last_arriver(int* counter, int participants, int* s_ticket);

__global__ __launch_bounds__(SJ_T) void sj_hist_kernel(const float* __restrict__ scores, const int* __restrict__ cu,
                                                       const int* __restrict__ bmap, SjState* __restrict__ st,
                                                       uint32_t* __restrict__ gh, const int* __restrict__ retain,
                                                       int* __restrict__ tickets, int H, int pass, int NS,
                                                       int reserved) {
  __shared__ uint32_t hist[SEL_BINS];
  __shared__ uint32_t s_state[2];
  __shared__ int s_wsum[SJ_T / 64];
  __shared__ int s_ticket;
  SEL_RT(pass, 0);
  const int b = blockIdx.x / NS, sl = blockIdx.x % NS;
  const int tid = threadIdx.x;
  const int n0 = cu[b];
  const int Lb = cu[b + 1] - n0;
  if (Lb <= 0 || bmap[b] == reserved) return;
  const int n = Lb * H;
  const int beg = sl * SJ_SLICE;
  if (beg >= n) return;
  const int end = min(n, beg + SJ_SLICE);
  const SjState s = st[b];
  SEL_RT(pass, 1);
  const int fixed_bits = pass == 0 ? 0 : s.fixed_bits;
  if (fixed_bits >= 32) return;  // retain == 0: nothing to select
  const int bits = pass < 2 ? 12 : 8;
  const int shift = 32 - fixed_bits - bits;
  const uint32_t dmask = (1u << bits) - 1u;
  const uint32_t prefix = s.prefix;
  for (int i = tid; i < SEL_BINS; i += SJ_T) hist[i] = 0;
  __syncthreads();
  const float* base = scores + (size_t)n0 * H;
  for (int i0 = beg; i0 < end; i0 += SJ_T * SJ_E) {
    float x[SJ_E];
#pragma unroll
    for (int e = 0; e < SJ_E; ++e) {
      const int i = i0 + tid * SJ_E + e;
      x[e] = i < end ? base[i] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < SJ_E; ++e) {
      const int i = i0 + tid * SJ_E + e;
      const uint32_t key = order_key(x[e]);
      if (i < end && (fixed_bits == 0 || (key >> (32 - fixed_bits)) == prefix)) atomicAdd(&hist[(key >> shift) & dmask], 1u);
    }
  }
  __syncthreads();
  SEL_RT(pass, 2);
  uint32_t* g = gh + (size_t)b * SEL_BINS;
  for (int i = tid; i < (1 << bits); i += SJ_T) {
    const uint32_t c = hist[i];
    if (c) atomicAdd(&g[i], c);
  }
  SEL_RT(pass, 3);
  // the radix step on the finished histogram, by the last slice workgroup of the sequence to arrive (this used to be
  // a launch of its own: the selection is launch-latency bound, 18 launches of ~ 4.7 us at 32 K x 8)
  const int participants = (n + SJ_SLICE - 1) / SJ_SLICE;
  const bool last = last_arriver(tickets + b, participants, &s_ticket);
  SEL_RT(pass, 4);
  if (last) {
    int rr = retain[b];
    rr = rr < 0 ? 0 : (rr > n ? n : rr);
    radix_scan_step<SJ_T>(g, st + b, rr, pass, s_state, s_wsum);
    SEL_RT(pass, 5);
  }
}

// one radix step on a finished global histogram g[4096]: fixes the next digit of the r-th largest key, updates the
// state and zeroes the histogram for the next pass.  r0 = the rank to select (used in pass 0), r0 == 0 keeps nothing.
// All SEL_T threads of the workgroup must call.
template <int NT>
__device__ void radix_scan_step(uint32_t* __restrict__ g, SjState* __restrict__ stp, int r0, int pass,
                                uint32_t* s_state, int* s_wsum) {
  constexpr int BP = SEL_BINS / NT;  // bins per thread
  const int tid = threadIdx.x;
  SjState s = *stp;
  int remaining;
  if (pass == 0) {
    s.prefix = 0;
    s.fixed_bits = 0;
    remaining = r0;
    if (r0 == 0) {  // keep nothing: a threshold above every key, quota 0
      if (tid == 0) *stp = SjState{0xffffffffu, 0, 32, 0};
      for (int i = tid; i < SEL_BINS; i += NT) g[i] = 0;
      return;
    }
  } else {
    if (s.fixed_bits >= 32) return;
    remaining = s.remaining;
  }
  const int bits = pass < 2 ? 12 : 8;
  const int nb = 1 << bits;
  const int b0 = nb - 1 - BP * tid;  // thread t owns bins top - BP t .. top - BP t - (BP - 1)
  uint32_t c[BP];
  int mine = 0;
#pragma unroll
  for (int j = 0; j < BP; ++j) {
    c[j] = b0 - j >= 0 ? ld_agent(&g[b0 - j]) : 0u;
    mine += (int)c[j];
  }
  int tot;
  const int excl = block_excl_scan_cnt_t<NT>(mine, s_wsum, tot);
  if (excl < remaining && remaining <= excl + mine) {
    int acc = excl, d = b0;
    bool found = false;
#pragma unroll
    for (int j = 0; j < BP; ++j) {
      if (!found) {
        if (acc + (int)c[j] >= remaining) {
          d = b0 - j;
          found = true;
        } else {
          acc += (int)c[j];
        }
      }
    }
    s_state[0] = (uint32_t)d;
    s_state[1] = (uint32_t)(remaining - acc);
  }
  __syncthreads();
  if (tid == 0) *stp = SjState{(s.prefix << bits) | s_state[0], (int)s_state[1], s.fixed_bits + bits, 0};
  for (int i = tid; i < SEL_BINS; i += NT) g[i] = 0;  // ready for the next pass (everyone has read its bins)
}

// Last-arriver hand-off inside a launch: every participating workgroup calls this after its global atomics; exactly
// one of them - the last to arrive - gets `true`.  Everything handed over was written with agent-scope ATOMICS (they
// are performed at the level all XCDs share) and is read back with agent-scope loads (ld_agent): no cache write-back
// or invalidate is needed - a __threadfence() pair here (L2 write-back + invalidate per workgroup) made the fused
// selection slower than the 18 launches it replaced.  The ticket is taken after the workgroup's atomics have been
// acknowledged.  The counter returns to zero for the next launch.  All threads must call.
__device__ __forceinline__ bool last_arriver(int* counter, int participants, int* s_ticket) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's atomics are performed (CDNA4 vmcnt counts them)
  __syncthreads();
  if (threadIdx.x == 0) *s_ticket = atomicAdd(counter, 1);
  __syncthreads();
  const bool last = *s_ticket == participants - 1;
  if (last && threadIdx.x == 0) atomicExch(counter, 0);
  return last;
}



// per-head counts of keys above / equal to the threshold over one slice
// Forward declaration of the tail that the last slice workgroup runs (defined below)
template <int NT>
__device__ void sj_final_body(int b, const float* __restrict__ scores, const int* __restrict__ cu,
                              const int* __restrict__ bh_lens0, const SjState* __restrict__ st,
                              const int* __restrict__ cnt_gt, const int* __restrict__ cnt_eq, int* __restrict__ target,
                              int* __restrict__ new_lens, int H, int PS, int pad, int* s_wsum, int* s_cnt,
                              int* s_tot_eq_p);

__global__ __launch_bounds__(SJ_T) void sj_count_kernel(const float* __restrict__ scores, const int* __restrict__ cu,
                                                        const int* __restrict__ bmap, const SjState* __restrict__ st,
                                                        int* __restrict__ cnt_gt, int* __restrict__ cnt_eq,
                                                        const int* __restrict__ bh_lens0, int* __restrict__ target,
                                                        int* __restrict__ new_lens, int* __restrict__ tickets, int H,
                                                        int NS, int PS, int pad, int reserved) {
  __shared__ int s_gt[SEL_MAXH], s_eq[SEL_MAXH];
  __shared__ int s_wsum[SJ_T / 64];
  __shared__ int s_tot_eq;
  __shared__ int s_ticket;
  const int b = blockIdx.x / NS, sl = blockIdx.x % NS;
  const int tid = threadIdx.x;
  const int n0 = cu[b];
  const int Lb = cu[b + 1] - n0;
  if (Lb <= 0 || bmap[b] == reserved) {  // nothing to select: the first slice writes the trivial result
    if (sl == 0 && tid < H) {
      target[b * H + tid] = 0;
      new_lens[b * H + tid] = bh_lens0[b * H + tid];
    }
    return;
  }
  const int n = Lb * H;
  const int beg = sl * SJ_SLICE;
  if (beg >= n) return;
  const int end = min(n, beg + SJ_SLICE);
  const uint32_t v = st[b].prefix;
  if (tid < SEL_MAXH) s_gt[tid] = 0, s_eq[tid] = 0;
  __syncthreads();
  const float* base = scores + (size_t)n0 * H;
  // (SJ_T * SJ_E) % H == 0 and SJ_SLICE % H == 0 for H a power of two: element e of thread t is always head
  // (4t + e) % H, so the counts stay in registers; other H go through LDS atomics
  const bool fixed_head = ((SJ_T * SJ_E) % H) == 0 && (SJ_SLICE % H) == 0;
  int lg[SJ_E], le[SJ_E];
#pragma unroll
  for (int e = 0; e < SJ_E; ++e) lg[e] = 0, le[e] = 0;
  for (int i0 = beg; i0 < end; i0 += SJ_T * SJ_E) {
#pragma unroll
    for (int e = 0; e < SJ_E; ++e) {
      const int i = i0 + tid * SJ_E + e;
      if (i < end) {
        const uint32_t key = order_key(base[i]);
        if (fixed_head) {
          lg[e] += key > v ? 1 : 0;
          le[e] += key == v ? 1 : 0;
        } else {
          if (key > v) atomicAdd(&s_gt[i % H], 1);
          if (key == v) atomicAdd(&s_eq[i % H], 1);
        }
      }
    }
  }
  if (fixed_head) {
#pragma unroll
    for (int e = 0; e < SJ_E; ++e) {
      const int hh = (tid * SJ_E + e) % H;
      if (lg[e]) atomicAdd(&s_gt[hh], lg[e]);
      if (le[e]) atomicAdd(&s_eq[hh], le[e]);
    }
  }
  __syncthreads();
  if (tid < H) {
    if (s_gt[tid]) atomicAdd(&cnt_gt[b * SEL_MAXH + tid], s_gt[tid]);
    if (s_eq[tid]) atomicAdd(&cnt_eq[b * SEL_MAXH + tid], s_eq[tid]);
  }
  // counts -> t_h by the last slice workgroup of the sequence to arrive (was a launch of its own)
  if (last_arriver(tickets + b, (n + SJ_SLICE - 1) / SJ_SLICE, &s_ticket))
    sj_final_body<SJ_T>(b, scores, cu, bh_lens0, st, cnt_gt, cnt_eq, target, new_lens, H, PS, pad, s_wsum, s_gt,
                        &s_tot_eq);
}

// per-head counts of sequence b -> t_h (target) and the new lengths; all NT threads of one workgroup call
template <int NT>
__device__ void sj_final_body(int b, const float* __restrict__ scores, const int* __restrict__ cu,
                              const int* __restrict__ bh_lens0, const SjState* __restrict__ st,
                              const int* __restrict__ cnt_gt, const int* __restrict__ cnt_eq, int* __restrict__ target,
                              int* __restrict__ new_lens, int H, int PS, int pad, int* s_wsum, int* s_cnt,
                              int* s_tot_eq_p) {
  const int tid = threadIdx.x;
  const int n0 = cu[b];
  const int Lb = cu[b + 1] - n0;
  int& s_tot_eq = *s_tot_eq_p;
  const int n = Lb * H;
  const uint32_t v = st[b].prefix;
  const int quota = st[b].remaining;
  if (tid == 0) {
    int t = 0;
    for (int hh = 0; hh < H; ++hh) t += ld_agent(&cnt_eq[b * SEL_MAXH + hh]);
    s_tot_eq = t;
  }
  if (tid < H) s_cnt[tid] = ld_agent(&cnt_gt[b * SEL_MAXH + tid]);
  __syncthreads();
  const int tot_eq = s_tot_eq;
  if (quota >= tot_eq) {
    if (tid < H) s_cnt[tid] += ld_agent(&cnt_eq[b * SEL_MAXH + tid]);
  } else if (quota > 0) {
    // partial ties: the first `quota` elements equal to v in ascending flat index are kept
    const float* base = scores + (size_t)n0 * H;
    int ties_before = 0;
    for (int i0 = 0; i0 < n && ties_before < quota; i0 += NT * SEL_E) {
      bool tie[SEL_E];
      int ntie = 0;
#pragma unroll
      for (int e = 0; e < SEL_E; ++e) {
        const int i = i0 + tid * SEL_E + e;
        tie[e] = i < n && order_key(base[i]) == v;
        ntie += tie[e] ? 1 : 0;
      }
      int tot;
      int tr = ties_before + block_excl_scan_cnt_t<NT>(ntie, s_wsum, tot);
#pragma unroll
      for (int e = 0; e < SEL_E; ++e) {
        if (tie[e] && tr < quota) atomicAdd(&s_cnt[(i0 + tid * SEL_E + e) % H], 1);
        tr += tie[e] ? 1 : 0;
      }
      ties_before += tot;
    }
  }
  __syncthreads();
  if (tid < H) {
    const int c = s_cnt[tid];
    const int L0 = bh_lens0[b * H + tid];
    const int L = L0 + c;
    int take = c;
    if (pad && (L % PS) != 0) {
      const int need = PS - L % PS;
      int extra = min(need, min(Lb - L, Lb - c));  // store_kv_cache.py:209-220
      take += extra > 0 ? extra : 0;
    }
    target[b * H + tid] = take;
    new_lens[b * H + tid] = L0 + take;
  }
}

// -----------------------------------------------------------------------------------------------------------------
// Multi-workgroup per-head selection + ordered compaction (max_seqlen >= SH_MIN): the same slice-histogram / scan
// passes per (sequence, head) column, then per-slice kept counts and a write kernel that places every slice's kept
// token indices behind those of the slices before it (token order is preserved: the list stays ascending).
constexpr int SH_SLICE = 4096;  // tokens per slice workgroup
constexpr int SH_MIN = 8192;    // shorter batches keep the one-workgroup-per-head kernel
constexpr int SH_MAXP = 64;     // slices per column (max_seqlen <= 256 K tokens)

__device__ __forceinline__ int sh_rank(const int* __restrict__ target, int bh, int Lb) {
  const int t = target[bh];
  return t < 0 ? 0 : (t > Lb ? Lb : t);
}

__global__ __launch_bounds__(SJ_T) void sh_hist_kernel(const float* __restrict__ scores, const int* __restrict__ cu,
                                                       const int* __restrict__ target, SjState* __restrict__ st,
                                                       uint32_t* __restrict__ gh, int* __restrict__ tickets, int H,
                                                       int pass, int P) {
  __shared__ uint32_t hist[SEL_BINS];
  __shared__ uint32_t s_state[2];
  __shared__ int s_wsum[SJ_T / 64];
  __shared__ int s_ticket;
  const int p = blockIdx.x % P, bh = blockIdx.x / P;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x;
  const int n0 = cu[b];
  const int Lb = cu[b + 1] - n0;
  const int beg = p * SH_SLICE;
  if (Lb <= 0 || beg >= Lb) return;
  const int t = sh_rank(target, bh, Lb);
  if (t <= 0 || t >= Lb) return;  // nothing / everything kept: no threshold needed
  const int end = min(Lb, beg + SH_SLICE);
  const SjState s = st[bh];
  const int fixed_bits = pass == 0 ? 0 : s.fixed_bits;
  const int bits = pass < 2 ? 12 : 8;
  const int shift = 32 - fixed_bits - bits;
  const uint32_t dmask = (1u << bits) - 1u;
  for (int i = tid; i < SEL_BINS; i += SJ_T) hist[i] = 0;
  __syncthreads();
  const float* base = scores + (size_t)n0 * H + h;
  for (int i = beg + tid; i < end; i += SJ_T) {
    const uint32_t key = order_key(base[(size_t)i * H]);
    if (fixed_bits == 0 || (key >> (32 - fixed_bits)) == s.prefix) atomicAdd(&hist[(key >> shift) & dmask], 1u);
  }
  __syncthreads();
  uint32_t* g = gh + (size_t)bh * SEL_BINS;
  for (int i = tid; i < (1 << bits); i += SJ_T) {
    const uint32_t c = hist[i];
    if (c) atomicAdd(&g[i], c);
  }
  // radix step by the last slice workgroup of the column to arrive (see sj_hist_kernel)
  if (last_arriver(tickets + bh, (Lb + SH_SLICE - 1) / SH_SLICE, &s_ticket))
    radix_scan_step<SJ_T>(g, st + bh, t, pass, s_state, s_wsum);
}


// Per-slice kept counts and the ordered write in ONE launch (they were two: sh_count_kernel, sh_write_kernel).  A slice
// workgroup counts its keys above / equal to the column's threshold (keys stay in registers), publishes the pair as one
// flagged 64-bit word, reads the words of the slices BEFORE it in the column (look-back: those workgroups have lower
// block ids, so the in-order dispatcher has started every one of them before this one - the wait cannot deadlock; it is
// bounded all the same) and places its kept token indices behind theirs: the list stays ascending.
constexpr unsigned long long SH_FLAG = 1ull << 63;
// Sticky error word of the selection kernels (per device): bit 0 = a look-back wait timed out.  Read and cleared by
// cvllm_select_status; the product library never traps.
__device__ unsigned g_select_err = 0;
#ifdef CVLLM_SEL_WITHHOLD  // tools/dbg test build only: slice 0 of every column withholds its word, short timeout
constexpr unsigned long long SH_WAIT_TICKS = 200000ull;  // 2 ms at 100 MHz
#else
constexpr unsigned long long SH_WAIT_TICKS = 50000000ull;  // 0.5 s at 100 MHz
#endif
static_assert(SH_SLICE == SJ_T * 8, "eight keys of a slice per thread");
__global__ __launch_bounds__(SJ_T) void sh_write_kernel(const float* __restrict__ scores, const int* __restrict__ cu,
                                                        const int* __restrict__ target, const SjState* __restrict__ st,
                                                        unsigned long long* __restrict__ slice_cnt,
                                                        int* __restrict__ kept_idx, int H, int P, int max_seqlen) {
  __shared__ int s_wsum[SJ_T / 64];
  __shared__ int s_before[2];
  const int p = blockIdx.x % P, bh = blockIdx.x / P;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x;
  const int n0 = cu[b];
  const int Lb = cu[b + 1] - n0;
  const int beg = p * SH_SLICE;
  if (Lb <= 0 || beg >= Lb) return;
  const int t = sh_rank(target, bh, Lb);
  if (t <= 0) return;
  const int end = min(Lb, beg + SH_SLICE);
  int* list = kept_idx + (size_t)bh * max_seqlen;
  if (t >= Lb) {
    for (int i = beg + tid; i < end; i += SJ_T) list[i] = i;
    return;
  }
  const uint32_t v = st[bh].prefix;
  const int quota = st[bh].remaining;
  // thread t owns the CONSECUTIVE tokens beg + 8 t .. + 7 (thread order = token order for the block scans)
  constexpr int E = SH_SLICE / SJ_T;
  const float* base = scores + (size_t)n0 * H + h;
  uint32_t key[E];
  int gt = 0, eq = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = beg + tid * E + e;
    key[e] = i < end ? order_key(base[(size_t)i * H]) : 0u;
    gt += (i < end && key[e] > v) ? 1 : 0;
    eq += (i < end && key[e] == v) ? 1 : 0;
  }
  int tg, te;
  const int gt_before_me = block_excl_scan_cnt_t<SJ_T>(gt, s_wsum, tg);
  const int eq_before_me = block_excl_scan_cnt_t<SJ_T>(eq, s_wsum, te);
  unsigned long long* col = slice_cnt + (size_t)bh * P;
#ifdef CVLLM_SEL_WITHHOLD
  if (tid == 0 && p != 0)
#else
  if (tid == 0)
#endif
    __hip_atomic_store(&col[p], SH_FLAG | ((unsigned long long)(uint32_t)tg << 31) | (unsigned long long)(uint32_t)te,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // look-back over the slices before this one (P <= 64: one thread each), then a serial fold in slice order by thread 0
  __shared__ unsigned long long s_prev[SH_MAXP];
  if (tid < p) {
    unsigned long long wv = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    do {
      wv = __hip_atomic_load(&col[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } while (!(wv & SH_FLAG) && __builtin_amdgcn_s_memrealtime() - t0 < SH_WAIT_TICKS);
    if (!(wv & SH_FLAG)) {
      // A slice before this one never published (dispatch order is not a contract; a preempted or masked-off workgroup
      // could do it).  Never trap and never spin on: raise the sticky error word (cvllm_select_status) and carry on with
      // that slice counted as empty - every index written below is then a valid token and lies inside the list (the
      // offsets can only be too SMALL), entries left unwritten are clamped by the consumers (store_kv.hip).
      __hip_atomic_fetch_or(&g_select_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wv = SH_FLAG;
    }
    s_prev[tid] = wv;
  }
  __syncthreads();
  if (tid == 0) {
    int ties_before = 0, kept_before = 0;
    for (int q = 0; q < p; ++q) {
      const unsigned long long wv = s_prev[q];
      const int g = (int)((wv >> 31) & 0x7fffffffu), e = (int)(wv & 0x7fffffffu);
      const int room = quota - ties_before;
      kept_before += g + (room > 0 ? min(e, room) : 0);
      ties_before += e;
    }
    s_before[0] = ties_before;
    s_before[1] = kept_before;
  }
  __syncthreads();
  int tr = s_before[0] + eq_before_me;  // ties before this thread's first token, in the whole column
  bool keep[E];
  int nkeep = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = beg + tid * E + e;
    const bool in = i < end;
    const bool tie = in && key[e] == v;
    keep[e] = in && (key[e] > v || (tie && tr < quota));
    tr += tie ? 1 : 0;
    nkeep += keep[e] ? 1 : 0;
  }
  (void)gt_before_me;
  int tot_k;
  int slot = s_before[1] + block_excl_scan_cnt_t<SJ_T>(nkeep, s_wsum, tot_k);
#pragma unroll
  for (int e = 0; e < E; ++e)
    if (keep[e]) list[slot++] = beg + tid * E + e;
}

}  // namespace cvllm

using namespace cvllm;

// workspace: target[B, H] | joint path: global histograms, states, per-head counts, tickets | per-head path: per-column
// histograms, states, tickets, per-slice count words.  All of it is zeroed by ONE memset.
static size_t sel_target_bytes(int B, int H) { return ((size_t)B * H * sizeof(int32_t) + 15) / 16 * 16; }
static size_t sel_joint_bytes(int B) {
  return (size_t)B * (SEL_BINS * sizeof(uint32_t) + sizeof(SjState) + 2 * SEL_MAXH * sizeof(int32_t) + 4 * sizeof(int32_t));
}
static size_t sel_head_zero_bytes(int B, int H) {
  return (size_t)B * H * (SEL_BINS * sizeof(uint32_t) + sizeof(SjState) + 4 * sizeof(int32_t));
}
static size_t sel_slice_bytes(int B, int H, size_t P) { return (size_t)B * H * P * sizeof(unsigned long long); }

extern "C" size_t cvllm_select_workspace_bytes(int B, int H, int max_seqlen) {
  if (B <= 0 || H <= 0) return 0;
  size_t bytes = sel_target_bytes(B, H);
  if ((long)max_seqlen * H >= SJ_MIN) bytes += sel_joint_bytes(B);
  if (max_seqlen >= SH_MIN) {
    const size_t P = ((size_t)max_seqlen + SH_SLICE - 1) / SH_SLICE;
    bytes += sel_head_zero_bytes(B, H) + sel_slice_bytes(B, H, P);
  }
  return bytes;
}

extern "C" int cvllm_select_topk(const float* scores, const int32_t* cu_seqlens_k, const int32_t* retain,
                                 const int32_t* bh_lens0, const int32_t* batch_mapping, int32_t* kept_idx,
                                 int32_t* new_lens, int B, int H, int max_seqlen, int page_size, int pad_to_page,
                                 int reserved_batch, void* workspace, size_t workspace_bytes,
                                 cvllm_stream_t stream) {
  if (!scores || !cu_seqlens_k || !retain || !bh_lens0 || !batch_mapping || !kept_idx || !new_lens)
    return CVLLM_ERR_ARG;
  if (B <= 0 || H <= 0 || max_seqlen <= 0 || page_size <= 0) return CVLLM_ERR_ARG;
  if (H > SEL_MAXH || (SEL_T % H) != 0 && H > SEL_T) return CVLLM_ERR_SHAPE;
  if (!workspace || workspace_bytes < cvllm_select_workspace_bytes(B, H, max_seqlen)) return CVLLM_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int* target = (int*)workspace;
  const bool joint_multi = (long)max_seqlen * H >= SJ_MIN;
  const int P = (max_seqlen + SH_SLICE - 1) / SH_SLICE;
  const bool head_multi = max_seqlen >= SH_MIN && P <= SH_MAXP;
  char* pj = (char*)workspace + sel_target_bytes(B, H);
  char* ph = pj + (joint_multi ? sel_joint_bytes(B) : 0);
  {
    // one memset for both multi-workgroup paths (their regions are adjacent)
    const size_t zero = (joint_multi ? sel_joint_bytes(B) : 0) +
                        (head_multi ? sel_head_zero_bytes(B, H) + sel_slice_bytes(B, H, (size_t)P) : 0);
    char* z0 = joint_multi ? pj : ph;
    if (zero && hipMemsetAsync(z0, 0, zero, st) != hipSuccess) return CVLLM_ERR_LAUNCH;
  }
  if (joint_multi) {
    uint32_t* gh = (uint32_t*)pj;
    SjState* sst = (SjState*)(gh + (size_t)B * SEL_BINS);
    int* cnt_gt = (int*)(sst + B);
    int* cnt_eq = cnt_gt + (size_t)B * SEL_MAXH;
    int* tickets = cnt_eq + (size_t)B * SEL_MAXH;
    const int NS = (int)(((long)max_seqlen * H + SJ_SLICE - 1) / SJ_SLICE);
    // 3 radix passes (histogram + the radix step by the last workgroup to arrive), then counts + t_h likewise
    for (int pass = 0; pass < 3; ++pass)
      hipLaunchKernelGGL(sj_hist_kernel, dim3(B * NS), dim3(SJ_T), 0, st, scores, cu_seqlens_k, batch_mapping, sst, gh,
                         retain, tickets, H, pass, NS, reserved_batch);
    hipLaunchKernelGGL(sj_count_kernel, dim3(B * NS), dim3(SJ_T), 0, st, scores, cu_seqlens_k, batch_mapping, sst,
                       cnt_gt, cnt_eq, bh_lens0, target, new_lens, tickets, H, NS, page_size, pad_to_page,
                       reserved_batch);
  } else {
    hipLaunchKernelGGL(select_joint_kernel, dim3(B), dim3(SEL_T), 0, st, scores, cu_seqlens_k, retain, bh_lens0,
                       batch_mapping, target, new_lens, H, page_size, pad_to_page, reserved_batch);
  }
  if (head_multi) {
    uint32_t* gh2 = (uint32_t*)ph;
    SjState* st2 = (SjState*)(gh2 + (size_t)B * H * SEL_BINS);
    int* tickets2 = (int*)(st2 + (size_t)B * H);
    unsigned long long* slice_cnt = (unsigned long long*)(ph + sel_head_zero_bytes(B, H));
    for (int pass = 0; pass < 3; ++pass)
      hipLaunchKernelGGL(sh_hist_kernel, dim3(B * H * P), dim3(SJ_T), 0, st, scores, cu_seqlens_k, target, st2, gh2,
                         tickets2, H, pass, P);
    hipLaunchKernelGGL(sh_write_kernel, dim3(B * H * P), dim3(SJ_T), 0, st, scores, cu_seqlens_k, target, st2, slice_cnt,
                       kept_idx, H, P, max_seqlen);
  } else {
    hipLaunchKernelGGL(select_head_kernel, dim3(B * H), dim3(SEL_T), 0, st, scores, cu_seqlens_k, target, kept_idx, H,
                       max_seqlen);
  }
  return check_launch();
}

// Health check of the selection kernels on the current device: 0 = fine, 1 = a look-back wait of the per-head ordered
// write timed out since the last check (the kept lists of that call are incomplete; the word is cleared here).
// Synchronises the stream - for tests and for the engine's one check per prefill, not part of the data path.
extern "C" int cvllm_select_status(cvllm_stream_t stream) {
  unsigned w = 0;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemcpyFromSymbolAsync(&w, HIP_SYMBOL(cvllm::g_select_err), sizeof(w), 0, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return CVLLM_ERR_LAUNCH;
  if (w) {
    const unsigned z = 0;
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(cvllm::g_select_err), &z, sizeof(z), 0, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return CVLLM_ERR_LAUNCH;
  }
  return (int)(w & 1u);
}

#ifdef CVLLM_SEL_TS
extern "C" void cvllm_debug_select_stamps(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(cvllm::g_sel_rt), sizeof(unsigned long long) * 8 * 64 * 8);
}
#endif
