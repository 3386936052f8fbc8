// a1 — causal varlen prefill attention over [paged per-head prefix || appended block], GQA, gfx950 MFMA.
//
// Replaces cv/attention/sparse_varlen_kernel.py:277-519 (_causal_head_sparse_varlen_with_cache).
//
// Tiling (wave64, 8 waves = 512 threads per workgroup, 2 waves per SIMD):
//   * a workgroup owns one (sequence b, kv-head g, query tile): 256 rows = (256/G tokens) x G query heads, so
//     every K/V tile staged in LDS is shared by all G heads of the group (the reference's M = BLOCK_M*G packing);
//     a wave owns 32 rows and keeps their Q fragments in registers (B operand of the swapped product).
//   * K/V are consumed in 64-key tiles staged through registers into LDS: K double-, V triple-buffered (96 KB), so
//     that tile t+2 (global loads issued at the top of iteration t) is written at the end of iteration t behind ONE
//     barrier per tile.
//   * S^T = K Q^T with v_mfma_f32_32x32x16 (keys on the accumulator ROW, the query on the lane): the row softmax
//     of a query is in-lane (+ one exchange with lane^32), and the fp32 accumulator registers 8s..8s+7 converted
//     to 16-bit ARE the B operand of the next product O^T += V^T P^T (no LDS round trip for P).
//   * In-wave software pipeline: iteration t issues the MFMAs of S(t+1) next to the exp / pack work of tile t, then
//     the MFMAs of O += P(t) V(t) next to the row max of tile t+1 (see the main loop).
//   * V^T fragments come from the row-major V tile with ds_read_b64_tr_b16 (hardware transpose); the K tile is
//     read row-wise with ds_read_b128.  Both images are PADDED (K rows 272 B, V rows 320 B): conflict free for
//     their read instruction and every fragment address = one lane-constant VGPR + an immediate.
//   * online softmax in the exp2 domain (scale*log2e folded into one FMA per logit), masked logits = -inf,
//     P rounded to the model dtype before PV like the reference (:398), fp32 accumulate, one divide at the end.
//   * O^T goes through LDS so that the global store is whole 256-byte rows.
// Grid order: kv-head fastest (blockIdx % 8 = XCD label -> all query tiles of one kv-head share an L2),
// heaviest (last) query tiles first for causal load balance.
#include "common.h"

#include <type_traits>

namespace cvllm {

typedef __attribute__((ext_vector_type(4))) short s16x4;

template <typename T>
__device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c);
template <>
__device__ __forceinline__ f32x16 mfma32<F16>(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mfma32<BF16>(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}

// The 4-wave kernel's output accumulators O^T (2 query blocks x 4 head-dim blocks x 16 registers) live in the ACC
// registers a[128:255], owned by inline asm (the structure the CDNA guide names for this kernel shape).  Why: the kernel
// is built with -amdgpu-mfma-vgpr-form so that hipcc's own MFMAs - the QK^T chains, whose results the VALU reads - take
// and deliver arch VGPRs; an MFMA's C and D share one register-file bit, so hipcc would keep the 128 output accumulators
// in arch VGPRs as well: half of the file.  Here the PV MFMAs (pv_mfma_w) name their accumulator registers literally; the
// compiler allocates ITS AGPRs from a0 upwards and must stay below a128 (build.py audits the code object: no
// compiler-emitted instruction may touch a128+).  hipcc pads no hazards around an asm statement: the two places that access these registers with
// v_accvgpr_* (rare rescale, epilogue) wait out the MFMA pipeline themselves (acc_settle) and pad their writes.
constexpr int P4_ACC0 = 128;
template <int R>
__device__ __forceinline__ float acc_read() {
  float t;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(t) : "i"(R));
  return t;
}
template <int R>
__device__ __forceinline__ void acc_write(float t) {
  asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(t), "i"(R));
}
// a[128:255] = 0; the clobber list is what makes the kernel descriptor allocate the registers
__device__ __forceinline__ void acc_zero_all() {
  asm volatile(
      "v_accvgpr_write_b32 a128, 0\n\t"
      "v_accvgpr_write_b32 a129, 0\n\t"
      "v_accvgpr_write_b32 a130, 0\n\t"
      "v_accvgpr_write_b32 a131, 0\n\t"
      "v_accvgpr_write_b32 a132, 0\n\t"
      "v_accvgpr_write_b32 a133, 0\n\t"
      "v_accvgpr_write_b32 a134, 0\n\t"
      "v_accvgpr_write_b32 a135, 0\n\t"
      "v_accvgpr_write_b32 a136, 0\n\t"
      "v_accvgpr_write_b32 a137, 0\n\t"
      "v_accvgpr_write_b32 a138, 0\n\t"
      "v_accvgpr_write_b32 a139, 0\n\t"
      "v_accvgpr_write_b32 a140, 0\n\t"
      "v_accvgpr_write_b32 a141, 0\n\t"
      "v_accvgpr_write_b32 a142, 0\n\t"
      "v_accvgpr_write_b32 a143, 0\n\t"
      "v_accvgpr_write_b32 a144, 0\n\t"
      "v_accvgpr_write_b32 a145, 0\n\t"
      "v_accvgpr_write_b32 a146, 0\n\t"
      "v_accvgpr_write_b32 a147, 0\n\t"
      "v_accvgpr_write_b32 a148, 0\n\t"
      "v_accvgpr_write_b32 a149, 0\n\t"
      "v_accvgpr_write_b32 a150, 0\n\t"
      "v_accvgpr_write_b32 a151, 0\n\t"
      "v_accvgpr_write_b32 a152, 0\n\t"
      "v_accvgpr_write_b32 a153, 0\n\t"
      "v_accvgpr_write_b32 a154, 0\n\t"
      "v_accvgpr_write_b32 a155, 0\n\t"
      "v_accvgpr_write_b32 a156, 0\n\t"
      "v_accvgpr_write_b32 a157, 0\n\t"
      "v_accvgpr_write_b32 a158, 0\n\t"
      "v_accvgpr_write_b32 a159, 0\n\t"
      "v_accvgpr_write_b32 a160, 0\n\t"
      "v_accvgpr_write_b32 a161, 0\n\t"
      "v_accvgpr_write_b32 a162, 0\n\t"
      "v_accvgpr_write_b32 a163, 0\n\t"
      "v_accvgpr_write_b32 a164, 0\n\t"
      "v_accvgpr_write_b32 a165, 0\n\t"
      "v_accvgpr_write_b32 a166, 0\n\t"
      "v_accvgpr_write_b32 a167, 0\n\t"
      "v_accvgpr_write_b32 a168, 0\n\t"
      "v_accvgpr_write_b32 a169, 0\n\t"
      "v_accvgpr_write_b32 a170, 0\n\t"
      "v_accvgpr_write_b32 a171, 0\n\t"
      "v_accvgpr_write_b32 a172, 0\n\t"
      "v_accvgpr_write_b32 a173, 0\n\t"
      "v_accvgpr_write_b32 a174, 0\n\t"
      "v_accvgpr_write_b32 a175, 0\n\t"
      "v_accvgpr_write_b32 a176, 0\n\t"
      "v_accvgpr_write_b32 a177, 0\n\t"
      "v_accvgpr_write_b32 a178, 0\n\t"
      "v_accvgpr_write_b32 a179, 0\n\t"
      "v_accvgpr_write_b32 a180, 0\n\t"
      "v_accvgpr_write_b32 a181, 0\n\t"
      "v_accvgpr_write_b32 a182, 0\n\t"
      "v_accvgpr_write_b32 a183, 0\n\t"
      "v_accvgpr_write_b32 a184, 0\n\t"
      "v_accvgpr_write_b32 a185, 0\n\t"
      "v_accvgpr_write_b32 a186, 0\n\t"
      "v_accvgpr_write_b32 a187, 0\n\t"
      "v_accvgpr_write_b32 a188, 0\n\t"
      "v_accvgpr_write_b32 a189, 0\n\t"
      "v_accvgpr_write_b32 a190, 0\n\t"
      "v_accvgpr_write_b32 a191, 0\n\t"
      "v_accvgpr_write_b32 a192, 0\n\t"
      "v_accvgpr_write_b32 a193, 0\n\t"
      "v_accvgpr_write_b32 a194, 0\n\t"
      "v_accvgpr_write_b32 a195, 0\n\t"
      "v_accvgpr_write_b32 a196, 0\n\t"
      "v_accvgpr_write_b32 a197, 0\n\t"
      "v_accvgpr_write_b32 a198, 0\n\t"
      "v_accvgpr_write_b32 a199, 0\n\t"
      "v_accvgpr_write_b32 a200, 0\n\t"
      "v_accvgpr_write_b32 a201, 0\n\t"
      "v_accvgpr_write_b32 a202, 0\n\t"
      "v_accvgpr_write_b32 a203, 0\n\t"
      "v_accvgpr_write_b32 a204, 0\n\t"
      "v_accvgpr_write_b32 a205, 0\n\t"
      "v_accvgpr_write_b32 a206, 0\n\t"
      "v_accvgpr_write_b32 a207, 0\n\t"
      "v_accvgpr_write_b32 a208, 0\n\t"
      "v_accvgpr_write_b32 a209, 0\n\t"
      "v_accvgpr_write_b32 a210, 0\n\t"
      "v_accvgpr_write_b32 a211, 0\n\t"
      "v_accvgpr_write_b32 a212, 0\n\t"
      "v_accvgpr_write_b32 a213, 0\n\t"
      "v_accvgpr_write_b32 a214, 0\n\t"
      "v_accvgpr_write_b32 a215, 0\n\t"
      "v_accvgpr_write_b32 a216, 0\n\t"
      "v_accvgpr_write_b32 a217, 0\n\t"
      "v_accvgpr_write_b32 a218, 0\n\t"
      "v_accvgpr_write_b32 a219, 0\n\t"
      "v_accvgpr_write_b32 a220, 0\n\t"
      "v_accvgpr_write_b32 a221, 0\n\t"
      "v_accvgpr_write_b32 a222, 0\n\t"
      "v_accvgpr_write_b32 a223, 0\n\t"
      "v_accvgpr_write_b32 a224, 0\n\t"
      "v_accvgpr_write_b32 a225, 0\n\t"
      "v_accvgpr_write_b32 a226, 0\n\t"
      "v_accvgpr_write_b32 a227, 0\n\t"
      "v_accvgpr_write_b32 a228, 0\n\t"
      "v_accvgpr_write_b32 a229, 0\n\t"
      "v_accvgpr_write_b32 a230, 0\n\t"
      "v_accvgpr_write_b32 a231, 0\n\t"
      "v_accvgpr_write_b32 a232, 0\n\t"
      "v_accvgpr_write_b32 a233, 0\n\t"
      "v_accvgpr_write_b32 a234, 0\n\t"
      "v_accvgpr_write_b32 a235, 0\n\t"
      "v_accvgpr_write_b32 a236, 0\n\t"
      "v_accvgpr_write_b32 a237, 0\n\t"
      "v_accvgpr_write_b32 a238, 0\n\t"
      "v_accvgpr_write_b32 a239, 0\n\t"
      "v_accvgpr_write_b32 a240, 0\n\t"
      "v_accvgpr_write_b32 a241, 0\n\t"
      "v_accvgpr_write_b32 a242, 0\n\t"
      "v_accvgpr_write_b32 a243, 0\n\t"
      "v_accvgpr_write_b32 a244, 0\n\t"
      "v_accvgpr_write_b32 a245, 0\n\t"
      "v_accvgpr_write_b32 a246, 0\n\t"
      "v_accvgpr_write_b32 a247, 0\n\t"
      "v_accvgpr_write_b32 a248, 0\n\t"
      "v_accvgpr_write_b32 a249, 0\n\t"
      "v_accvgpr_write_b32 a250, 0\n\t"
      "v_accvgpr_write_b32 a251, 0\n\t"
      "v_accvgpr_write_b32 a252, 0\n\t"
      "v_accvgpr_write_b32 a253, 0\n\t"
      "v_accvgpr_write_b32 a254, 0\n\t"
      "v_accvgpr_write_b32 a255, 0\n\t"
      ""
      :
      :
      : "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
}
// every asm MFMA issued so far has written its accumulator (8-pass XDL write -> v_accvgpr_read: 18 wait states cover it)
__device__ __forceinline__ void acc_settle() { asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); }

// The 4-wave kernel's softmax state that its RARE path rewrites - reference points, row sums, packed P words - lives in
// arch VGPRs v230..v255 owned by inline asm in the same way (the kernel is compiled with amdgpu_num_vgpr(230): hipcc
// allocates v0..v229).  Why: a value the compiler can see being rewritten inside a conditional block gets a second home
// register there, and the copies that reconcile the two homes land on the COMMON path (18-26 v_mov per check, measured:
// as many instructions as the check scheme saves).  Registers nothing visible ever writes need no reconciling.
//   v222+qb  minus the reference point of query block qb      v224+qb / v226+qb  running sums l (even / odd logits)
//   v228+qb / v230+qb  the sums of the unit in flight
//   v232 + 8 par + 4 qb + word  packed P words of keys 0-15 (two sets, by the parity of the unit: the next unit's stream
//                               writes its set while the PV MFMAs still read this unit's)
//   v248 + 4 qb + word          packed P words of keys 16-31
// Hazards are the placement's business, as with the accumulators: a word is packed at least one MFMA shadow before the
// MFMA that reads it, a v_exp result is read by an asm add two instructions later, the logits an asm fma reads were
// written by MFMAs at least two shadows earlier.
constexpr int P4_VNM = 222, P4_VL = 224, P4_VL2 = 226, P4_VLU = 228, P4_VLU2 = 230, P4_VPW0 = 232, P4_VPW1 = 248;
constexpr int P4_NUM_VGPR = 222;
// register of the packed word of item j (= 8 s2 + 4 qb + jp) of a unit of parity par
constexpr int p4_word_reg(int par, int j) {
  return (j >> 3) ? P4_VPW1 + 4 * ((j >> 2) & 1) + (j & 3) : P4_VPW0 + 8 * par + 4 * ((j >> 2) & 1) + (j & 3);
}
template <typename T, int X, int W>
__device__ __forceinline__ void pv_mfma_w(s16x8 a) {
  static_assert(X >= P4_ACC0 && X + 15 < 256 && (X % 16) == 0 && W >= P4_VPW0 && W + 3 < 256 && (W % 4) == 0, "registers");
  if constexpr (std::is_same<T, BF16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, v[%c3:%c4], a[%c1:%c2]" ::"v"(a), "i"(X), "i"(X + 15), "i"(W),
                 "i"(W + 3));
  else
    asm volatile("v_mfma_f32_32x32x16_f16 a[%c1:%c2], %0, v[%c3:%c4], a[%c1:%c2]" ::"v"(a), "i"(X), "i"(X + 15), "i"(W),
                 "i"(W + 3));
}
// v222..v255 = 0 except v222 / v223 = +1e30 (reference point -1e30: the first unit's row_fix always takes); the clobber
// list makes the kernel descriptor count the registers
__device__ __forceinline__ void p4_state_init() {
  asm volatile(
      "v_mov_b32 v222, 0x7149f2ca\n\tv_mov_b32 v223, 0x7149f2ca\n\t"
      "v_mov_b32 v224, 0\n\tv_mov_b32 v225, 0\n\tv_mov_b32 v226, 0\n\tv_mov_b32 v227, 0\n\t"
      "v_mov_b32 v228, 0\n\tv_mov_b32 v229, 0\n\tv_mov_b32 v230, 0\n\tv_mov_b32 v231, 0\n\t"
      "v_mov_b32 v232, 0\n\tv_mov_b32 v233, 0\n\tv_mov_b32 v234, 0\n\tv_mov_b32 v235, 0\n\t"
      "v_mov_b32 v236, 0\n\tv_mov_b32 v237, 0\n\tv_mov_b32 v238, 0\n\tv_mov_b32 v239, 0\n\t"
      "v_mov_b32 v240, 0\n\tv_mov_b32 v241, 0\n\tv_mov_b32 v242, 0\n\tv_mov_b32 v243, 0\n\t"
      "v_mov_b32 v244, 0\n\tv_mov_b32 v245, 0\n\tv_mov_b32 v246, 0\n\tv_mov_b32 v247, 0\n\t"
      "v_mov_b32 v248, 0\n\tv_mov_b32 v249, 0\n\tv_mov_b32 v250, 0\n\tv_mov_b32 v251, 0\n\t"
      "v_mov_b32 v252, 0\n\tv_mov_b32 v253, 0\n\tv_mov_b32 v254, 0\n\tv_mov_b32 v255, 0"
      :
      :
      : "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235",
        "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249",
        "v250", "v251", "v252", "v253", "v254", "v255");
}

// rescale threshold of both kernels, exp2 domain (probabilities reach at most 2^PF_THR; 0 = the textbook rule)
#ifndef PF_THR
#define PF_THR 8
#endif
constexpr int PF_ROWS = 256;   // query rows per workgroup
constexpr int PF_KT = 64;      // keys per tile
constexpr int PF_THREADS = 512;


constexpr int PF_KSTR = 272;                      // K image row stride: 256 + 16 B pad -> conflict-free ds_read_b128
constexpr int PF_VSTR = 320;                      // V image row stride: 256 + 64 B pad -> conflict-free ds_read_b64_tr_b16
constexpr int PF_KTILE = PF_KT * PF_KSTR;         // 17,408 B
constexpr int PF_VTILE = PF_KT * PF_VSTR;         // 20,480 B
constexpr int PF_OSTRIDE = 272;                   // padded row stride of the O staging image
constexpr int PF_SMEM = 2 * PF_KTILE + 3 * PF_VTILE;  // 2 K tiles + 3 V tiles = 96,256 B
static_assert(PF_SMEM >= 8 * 32 * PF_OSTRIDE, "O staging image must fit");

// Padded (not XOR-swizzled) images: every fragment address of the main loop is ONE lane-constant VGPR plus an
// immediate offset, so the loop spends no VALU on LDS addressing (VALU issue, shared by the two waves of a SIMD,
// is what bounds this kernel: MI355X guide, 'vector-instruction ISSUE cost').
//   K: lanes r = 0..15 of a ds_read_b128 group read rows r at the same column -> bank (68 r) % 64 = 4 r: distinct.
//   V: a 32-lane half of ds_read_b64_tr_b16 reads 4 rows x 64 contiguous bytes -> row shift 80 % 64 = 16 banks.

// max(a, b, c).  This file is built with -fno-honor-nans (build.py): without it hipcc puts a v_max x,x
// canonicalisation in front of every fmaxf of an MFMA result (3 VALU per pair instead of 1 v_max3 per two values).
// NOT inline asm: hipcc's hazard recogniser does not see an asm statement as a VALU read of an MFMA result and
// leaves out the wait states the read needs - stale maxima, run-to-run different roundings.
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
// row max of one query over the wave's 64 keys: 32 in-lane values, then the partner lane (l ^ 32)
__device__ __forceinline__ float tile_max(const f32x16& a, const f32x16& b) {
  float m0 = max3(a[0], a[1], a[2]), m1 = max3(b[0], b[1], b[2]);
#pragma unroll
  for (int i = 3; i < 15; i += 2) m0 = max3(m0, a[i], a[i + 1]), m1 = max3(m1, b[i], b[i + 1]);
  const float mx = max3(m0, m1, fmaxf(a[15], b[15]));
  // partner lane l ^ 32 by v_permlane32_swap (VALU; no LDS round trip as with ds_bpermute)
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
  return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}

// LDS-visibility barrier without hipcc's vmcnt(0) drain: only LDS traffic has to be complete at the rendezvous.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T, int D, int G>
__global__ __launch_bounds__(PF_THREADS) void prefill_attn_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v, int64_t sq_n,
    int64_t sk_n, int64_t sk_h, int64_t sv_n, int64_t sv_h, const uint16_t* __restrict__ kc,
    const uint16_t* __restrict__ vc, uint16_t* __restrict__ out, const int* __restrict__ seq_lens,
    const int* __restrict__ page_table, const int* __restrict__ bmap, const int* __restrict__ cu, int B, int HKV,
    int PS, int NLP, float scale_log2e) {
  constexpr int BM = PF_ROWS / G;  // tokens per query tile
  constexpr int KS = D / 16;       // k-steps of the QK^T product
  constexpr int DB = D / 32;       // 32-wide blocks of the head dim
  constexpr int CH = D / 8;        // 16-byte chunks per row

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int bid = blockIdx.x;
  const int g = bid % HKV;
  const int b = (bid / HKV) % B;
  const int tile_rev = bid / (HKV * B);
  const int s0 = cu[b];
  const int La = cu[b + 1] - s0;
  const int nqt = (La + BM - 1) / BM;
  if (tile_rev >= nqt) return;
  const int qt = nqt - 1 - tile_rev;  // heaviest tiles first
  const int m0 = qt * BM;
  const int HQ = HKV * G;

  const int Lc = seq_lens[b * HKV + g];
  const int* pt = page_table + ((size_t)bmap[b] * HKV + g) * NLP;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int row = wave * 32 + r;            // query row inside the tile
  const int head_local = row / BM;
  const int tok = m0 + row % BM;            // local token index of this lane's query
  const bool valid_q = tok < La;
  const int hq = g * G + head_local;

  // ---- Q fragments (B operand: lane holds Q[q = r][dims 16s + 8h .. +8]) ---------------------------------
  s16x8 qf[KS];
  {
    const uint16_t* qp = q + (size_t)(s0 + (valid_q ? tok : 0)) * sq_n + (size_t)hq * D + 8 * h;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      uint4 t = valid_q ? *reinterpret_cast<const uint4*>(qp + 16 * s) : make_uint4(0, 0, 0, 0);
      qf[s] = __builtin_bit_cast(s16x8, t);
    }
  }

  const int ntc = (Lc + PF_KT - 1) / PF_KT;                       // cached-prefix tiles
  const int la_vis = min(m0 + BM, La);                            // appended keys visible to this query tile
  const int nta = (la_vis + PF_KT - 1) / PF_KT;                   // appended tiles (last ones are on the diagonal)
  const int ntiles = ntc + nta;

  // ---- staging: thread -> (rows srow, srow+32; chunk sch) of a K or V tile ------------------------------------
  const int sch = tid & 15;
  const int srow = tid >> 4;  // 0..31
  const bool stage_active = sch < CH;
  uint4 kr0, kr1, vr0, vr1;  // staged tile chunks (named scalars: arrays captured by the lambdas went to scratch)

  // Source row offsets (in elements) of this thread's two rows of tile t (cached prefix first, then the appended
  // block).  Rows past the end of the tile are CLAMPED to the last valid row instead of being skipped: their
  // logits are masked to -inf (P = 0) and the clamped row is finite data, so 0 * V stays 0 - and the loads stay
  // unconditional (a per-row "load or zero" select makes hipcc branch around every load and serialise them).
  // Pages of PS % 64 == 0 keep a whole 64-key tile inside one page: the page id is a wave-uniform scalar load issued
  // one tile ahead and the page walk is incremental (no per-row division, no dependent vector load of the page table
  // at the top of every iteration).  Other page sizes take the per-row path.
  const bool ps64 = (PS % PF_KT) == 0;
  int ld_pi = 0, ld_po = 0;                 // page index / offset in page of the next cached tile to load
  int ld_pg = (ntc > 0) ? pt[0] : 0;        // its physical page
  const uint32_t skn = (uint32_t)sk_n, svn = (uint32_t)sv_n;  // row strides < 2^31 elements (checked on the host)
  auto gload = [&](int t) {
    if (t < ntc) {
      size_t o0, o1;
      if (ps64) {
        const int last = Lc - 1 - t * PF_KT;  // >= 0
        const size_t rowbase = (size_t)ld_pg * PS + ld_po;
        o0 = (rowbase + min(srow, last)) * D + sch * 8;  // int64 row offset (reference :371)
        o1 = (rowbase + min(srow + 32, last)) * D + sch * 8;
        ld_po += PF_KT;
        if (ld_po >= PS) {
          ld_po = 0;
          ld_pi += 1;
          ld_pg = pt[min(ld_pi, NLP - 1)];
        }
      } else {
        const int p0 = min(t * PF_KT + srow, Lc - 1), p1 = min(t * PF_KT + srow + 32, Lc - 1);
        o0 = ((size_t)pt[p0 / PS] * PS + p0 % PS) * D + sch * 8;
        o1 = ((size_t)pt[p1 / PS] * PS + p1 % PS) * D + sch * 8;
      }
      if (stage_active) {
        kr0 = *reinterpret_cast<const uint4*>(kc + o0);
        kr1 = *reinterpret_cast<const uint4*>(kc + o1);
        vr0 = *reinterpret_cast<const uint4*>(vc + o0);
        vr1 = *reinterpret_cast<const uint4*>(vc + o1);
      }
    } else if (stage_active) {
      const uint32_t n0 = (uint32_t)(s0 + min((t - ntc) * PF_KT + srow, la_vis - 1));
      const uint32_t n1 = (uint32_t)(s0 + min((t - ntc) * PF_KT + srow + 32, la_vis - 1));
      const uint16_t* kg = k + (size_t)g * sk_h + sch * 8;
      const uint16_t* vg = v + (size_t)g * sv_h + sch * 8;
      kr0 = *reinterpret_cast<const uint4*>(kg + (uint64_t)n0 * skn);
      kr1 = *reinterpret_cast<const uint4*>(kg + (uint64_t)n1 * skn);
      vr0 = *reinterpret_cast<const uint4*>(vg + (uint64_t)n0 * svn);
      vr1 = *reinterpret_cast<const uint4*>(vg + (uint64_t)n1 * svn);
    }
  };

  // ---- accumulators --------------------------------------------------------------------------------------------
  f32x16 oacc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;
  float m_run = -INFINITY;  // running max (exp2 domain), identical in lanes l and l^32
  float l_run = 0.f;        // running sum of this lane's half of the keys

  // per-lane constant parts of the LDS addresses
  const int gi = lane >> 4;       // 16-lane group
  const int li = lane & 15;
  const int tq = li >> 2, tp = li & 3;

  // ---- main loop: one 64-key tile per iteration, software-pipelined by one tile inside every wave -----------------
  // Iteration t holds the raw logits of tile t in registers (computed one iteration earlier) and issues, in ONE
  // basic block, the 16 MFMAs of S(t+1) = K(t+1) Q^T next to the exp / pack VALU work of tile t, then the 16 MFMAs
  // of O += P(t) V(t): the matrix pipe has work while the wave's own softmax runs, without relying on a SIMD
  // partner being in the opposite phase.  K is double- and V triple-buffered in LDS so that tile t+2 can be written
  // during iteration t with ONE barrier per tile: K(t+2) replaces K(t) (last read in iteration t-1), V(t+2)
  // replaces V(t-1).  Global loads of tile t+2 are issued at the top of iteration t and land in LDS at its end.
  // Measured (32K tokens, bf16, TFLOP/s): this loop 1043.  Lock-step phases (QK, softmax, PV of one tile) 840; a
  // half-tile stagger of waves 4-7 with two barriers per tile 786; row sums by an all-ones MFMA instead of 32 v_add
  // 958 (+16 VGPRs spill); the row-sum adds pinned into this block, where hipcc then forms one VALU lump between the
  // MFMA runs, 968; sched_group_barrier pipelines over the whole block or over two sched_barrier-separated regions 968;
  // a static s_setprio 1 for waves 4-7: no difference (1040 +- 2 in an A/B on one box).
  char* const kbase = smem;
  char* const vbase = smem + 2 * PF_KTILE;
  const uint32_t kst = srow * PF_KSTR + sch * 16, vst = srow * PF_VSTR + sch * 16;
  auto lstore = [&](int kbuf, int vbuf) {
    if (stage_active) {
      char* kb = kbase + kbuf * PF_KTILE + kst;
      char* vb = vbase + vbuf * PF_VTILE + vst;
      *reinterpret_cast<uint4*>(kb) = kr0;
      *reinterpret_cast<uint4*>(kb + 32 * PF_KSTR) = kr1;
      *reinterpret_cast<uint4*>(vb) = vr0;
      *reinterpret_cast<uint4*>(vb + 32 * PF_VSTR) = vr1;
    }
  };
  const uint32_t k_lane = r * PF_KSTR + h * 16;                                           // + 32 s (+ 32 rows)
  const uint32_t v_lane = (4 * (gi >> 1) + tq) * PF_VSTR + (gi & 1) * 32 + tp * 8;       // + rows, + 64 db
  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)smem;  // LDS byte address of the dynamic segment
  // The tile base is added to the lane constant ONCE and laundered, so that every read below is that VGPR plus an
  // immediate (left alone, hipcc folds the immediates into the uniform tile base and spends one v_add per read).
  auto qk_tile = [&](int kbuf, f32x16& s0_, f32x16& s1_) {
    uint32_t ka = lds0 + kbuf * PF_KTILE + k_lane;
    asm volatile("" : "+v"(ka));
    const lds_char* kb = (const lds_char*)(uintptr_t)ka;
#pragma unroll
    for (int i = 0; i < 16; ++i) s0_[i] = 0.f, s1_[i] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const s16x8 a0 = *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(kb + 32 * s);
      const s16x8 a1 = *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(kb + 32 * s + 32 * PF_KSTR);
      s0_ = mfma32<T>(a0, qf[s], s0_);
      s1_ = mfma32<T>(a1, qf[s], s1_);
    }
  };

  kr0 = kr1 = vr0 = vr1 = make_uint4(0, 0, 0, 0);
  gload(0);
  lstore(0, 0);
  if (ntiles > 1) gload(1);
  lds_barrier();
  f32x16 sc0, sc1;  // raw logits of the current tile (keys 0-31 / 32-63)
  qk_tile(0, sc0, sc1);
  float mx_raw = tile_max(sc0, sc1);
  if (ntiles > 1) lstore(1, 1);
  lds_barrier();

  int vcur = 0;  // t % 3
  for (int t = 0; t < ntiles; ++t) {
    if (t + 2 < ntiles) gload(t + 2);

    const int vnext2 = vcur == 0 ? 2 : vcur - 1;  // (t + 2) % 3

    const bool cached = t < ntc;
    const int j0 = cached ? t * PF_KT : (t - ntc) * PF_KT;
    const int count = cached ? min(PF_KT, Lc - j0) : min(PF_KT, la_vis - j0);
    const bool need_mask = count < PF_KT || (!cached && j0 + PF_KT - 1 > m0);  // wave-uniform
    if (need_mask) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kk = (i & 3) + 8 * (i >> 2) + 4 * h;
        const bool vis0 = kk < count && (cached || (j0 + kk) <= tok);
        const bool vis1 = kk + 32 < count && (cached || (j0 + kk + 32) <= tok);
        sc0[i] = vis0 ? sc0[i] : -INFINITY;
        sc1[i] = vis1 ? sc1[i] : -INFINITY;
      }
    }
    // max over the RAW logits (scale > 0 commutes with max), then p = exp2(s*c - m*c): one FMA per logit.
    // The unmasked max was computed next to the previous tile's PV MFMAs; masked tiles (rare) redo it here.
    if (need_mask) mx_raw = tile_max(sc0, sc1);
    const float mx = mx_raw * scale_log2e;
    // Deferred rescale (MI355X guide T13; the decision sits before the tile's exp work and after the previous tile's
    // PV, so no P is pending): the running max follows the row max only once it has grown by more than PF_THR in
    // the exp2 domain.  A row's first finite max always takes (-inf + PF_THR = -inf); a row that is still all -inf never
    // does, and no NaN is formed on the way (the file is built with -fno-honor-nans).
    const bool grew = mx > m_run + (float)PF_THR;
    const float m_new = grew ? mx : m_run;
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
    m_run = m_new;
    if (__any(grew)) {  // alpha == 1 for every lane otherwise
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[db][i] *= alpha;
    }

    // ---- S(t+1) = K(t+1) Q^T  (MFMA)   ||   P(t) = exp2(S(t) c - m) packed to the model dtype  (VALU) -----------
    // On the last tile this recomputes an old K tile; the result is never used.
    f32x16 sn0, sn1;
    qk_tile((t + 1) & 1, sn0, sn1);

    float psum = 0.f;
    s16x8 pf[4];
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        uint32_t w[4];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
          const float x0 = kb2 ? sc1[8 * s2 + 2 * jp] : sc0[8 * s2 + 2 * jp];
          const float x1 = kb2 ? sc1[8 * s2 + 2 * jp + 1] : sc0[8 * s2 + 2 * jp + 1];
          const float p0 = __builtin_amdgcn_exp2f(fmaf(x0, scale_log2e, -m_safe));
          const float p1 = __builtin_amdgcn_exp2f(fmaf(x1, scale_log2e, -m_safe));
          w[jp] = pack2<T>(p0, p1);
          // row sum from the fp32 probabilities, P itself rounded to the model dtype (reference :398-400)
          psum += p0 + p1;
        }
        pf[kb2 * 2 + s2] = __builtin_bit_cast(s16x8, make_uint4(w[0], w[1], w[2], w[3]));
      }
    }
    l_run = l_run * alpha + psum;

    // ---- O^T += V(t)^T P(t)^T ------------------------------------------------------------------------------------
    uint32_t va = lds0 + 2 * PF_KTILE + vcur * PF_VTILE + v_lane;
    asm volatile("" : "+v"(va));
    const lds_char* vb = (const lds_char*)(uintptr_t)va;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const uint32_t a0 = ((ks >> 1) * 32 + (ks & 1) * 16) * PF_VSTR + 64 * db;  // immediate
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + a0));
        const s16x4 hi =
            __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + a0 + 8 * PF_VSTR));
        const s16x8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        oacc[db] = mfma32<T>(a, pf[ks], oacc[db]);
      }
    }
    mx_raw = tile_max(sn0, sn1);  // next tile, unmasked; overlaps the PV MFMAs above
    if (t + 2 < ntiles) lstore(t & 1, vnext2);
    lds_barrier();
    sc0 = sn0;
    sc1 = sn1;
    vcur = vcur == 2 ? 0 : vcur + 1;
  }

  // ---- epilogue: normalise, stage O through LDS (wave-private region), store whole rows ------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  char* ob = smem + wave * (32 * PF_OSTRIDE);
#pragma unroll
  for (int db = 0; db < DB; ++db) {
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const int d = db * 32 + 8 * i4 + 4 * h;  // accumulator rows (i&3)+8*(i>>2)+4h for i = 4*i4 .. 4*i4+3
      const uint32_t w0 = pack2<T>(oacc[db][4 * i4] * inv, oacc[db][4 * i4 + 1] * inv);
      const uint32_t w1 = pack2<T>(oacc[db][4 * i4 + 2] * inv, oacc[db][4 * i4 + 3] * inv);
      *reinterpret_cast<uint2*>(ob + r * PF_OSTRIDE + d * 2) = make_uint2(w0, w1);
    }
  }
  // the wave re-reads only what it wrote itself: LDS ops of one wave execute in order, no barrier needed
  constexpr int RPI = 64 / CH;          // rows per store instruction
#pragma unroll
  for (int it = 0; it < 32 / RPI; ++it) {
    const int rr = it * RPI + lane / CH;
    const int ch = lane % CH;
    const int grow = wave * 32 + rr;
    const int ghead = g * G + grow / BM;
    const int gtok = m0 + grow % BM;
    if (gtok < La) {
      const uint4 val = *reinterpret_cast<const uint4*>(ob + rr * PF_OSTRIDE + ch * 16);
      *reinterpret_cast<uint4*>(out + ((size_t)(s0 + gtok) * HQ + ghead) * D + ch * 8) = val;
    }
  }
}


// =====================================================================================================================
// 4-wave structure (D = 128, page size a multiple of 64): one wave per SIMD, 64 query rows per wave.
//
// Why a second structure: with 32 rows per wave (kernel above) every wave re-reads the whole K and V tile from LDS for
// 32 MFMAs' worth of work, and the two waves of a SIMD, released by the same barrier, contend for the one vector-issue
// port in lock-step (SQ counters, round 2: 31 % of wave cycles parked, 32 % issue-stalled, matrix pipe 50 % busy).
// Here a wave owns TWO 32-row query blocks: every K / V fragment read from LDS feeds two MFMAs (half the LDS bytes
// per flop), there is no co-resident wave, and the wave's own softmax VALU work is hand-placed into the 32-cycle
// shadows of its own MFMAs (see the loop).  What bounds it: with one wave per SIMD EVERY instruction - VALU, scalar,
// LDS, wait - takes an issue slot (>= 4 cycles) from the same in-order stream; a 64-key tile is 64 MFMAs (2048
// cycles of matrix pipe) and 480 instructions; measured 2 650 cycles per tile (round 2: 4 100; the round's steps and what
// each bought: profiles/r03_prefill_slot_timeline.txt).
//   * workgroup = 4 waves = the same 256 rows ((256/G) tokens x G heads of one kv-head) as above, same grid order;
//   * registers: O in a[128:255] and the softmax state the rare path rewrites in v[222:255], both owned by inline asm
//     (see above); Q fragments 64, two 32-key logit blocks 64, fragment rings 32, DMA source offsets 16;
//   * K/V tiles come in by LDS-DMA (buffer_load_dwordx4 ... lds: tile base and valid bytes in the resource descriptor,
//     lane-constant source offsets, rows past the end of the tile arrive as zeros) into read-order LDS images, four
//     buffers each for K and V, issued three tiles ahead (-DP4_DMA=0: register staging into padded images, three buffers);
//   * the softmax differs from the 8-wave kernel's: probabilities are taken against a per-row reference point that is
//     moved only when a check of the packed probabilities finds one >= 2 ("optimistic probabilities", in the kernel),
//     per 32-key unit.  Both kernels meet the attention tolerance of the parity tests (tests/test_gpu_prefill.py runs
//     every case on both).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// LDS fragment reads run P4_RA fragments ahead of the MFMAs that consume them (tunable, A/B builds)
#ifndef P4_RA
#define P4_RA 3
#endif
// issue cycles of exp work dealt to one MFMA shadow (A/B builds; the stream's 628 cycles fill 30 shadows at 21: the check's
// ballot is then two shadows old when it is tested)
#ifndef P4_CAP
#define P4_CAP 21
#endif

// ---- the softmax work of one 32-key unit as ONE stream of 125 micro-ops, dealt evenly over 32 MFMA shadows ----------
// Item j = 8 s2 + 4 qb + jp turns two logits of query block qb into two probabilities and one packed word: F0 F1 (the
// logits into the exp2 domain, minus the row's reference point), E0 E1 (exp2), A0 A1 (sum into the unit's own row sums)
// and PK (round both to the model dtype).  Stream order: the four FOLD ops (the PREVIOUS unit's sums into the running
// sums l), item 0's F F E E, then for b = 1..15: F F of item b, A A PK of item b-1, E E of item b - an exp2 never directly
// follows the fma that feeds it and an add never directly follows the exp2 it consumes - and, wound into the tail, the nine
// ops of the CHECK (see the kernel: "optimistic probabilities"): five ORs over the first 15 words as soon as they exist,
// A A PK of item 15, then the three ops that depend on its word.
// The stream of unit u runs in the 16 shadows of phase B of unit u-1 (its logits S(u) are complete when phase A ends) and
// the 16 shadows of phase A of unit u; the check is tested between the phases, before the first PV MFMA of the unit.
// 32 shadows for 628 issue cycles (exp2 8, the others 4), ~20 per shadow: with the shadow's MFMA (8) and its one or two
// LDS reads a shadow's issue work fits its 32 cycles of matrix pipe.  (Rounds 2-3 computed the exact row max of S(u)
// first - the last 8 shadows of a unit - and dealt the exp work over 24 shadows at 24 cycles each: every one of those 24
// over its budget, the other 8 under it; profiles/r03_prefill_slot_timeline.txt.)
struct P4Op {
  int kind, arg;
};
constexpr int P4_F0 = 0, P4_F1 = 1, P4_E0 = 2, P4_E1 = 3, P4_A0 = 4, P4_A1 = 5, P4_PK = 6, P4_CHK = 7, P4_FOLD = 8;
constexpr int P4_NOPS = 125;
struct P4Stream {
  P4Op op[P4_NOPS];
  int first[33];  // first op of stream shadow sl (0-15: phase B of the previous unit, 16-31: phase A)
};
constexpr P4Stream p4_build() {
  P4Stream s{};
  int n = 0;
  for (int i = 0; i < 4; ++i) s.op[n++] = P4Op{P4_FOLD, i};
  s.op[n++] = P4Op{P4_F0, 0};
  s.op[n++] = P4Op{P4_F1, 0};
  s.op[n++] = P4Op{P4_E0, 0};
  s.op[n++] = P4Op{P4_E1, 0};
  for (int b = 1; b < 16; ++b) {
    s.op[n++] = P4Op{P4_F0, b};
    s.op[n++] = P4Op{P4_F1, b};
    s.op[n++] = P4Op{P4_A0, b - 1};
    s.op[n++] = P4Op{P4_A1, b - 1};
    s.op[n++] = P4Op{P4_PK, b - 1};
    s.op[n++] = P4Op{P4_E0, b};
    s.op[n++] = P4Op{P4_E1, b};
  }
  for (int i = 0; i < 4; ++i) s.op[n++] = P4Op{P4_CHK, i};  // words 0-11
  s.op[n++] = P4Op{P4_A0, 15};
  s.op[n++] = P4Op{P4_A1, 15};
  s.op[n++] = P4Op{P4_CHK, 4};  // words 12-14
  s.op[n++] = P4Op{P4_PK, 15};
  for (int i = 5; i < 9; ++i) s.op[n++] = P4Op{P4_CHK, i};
  // deal by issue cost
  int cost = 0, sl = 0;
  s.first[0] = 0;
  for (int g = 0; g < P4_NOPS; ++g) {
    int slot = cost / P4_CAP;
    if (slot > 31) slot = 31;
    while (sl < slot) s.first[++sl] = g;
    cost += (s.op[g].kind == P4_E0 || s.op[g].kind == P4_E1) ? 8 : 4;
  }
  while (sl < 32) s.first[++sl] = P4_NOPS;
  return s;
}
constexpr P4Stream P4S = p4_build();
static_assert(P4S.first[32] == P4_NOPS && P4S.op[P4_NOPS - 1].kind == P4_CHK && P4S.op[P4_NOPS - 1].arg == 8, "stream");
static_assert(P4S.first[16] > 4 + 4 + 7 * 3 + 5,
              "the words of items 0-3 are packed in phase B: at least a whole phase before the MFMAs that read them");
// -DCVLLM_PF_TS (debug builds of tools/dbg only): s_memrealtime at the phase boundaries of every workgroup
#ifdef CVLLM_PF_TS
__device__ unsigned long long g_pf_rt[8192 * 8];  // [workgroup][4 x s_memrealtime (100 MHz) | 4 x s_memtime (shader clock)]
#define PF_RT(i)                                                                                              \
  do {                                                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < 8192) {                                                              \
      g_pf_rt[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime();                                       \
      g_pf_rt[blockIdx.x * 8 + 4 + (i)] = __builtin_amdgcn_s_memtime();                                       \
    }                                                                                                         \
  } while (0)
#else
#define PF_RT(i) \
  do {           \
  } while (0)
#endif
// -DP4_TS_SLOT=k (debug builds of tools/dbg/slot_ts.sh only, k in 0..64): s_memtime at the entry of a steady-state tile, in
// front of MFMA slot k of it (0-31 first unit, 32-63 second unit, 64 = before the barrier) and behind the barrier - the
// per-slot timeline of one wave.  The stamps are inline asm, i.e. invisible to hipcc's wait-count pass: a counted LDS wait
// that follows one may be a count too loose (results of such a build are not checked), the timing is what it is for.
#ifdef P4_TS_SLOT
__device__ unsigned long long g_p4_slot[8192 * 4];
#ifndef P4_TS_WAVE
#define P4_TS_WAVE 0
#endif
#define P4_STAMP(var) asm volatile("s_memtime %0" : "=s"(var))
#define P4_SLOT_STAMP(slot)                          \
  do {                                               \
    if constexpr ((slot) == P4_TS_SLOT) P4_STAMP(ts_k); \
  } while (0)
#else
#define P4_SLOT_STAMP(slot) \
  do {                      \
  } while (0)
#endif
constexpr int P4_THREADS = 256;
// K / V tiles reach LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write - the eight
// ds_write_b128 of a tile cost a wave ~290 of its ~2,950 cycles per tile, profiles/r03_prefill_slot_timeline.txt).  A DMA
// instruction writes 64 x 16 B to CONSECUTIVE LDS bytes, so the images are in READ ORDER: the 1 KB a ds_read_b128 of K
// k-step s (or the two ds_read_b64_tr_b16 of a V fragment) takes is one contiguous block, lane-linear - conflict free,
// every fragment address = one lane-constant VGPR + an immediate - and each lane's SOURCE offset picks the 16 B that
// belong there.  Four buffers of K and of V (128 KB): the DMAs of tile t+3 are issued during tile t and waited for
// (counted vmcnt) at the end of tile t+1.
#ifndef P4_DMA
#define P4_DMA 1
#endif
#if P4_DMA
constexpr int P4_NBUF = 4;
constexpr int P4_KTILE = PF_KT * 256, P4_VTILE = PF_KT * 256;  // 16 KB each, no padding
#else
constexpr int P4_NBUF = 3;
constexpr int P4_KTILE = PF_KTILE, P4_VTILE = PF_VTILE;
#endif
constexpr int P4_SMEM = P4_NBUF * (P4_KTILE + P4_VTILE);  // DMA: 131,072 B; register staging: 113,664 B
static_assert(P4_SMEM >= 4 * 64 * PF_OSTRIDE && P4_SMEM <= 160 * 1024, "O staging image must fit");


template <typename T, int G>
__global__ __launch_bounds__(P4_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1), amdgpu_num_vgpr(P4_NUM_VGPR))) void
prefill_attn_w4_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v, int64_t sq_n,
    int64_t sk_n, int64_t sk_h, int64_t sv_n, int64_t sv_h, const uint16_t* __restrict__ kc,
    const uint16_t* __restrict__ vc, uint16_t* __restrict__ out, const int* __restrict__ seq_lens,
    const int* __restrict__ page_table, const int* __restrict__ bmap, const int* __restrict__ cu, int B, int HKV,
    int PS, int NLP, float scale_log2e) {
  constexpr int D = 128;
  constexpr int BM = PF_ROWS / G;  // tokens per query tile
  constexpr int KS = D / 16;       // k-steps of the QK^T product
  constexpr int DB = D / 32;       // 32-wide blocks of the head dim

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int bid = blockIdx.x;
  const int g = bid % HKV;
  const int b = (bid / HKV) % B;
  const int tile_rev = bid / (HKV * B);
  const int s0 = cu[b];
  const int La = cu[b + 1] - s0;
  const int nqt = (La + BM - 1) / BM;
  if (tile_rev >= nqt) return;
  PF_RT(0);
  const int qt = nqt - 1 - tile_rev;  // heaviest tiles first
  const int m0 = qt * BM;
  const int HQ = HKV * G;

  const int Lc = seq_lens[b * HKV + g];
  const int* pt = page_table + ((size_t)bmap[b] * HKV + g) * NLP;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;

  // ---- the wave's two query blocks: rows wave*64 + 32*qb + r -----------------------------------------------------
  int tok[2];
  s16x8 qf[2][KS];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int row = wave * 64 + qb * 32 + r;
    const int head_local = row / BM;
    tok[qb] = m0 + row % BM;
    const bool valid_q = tok[qb] < La;
    const uint16_t* qp = q + (size_t)(s0 + (valid_q ? tok[qb] : 0)) * sq_n + (size_t)(g * G + head_local) * D + 8 * h;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      uint4 t = valid_q ? *reinterpret_cast<const uint4*>(qp + 16 * s) : make_uint4(0, 0, 0, 0);
      qf[qb][s] = __builtin_bit_cast(s16x8, t);
    }
  }

  const int ntc = (Lc + PF_KT - 1) / PF_KT;      // cached-prefix tiles
  const int la_vis = min(m0 + BM, La);           // appended keys visible to this query tile
  const int nta = (la_vis + PF_KT - 1) / PF_KT;  // appended tiles (last ones are on the diagonal)
  const int ntiles = ntc + nta;

  // ---- accumulators and LDS addressing ------------------------------------------------------------------------------
  acc_zero_all();  // O^T accumulators a[128:255] (see pv_mfma_w): block (qb, db) = a[128 + 16 (4 qb + db) .. + 15]
  // Optimistic probabilities.  A probability is exp2(s c - m_ref) with m_ref the row's REFERENCE POINT (identical in
  // lanes l and l^32): the row max as of the last update plus P4_BIAS.  Everything accumulated (O, l) and every pending
  // packed word is relative to m_ref; the row max of a unit is NOT computed on the common path.  Instead the packed
  // words are checked before any MFMA consumes them: bit 14 of a bf16 / fp16 pattern is the top exponent bit, set iff
  // the value is >= 2 (or inf / NaN), so OR-ing a unit half's 8 words and testing 0x40004000 tells whether some logit
  // has climbed P4_BIAS + 1 above the row max of the last update (rare - the deferred rescale of the MI355X guide, T13,
  // with the test moved from the logits to the probabilities: 6 VALU per check instead of ~15 for a row max).  Then, and
  // only then (`row_fix`): exact row max of the unit, new m_ref, O and l rescaled, and the unit's stream replayed from
  // its logits, which are still in registers.  The unit's own row sums (lu, lu2) are folded into l only after its second
  // check: a probability that overflowed fp32 never reaches l, and one that overflowed the 16-bit type never reaches O.
  // Probabilities are at most 2 and at least 2^-P4_BIAS at the old row max: bf16 keeps its relative precision at any
  // magnitude, fp16 (normal down to 2^-14) takes a smaller bias.
  constexpr float P4_BIAS = std::is_same<T, BF16>::value ? 7.f : 2.f;
  // reference points, running sums l (two per query block: even / odd logits of an item, so that consecutive adds never
  // depend on each other), the sums of the unit in flight and the packed words: asm-owned registers, see P4_VNM
  p4_state_init();
  uint32_t chk_a = 0, chk_b = 0, chk_c = 0, chk_d = 0, chk_e = 0;
  unsigned long long chk_bal = 0;    // ballot of the unit's check (SGPR pair)
  unsigned long long rare_flag = 0;  // ... or-ed with "the next unit needs the mask": what the phase boundary branches on

  const int gi = lane >> 4;
  const int li = lane & 15;
  const int tq = li >> 2, tp = li & 3;
#if P4_DMA
  const uint32_t k_lane = lane * 16;  // read-order images: lane-linear
  const uint32_t v_lane = lane * 8;
  (void)gi, (void)tq, (void)tp;
#else
  const uint32_t k_lane = r * PF_KSTR + h * 16;
  const uint32_t v_lane = (4 * (gi >> 1) + tq) * PF_VSTR + (gi & 1) * 32 + tp * 8;
#endif
  typedef __attribute__((address_space(3))) char lds_char;
  typedef __attribute__((address_space(3))) s16x8 lds_s16x8;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)smem;

#if P4_DMA
  // ---- staging by LDS-DMA ---------------------------------------------------------------------------------------------
  // A tile = 16 K blocks + 16 V blocks of 1 KB; wave w issues blocks 4w .. 4w+3 of each (four K and four V DMAs per tile).
  //   K block (kb, s) = bytes of ds_read_b128 for k-step s of unit kb: lane L holds key 32 kb + (L & 31), 16-byte chunk
  //                     2 s + (L >> 5) of its row;  block index 8 kb + s.
  //   V block (kb, i) = bytes of the two ds_read_b64_tr_b16 of fragment i = 4 s2 + db of unit kb (lo | hi, 512 B each):
  //                     DMA lane L covers the 8-byte pieces of read lanes 2 m and 2 m + 1 (m = L & 31) of the lo (L < 32)
  //                     or hi read - 16 contiguous bytes of key 32 kb + 16 s2 + 8 hi + 4 (m >> 4) + ((m & 7) >> 1) at column
  //                     64 db + 32 ((m >> 3) & 1) + 16 (m & 1);  block index 8 kb + i.
  // The tile's base and valid byte count sit in the resource descriptor, the lane's (row, column) in loop-invariant VGPR
  // offsets, one set per layout (cached rows are D elements apart, appended rows sk_n / sv_n): rows past the end of the
  // tile are out of range and arrive as zeros.
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  uint32_t voffKc[4], voffKa[4], voffVc[4], voffVa[4];
  {
    const int wkb = wave >> 1, w1 = wave & 1, m = lane & 31, hi = lane >> 5;
    const uint32_t krow = 32 * wkb + m;
    const uint32_t vrow = 32 * wkb + 16 * w1 + 8 * hi + 4 * (m >> 4) + ((m & 7) >> 1);
    const uint32_t vcol = 32 * ((m >> 3) & 1) + 16 * (m & 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t kcol = (2 * (4 * w1 + j) + hi) * 16;
      voffKc[j] = krow * (D * 2) + kcol;
      voffKa[j] = krow * (uint32_t)sk_n * 2 + kcol;
      voffVc[j] = vrow * (D * 2) + 64 * j + vcol;
      voffVa[j] = vrow * (uint32_t)sv_n * 2 + 64 * j + vcol;
    }
  }
  int ld_pi = 0, ld_po = 0;
  int ld_pg = (ntc > 0) ? pt[0] : 0;
  bool ld_app = false;  // layout of the tile whose descriptors are current (workgroup-uniform)
  u32x4 rk, rv;         // buffer resource descriptors (SGPRs): base, stride 0, valid bytes, raw-buffer flags
  auto mk_rsrc = [&](const uint16_t* p, int nbytes) __attribute__((always_inline)) {
    const uint64_t a = (uint64_t)p;
    const u32x4 d = {(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, (uint32_t)nbytes, 0x00020000u};
    return d;
  };
  // running state of the two kinds: the current page's base (cached) / the next tile's base and the valid bytes from
  // there to the end of the visible rows (appended) - a tile costs two 64-bit adds and a subtract, not a 64-bit multiply
  const uint16_t* c_kp = kc + (int64_t)ld_pg * PS * D;
  const uint16_t* c_vp = vc + (int64_t)ld_pg * PS * D;
  const uint16_t* a_kp = k + (int64_t)s0 * sk_n + (int64_t)g * sk_h;
  const uint16_t* a_vp = v + (int64_t)s0 * sv_n + (int64_t)g * sv_h;
  int a_remk = la_vis > 0 ? (int)(((uint32_t)(la_vis - 1) * (uint32_t)sk_n + D) * 2) : 0;
  int a_remv = la_vis > 0 ? (int)(((uint32_t)(la_vis - 1) * (uint32_t)sv_n + D) * 2) : 0;
  const int a_stepk = PF_KT * (int)sk_n, a_stepv = PF_KT * (int)sv_n;  // elements per appended tile
  auto tile_desc_app = [&]() __attribute__((always_inline)) {
    // valid bytes reach to the END of the visible rows (the row offsets cover 64 rows anyway); tiles past the last one
    // have nothing left: every row out of range, nothing is fetched
    rk = mk_rsrc(a_kp, max(a_remk, 0));
    rv = mk_rsrc(a_vp, max(a_remv, 0));
    a_kp += a_stepk;
    a_vp += a_stepv;
    a_remk -= 2 * a_stepk;
    a_remv -= 2 * a_stepv;
  };
  auto tile_desc = [&](int tt) __attribute__((always_inline)) {
    ld_app = tt >= ntc;
    if (!ld_app) {
      const int count = min(PF_KT, Lc - tt * PF_KT);  // > 0
      rk = mk_rsrc(c_kp + ld_po * D, count * D * 2);
      rv = mk_rsrc(c_vp + ld_po * D, count * D * 2);
      ld_po += PF_KT;
      if (ld_po >= PS) {
        ld_po = 0;
        ld_pi += 1;
        ld_pg = pt[min(ld_pi, NLP - 1)];
        c_kp = kc + (int64_t)ld_pg * PS * D;  // int64 row offset (reference :371)
        c_vp = vc + (int64_t)ld_pg * PS * D;
      }
    } else {
      tile_desc_app();
    }
  };
  // one DMA: 64 lanes x 16 B from rsrc + voff to LDS bytes [dst + 1024 j, + 1024).  Inline asm: M0 (the LDS base of the
  // DMA) is written in the statement that uses it, and hipcc's wait-count pass does not see the instruction - it would
  // otherwise drain vmcnt to 0 in front of every ds_read that may alias the destination.  The waits are placed by hand
  // (dma_wait).
  auto dma16 = [&](const u32x4& rs, uint32_t voff, uint32_t dst, auto j_c) __attribute__((always_inline)) {
#if defined(P4_NO_DMA)  // timing-only A/B builds (wrong results): what the loop's DMAs cost, and what their M0 writes cost
    (void)rs, (void)voff, (void)dst;
#elif defined(P4_DMA_SAMEM0)
    asm volatile("buffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(dst), "v"(voff), "s"(rs) : "memory");
#elif defined(P4_DMA_SPLIT)  // A/B: M0 written one shadow ahead of the DMA (dma_m0 below), no s_nop
    asm volatile("buffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(dst), "v"(voff), "s"(rs) : "memory");
#else
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(dst), "v"(voff), "s"(rs), "i"(1024 * decltype(j_c)::value)
                 : "memory", "scc");
#endif
  };
  auto dma_m0 = [&](uint32_t dst, auto j_c) __attribute__((always_inline)) {
#ifdef P4_DMA_SPLIT
    asm volatile("s_add_u32 m0, %0, %1" : : "s"(dst), "i"(1024 * decltype(j_c)::value) : "memory", "scc");
#else
    (void)dst;
#endif
  };
  // LDS byte address of the wave's first K / V block of buffer `buf`
  auto dma_dst_k = [&](int buf) __attribute__((always_inline)) { return lds0 + (uint32_t)(buf * P4_KTILE + wv * 4096); };
  auto dma_dst_v = [&](int buf) __attribute__((always_inline)) {
    return lds0 + (uint32_t)(P4_NBUF * P4_KTILE + buf * P4_VTILE + wv * 4096);
  };
  auto dma_k = [&](uint32_t dst, auto j_c, auto app_c) __attribute__((always_inline)) {
    constexpr int j = decltype(j_c)::value;
    if constexpr (decltype(app_c)::value)
      dma16(rk, voffKa[j], dst, j_c);
    else
    {
      const uint32_t oa = voffKa[j], oc = voffKc[j];
      dma16(rk, ld_app ? oa : oc, dst, j_c);
    }
  };
  auto dma_v = [&](uint32_t dst, auto j_c, auto app_c) __attribute__((always_inline)) {
    constexpr int j = decltype(j_c)::value;
    if constexpr (decltype(app_c)::value)
      dma16(rv, voffVa[j], dst, j_c);
    else
    {
      const uint32_t oa = voffVa[j], oc = voffVc[j];
      dma16(rv, ld_app ? oa : oc, dst, j_c);
    }
  };
#else
  // ---- staging: thread -> rows srow + 16 i (i = 0..3), 16-byte chunk sch of a K and of a V tile -------------------
  // Buffer loads: the tile's base and size sit in the resource descriptor (SGPRs), the row block in the scalar offset,
  // the lane's (row, chunk) in ONE loop-invariant VGPR - no per-load address arithmetic on the VALU; rows past the end
  // of the tile are out of range and read as zero (their logits are masked).  Cached and appended tiles differ only
  // in scalars, so a tile's 8 loads and 8 LDS writes can be dealt out one by one to MFMA slots without branches.
  const int sch = tid & 15;
  const int srow = tid >> 4;  // 0..15
  u32x4 st[2][8];  // two register sets (K rows 0-3, V rows 4-7), every index a compile-time constant
  int ld_pi = 0, ld_po = 0;
  int ld_pg = (ntc > 0) ? pt[0] : 0;
  // The bounds check of a buffer load covers the VGPR offset only (not the scalar offset), so the row offset lives
  // in the VGPR - four loop-invariant offsets per layout (cached rows are D elements apart, appended rows sk_n / sv_n),
  // all three sets live for the whole kernel (the output accumulators sit in AGPRs, arch VGPRs are plentiful) and ONE
  // v_cndmask per load picks the tile's layout (keeping a "current" set that is rewritten at the cached -> appended
  // change made hipcc copy a set in every tile: 16-33 v_mov) - and the tile's base and valid byte count in the descriptor.
  uint32_t voffc[4], voffak[4], voffav[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    voffc[i] = (uint32_t)(srow + 16 * i) * (D * 2) + sch * 16;
    voffak[i] = (uint32_t)(srow + 16 * i) * (uint32_t)sk_n * 2 + sch * 16;
    voffav[i] = (uint32_t)(srow + 16 * i) * (uint32_t)sv_n * 2 + sch * 16;
  }
  bool ld_app = false;  // layout of the tile whose descriptors are current (workgroup-uniform)
  __amdgpu_buffer_rsrc_t rk, rv;
  // running state of the two kinds: the current page's base (cached) / the next tile's base and the valid bytes from
  // there to the end of the visible rows (appended) - a tile costs two 64-bit adds and a subtract, not a 64-bit multiply
  const uint16_t* c_kp = kc + (int64_t)ld_pg * PS * D;
  const uint16_t* c_vp = vc + (int64_t)ld_pg * PS * D;
  const uint16_t* a_kp = k + (int64_t)s0 * sk_n + (int64_t)g * sk_h;
  const uint16_t* a_vp = v + (int64_t)s0 * sv_n + (int64_t)g * sv_h;
  int a_remk = la_vis > 0 ? (int)(((uint32_t)(la_vis - 1) * (uint32_t)sk_n + D) * 2) : 0;
  int a_remv = la_vis > 0 ? (int)(((uint32_t)(la_vis - 1) * (uint32_t)sv_n + D) * 2) : 0;
  const int a_stepk = PF_KT * (int)sk_n, a_stepv = PF_KT * (int)sv_n;  // elements per appended tile
  auto tile_desc = [&](int tt) __attribute__((always_inline)) {
    ld_app = tt >= ntc;
    if (!ld_app) {
      const int count = min(PF_KT, Lc - tt * PF_KT);  // > 0
      rk = __builtin_amdgcn_make_buffer_rsrc((void*)(c_kp + ld_po * D), 0, count * D * 2, 0x00020000);
      rv = __builtin_amdgcn_make_buffer_rsrc((void*)(c_vp + ld_po * D), 0, count * D * 2, 0x00020000);
      ld_po += PF_KT;
      if (ld_po >= PS) {
        ld_po = 0;
        ld_pi += 1;
        ld_pg = pt[min(ld_pi, NLP - 1)];
        c_kp = kc + (int64_t)ld_pg * PS * D;  // int64 row offset (reference :371)
        c_vp = vc + (int64_t)ld_pg * PS * D;
      }
    } else {
      // valid bytes reach to the END of the visible rows (the four row offsets cover 64 rows anyway); tiles past the
      // last one have nothing left: every row out of range, nothing is fetched
      rk = __builtin_amdgcn_make_buffer_rsrc((void*)a_kp, 0, max(a_remk, 0), 0x00020000);
      rv = __builtin_amdgcn_make_buffer_rsrc((void*)a_vp, 0, max(a_remv, 0), 0x00020000);
      a_kp += a_stepk;
      a_vp += a_stepv;
      a_remk -= 2 * a_stepk;
      a_remv -= 2 * a_stepv;
    }
  };
  auto gload_piece = [&](int set, int i) __attribute__((always_inline)) {
    st[set][i] = (i < 4) ? __builtin_amdgcn_raw_buffer_load_b128(rk, ld_app ? voffak[i & 3] : voffc[i & 3], 0, 0)
                         : __builtin_amdgcn_raw_buffer_load_b128(rv, ld_app ? voffav[i & 3] : voffc[i & 3], 0, 0);
  };
  // the same two for loops in which every tile still to be loaded is an appended one (all of a prefill without a cached
  // prefix): no layout test, no select - two 64-bit adds, two subtracts, eight loads with loop-invariant offsets
  auto tile_desc_app = [&]() __attribute__((always_inline)) {
    rk = __builtin_amdgcn_make_buffer_rsrc((void*)a_kp, 0, max(a_remk, 0), 0x00020000);
    rv = __builtin_amdgcn_make_buffer_rsrc((void*)a_vp, 0, max(a_remv, 0), 0x00020000);
    a_kp += a_stepk;
    a_vp += a_stepv;
    a_remk -= 2 * a_stepk;
    a_remv -= 2 * a_stepv;
  };
#ifdef P4_DECOUPLE_LOADS  // timing-only A/B build (wrong results): the loop's loads land in registers nothing waits for
  u32x4 sink[8];
  auto gload_piece_app = [&](int set, int i) __attribute__((always_inline)) {
    sink[i] = (i < 4) ? __builtin_amdgcn_raw_buffer_load_b128(rk, voffak[i & 3], 0, 0)
                      : __builtin_amdgcn_raw_buffer_load_b128(rv, voffav[i & 3], 0, 0);
  };
#else
  auto gload_piece_app = [&](int set, int i) __attribute__((always_inline)) {
    st[set][i] = (i < 4) ? __builtin_amdgcn_raw_buffer_load_b128(rk, voffak[i & 3], 0, 0)
                         : __builtin_amdgcn_raw_buffer_load_b128(rv, voffav[i & 3], 0, 0);
  };
#endif
  const uint32_t kst = srow * PF_KSTR + sch * 16, vst = srow * PF_VSTR + sch * 16;
  auto lstore_piece = [&](int set, int buf, int i) __attribute__((always_inline)) {
    char* kb = smem + buf * PF_KTILE + kst;
    char* vb = smem + 3 * PF_KTILE + buf * PF_VTILE + vst;
    if (i < 4)
      *reinterpret_cast<u32x4*>(kb + 16 * i * PF_KSTR) = st[set][i];
    else
      *reinterpret_cast<u32x4*>(vb + 16 * (i - 4) * PF_VSTR) = st[set][i];
  };
#if defined(P4_NO_LSTORE) || defined(P4_LSTORE_CONST)  // timing-only A/B builds (wrong results): what the loop's LDS writes cost
  const u32x4 lst_const = {(uint32_t)tid, (uint32_t)tid, (uint32_t)tid, (uint32_t)tid};
  auto lstore_loop = [&](int set, int buf, int i) __attribute__((always_inline)) {
#ifdef P4_LSTORE_CONST
    char* kb = smem + buf * PF_KTILE + kst;
    char* vb = smem + 3 * PF_KTILE + buf * PF_VTILE + vst;
    if (i < 4)
      *reinterpret_cast<u32x4*>(kb + 16 * i * PF_KSTR) = lst_const;
    else
      *reinterpret_cast<u32x4*>(vb + 16 * (i - 4) * PF_VSTR) = lst_const;
#endif
  };
#else
  auto lstore_loop = lstore_piece;
#endif

#endif

  // The loop works in UNITS of 32 keys (half a staged tile).  For unit u (logits S(u), 16 registers per query block):
  //   phase A: 16 MFMAs  S(u+1) = K[unit u+1] Q^T      (every K fragment feeds both query blocks)
  //   phase B: 16 MFMAs  O^T   += V[unit u]^T P(u)^T   (every V fragment feeds both query blocks)
  // and the VALU / LDS / memory work is HAND-PLACED into the 32 MFMA shadows of the unit (a shadow = one MFMA + its
  // fillers, closed by sched_barrier(0) so that hipcc keeps the placement; left to itself it forms one VALU lump per
  // phase, which a single wave per SIMD cannot hide): the softmax stream of unit u+1 over phase B of unit u and phase A
  // of unit u+1 (P4Stream), the unit's check between its phases, the DMAs of tile t+3 over the first unit of tile t;
  // fragment reads run P4_RA fragments ahead of their MFMAs.  Budget per MFMA: 32 cycles of matrix pipe, 8 of them
  // holding the vector issue port.  Only two logit blocks (64 registers) are live (this file is built with
  // -amdgpu-mfma-vgpr-form: the QK^T results land in arch VGPRs, no v_accvgpr_read in front of the softmax) and nothing
  // spills.
#if P4_DMA
  constexpr int K_STEP = 1024, K_UNIT = 8192, V_FRAG = 1024, V_HI = 512, V_UNIT = 8192;
#else
  constexpr int K_STEP = 32, K_UNIT = 32 * PF_KSTR, V_UNIT = 32 * PF_VSTR;
#endif
  auto k_read = [&](const lds_char* kbp, int s) __attribute__((always_inline)) {
    return *reinterpret_cast<const lds_s16x8*>(kbp + K_STEP * s);
  };
  auto v_read = [&](const lds_char* vbp, int i) __attribute__((always_inline)) {  // i = 4 * s2 + db
#if P4_DMA
    const uint32_t a0 = V_FRAG * i, a1 = a0 + V_HI;
#else
    const uint32_t a0 = (i >> 2) * 16 * PF_VSTR + 64 * (i & 3), a1 = a0 + 8 * PF_VSTR;
#endif
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vbp + a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vbp + a1));
    const s16x8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return a;
  };
  auto k_ptr = [&](int buf, int kb) __attribute__((always_inline)) {
    uint32_t ka = lds0 + buf * P4_KTILE + kb * K_UNIT + k_lane;
    asm volatile("" : "+v"(ka));  // one lane-constant VGPR + immediates for every read of the unit
    return (const lds_char*)(uintptr_t)ka;
  };
  auto v_ptr = [&](int buf, int kb) __attribute__((always_inline)) {
    uint32_t va = lds0 + P4_NBUF * P4_KTILE + buf * P4_VTILE + kb * V_UNIT + v_lane;
    asm volatile("" : "+v"(va));
    return (const lds_char*)(uintptr_t)va;
  };

  // fragment rings: LDS reads run P4_RA fragments ahead of their MFMAs (lgkmcnt is a 4-bit counter: K reads + V
  // preloads + staging writes in flight must stay below 16)
  constexpr int RA = P4_RA, RING = P4_RA + 1;
  s16x8 kfr[RING], vfr[RING];
#ifdef P4_TS_SLOT
  unsigned long long ts_a = 0, ts_k = 0, ts_e = 0;
#endif

  // One micro-op of the stream (P4Op above).  The ops that touch the asm-owned state are asm statements; the others end in
  // an empty asm anchor: hipcc's IR passes otherwise sink them to their first use, out of their shadow.  REPLAY (the rare
  // path re-running a unit's stream from its logits): the same ops without the folds and the checks.
  float me0 = 0.f, me1 = 0.f, mp[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  auto micro = [&](const f32x16(&sc_)[2], auto g_c, auto par_c, auto replay_c) __attribute__((always_inline)) {
    constexpr P4Op op = P4S.op[decltype(g_c)::value];
    constexpr int kind = op.kind;
    constexpr int PAR = decltype(par_c)::value;  // parity of the unit: which set of keys-0-15 words it writes
    constexpr bool REPLAY = decltype(replay_c)::value;
    if constexpr (kind == P4_FOLD) {
      if constexpr (!REPLAY) {  // the previous unit's sums: not part of a replay
        constexpr int qb = op.arg >> 1;
        if constexpr ((op.arg & 1) == 0)
          asm volatile("v_add_f32 v%c0, v%c0, v%c1" ::"i"(P4_VL + qb), "i"(P4_VLU + qb));
        else
          asm volatile("v_add_f32 v%c0, v%c0, v%c1" ::"i"(P4_VL2 + qb), "i"(P4_VLU2 + qb));
      }
    } else if constexpr (kind == P4_CHK) {
      if constexpr (!REPLAY) {  // a replayed unit is not checked again
        constexpr int st = op.arg;
        auto or3w = [&](uint32_t& d, auto j0_c) __attribute__((always_inline)) {  // d = w[j0] | w[j0+1] | w[j0+2]
          constexpr int j0 = decltype(j0_c)::value;
          asm volatile("v_or3_b32 %0, v%c1, v%c2, v%c3"
                       : "=v"(d)
                       : "i"(p4_word_reg(PAR, j0)), "i"(p4_word_reg(PAR, j0 + 1)), "i"(p4_word_reg(PAR, j0 + 2)));
        };
        if constexpr (st == 0) {
          or3w(chk_a, std::integral_constant<int, 0>{});
        } else if constexpr (st == 1) {
          or3w(chk_b, std::integral_constant<int, 3>{});
        } else if constexpr (st == 2) {
          or3w(chk_c, std::integral_constant<int, 6>{});
        } else if constexpr (st == 3) {
          or3w(chk_d, std::integral_constant<int, 9>{});
        } else if constexpr (st == 4) {
          or3w(chk_e, std::integral_constant<int, 12>{});
        } else if constexpr (st == 5) {
          asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(chk_a) : "v"(chk_b), "v"(chk_c));
        } else if constexpr (st == 6) {
          asm volatile("v_or3_b32 %0, %0, %1, v%c2" : "+v"(chk_d) : "v"(chk_e), "i"(p4_word_reg(PAR, 15)));
        } else if constexpr (st == 7) {  // (a | d) & 0x40004000 in one v_bitop3 (truth table (S0 | S1) & S2 = 0xa8)
          asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xa8" : "+v"(chk_a) : "v"(chk_d), "s"(0x40004000u));
        } else {
          chk_bal = __builtin_amdgcn_ballot_w64(chk_a != 0);
          asm volatile("" : "+s"(chk_bal));
        }
      }
    } else {
      constexpr int j = op.arg;
      constexpr int s2 = j >> 3, qb = (j >> 2) & 1, jp = j & 3;
      if constexpr (kind == P4_F0) {
        asm volatile("v_fma_f32 %0, %1, %2, v%c3" : "=v"(me0) : "v"(sc_[qb][8 * s2 + 2 * jp]), "s"(scale_log2e), "i"(P4_VNM + qb));
      } else if constexpr (kind == P4_F1) {
        asm volatile("v_fma_f32 %0, %1, %2, v%c3"
                     : "=v"(me1)
                     : "v"(sc_[qb][8 * s2 + 2 * jp + 1]), "s"(scale_log2e), "i"(P4_VNM + qb));
      } else if constexpr (kind == P4_E0) {
        // asm like its neighbours: between a compiler-emitted v_exp and an asm statement that reads its result hipcc's
        // hazard recogniser counts the asm statements in between as no wait states and pads with s_nop (7 per unit); the
        // stream order itself keeps three instructions between an exp2 and its first reader
        asm volatile("v_exp_f32 %0, %1" : "=v"(mp[j & 1][0]) : "v"(me0));
      } else if constexpr (kind == P4_E1) {
        asm volatile("v_exp_f32 %0, %1" : "=v"(mp[j & 1][1]) : "v"(me1));
      } else if constexpr (kind == P4_A0) {  // the row sum is taken from the fp32 probabilities ...
        if constexpr (s2 == 0 && jp == 0)    // (the unit's first item of this query block starts its sums)
          asm volatile("v_mov_b32 v%c0, %1" ::"i"(P4_VLU + qb), "v"(mp[j & 1][0]));
        else
          asm volatile("v_add_f32 v%c0, v%c0, %1" ::"i"(P4_VLU + qb), "v"(mp[j & 1][0]));
      } else if constexpr (kind == P4_A1) {
        if constexpr (s2 == 0 && jp == 0)
          asm volatile("v_mov_b32 v%c0, %1" ::"i"(P4_VLU2 + qb), "v"(mp[j & 1][1]));
        else
          asm volatile("v_add_f32 v%c0, v%c0, %1" ::"i"(P4_VLU2 + qb), "v"(mp[j & 1][1]));
      } else {  // ... P itself is rounded to the model dtype, round to nearest even (reference :398-400)
        if constexpr (std::is_same<T, BF16>::value)
          asm volatile("v_cvt_pk_bf16_f32 v%c0, %1, %2" ::"i"(p4_word_reg(PAR, j)), "v"(mp[j & 1][0]), "v"(mp[j & 1][1]));
        else
          asm volatile("v_cvt_pk_f16_f32 v%c0, %1, %2" ::"i"(p4_word_reg(PAR, j)), "v"(mp[j & 1][0]), "v"(mp[j & 1][1]));
      }
    }
  };
  // RARE: a check found a probability >= 2 in the unit whose logits are sc_ (or this is the sequence's first unit).
  // Exact row max of the unit (this lane's 16 keys, then the partner lane l ^ 32), and for every row that has climbed:
  // reference point = row max + P4_BIAS, O and l scaled once by 2^(old - new).  Rows that have not climbed keep
  // everything.  The caller replays the unit's stream afterwards.  A fully masked row has row max -inf: never taken.
  auto row_fix = [&](const f32x16(&sc_)[2], auto first_c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value;  // the sequence's first unit: O and l are still zero, nothing to scale
    if constexpr (!FIRST) acc_settle();
    static_for<0, 2>([&](auto qb_c) __attribute__((always_inline)) {
      constexpr int qb = decltype(qb_c)::value;
      const f32x16& a = sc_[qb];
      float a0 = a[0], a3 = a[3];
      asm volatile("" : "+v"(a0), "+v"(a3));  // keeps the max chain inside this (cold) block: hipcc otherwise speculates it
      float x0 = max3(a0, a[1], a[2]), x1 = max3(a3, a[4], a[5]);
      x0 = max3(x0, a[6], a[7]), x1 = max3(x1, a[8], a[9]);
      x0 = max3(x0, a[10], a[11]), x1 = max3(x1, a[12], a[13]);
      const float mx = max3(x0, x1, fmaxf(a[14], a[15]));
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      const float mxs = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * scale_log2e;
      float nm_old;  // minus the old reference point
      asm volatile("v_mov_b32 %0, v%c1" : "=v"(nm_old) : "i"(P4_VNM + qb));
      const bool take = mxs + nm_old > 0.5f;
      const float nm_new = take ? -(mxs + P4_BIAS) : nm_old;
      const float alpha = take ? __builtin_amdgcn_exp2f(nm_new - nm_old) : 1.f;  // 2^(old point - new point)
      asm volatile("v_mov_b32 v%c0, %1\n\tv_mul_f32 v%c2, v%c2, %3\n\tv_mul_f32 v%c4, v%c4, %3" ::"i"(P4_VNM + qb),
                   "v"(nm_new), "i"(P4_VL + qb), "v"(alpha), "i"(P4_VL2 + qb));
      if constexpr (!FIRST)
        static_for<64 * qb, 64 * qb + 64>([&](auto r_c) __attribute__((always_inline)) {
          constexpr int R = P4_ACC0 + decltype(r_c)::value;
          acc_write<R>(acc_read<R>() * alpha);
        });
    });
    asm volatile("s_nop 3" ::: "memory");  // v_accvgpr_write -> MFMA reads the register as its accumulator
  };

  // Which 32-key units need the mask (workgroup-uniform), as two scalar thresholds on the unit index u = 2 t + kb:
  // cached units from the first one that is not full (u >= Lc / 32) and appended units from the first one that reaches
  // past the tile's first query (32 ua + 31 > m0  <=>  ua >= (m0 + 1) / 32; a short last unit lies behind that one).
  const int u_cmask = Lc / 32, u_cend = 2 * ntc, u_amask = 2 * ntc + (m0 + 1) / 32;
  // The units are visited in order, so ONE running threshold serves both intervals ([u_cmask, u_cend) and [u_amask, ..)):
  // a unit needs the mask iff its index has reached mask_lo; the last masked cached unit moves mask_lo on to u_amask
  // (inside the rare branch).  Hot path per unit: one compare and one branch.
  int mask_lo = u_cmask < u_cend ? u_cmask : u_amask;
  auto unit_mask = [&](int t, int kb, f32x16(&sc_)[2]) __attribute__((always_inline)) {
    const bool cached = t < ntc;
    const int j0 = (cached ? t * PF_KT : (t - ntc) * PF_KT) + 32 * kb;
    const int count = (cached ? Lc : la_vis) - j0;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int lim = (cached ? count : min(count, tok[qb] - j0 + 1)) - 4 * h;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kk = (i & 3) + 8 * (i >> 2);
        const uint32_t vis = (uint32_t)((kk - lim) >> 31);  // ~0 = visible
        sc_[qb][i] = __uint_as_float((__float_as_uint(sc_[qb][i]) & vis) | (0xff800000u & ~vis));
      }
    }
  };

  // The stream of the unit (parity PAR) whose logits are sc_: the ops of stream shadow SL.
  auto stream_ops = [&](const f32x16(&sc_)[2], auto sl_c, auto par_c) __attribute__((always_inline)) {
    constexpr int SL = decltype(sl_c)::value;
#ifdef P4_NO_STREAM  // timing-only A/B build (wrong results): the loop without its softmax work
    if constexpr (SL == 0) asm volatile("" ::"v"(sc_[0]), "v"(sc_[1]));
#else
    static_for<P4S.first[SL], P4S.first[SL + 1]>(
        [&](auto g_c) __attribute__((always_inline)) { micro(sc_, g_c, par_c, std::false_type{}); });
#endif
  };
  // (rare) the check fired: new reference points, then the unit's whole stream once more from its logits (without the
  // four FOLD ops, which belong to the previous unit's sums, and without the check)
  auto fix_and_replay = [&](const f32x16(&sc_)[2], auto par_c) __attribute__((always_inline)) {
    row_fix(sc_, std::false_type{});
    static_for<4, P4_NOPS>([&](auto g_c) __attribute__((always_inline)) { micro(sc_, g_c, par_c, std::true_type{}); });
    asm volatile("s_nop 1" ::: "memory");  // VALU-written packed words -> (asm) MFMA operand
  };
  // phase A of a unit of parity U (shadows 0-15): S(next) = K[next unit] Q^T beside the second half of the unit's stream.
  // Entry: kfr[0..RA-1] hold the first k-steps of the next unit's K block.  Exit: vfr[0..RA-1] hold this unit's first V
  // fragments.
  auto phase_a = [&](auto u_c, const lds_char* kbp, const lds_char* vbp, const f32x16(&sc_)[2], f32x16(&sn_)[2],
                     unsigned long long need_mask, auto extra) __attribute__((always_inline)) {
    static_for<0, 16>([&](auto kk_c) __attribute__((always_inline)) {
      constexpr int kk = decltype(kk_c)::value;
      constexpr int s = kk >> 1, qb = kk & 1;
      P4_SLOT_STAMP(32 * decltype(u_c)::value + kk);
      extra(kk_c);  // (the staging work of the shadow first: a DMA issued behind the shadow's LDS reads costs ~4 cycles more)
      if (qb == 0 && s + RA < KS) kfr[(s + RA) % RING] = k_read(kbp, s + RA);
      if (s == 0) {
        f32x16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0.f;
        sn_[qb] = mfma32<T>(kfr[0], qf[qb][0], z);
      } else {
        sn_[qb] = mfma32<T>(kfr[s % RING], qf[qb][s], sn_[qb]);
      }
      stream_ops(sc_, std::integral_constant<int, 16 + kk>{}, u_c);
      if constexpr (kk == 15) {  // the branch condition of the phase boundary, in the shadow the stream leaves empty
        rare_flag = chk_bal | need_mask;
        asm volatile("" : "+s"(rare_flag));
      }
      if constexpr ((kk & 1) && kk >= 17 - 2 * RA) vfr[(kk - (17 - 2 * RA)) / 2] = v_read(vbp, (kk - (17 - 2 * RA)) / 2);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // phase B of a unit of parity U (shadows 16-31): first the unit's check (its stream is complete), then the mask of the
  // next unit where it needs one; O^T += V[unit]^T P^T beside the first half of the NEXT unit's stream, whose logits
  // are in sn_ ((tn, kbn) = that unit).
  // Exit: kfr[0..RA-1] hold the first k-steps of the K block at kbp_next (the unit after the next).
  auto phase_b = [&](auto u_c, const lds_char* vbp, const lds_char* kbp_next, const f32x16(&sc_)[2], f32x16(&sn_)[2],
                     int tn, int kbn, unsigned long long need_mask, auto extra) __attribute__((always_inline)) {
    constexpr int U = decltype(u_c)::value;
    if (__builtin_expect(rare_flag != 0, 0)) {  // one compare and one branch on the common path for both
      if (chk_bal != 0) fix_and_replay(sc_, u_c);
      if (need_mask != 0) {
        unit_mask(tn, kbn, sn_);
        if (mask_lo < u_cend && 2 * tn + kbn + 1 >= u_cend) mask_lo = u_amask;
      }
    }
    static_for<0, 16>([&](auto kk_c) __attribute__((always_inline)) {
      constexpr int kk = decltype(kk_c)::value;
      constexpr int i = kk >> 1, qb = kk & 1;
      P4_SLOT_STAMP(32 * U + 16 + kk);
      extra(kk_c);  // (the staging work of the shadow first: a DMA issued behind the shadow's LDS reads costs ~4 cycles more)
      if (qb == 0 && i + RA < 8) vfr[(i + RA) % RING] = v_read(vbp, i + RA);
      pv_mfma_w<T, P4_ACC0 + 16 * (4 * qb + (i & 3)), (kk < 8 ? P4_VPW0 + 8 * U : P4_VPW1) + 4 * qb>(vfr[i % RING]);
      stream_ops(sn_, kk_c, std::integral_constant<int, 1 - U>{});
      // the next phase A's first RA K fragments, one per odd shadow up to the last one
      if constexpr ((kk & 1) && kk >= 17 - 2 * RA) kfr[(kk - (17 - 2 * RA)) / 2] = k_read(kbp_next, (kk - (17 - 2 * RA)) / 2);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // unit index >= mask_lo as an all-ones / zero SGPR pair, by scalar instructions (written in C the compare lands on
  // the VALU, in front of the branch that waits for it)
  auto unit_needs_mask = [&](int u) __attribute__((always_inline)) {
    unsigned long long m;
    asm volatile("s_cmp_ge_i32 %1, %2\n\ts_cselect_b64 %0, -1, 0" : "=s"(m) : "s"(u), "s"(mask_lo) : "scc");
    return m;
  };
  auto no_extra = [](int) {};
  (void)no_extra;
  constexpr std::integral_constant<int, 0> U0{};
  constexpr std::integral_constant<int, 1> U1{};

#if P4_DMA
  // ---- prologue: tiles 0, 1 and 2 on their way to LDS; S(0, keys 0-31) and its bookkeeping unpipelined ----------------
  // (An out-of-range lane of a DMA writes ZEROS to its LDS bytes - tools/dbg/probe_dma.cpp - so every byte a fragment read
  // takes has been written by a DMA of its tile: rows past the end of a tile are zero rows, as with register staging.)
  // waits for the wave's own DMAs are counted by hand (the instructions are inline asm); `left` = DMAs that may stay in
  // flight.  The barrier that follows makes every wave's blocks of the awaited tile visible to all.
  auto dma_tile = [&](int buf) __attribute__((always_inline)) {
    const uint32_t dk = dma_dst_k(buf), dv = dma_dst_v(buf);
    static_for<0, 4>([&](auto j_c) __attribute__((always_inline)) { dma_k(dk, j_c, std::false_type{}); });
    static_for<0, 4>([&](auto j_c) __attribute__((always_inline)) { dma_v(dv, j_c, std::false_type{}); });
  };
  tile_desc(0);
  dma_tile(0);
  tile_desc(1);
  dma_tile(1);
  tile_desc(2);
  dma_tile(2);
  asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // tiles 0 and 1 have landed
#else
  // ---- prologue: tiles 0 and 1 staged, tiles 2 and 3 in flight; S(0, keys 0-31) and its bookkeeping unpipelined ----
  // A tile's loads are issued two iterations before its LDS writes (two register sets): every workgroup of a
  // kv-head walks the same K/V stream at the same pace, so each fetch is a miss somewhere and all of them wait for it.
  tile_desc(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) gload_piece(0, i);
  tile_desc(1);
#pragma unroll
  for (int i = 0; i < 8; ++i) gload_piece(1, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) lstore_piece(0, 0, i);
  tile_desc(2);
#pragma unroll
  for (int i = 0; i < 8; ++i) gload_piece(0, i);
  lds_barrier();
#endif
  f32x16 sX[2], sY[2];
  {
    const lds_char* kbp = k_ptr(0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) sX[0][i] = 0.f, sX[1][i] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const s16x8 a0 = k_read(kbp, s);
      sX[0] = mfma32<T>(a0, qf[0][s], sX[0]);
      sX[1] = mfma32<T>(a0, qf[1][s], sX[1]);
    }
    if (0 >= mask_lo) {
      unit_mask(0, 0, sX);
      if (mask_lo < u_cend && 1 >= u_cend) mask_lo = u_amask;
    }
    // the first unit sets the reference points (O and l are still zero: nothing else moves; every query sees key 0 of
    // its sequence, so every row takes), then runs the first half of its stream outside any MFMA shadow
    row_fix(sX, std::true_type{});
    static_for<0, 16>([&](auto sl_c) __attribute__((always_inline)) { stream_ops(sX, sl_c, std::integral_constant<int, 0>{}); });
    const lds_char* kb1 = k_ptr(0, 1);
#pragma unroll
    for (int f = 0; f < RA; ++f) kfr[f] = k_read(kb1, f);
  }
#if P4_DMA
  // one tile = two units; cur = t % 4 = the LDS buffer of tile t.  The DMAs of tile t+3 (buffer (t+3) % 4, which tile t-1
  // left at the last barrier) are issued in the first unit - K blocks in phase A, V blocks in phase B - and the tile ends
  // with the counted wait that retires tile t+2 (its eight DMAs per wave were issued a tile ago) in front of the barrier.
  auto tile_body = [&](int t, int cur, auto app_c) __attribute__((always_inline)) {
    constexpr bool APP = decltype(app_c)::value;  // every tile loaded from here on is an appended one
    const int nxt = (cur + 1) & 3;   // (t + 1) % 4
    const int nxt3 = (cur + 3) & 3;  // (t + 3) % 4
#ifdef P4_TS_SLOT
    P4_STAMP(ts_a);
#endif
    // unit (t, keys 0-31): logits in sX; the next unit is (t, keys 32-63), then (t + 1, keys 0-31)
    {
      const lds_char* vbp = v_ptr(cur, 0);
      const unsigned long long nm0 = unit_needs_mask(2 * t + 1);  // the unit after this one (see phase_b)
      const uint32_t dk = dma_dst_k(nxt3), dv = dma_dst_v(nxt3);
      phase_a(U0, k_ptr(cur, 1), vbp, sX, sY, nm0, [&](auto kk_c) __attribute__((always_inline)) {
        constexpr int kk = decltype(kk_c)::value;
        if constexpr (kk == 0) {
          if constexpr (APP)
            tile_desc_app();
          else
            tile_desc(t + 3);
        }
        if constexpr ((kk & 3) == 2) dma_m0(dk, std::integral_constant<int, (kk >> 2)>{});
        if constexpr ((kk & 3) == 3) dma_k(dk, std::integral_constant<int, (kk >> 2)>{}, app_c);
      });
      phase_b(U0, vbp, k_ptr(nxt, 0), sX, sY, t, 1, nm0, [&](auto kk_c) __attribute__((always_inline)) {
        constexpr int kk = decltype(kk_c)::value;
        if constexpr ((kk & 3) == 2) dma_m0(dv, std::integral_constant<int, (kk >> 2)>{});
        if constexpr ((kk & 3) == 3) dma_v(dv, std::integral_constant<int, (kk >> 2)>{}, app_c);
      });
    }
    // unit (t, keys 32-63): logits in sY.  On the last tile S(t + 1) is computed from a stale K tile and never used.
    {
      const lds_char* vbp = v_ptr(cur, 1);
      const unsigned long long nm1 = unit_needs_mask(2 * t + 2);
      phase_a(U1, k_ptr(nxt, 0), vbp, sY, sX, nm1, [](auto) {});
      phase_b(U1, vbp, k_ptr(nxt, 1), sY, sX, t + 1, 0, nm1, [](auto) {});
    }
    P4_SLOT_STAMP(64);
#ifdef P4_NO_BARRIER  // timing-only A/B build (results may be wrong): what the one barrier per tile costs
    asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#ifdef P4_TS_SLOT
    P4_STAMP(ts_e);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // s_memtime lands in its SGPR pair asynchronously: the pairs must stay reserved until the wait above, in EVERY copy
    // of the tile body (a dead stamp's pair would otherwise be reused at once and overwritten late by the counter)
    asm volatile("" ::"s"(ts_a), "s"(ts_k), "s"(ts_e));
    if (t == 16 && tid == P4_TS_WAVE * 64 && bid < 8192) {
      g_p4_slot[bid * 4 + 0] = ts_a;
      g_p4_slot[bid * 4 + 1] = ts_k;
      g_p4_slot[bid * 4 + 2] = ts_e;
      g_p4_slot[bid * 4 + 3] = ntiles;
    }
#endif
  };
  int cur = 0;  // t % 4
  PF_RT(1);
  int t = 0;
  // tiles whose DMA (tile t + 3) may still be a cached tile: the generic descriptor step
  for (; t < ntiles && t + 3 < ntc; ++t) {
    tile_body(t, cur, std::false_type{});
    cur = (cur + 1) & 3;
  }
  // t + 3 >= ntc from here on: only appended tiles (or nothing) are left to load
  for (; t < ntiles; ++t) {
    tile_body(t, cur, std::true_type{});
    cur = (cur + 1) & 3;
  }
  PF_RT(2);
  // no DMA may land in the O staging image: the last tiles' (empty) DMAs are retired, and every wave has passed the last
  // tile's barrier, before anything is written
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#else
  // tile 1 (set 1) to LDS; its set then receives tile 3
#pragma unroll
  for (int i = 0; i < 8; ++i) lstore_piece(1, 1, i);
  tile_desc(3);
#pragma unroll
  for (int i = 0; i < 8; ++i) gload_piece(1, i);
  lds_barrier();

  // one tile = two units.  SET (compile-time) = t & 1: the register set that holds tile t+2 on entry and receives
  // tile t+4; cur = t % 3 = the LDS buffer of tile t.
  auto tile_body = [&](int t, int cur, auto set_c, auto app_c) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    constexpr bool APP = decltype(app_c)::value;  // every tile loaded from here on is an appended one
    const int nxt = cur == 2 ? 0 : cur + 1;   // (t + 1) % 3
    const int nxt2 = cur == 0 ? 2 : cur - 1;  // (t + 2) % 3
#ifdef P4_TS_SLOT
    P4_STAMP(ts_a);
#endif
    // unit (t, keys 0-31): logits in sX; the next unit is (t, keys 32-63), then (t + 1, keys 0-31)
    {
      const lds_char* vbp = v_ptr(cur, 0);
      const unsigned long long nm0 = unit_needs_mask(2 * t + 1);  // the unit after this one (see phase_b)
      phase_a(U0, k_ptr(cur, 1), vbp, sX, sY, nm0, [&](int kk) __attribute__((always_inline)) {
        if (kk == 5) lstore_loop(SET, nxt2, 0);
        if (kk == 11) lstore_loop(SET, nxt2, 1);
      });
      phase_b(U0, vbp, k_ptr(nxt, 0), sX, sY, t, 1, nm0, [&](int kk) __attribute__((always_inline)) {
        if (kk == 5) lstore_loop(SET, nxt2, 2);
        if (kk == 11) lstore_loop(SET, nxt2, 3);
      });
    }
    // unit (t, keys 32-63): logits in sY.  On the last tile S(t + 1) is computed from a stale K tile and never used.
    // The staging rides along: the eight LDS writes of tile t+2 (buffer (t+2) % 3 = the one tile t-1 left at the last
    // barrier) are spread over the tile, two per phase - eight in one phase cost that phase 330 cycles, the LDS pipe being
    // ~60 % busy with fragment reads as it is (profiles/r03_prefill_slot_timeline.txt) - and the loads of tile t+4 into the
    // same register set follow in the last phase, each after the write of its piece.
    {
      const lds_char* vbp = v_ptr(cur, 1);
      const unsigned long long nm1 = unit_needs_mask(2 * t + 2);
      phase_a(U1, k_ptr(nxt, 0), vbp, sY, sX, nm1, [&](int kk) __attribute__((always_inline)) {
        if (kk == 5) lstore_loop(SET, nxt2, 4);
        if (kk == 11) lstore_loop(SET, nxt2, 5);
      });
      phase_b(U1, vbp, k_ptr(nxt, 1), sY, sX, t + 1, 0, nm1, [&](int kk) __attribute__((always_inline)) {
        if (kk == 0) lstore_loop(SET, nxt2, 6);
        if (kk == 2) lstore_loop(SET, nxt2, 7);
        if constexpr (APP) {
          if (kk == 0) tile_desc_app();
          if (kk & 1) gload_piece_app(SET, kk >> 1);
        } else {
          if (kk == 0) tile_desc(t + 4);
          if (kk & 1) gload_piece(SET, kk >> 1);
        }
      });
    }
    P4_SLOT_STAMP(64);
#ifndef P4_NO_BARRIER  // timing-only A/B build (wrong results): what the one barrier per tile costs
    lds_barrier();
#endif
#ifdef P4_TS_SLOT
    P4_STAMP(ts_e);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // s_memtime lands in its SGPR pair asynchronously: the pairs must stay reserved until the wait above, in EVERY copy
    // of the tile body (a dead stamp's pair would otherwise be reused at once and overwritten late by the counter)
    asm volatile("" ::"s"(ts_a), "s"(ts_k), "s"(ts_e));
    if (t == 16 && tid == P4_TS_WAVE * 64 && bid < 8192) {
      g_p4_slot[bid * 4 + 0] = ts_a;
      g_p4_slot[bid * 4 + 1] = ts_k;
      g_p4_slot[bid * 4 + 2] = ts_e;
      g_p4_slot[bid * 4 + 3] = ntiles;
    }
#endif
  };
  // Pairs of tiles, then an odd last one: with the second body under an `if` inside the loop, hipcc's wait-count
  // pass merges "set 0 loaded last" into the loop header and drains vmcnt to 0 in front of every LDS write of set 0.
  int cur = 0;  // t % 3
  PF_RT(1);
  int t = 0;
  // pairs whose loads (tiles t + 4, t + 5) may still be cached tiles: the generic descriptor step
  for (; t + 1 < ntiles && t + 4 < ntc; t += 2) {
    tile_body(t, cur, std::integral_constant<int, 0>{}, std::false_type{});
    cur = cur == 2 ? 0 : cur + 1;
    tile_body(t + 1, cur, std::integral_constant<int, 1>{}, std::false_type{});
    cur = cur == 2 ? 0 : cur + 1;
  }
  // t + 4 >= ntc from here on: only appended tiles (or nothing) are left to load
  for (; t + 1 < ntiles; t += 2) {
    tile_body(t, cur, std::integral_constant<int, 0>{}, std::true_type{});
    cur = cur == 2 ? 0 : cur + 1;
    tile_body(t + 1, cur, std::integral_constant<int, 1>{}, std::true_type{});
    cur = cur == 2 ? 0 : cur + 1;
  }
  if (t < ntiles) tile_body(t, cur, std::integral_constant<int, 0>{}, std::true_type{});
  PF_RT(2);

#endif
#ifdef P4_DECOUPLE_LOADS
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(sink[i]));
#endif
  // ---- epilogue: normalise, stage O through LDS (wave-private region), store whole rows ---------------------------
  acc_settle();
  char* ob = smem + wave * (64 * PF_OSTRIDE);
  static_for<0, 2>([&](auto qb_c) __attribute__((always_inline)) {
    constexpr int qb = decltype(qb_c)::value;
    float l_lane;
    asm volatile("v_add_f32 %0, v%c1, v%c2" : "=v"(l_lane) : "i"(P4_VL + qb), "i"(P4_VL2 + qb));
    const float l_tot = l_lane + __shfl_xor(l_lane, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    static_for<0, 16>([&](auto j_c) __attribute__((always_inline)) {
      constexpr int db = decltype(j_c)::value >> 2, i4 = decltype(j_c)::value & 3;
      constexpr int R = P4_ACC0 + 16 * (4 * qb + db) + 4 * i4;
      const int d = db * 32 + 8 * i4 + 4 * h;
      const uint32_t w0 = pack2<T>(acc_read<R>() * inv, acc_read<R + 1>() * inv);
      const uint32_t w1 = pack2<T>(acc_read<R + 2>() * inv, acc_read<R + 3>() * inv);
      *reinterpret_cast<uint2*>(ob + (qb * 32 + r) * PF_OSTRIDE + d * 2) = make_uint2(w0, w1);
    });
  });
  // the wave re-reads only what it wrote itself: LDS ops of one wave execute in order, no barrier needed
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int rr = it * 4 + lane / 16;
    const int ch = lane % 16;
    const int grow = wave * 64 + rr;
    const int ghead = g * G + grow / BM;
    const int gtok = m0 + grow % BM;
    if (gtok < La) {
      const uint4 val = *reinterpret_cast<const uint4*>(ob + rr * PF_OSTRIDE + ch * 16);
      *reinterpret_cast<uint4*>(out + ((size_t)(s0 + gtok) * HQ + ghead) * D + ch * 8) = val;
    }
  }
  PF_RT(3);
}

// CVLLM_PREFILL=8wave selects the 8-wave kernel for every shape (A/B measurements and the cross-check test)
static bool prefill_force_8wave() {
  const char* e = getenv("CVLLM_PREFILL");
  return e && e[0] == '8';
}

template <typename T, int D, int G>
static int launch_prefill(const void* q, const void* k, const void* v, int64_t sq_n, int64_t sk_n, int64_t sk_h,
                          int64_t sv_n, int64_t sv_h, const void* kc, const void* vc, void* out, const int* seq_lens,
                          const int* page_table, const int* bmap, const int* cu, int B, int max_seqlen_q, int HKV,
                          int PS, int NLP, float scale, hipStream_t st) {
  constexpr int BM = PF_ROWS / G;
  const int nqt = (max_seqlen_q + BM - 1) / BM;
  if constexpr (D == 128) {
    // 4-wave structure unless a 64-key tile would straddle pages (page size not a multiple of 64), a sequence's
    // appended block does not fit the kernel's 31-bit buffer byte counts, or CVLLM_PREFILL=8wave
    if (!prefill_force_8wave() && PS % PF_KT == 0 && sk_n * 2 * ((int64_t)max_seqlen_q + 64) < 0x7fffffffLL &&
        sv_n * 2 * ((int64_t)max_seqlen_q + 64) < 0x7fffffffLL) {
      auto kern4 = prefill_attn_w4_kernel<T, G>;
      set_dyn_lds_once(kern4, P4_SMEM);
      hipLaunchKernelGGL(kern4, dim3(nqt * B * HKV), dim3(P4_THREADS), P4_SMEM, st, (const uint16_t*)q,
                         (const uint16_t*)k, (const uint16_t*)v, sq_n, sk_n, sk_h, sv_n, sv_h, (const uint16_t*)kc,
                         (const uint16_t*)vc, (uint16_t*)out, seq_lens, page_table, bmap, cu, B, HKV, PS, NLP,
                         scale * 1.4426950408889634f);
      return check_launch();
    }
  }
  auto kern = prefill_attn_kernel<T, D, G>;
  set_dyn_lds_once(kern, PF_SMEM);
  hipLaunchKernelGGL(kern, dim3(nqt * B * HKV), dim3(PF_THREADS), PF_SMEM, st, (const uint16_t*)q, (const uint16_t*)k,
                     (const uint16_t*)v, sq_n, sk_n, sk_h, sv_n, sv_h, (const uint16_t*)kc, (const uint16_t*)vc,
                     (uint16_t*)out, seq_lens, page_table, bmap, cu, B, HKV, PS, NLP, scale * 1.4426950408889634f);
  return check_launch();
}

template <typename T, int D>
static int prefill_dispatch_g(int G, const void* q, const void* k, const void* v, int64_t sq_n, int64_t sk_n,
                              int64_t sk_h, int64_t sv_n, int64_t sv_h, const void* kc, const void* vc, void* out,
                              const int* seq_lens, const int* page_table, const int* bmap, const int* cu, int B,
                              int max_seqlen_q, int HKV, int PS, int NLP, float scale, hipStream_t st) {
#define PF_CALL(G_) \
  launch_prefill<T, D, G_>(q, k, v, sq_n, sk_n, sk_h, sv_n, sv_h, kc, vc, out, seq_lens, page_table, bmap, cu, B, \
                           max_seqlen_q, HKV, PS, NLP, scale, st)
  switch (G) {
    case 1: return PF_CALL(1);
    case 2: return PF_CALL(2);
    case 4: return PF_CALL(4);
    case 8: return PF_CALL(8);
    default: return CVLLM_ERR_SHAPE;
  }
#undef PF_CALL
}

}  // namespace cvllm

using namespace cvllm;


extern "C" int cvllm_prefill_attn(const void* q, const void* k, const void* v, int64_t sq_n, int64_t sk_n,
                                  int64_t sk_h, int64_t sv_n, int64_t sv_h, const void* k_cache, const void* v_cache,
                                  void* out, const int32_t* seq_lens_bh, const int32_t* page_table,
                                  const int32_t* batch_mapping, const int32_t* cu_seqlens_q, int B, int total_tokens,
                                  int max_seqlen_q, int HQ, int HKV, int D, int page_size, int n_logical_pages_max,
                                  float sm_scale, int dtype, cvllm_stream_t stream) {
  if (!q || !k || !v || !k_cache || !v_cache || !out || !seq_lens_bh || !page_table || !batch_mapping ||
      !cu_seqlens_q)
    return CVLLM_ERR_ARG;
  if (B <= 0 || total_tokens < 0 || HQ <= 0 || HKV <= 0 || page_size <= 0 || n_logical_pages_max <= 0)
    return CVLLM_ERR_ARG;
  if (HQ % HKV != 0) return CVLLM_ERR_SHAPE;
  if (total_tokens == 0 || max_seqlen_q <= 0) return CVLLM_OK;
  // 16-byte vector loads: every row start must be 16-byte aligned
  if ((sq_n % 8) || (sk_n % 8) || (sk_h % 8) || (sv_n % 8) || (sv_h % 8)) return CVLLM_ERR_SHAPE;
  if (sk_n < 0 || sv_n < 0 || sk_n > 0x7fffffff || sv_n > 0x7fffffff) return CVLLM_ERR_SHAPE;  // 32-bit row strides in the kernel
  const int G = HQ / HKV;
  hipStream_t st = (hipStream_t)stream;
#ifdef P4_TS_SLOT  // debug build: the one instantiation the stamps are read from
  if (dtype != CVLLM_BF16 || D != 128 || G != 4) return CVLLM_ERR_SHAPE;
  return launch_prefill<BF16, 128, 4>(q, k, v, sq_n, sk_n, sk_h, sv_n, sv_h, k_cache, v_cache, out, seq_lens_bh, page_table,
                                      batch_mapping, cu_seqlens_q, B, max_seqlen_q, HKV, page_size, n_logical_pages_max,
                                      sm_scale, st);
#else
#define PF_D(T_, D_)                                                                                              \
  return prefill_dispatch_g<T_, D_>(G, q, k, v, sq_n, sk_n, sk_h, sv_n, sv_h, k_cache, v_cache, out, seq_lens_bh, \
                                    page_table, batch_mapping, cu_seqlens_q, B, max_seqlen_q, HKV, page_size,     \
                                    n_logical_pages_max, sm_scale, st)
  if (dtype == CVLLM_F16) {
    if (D == 128) PF_D(F16, 128);
    if (D == 64) PF_D(F16, 64);
  } else if (dtype == CVLLM_BF16) {
    if (D == 128) PF_D(BF16, 128);
    if (D == 64) PF_D(BF16, 64);
  }
#undef PF_D
  return CVLLM_ERR_SHAPE;
#endif
}

#ifdef P4_TS_SLOT
extern "C" void cvllm_debug_prefill_slot_stamps(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(cvllm::g_p4_slot), sizeof(unsigned long long) * 8192 * 4);
}
#endif
#ifdef CVLLM_PF_TS
extern "C" void cvllm_debug_prefill_stamps(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(cvllm::g_pf_rt), sizeof(unsigned long long) * 8192 * 8);
}
#endif
