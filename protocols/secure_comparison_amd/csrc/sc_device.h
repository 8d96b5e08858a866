// Device-side big-integer primitives for gfx950 (CDNA4).
//
// Number format inside a kernel ("limb form"): radix 2^29 limbs, S = G*L limbs per number, spread
// over G adjacent lanes (a "group"), lane j of the group holding limbs j*L .. j*L+L-1 in VGPRs.
// A 64-lane wave therefore works on 64/G numbers at once.  Why 29-bit limbs: on gfx950
// v_mad_u64_u32 issues every 4 cycles per wave64 and nearly every other integer instruction (add
// with carry, DPP move, 64-bit shift) costs about as much (tools/microbench/probe2.hip), so the
// inner loop must be (almost) nothing but v_mad_u64_u32.  With 29-bit limbs a 64-bit column
// accumulator absorbs 2L <= 54 products (54 * 2^58 < 2^64) with NO carry handling; carries are
// resolved once per column when it leaves the lane.
//
// Montgomery form: R = 2^(29*S) >= 2^8 * n for every modulus a configuration accepts (2^40 * n for the
// 2048/4096-bit cases), so operands may stay in [0, 2n) between multiplications -- even [0, 4n) after a lazy
// addition -- with no conditional subtraction inside exponentiation loops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Every kernel of the library runs 64-thread workgroups, i.e. ONE wave: the lanes of a wave execute LDS instructions in
// program order, so a compiler-level wave barrier would be enough to order the staging writes before the reads.  Measured on the
// MI355X (-DSC_WAVE_BARRIER_ONLY, tools/gpu_kernel_rates.py): no difference to __syncthreads() in any launch group -- an s_barrier
// of a one-wave workgroup costs nothing -- so the hardware barrier stays.
#ifdef SC_WAVE_BARRIER_ONLY
#define SC_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define SC_WAVE_SYNC() __syncthreads()
#endif

namespace sc {

constexpr int W_DEFAULT = 29;  // limb width of the standard configurations (a 28-bit family exists for L = 37)

// ---------------------------------------------------------------------------------------------
// cross-lane helpers (DPP; a group never straddles a 16-lane DPP row because G divides 16)
// ---------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ uint32_t bcast0(uint32_t v) {  // value of the group's lane 0
  if constexpr (G == 1) {
    return v;
  } else if constexpr (G == 2) {
    return __builtin_amdgcn_update_dpp(0u, v, 0xA0, 0xf, 0xf, true);  // quad_perm:[0,0,2,2]; bound_ctrl so that a following mask can fold in
  } else if constexpr (G == 4) {
    return __builtin_amdgcn_update_dpp(0u, v, 0x00, 0xf, 0xf, true);  // quad_perm:[0,0,0,0]
  } else if constexpr (G == 16) {
    return __builtin_amdgcn_update_dpp(0u, v, 0x150, 0xf, 0xf, true);  // row_newbcast:0
  } else {
    static_assert(G == 8, "unsupported group size");
    // two DPP moves; a single ds_swizzle (LDS crossbar, no VALU slot) was measured 4 % slower: its latency sits on
    // the quotient-digit critical path
    uint32_t lo = __builtin_amdgcn_update_dpp(v, v, 0x150, 0xf, 0x3, false);   // lanes 0-7  <- lane 0 (others keep v for now)
    return __builtin_amdgcn_update_dpp(lo, v, 0x158, 0xf, 0xc, false);         // lanes 8-15 <- lane 8
  }
}
// lane i receives lane i+1 of its 16-lane row (row end: 0)
__device__ __forceinline__ uint32_t row_from_above(uint32_t v) {
  return __builtin_amdgcn_mov_dpp(v, 0x101, 0xf, 0xf, true);  // row_shl:1 (bound_ctrl: out-of-row source reads 0)
}
// lane i receives lane i-1 of its 16-lane row (row start: 0)
__device__ __forceinline__ uint32_t row_from_below(uint32_t v) {
  return __builtin_amdgcn_mov_dpp(v, 0x111, 0xf, 0xf, true);  // row_shr:1
}

// ---------------------------------------------------------------------------------------------
// Per-thread view of the group it belongs to.
// ---------------------------------------------------------------------------------------------
// NEG1: the modulus satisfies n = -1 (mod 2^W), i.e. -n^-1 = 1: the Montgomery quotient digit is the low limb of the running sum
// itself and the v_mul_lo_u32 of every limb step disappears from the dependent chain.  An arbitrary odd n gets there through the
// multiple M = (-n^-1 mod 2^W) n, whose residues reduce to residues modulo n (sc_lib.hip::neg1_twin).
template <int G_, int L_, int W_ = W_DEFAULT, bool NEG1_ = false>
struct Grp {
  static constexpr bool NEG1 = NEG1_;
  static constexpr int G = G_, L = L_, S = G_ * L_, NG = 64 / G_;
  static constexpr int W = W_;                          // bits per limb: 2L products of 2W bits must fit 64 bits
  static constexpr uint32_t LMASK = (1u << W_) - 1;
  static_assert(2 * W_ + 6 <= 64 && (2 * L_ + 2) <= (1 << (64 - 2 * W_ - 1)) * 2, "column accumulators would overflow");
  static constexpr int SP = (S + 3) | 1;                // padded limb-array stride in LDS (odd: conflict-free across groups)
  static constexpr int WP = (W * S + 31) / 32 + 2;      // 32-bit-word scratch stride in LDS
  int lane, g, j;                                       // lane in wave, group in wave, lane in group
  uint32_t notTop, notBot;                              // 0 for the top / bottom lane of the group
  uint32_t n[L];                                        // modulus limbs of this lane
  uint32_t n0inv;                                       // -n^-1 mod 2^29
  uint32_t lmask_v;                                     // LMASK held in a VGPR the optimiser cannot see through: lets the masks
                                                        // that follow a DPP move fold into one v_and_b32_dpp (no literal allowed there)

  __device__ __forceinline__ void init(const uint32_t* __restrict__ n_limbs, uint32_t n0inv_) {
    lane = threadIdx.x & 63;
    g = lane / G;
    j = lane % G;
    notTop = (j == G - 1) ? 0u : ~0u;
    notBot = (j == 0) ? 0u : ~0u;
    n0inv = n0inv_;
    uint32_t m = LMASK;
    asm volatile("" : "+v"(m));
    lmask_v = m;
#pragma unroll
    for (int l = 0; l < L; l++) {
      n[l] = n_limbs[j * L + l];
      // one-lane numbers: every lane holds the same modulus limbs -- pin them in scalar registers (a v_mad_u64_u32 takes one
      // scalar operand), which frees L vector registers; the compiler cannot prove the load uniform and read-only by itself
      if constexpr (G == 1) n[l] = __builtin_amdgcn_readfirstlane(n[l]);
    }
  }

  __device__ __forceinline__ uint32_t from_above(uint32_t v) const {
    if constexpr (G == 1) return 0u; else return row_from_above(v) & notTop;
  }
  __device__ __forceinline__ uint32_t from_above_raw(uint32_t v) const {
    if constexpr (G == 1) return 0u; else return row_from_above(v);
  }
  __device__ __forceinline__ uint32_t from_below(uint32_t v) const {
    if constexpr (G == 1) return 0u; else return row_from_below(v) & notBot;
  }

  // -------------------------------------------------------------------------------------------
  // Montgomery product r = a * b / R mod n (lazily reduced: r < 2n, limbs < 2^29 + 2^7).
  //   a: S limbs in LDS (the group's staging area), b: this lane's L limbs.
  //   MODE 0: full product.  MODE 3: square (a == b).  MODE 1: reduction only (a ignored, computes b / R mod n).
  //   MODE 2: like 1 and additionally collects the Montgomery quotient digits into quot[] (lane k
  //           keeps digits k*L..k*L+L-1); quot = -b / n mod R, which is how exact division is done.
  // -------------------------------------------------------------------------------------------
  //   COLLECT (default: MODE 2): record the quotient digits.  INIT: the running sum starts at init[] instead of 0
  //   (MODE 1/2 always start at b).  Both are used by the pair arithmetic modulo n^2 further down.
  //   ADDN: the result gets n - 1 added inside the final carry pass (the pair arithmetic's correction for the start value
  //   R - q, see pair_fix) instead of by a separate lazy addition and re-normalisation.
  template <int MODE, bool COLLECT = (MODE == 2), bool INIT = false, bool PRELOAD_A = true, bool ADDN = false>
  __device__ __forceinline__ void mont(uint32_t (&r)[L], const uint32_t* a_lds, const uint32_t (&b)[L],
                                       uint32_t (&quot)[L], const uint32_t* a2_lds = nullptr,
                                       const uint32_t* init = nullptr) const {
    uint64_t T[L];
#pragma unroll
    for (int i = 0; i < L; i++) T[i] = (MODE == 1 || MODE == 2) ? (uint64_t)b[i] : (INIT ? (uint64_t)init[i] : 0ull);
#pragma unroll 1
    for (int k = 0; k < G; k++) {
      // L <= 18: fetch the whole block of a-limbs up front (registers to spare).  Larger L: fetch limb by limb so the
      // block does not pin another L registers (the L = 27 configurations otherwise spill into scratch).
      constexpr bool PRELOAD = (L <= 18) && PRELOAD_A;
      uint32_t av[PRELOAD ? L : 1];
      if constexpr (MODE == 0 && PRELOAD) {
#pragma unroll
        for (int l = 0; l < L; l++) av[l] = a_lds[k * L + l];
      }
      // MODE 3 (squaring, a == b): the L x L block a_k (x) a_j and its mirror a_j (x) a_k (computed by lane k at block
      // step j) hold the same products, so every lane takes only the entries (l, c) with c > l of each of its blocks,
      // doubled, plus the block diagonal c == l once: a diagonal entry of an off-diagonal block is then counted once
      // here and once in the mirror block (= twice), a true square term a_i^2 (k == j) exactly once.  Every product
      // still reaches its column before that column is consumed (it is added at a row <= its column index), the
      // instruction stream is identical in all lanes, and the a*a part costs L(L+1)/2 instead of L^2 multiply-adds per
      // block.  The doubled limbs 2*a_i are staged in LDS next to a_i (a2_lds), so the loop has no extra VALU work.
      // Column bound: a column lives L limb steps in a lane; it takes a q*n product (< 2^58) at each of them and a
      // doubled a*a product (< 2^59) only while its position is >= the row index, i.e. at <= L/2 + 1 of them:
      // (L + 2) 2^58 + L 2^58 <= 56 * 2^58 < 2^64 for L <= 27 -- the same bound as the general product (2L * 2^58).
#pragma unroll
      for (int l = 0; l < L; l++) {
        if constexpr (MODE == 0) {
          const uint32_t ai = PRELOAD ? av[PRELOAD ? l : 0] : a_lds[k * L + l];
#pragma unroll
          for (int c = 0; c < L; c++) T[(l + c) % L] += (uint64_t)ai * b[c];
        }
        if constexpr (MODE == 3) {
          const uint32_t ai = a_lds[k * L + l], ai2 = a2_lds[k * L + l];
          T[(l + l) % L] += (uint64_t)ai * b[l];
#pragma unroll
          for (int c = l + 1; c < L; c++) T[(l + c) % L] += (uint64_t)ai2 * b[c];
        }
        const uint32_t q = bcast0<G>(NEG1 ? (uint32_t)T[l] : (uint32_t)T[l] * n0inv) & lmask_v;   // broadcast first: the mask folds into the DPP op
        if constexpr (COLLECT) quot[l] = (j == k) ? q : quot[l];
#pragma unroll
        for (int c = 0; c < L; c++) T[(l + c) % L] += (uint64_t)q * n[c];
        const uint64_t t0 = T[l];                 // column 0: its low 29 bits are 0 in lane 0
        T[(l + 1) % L] += t0 >> W;                // carry into column 1
        // The low limb moves to the lane below and becomes its new top column.  No group-boundary mask is needed:
        // the bottom lane of the group above contributes exactly 0 (its column 0 was just made divisible by 2^29),
        // and the DPP row end reads 0 (bound_ctrl).
        T[l] = (uint64_t)(from_above_raw((uint32_t)t0) & lmask_v);  // mask after the move: folds into v_and_b32_dpp
      }
    }
    // one local carry pass + hand the lane carry to the next lane (result "almost normalised")
    uint64_t c = ADDN ? (uint64_t)(n[0] - ((j == 0) ? 1u : 0u)) : 0ull;   // n is odd: limb 0 of lane 0 is >= 1
#pragma unroll
    for (int l = 0; l < L; l++) {
      const uint64_t v = T[l] + c + ((ADDN && l > 0) ? (uint64_t)n[l] : 0ull);
      r[l] = (uint32_t)v & LMASK;
      c = v >> W;
    }
    if constexpr (G > 1) {
      const uint32_t clo = from_below((uint32_t)c), chi = from_below((uint32_t)(c >> 32));
      const uint64_t v = (uint64_t)r[0] + (((uint64_t)chi << 32) | clo);
      r[0] = (uint32_t)v & LMASK;
      r[1] += (uint32_t)(v >> W);
    }
  }
  // -------------------------------------------------------------------------------------------
  // G == 1 ("one lane per number"): the whole number lives in this lane, so a product needs no LDS staging, no cross-lane
  // move and no quotient broadcast, and the modulus limbs are wave-uniform (the compiler keeps them in scalar registers).
  // Per limb step 2L multiply-adds + 4 other instructions, paid by ONE lane per number -- the G-lane form pays its per-step
  // bookkeeping in every lane of the group (G S steps per number instead of S).
  //   MODE 0: r = a * b / R.  MODE 3: r = b * b / R (a ignored).  DOUBLE_A: the multiplier is 2 a (second pass of a pair
  //   squaring: 2 x0 x1).  COLLECT / INIT / ADDN as in mont().  r may alias a or b (written after the last read).
  // Column bound: L reduction products plus L (possibly doubled) operand products per column: 3 L 2^(2W) <= 2^64 holds
  // for (L, W) = (37, 28) and (18, 29).
  // -------------------------------------------------------------------------------------------
  template <int MODE, bool COLLECT = false, bool INIT = false, bool ADDN = false, bool DOUBLE_A = false>
  __device__ __forceinline__ void mont_r(uint32_t (&r)[L], const uint32_t (&a)[L], const uint32_t (&b)[L], uint32_t (&quot)[L],
                                         const uint32_t (&init)[L]) const {
    static_assert(G == 1, "register-operand products exist for one-lane numbers only");
    static_assert((uint64_t)3 * L <= ((uint64_t)1 << (64 - 2 * W)), "column accumulators would overflow");   // L q n products + L doubled ones
    uint64_t T[L];
#pragma unroll
    for (int i = 0; i < L; i++) T[i] = INIT ? (uint64_t)init[i] : 0ull;
#pragma unroll
    for (int l = 0; l < L; l++) {
      if constexpr (MODE == 0) {
        const uint32_t ai = DOUBLE_A ? (a[l] << 1) : a[l];
#pragma unroll
        for (int c = 0; c < L; c++) T[(l + c) % L] += (uint64_t)ai * b[c];
      }
      if constexpr (MODE == 3) {
        const uint32_t ai = b[l], ai2 = b[l] << 1;
        T[(l + l) % L] += (uint64_t)ai * b[l];
#pragma unroll
        for (int c = l + 1; c < L; c++) T[(l + c) % L] += (uint64_t)ai2 * b[c];
      }
      const uint32_t q = ((uint32_t)T[l] * n0inv) & LMASK;
      if constexpr (COLLECT) quot[l] = q;
#pragma unroll
      for (int c = 0; c < L; c++) T[(l + c) % L] += (uint64_t)q * n[c];
      T[(l + 1) % L] += T[l] >> W;     // column 0 is now divisible by 2^W: only its carry survives
      T[l] = 0;                        // and the register becomes the new top column
    }
    uint64_t c = ADDN ? (uint64_t)(n[0] - 1u) : 0ull;   // n is odd: limb 0 is >= 1
#pragma unroll
    for (int l = 0; l < L; l++) {
      const uint64_t v = T[l] + c + ((ADDN && l > 0) ? (uint64_t)n[l] : 0ull);
      r[l] = (uint32_t)v & LMASK;
      c = v >> W;
    }
  }
  // One local carry pass plus the hand-over of the lane carry: brings limbs that grew by lazy additions (entries
  // < 2^31) back to <= 2^29 + 1, so that the accumulator bounds of mont() hold for the next product.
  __device__ __forceinline__ void renorm(uint32_t (&r)[L]) const {
    uint32_t c = 0;
#pragma unroll
    for (int l = 0; l < L; l++) {
      const uint32_t v = r[l] + c;
      r[l] = v & LMASK;
      c = v >> W;
    }
    // The group's top limb keeps its excess: a lazy sum may reach R and beyond (a raw chunk of 32 * nwords = 29 * S bits added
    // to a residue, configurations (16,14) and (16,18)); the next product reads the limb as it is (< 2^32).
    r[L - 1] += (c << W) & ~notTop;
    if constexpr (G > 1) {
      const uint32_t v = r[0] + from_below(c);
      r[0] = v & LMASK;
      r[1] += v >> W;
    }
  }
  // Two products sharing one reduction: r = (a1 * b1 + a2 * b2 + init + q n) / R.  Three products per column and limb
  // step: 3L * 2^58 < 2^64 holds for L <= 18 only (static_assert), which is where the pair arithmetic uses it.
  template <bool ADDN = false>
  __device__ __forceinline__ void mont2(uint32_t (&r)[L], const uint32_t* a1_lds, const uint32_t (&b1)[L], const uint32_t* a2_lds,
                                        const uint32_t (&b2)[L], const uint32_t (&init)[L]) const {
    uint64_t T[L];
#pragma unroll
    for (int i = 0; i < L; i++) T[i] = (uint64_t)init[i];
#pragma unroll 1
    for (int k = 0; k < G; k++) {
#pragma unroll
      for (int l = 0; l < L; l++) {
        const uint32_t a1 = a1_lds[k * L + l], a2 = a2_lds[k * L + l];
#pragma unroll
        for (int c = 0; c < L; c++) T[(l + c) % L] += (uint64_t)a1 * b1[c];
#pragma unroll
        for (int c = 0; c < L; c++) T[(l + c) % L] += (uint64_t)a2 * b2[c];
        const uint32_t q = bcast0<G>(NEG1 ? (uint32_t)T[l] : (uint32_t)T[l] * n0inv) & lmask_v;
#pragma unroll
        for (int c = 0; c < L; c++) T[(l + c) % L] += (uint64_t)q * n[c];
        const uint64_t t0 = T[l];
        T[(l + 1) % L] += t0 >> W;
        T[l] = (uint64_t)(from_above_raw((uint32_t)t0) & lmask_v);
      }
    }
    uint64_t c = ADDN ? (uint64_t)(n[0] - ((j == 0) ? 1u : 0u)) : 0ull;
#pragma unroll
    for (int l = 0; l < L; l++) {
      const uint64_t v = T[l] + c + ((ADDN && l > 0) ? (uint64_t)n[l] : 0ull);
      r[l] = (uint32_t)v & LMASK;
      c = v >> W;
    }
    if constexpr (G > 1) {
      const uint32_t clo = from_below((uint32_t)c), chi = from_below((uint32_t)(c >> 32));
      const uint64_t v = (uint64_t)r[0] + (((uint64_t)chi << 32) | clo);
      r[0] = (uint32_t)v & LMASK;
      r[1] += (uint32_t)(v >> W);
    }
  }
  __device__ __forceinline__ void mul(uint32_t (&r)[L], const uint32_t* a_lds, const uint32_t (&b)[L]) const {
    uint32_t dummy[L];
    mont<0>(r, a_lds, b, dummy);
  }
  // r = b * b / R mod n; a_lds / a2_lds must hold the limbs of b and of 2b (staged by the caller)
  __device__ __forceinline__ void sqr(uint32_t (&r)[L], const uint32_t* a_lds, const uint32_t* a2_lds, const uint32_t (&b)[L]) const {
    uint32_t dummy[L];
    mont<3>(r, a_lds, b, dummy, a2_lds);
  }
  __device__ __forceinline__ void redc(uint32_t (&r)[L], const uint32_t (&b)[L]) const {
    uint32_t dummy[L];
    mont<1>(r, nullptr, b, dummy);
  }

  // -------------------------------------------------------------------------------------------
  // Pair arithmetic modulo n^2 with products modulo n only.  An element X of Z_{n^2} is held as (x0, x1), each a
  // lazily reduced S-limb integer, with  X = (x0 + x1 n) / R  (mod n^2).  For the product of (x0,x1) and (y0,y1) the
  // x1 y1 n^2 term vanishes; x0 y0 is Montgomery-reduced with its quotient digits q recorded, x0 y0 + q n = R t
  // (an integer identity), hence x0 y0 / R = t - q n / R (mod n^2): the "overflow" of the low part is exactly -q,
  // which is folded into the n-part before ITS reduction:
  //      z0 = t,   z1 = (x0 y1 + x1 y0 - q) / R  (mod n).
  // A square costs 0.5 + 1 + 1 + 1 = 3.5 S^2 multiply-adds (S = limbs of n) instead of 6 S^2 for a direct Montgomery
  // square modulo the 2S-limb n^2; a general product 6 S^2 instead of 8 S^2.  The -q is introduced as the start value
  // R - q of the running sum (digit complement + 1), which makes the result one too large: corrected by adding n - 1.
  // Requires L <= 18 (the doubled operand of the square's second pass needs the 1.5 * 2^59 * L < 2^64 column bound).
  // -------------------------------------------------------------------------------------------
  // x1 <- x1 + n - 1 (lazy), the correction for the start value R - q
  __device__ __forceinline__ void pair_fix(uint32_t (&x1)[L]) const {
#pragma unroll
    for (int l = 0; l < L; l++) x1[l] += n[l];
    x1[0] -= (j == 0) ? 1u : 0u;  // n is odd, so limb 0 of v + n is >= 1
    renorm(x1);
  }
  // q <- digits of R - q (+ add[]): the start value that injects -q into the next reduction
  __device__ __forceinline__ void neg_quot_init(uint32_t (&q)[L]) const {
#pragma unroll
    for (int l = 0; l < L; l++) q[l] = LMASK - q[l];
    q[0] += (j == 0) ? 1u : 0u;
  }
  // (x0, x1) <- (x0, x1)^2 ;  a_lds / a2_lds hold x0 and 2 x0 (staged by the caller).  Outputs alias the inputs on purpose
  // (a result is only written after the last read of its operand) to keep the register footprint at x0, x1, q.
  __device__ __forceinline__ void pair_sqr(uint32_t (&x0)[L], uint32_t (&x1)[L], const uint32_t* a_lds, const uint32_t* a2_lds) const {
    uint32_t q[L];
#pragma unroll
    for (int l = 0; l < L; l++) q[l] = 0;
    mont<3, true, false>(x0, a_lds, x0, q, a2_lds);                  // x0 <- t = (x0^2 + q n) / R
    neg_quot_init(q);
    mont<0, false, true, false, true>(x1, a2_lds, x1, q, nullptr, q);   // x1 <- (2 x0 x1 + R - q + q' n) / R + n - 1
  }
  // The same pair squaring with its two passes INTERLEAVED limb step by limb step (small-batch configurations, L <= 9).  A small
  // batch is one dependent chain of ~2400 pair squarings per item on a chip with waves to spare: what a squaring costs there is the
  // LATENCY of its chain of limb steps (multiply-add -> quotient digit -> broadcast -> multiply-add -> carry -> hand-over, ~55
  // cycles a step), not its instruction count.  Pass 2 needs pass 1 only through the start value R - q, and digit d of q is born in
  // step d of pass 1 while position d of pass 2's running sum is consumed in ITS step d: so pass 2 runs in the shadow of pass 1,
  // one step behind inside the same loop iteration, with the complemented digit added to the bottom lane's lowest column at the
  // moment it is produced (the broadcast already brought it there) instead of to the start value.  The represented integer, the
  // quotient digits and the results are those of pair_sqr bit for bit; the two chains give the scheduler two independent
  // instruction streams per wave.
  __device__ __forceinline__ void pair_sqr_il(uint32_t (&x0)[L], uint32_t (&x1)[L], const uint32_t* a_lds, const uint32_t* a2_lds) const {
    static_assert(G > 1 && L <= 9, "interleaved pair squaring: multi-lane small-batch configurations");
    uint64_t T1[L], T2[L];
#pragma unroll
    for (int i = 0; i < L; i++) { T1[i] = 0ull; T2[i] = 0ull; }
    uint32_t first = (j == 0) ? 1u : 0u;                   // the "+ 1" of R - q = complement + 1, at digit 0
#pragma unroll 1
    for (int k = 0; k < G; k++) {
#pragma unroll
      for (int l = 0; l < L; l++) {
        const uint32_t ai = a_lds[k * L + l], ai2 = a2_lds[k * L + l];
        // ---- pass 1: t = (x0^2 + q n) / R, upper triangle doubled (mont<3>)
        T1[(l + l) % L] += (uint64_t)ai * x0[l];
#pragma unroll
        for (int c = l + 1; c < L; c++) T1[(l + c) % L] += (uint64_t)ai2 * x0[c];
        const uint32_t q1 = bcast0<G>(NEG1 ? (uint32_t)T1[l] : (uint32_t)T1[l] * n0inv) & lmask_v;
#pragma unroll
        for (int c = 0; c < L; c++) T1[(l + c) % L] += (uint64_t)q1 * n[c];
        const uint64_t t1 = T1[l];
        T1[(l + 1) % L] += t1 >> W;
        T1[l] = (uint64_t)(from_above_raw((uint32_t)t1) & lmask_v);
        // ---- pass 2: (2 x0 x1 + R - q + q' n) / R, digit d of R - q added where position d is about to be consumed
        T2[l] += (uint64_t)(((LMASK - q1) + first) & ~notBot);       // bottom lane only (notBot is 0 there)
        first = 0u;
#pragma unroll
        for (int c = 0; c < L; c++) T2[(l + c) % L] += (uint64_t)ai2 * x1[c];
        const uint32_t q2 = bcast0<G>(NEG1 ? (uint32_t)T2[l] : (uint32_t)T2[l] * n0inv) & lmask_v;
#pragma unroll
        for (int c = 0; c < L; c++) T2[(l + c) % L] += (uint64_t)q2 * n[c];
        const uint64_t t2 = T2[l];
        T2[(l + 1) % L] += t2 >> W;
        T2[l] = (uint64_t)(from_above_raw((uint32_t)t2) & lmask_v);
      }
    }
    // the two final carry passes of mont(): plain for t, with n - 1 added for the n-part (the correction for the start value R - q)
    uint64_t c1 = 0ull, c2 = (uint64_t)(n[0] - ((j == 0) ? 1u : 0u));
#pragma unroll
    for (int l = 0; l < L; l++) {
      const uint64_t v1 = T1[l] + c1;
      x0[l] = (uint32_t)v1 & LMASK;
      c1 = v1 >> W;
      const uint64_t v2 = T2[l] + c2 + ((l > 0) ? (uint64_t)n[l] : 0ull);
      x1[l] = (uint32_t)v2 & LMASK;
      c2 = v2 >> W;
    }
    {
      const uint32_t clo = from_below((uint32_t)c1), chi = from_below((uint32_t)(c1 >> 32));
      const uint64_t v = (uint64_t)x0[0] + (((uint64_t)chi << 32) | clo);
      x0[0] = (uint32_t)v & LMASK;
      x0[1] += (uint32_t)(v >> W);
    }
    {
      const uint32_t clo = from_below((uint32_t)c2), chi = from_below((uint32_t)(c2 >> 32));
      const uint64_t v = (uint64_t)x1[0] + (((uint64_t)chi << 32) | clo);
      x1[0] = (uint32_t)v & LMASK;
      x1[1] += (uint32_t)(v >> W);
    }
  }
  // (x0, x1) <- (x0, x1) * (y0, y1) ;  y0_lds / y1_lds hold the second operand
  __device__ __forceinline__ void pair_mul(uint32_t (&x0)[L], uint32_t (&x1)[L], const uint32_t* y0_lds, const uint32_t* y1_lds) const {
    uint32_t t[L], q[L];
#pragma unroll
    for (int l = 0; l < L; l++) q[l] = 0;
    mont<0, true, false, false>(t, y0_lds, x0, q);                   // t = (x0 y0 + q n) / R
    neg_quot_init(q);
    if constexpr (L <= 18) {
      mont2<true>(x1, y1_lds, x0, y0_lds, x1, q);                    // x1 <- (x0 y1 + x1 y0 + R - q + q' n) / R + n - 1, one reduction
#pragma unroll
      for (int l = 0; l < L; l++) x0[l] = t[l];
    } else {
      mont<0, false, true, false>(x0, y1_lds, x0, q, nullptr, q);    // x0 <- (x0 y1 + R - q + ..) / R
      mont<0, false, false, false>(x1, y0_lds, x1, q);               // x1 <- (x1 y0 + ..) / R
#pragma unroll
      for (int l = 0; l < L; l++) { x1[l] += x0[l]; x0[l] = t[l]; }
      pair_fix(x1);
    }
  }
  // ---- one-lane forms (G == 1) of the pair operations.  Register budget: x0, x1, the quotient digits and the 2L column
  // registers are 5 L = 185 registers; anything else that must survive a pass is parked (in the lane's LDS staging area, free
  // during a squaring; in a spare row of the slot's scratch table during a product) instead of spilled by the compiler.
  // (x0, x1) <- (x0, x1)^2 with both operands in registers: t = x0^2 with its quotient digits, then
  // x1 <- (2 x0 x1 + R - q + q' n) / R + n - 1 with the multiplier doubled on the fly.  `park`: L words of LDS (this lane's).
  __device__ __forceinline__ void pair_sqr_r(uint32_t (&x0)[L], uint32_t (&x1)[L], uint32_t* park) const {
    // x1 is idle during the first pass and t during the second: each waits its turn in `park`, so that at most x0, one of
    // them, the quotient digits and the columns are in registers at any time (4 L + misc instead of 6 L)
    uint32_t q[L];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int l = 0; l < L; l++) park[l] = x1[l];
    {
      uint32_t t[L];
      mont_r<3, true>(t, x0, x0, q, q);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int l = 0; l < L; l++) { x1[l] = park[l]; }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int l = 0; l < L; l++) { park[l] = t[l]; }
    }
    neg_quot_init(q);
    mont_r<0, false, true, true, true>(x1, x0, x1, q, q);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int l = 0; l < L; l++) x0[l] = park[l];
  }
  // (x0, x1) <- (x0, x1) * (y0, y1) with ONE staging area in LDS (a second one would cost the eighth wave per CU): three
  // passes t = x0 y0 (quotient recorded), v = x1 y0 (same staged operand), u = x0 y1 + R - q, then (x0, x1) <- (t, u + v + n - 1):
  // 6 S^2 multiply-adds.  y0_src / y1_src: limb arrays of this lane's operand, element stride YS (rows of the slot's table, or LDS
  // constants).  The three passes are ONE rolled loop over a single instance of the product code (the unrolled 37-limb product
  // is 29 KB of code), the operand is always x0 (x0 and x1 trade places around the second pass) and t waits in `t_park`
  // (L elements of stride TS in the slot's table).
  template <int TS>
  __device__ __forceinline__ void pair_mul_seq(uint32_t (&x0)[L], uint32_t (&x1)[L], const uint32_t* y0_src, const uint32_t* y1_src, int ys,
                                               uint32_t* area, uint32_t* t_park) const {
    static_assert(G == 1, "sequential pair product: one-lane configurations only");
    uint32_t q[L];
#pragma unroll
    for (int l = 0; l < L; l++) q[l] = 0;
#pragma unroll 1
    for (int pass = 0; pass < 3; pass++) {
      if (pass != 1) {
        const uint32_t* src = (pass == 0) ? y0_src : y1_src;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int l = 0; l < L; l++) area[l] = src[l * ys];
        __builtin_amdgcn_wave_barrier();
      }
      if (pass != 0) {
#pragma unroll
        for (int l = 0; l < L; l++) { const uint32_t v = x0[l]; x0[l] = x1[l]; x1[l] = v; }
      }
      uint32_t r[L];
      mont_a1(r, area, x0, q, pass == 2, pass == 0);
      if (pass == 0) {
#pragma unroll
        for (int l = 0; l < L; l++) t_park[l * TS] = r[l];
        neg_quot_init(q);
      } else if (pass == 1) {
#pragma unroll
        for (int l = 0; l < L; l++) x0[l] = r[l];                        // v replaces the dead operand (old x1)
      } else {
#pragma unroll
        for (int l = 0; l < L; l++) x1[l] += r[l];                       // the x1 slot held v after the second exchange
      }
    }
#pragma unroll
    for (int l = 0; l < L; l++) x0[l] = t_park[l * TS];
    pair_fix(x1);
  }
  // Product pass with the multiplier in this lane's LDS area: r = (a b + start + q n) / R, start = q_io when `use_init` else
  // 0; the quotient digits replace q_io when `collect`, otherwise q_io is left alone (both flags wave-uniform run-time values).
  // Code size matters here: next to the register-operand squaring passes (45 KB) a fully unrolled 37-limb product (29 KB) no
  // longer fits the 64 KB instruction cache two CUs share, and every pair product would stream its code from L2 (measured:
  // 2x the time of the products).  So the limb steps run in CHUNKS: CH steps unrolled (the column registers rotate by
  // renaming inside a chunk), then the columns -- and the quotient digits, kept as a shift register -- are physically rotated
  // by CH positions so that the next chunk can run the same code: 4 chunks of 8 steps and one of 5 (37 is prime), 8 KB of
  // code, about 100 extra moves per 592 multiply-adds.
  template <int CH>
  __device__ __forceinline__ void mont_chunk(uint64_t (&T)[L], const uint32_t* a_lds, const uint32_t (&b)[L], uint32_t (&qd)[L], bool collect) const {
#pragma unroll
    for (int i = 0; i < CH; i++) {
      const uint32_t ai = a_lds[i];
#pragma unroll
      for (int c = 0; c < L; c++) T[(i + c) % L] += (uint64_t)ai * b[c];
      const uint32_t q = ((uint32_t)T[i] * n0inv) & LMASK;
#pragma unroll
      for (int c = 0; c < L; c++) T[(i + c) % L] += (uint64_t)q * n[c];
      T[(i + 1) % L] += T[i] >> W;
      T[i] = 0;
      qd[i] = collect ? q : qd[i];        // digit of this chunk's step i (the low CH entries are free); rotated into place below
    }
    // rotate by CH: the lowest live column (register CH) becomes register 0; the CH registers that already hold the new top
    // columns follow at the top.  Same rotation for the digit shift register.
    uint64_t tsave[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) tsave[i] = T[i];
#pragma unroll
    for (int r = 0; r + CH < L; r++) T[r] = T[r + CH];
#pragma unroll
    for (int i = 0; i < CH; i++) T[L - CH + i] = tsave[i];
    // (the digit register rotates even when nothing is recorded: L positions in total bring a preserved q_io back into place,
    // and one copy of the chunk code serves both kinds of pass)
    uint32_t qsave[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) qsave[i] = qd[i];
#pragma unroll
    for (int r = 0; r + CH < L; r++) qd[r] = qd[r + CH];
#pragma unroll
    for (int i = 0; i < CH; i++) qd[L - CH + i] = qsave[i];
  }
  __device__ __forceinline__ void mont_a1(uint32_t (&r)[L], const uint32_t* a_lds, const uint32_t (&b)[L], uint32_t (&q_io)[L],
                                          bool use_init, bool collect) const {
    static_assert(G == 1, "one-lane product pass");
    constexpr int CH = 8, NFULL = L / CH, REM = L % CH;      // L = 37: chunks of 8, 8, 8, 8, 5 limb steps
    uint64_t T[L];
#pragma unroll
    for (int i = 0; i < L; i++) T[i] = use_init ? (uint64_t)q_io[i] : 0ull;
    // after L steps and rotations by L positions in total the registers are back in column order.  q_io itself is the digit shift
    // register (its start value has been consumed above): with `collect` digit l ends in q_io[l], without q_io is unchanged
#pragma unroll 1
    for (int k = 0; k < NFULL; k++) mont_chunk<CH>(T, a_lds + CH * k, b, q_io, collect);
    if constexpr (REM > 0) mont_chunk<(REM > 0 ? REM : 1)>(T, a_lds + CH * NFULL, b, q_io, collect);
    uint64_t c = 0;
#pragma unroll
    for (int l = 0; l < L; l++) {
      const uint64_t v = T[l] + c;
      r[l] = (uint32_t)v & LMASK;
      c = v >> W;
    }
  }
  // (x0, x1) <- (w0, w1) with  w0 + w1 n = (x0 + x1 n) / R  (mod n^2): leaves the pair form (w0, w1 < 2n + 1, lazy)
  __device__ __forceinline__ void pair_redc(uint32_t (&x0)[L], uint32_t (&x1)[L]) const {
    uint32_t q[L];
#pragma unroll
    for (int l = 0; l < L; l++) q[l] = 0;
    mont<2>(x0, nullptr, x0, q);                                     // x0 <- (x0 + q n) / R
    neg_quot_init(q);
#pragma unroll
    for (int l = 0; l < L; l++) x1[l] += q[l];
    mont<1>(x1, nullptr, x1, q);                                     // x1 <- (x1 + R - q + q' n) / R
    pair_fix(x1);
  }

  // -------------------------------------------------------------------------------------------
  // Exact normalisation: r (entries < 2^32, any lazy form) minus sub[] (may be all zero) ->
  // limbs in [0, 2^29).  Precondition: the represented value is in [0, R).
  // -------------------------------------------------------------------------------------------
  __device__ __forceinline__ void normalize(uint32_t (&r)[L], const uint32_t (&sub)[L]) const {
    int64_t c = 0;
#pragma unroll
    for (int l = 0; l < L; l++) {
      const int64_t v = (int64_t)r[l] - (int64_t)sub[l] + c;
      r[l] = (uint32_t)v & LMASK;
      c = v >> W;
    }
    if constexpr (G > 1) {
      int32_t cin = (int32_t)from_below((uint32_t)(int32_t)c);
      while (__any(cin != 0)) {  // wave-uniform loop; at most G further rounds
        int64_t cc = cin;
#pragma unroll
        for (int l = 0; l < L; l++) {
          const int64_t v = (int64_t)r[l] + cc;
          r[l] = (uint32_t)v & LMASK;
          cc = v >> W;
        }
        cin = (int32_t)from_below((uint32_t)(int32_t)cc);
      }
    }
  }
  // is the exactly-normalised value r >= n ?  (same answer in every lane of the group)
  __device__ __forceinline__ bool ge_n(const uint32_t (&r)[L]) const {
    int cmp = 0;
#pragma unroll
    for (int l = L - 1; l >= 0; l--) cmp = (cmp != 0) ? cmp : ((r[l] > n[l]) ? 1 : ((r[l] < n[l]) ? -1 : 0));
    const uint64_t gt = __ballot(cmp > 0), lt = __ballot(cmp < 0);
    constexpr uint64_t GM = (G == 64) ? ~0ull : ((1ull << G) - 1);
    const uint32_t ggt = (uint32_t)((gt >> (g * G)) & GM), glt = (uint32_t)((lt >> (g * G)) & GM);
    return ggt >= glt;  // disjoint bit sets: the more significant differing lane decides; equal -> true
  }
  // r (lazy, value < 2^k * n for small k) -> canonical residue in [0, n), exact limbs
  __device__ __forceinline__ void canonical(uint32_t (&r)[L]) const {
    uint32_t zero[L];
#pragma unroll
    for (int l = 0; l < L; l++) zero[l] = 0;
    normalize(r, zero);
    for (int it = 0; it < 4; it++) {
      const bool ge = ge_n(r);
      if (!__any(ge)) break;
      uint32_t sub[L];
#pragma unroll
      for (int l = 0; l < L; l++) sub[l] = ge ? n[l] : 0u;
      normalize(r, sub);
    }
  }
  // -------------------------------------------------------------------------------------------
  // Leaving a context whose modulus is the multiple M = c n (Grp::NEG1, sc_lib.hip::neg1_twin) with a residue modulo n:
  // for a in Z_M,  (a c) mod M = c (a mod n)  (a = r + n j  =>  a c = r c + M j, and r c < n c = M), so the out-conversion
  // multiplies by the small factor c inside the reduction pass (redc_scaled: r = b c / R mod M, the same S^2 multiply-adds as a
  // plain reduction plus L), the caller canonicalises modulo M, and the exact quotient by c is a mod n in [0, n) -- canonical
  // without ever touching n (exact_div_small).
  // -------------------------------------------------------------------------------------------
  // r = b * scale / R mod n (lazily reduced); scale < 2^W, limbs of b < 2^(W+2): every column starts below 2^(2W+2).
  __device__ __forceinline__ void redc_scaled(uint32_t (&r)[L], const uint32_t (&b)[L], uint32_t scale) const {
    uint64_t T[L];
#pragma unroll
    for (int i = 0; i < L; i++) T[i] = (uint64_t)b[i] * scale;
#pragma unroll 1
    for (int k = 0; k < G; k++) {
#pragma unroll
      for (int l = 0; l < L; l++) {
        const uint32_t q = bcast0<G>(NEG1 ? (uint32_t)T[l] : (uint32_t)T[l] * n0inv) & lmask_v;
#pragma unroll
        for (int c = 0; c < L; c++) T[(l + c) % L] += (uint64_t)q * n[c];
        const uint64_t t0 = T[l];
        T[(l + 1) % L] += t0 >> W;
        T[l] = (uint64_t)(from_above_raw((uint32_t)t0) & lmask_v);
      }
    }
    uint64_t c = 0;
#pragma unroll
    for (int l = 0; l < L; l++) {
      const uint64_t v = T[l] + c;
      r[l] = (uint32_t)v & LMASK;
      c = v >> W;
    }
    if constexpr (G > 1) {
      const uint32_t clo = from_below((uint32_t)c), chi = from_below((uint32_t)(c >> 32));
      const uint64_t v = (uint64_t)r[0] + (((uint64_t)chi << 32) | clo);
      r[0] = (uint32_t)v & LMASK;
      r[1] += (uint32_t)(v >> W);
    }
  }
  // x <- x / c for an exact multiple x of the odd c < 2^W (x canonical: limbs < 2^W), cinv = c^-1 mod 2^W.  Least significant
  // limb first (Jebelean): y_i = (x_i - b) c^-1 mod 2^W, and the next limb owes b = floor(y_i c / 2^W) + [x_i < b].  The chain runs
  // through the lanes of the group in order: in phase k every lane runs its L limbs with the debt it was handed, lane k keeps the
  // result and hands its final debt to lane k + 1 (identical instruction stream in all lanes).  b <= c < 2^W throughout.
  __device__ __forceinline__ void exact_div_small(uint32_t (&x)[L], uint32_t c, uint32_t cinv) const {
    uint32_t res[L], debt_in = 0;
#pragma unroll
    for (int l = 0; l < L; l++) res[l] = 0;
#pragma unroll 1
    for (int k = 0; k < G; k++) {
      uint32_t b = debt_in, y[L];
#pragma unroll
      for (int l = 0; l < L; l++) {
        const uint32_t xi = x[l];
        const uint32_t under = (xi < b) ? 1u : 0u;
        const uint32_t t = (xi - b) & LMASK;
        y[l] = (t * cinv) & LMASK;
        b = (uint32_t)(((uint64_t)y[l] * c) >> W) + under;
      }
#pragma unroll
      for (int l = 0; l < L; l++) res[l] = (j == k) ? y[l] : res[l];
      debt_in = from_below(b);
    }
#pragma unroll
    for (int l = 0; l < L; l++) x[l] = res[l];
  }
  __device__ __forceinline__ bool equal(const uint32_t (&a)[L], const uint32_t (&b)[L]) const {
    uint32_t d = 0;
#pragma unroll
    for (int l = 0; l < L; l++) d |= a[l] ^ b[l];
    const uint64_t ne = __ballot(d != 0);
    constexpr uint64_t GM = (G == 64) ? ~0ull : ((1ull << G) - 1);
    return ((ne >> (g * G)) & GM) == 0;
  }

  // -------------------------------------------------------------------------------------------
  // LDS staging and format conversion (single-wave workgroups: __syncthreads() is a wave barrier)
  // -------------------------------------------------------------------------------------------
  __device__ __forceinline__ void stage(uint32_t* dst_lds /*group area*/, const uint32_t (&x)[L]) const {
#pragma unroll
    for (int l = 0; l < L; l++) dst_lds[j * L + l] = x[l];
  }
  __device__ __forceinline__ void stage_doubled(uint32_t* dst_lds, const uint32_t (&x)[L]) const {
#pragma unroll
    for (int l = 0; l < L; l++) dst_lds[j * L + l] = x[l] << 1;
  }
  // copy S limbs (limb form, global) of one number into this lane's registers
  // TS: element stride (1 = one number's limbs are consecutive; 64 = the one-lane table layout [limb][lane], in which the
  // 64 numbers of a wave interleave so that a wave's access to limb l is one contiguous 256-byte request)
  template <int TS = 1>
  __device__ __forceinline__ void load_limbs(uint32_t (&x)[L], const uint32_t* __restrict__ src) const {
#pragma unroll
    for (int l = 0; l < L; l++) x[l] = src[(j * L + l) * TS];
  }
  template <int TS = 1>
  __device__ __forceinline__ void store_limbs(uint32_t* __restrict__ dst, const uint32_t (&x)[L]) const {
#pragma unroll
    for (int l = 0; l < L; l++) dst[(j * L + l) * TS] = x[l];
  }
  // canonical little-endian 32-bit words (global) -> limb form.  wtmp: group's WP-word LDS scratch.
  // Reads `nwords` words starting at word offset `woff`, i.e. the value floor(x / 2^(32 woff)) mod 2^(32 nwords).
  __device__ __forceinline__ void load_words(uint32_t (&x)[L], const uint32_t* __restrict__ src, int nwords,
                                             uint32_t* wtmp) const {
    SC_WAVE_SYNC();
    for (int t = j; t < WP; t += G) wtmp[t] = (t < nwords) ? src[t] : 0u;
    SC_WAVE_SYNC();
#pragma unroll
    for (int l = 0; l < L; l++) {
      const int bit = W * (j * L + l);
      const int w0 = bit >> 5, sh = bit & 31;
      const uint64_t v = ((uint64_t)wtmp[w0 + 1] << 32) | wtmp[w0];
      x[l] = (uint32_t)(v >> sh) & LMASK;
    }
  }
  // exact limbs -> canonical 32-bit words (global).  ltmp: group's SP-limb LDS scratch.
  __device__ __forceinline__ void store_words(uint32_t* __restrict__ dst, int nwords, const uint32_t (&x)[L],
                                              uint32_t* ltmp, bool pred) const {
    SC_WAVE_SYNC();
    stage(ltmp, x);
    if (j == 0) { ltmp[S] = 0; ltmp[S + 1] = 0; ltmp[S + 2] = 0; }
    SC_WAVE_SYNC();
    if (pred) {
      for (int t = j; t < nwords; t += G) {
        const int bit = 32 * t;
        const int i0 = bit / W, off = bit - i0 * W;
        uint32_t word = 0;
        if (i0 < S) {
          const uint64_t v = (uint64_t)ltmp[i0] | ((uint64_t)ltmp[i0 + 1] << W) | ((uint64_t)ltmp[i0 + 2] << (2 * W));
          word = (uint32_t)(v >> off);
        }
        dst[t] = word;
      }
    }
  }
};

}  // namespace sc
