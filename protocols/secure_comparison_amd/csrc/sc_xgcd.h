// Modular inversion of a few residues on the device: one 64-lane wave per residue, canonical 32-bit
// words spread over the lanes (WPL consecutive words per lane).  Batched division steps (see k_xgcd below); the
// exact multiword helpers of MW (ballot carry look-ahead) are only needed for the final canonical reduction.
// Used only at the top of the simultaneous-inversion tree (sc_modinv), i.e. for at most a few dozen residues
// per call, so it is latency- not throughput-tuned.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace sc {

template <int WPL>
struct MW {
  // carry-lookahead across lanes: gen/prop are per-lane predicates, cin the carry into lane 0
  static __device__ __forceinline__ bool lane_carry_in(bool gen, bool prop, uint32_t cin) {
    const uint64_t G = __ballot(gen), P = __ballot(prop);
    const uint64_t B = (G << 1) | (uint64_t)(cin & 1), T = P + B;
    const uint64_t C = B | (T ^ P ^ B);
    return (C >> (threadIdx.x & 63)) & 1;
  }
  // x = x + (neg ? ~y : y) + cin
  static __device__ __forceinline__ void add(uint32_t (&x)[WPL], const uint32_t (&y)[WPL], bool neg, uint32_t cin) {
    uint64_t c = 0;
    bool ones = true;
#pragma unroll
    for (int k = 0; k < WPL; k++) {
      const uint64_t t = (uint64_t)x[k] + (neg ? ~y[k] : y[k]) + c;
      x[k] = (uint32_t)t;
      c = t >> 32;
      ones = ones && (x[k] == 0xFFFFFFFFu);
    }
    uint32_t ci = lane_carry_in(c != 0, ones, cin) ? 1u : 0u;
#pragma unroll
    for (int k = 0; k < WPL; k++) {
      const uint64_t t = (uint64_t)x[k] + ci;
      x[k] = (uint32_t)t;
      ci = (uint32_t)(t >> 32);
    }
  }
  static __device__ __forceinline__ void shr1(uint32_t (&x)[WPL]) {
    // wave_shl:1 -- lane i reads lane i+1 across the whole wave (gfx9 DPP), lane 63 reads 0
    const uint32_t up = __builtin_amdgcn_update_dpp(0u, x[0], 0x130, 0xf, 0xf, true);
#pragma unroll
    for (int k = 0; k < WPL; k++) {
      const uint32_t nxt = (k + 1 < WPL) ? x[k + 1] : up;
      x[k] = (x[k] >> 1) | (nxt << 31);
    }
  }
  static __device__ __forceinline__ void shl1(uint32_t (&x)[WPL]) {
    // wave_shr:1 -- lane i reads lane i-1, lane 0 reads 0
    const uint32_t dn = __builtin_amdgcn_update_dpp(0u, x[WPL - 1], 0x138, 0xf, 0xf, true);
#pragma unroll
    for (int k = WPL - 1; k >= 0; k--) {
      const uint32_t prv = (k > 0) ? x[k - 1] : dn;
      x[k] = (x[k] << 1) | (prv >> 31);
    }
  }
  // sign of (x - y): 1, 0, -1 (wave-uniform)
  static __device__ __forceinline__ int cmp(const uint32_t (&x)[WPL], const uint32_t (&y)[WPL]) {
    int c = 0;
#pragma unroll
    for (int k = WPL - 1; k >= 0; k--) c = (c != 0) ? c : ((x[k] > y[k]) ? 1 : ((x[k] < y[k]) ? -1 : 0));
    const uint64_t gt = __ballot(c > 0), lt = __ballot(c < 0);
    return gt > lt ? 1 : (gt < lt ? -1 : 0);
  }
  static __device__ __forceinline__ bool is_zero(const uint32_t (&x)[WPL]) {
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < WPL; k++) o |= x[k];
    return __ballot(o != 0) == 0;
  }
  static __device__ __forceinline__ bool is_one(const uint32_t (&x)[WPL]) {
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < WPL; k++) o |= (k == 0 && (threadIdx.x & 63) == 0) ? (x[k] ^ 1u) : x[k];
    return __ballot(o != 0) == 0;
  }
  static __device__ __forceinline__ bool odd(const uint32_t (&x)[WPL]) { return (__builtin_amdgcn_readfirstlane(x[0]) & 1u) != 0; }
  // x = (x + m * n) / 2^32 with m = -x * n^-1 mod 2^32 (one word-level Montgomery step = 32 modular halvings at once);
  // needs one spare word above n (x + m n < 2^32 n + n)
  static __device__ __forceinline__ void halve32(uint32_t (&x)[WPL], const uint32_t (&n)[WPL], uint32_t n0inv) {
    const uint32_t m = __builtin_amdgcn_readfirstlane(x[0]) * n0inv;
    uint32_t lo[WPL], hi[WPL];
#pragma unroll
    for (int k = 0; k < WPL; k++) { lo[k] = n[k] * m; hi[k] = __umulhi(n[k], m); }
    add(x, lo, false, 0);                                   // word 0 becomes zero
    // the high halves enter one word higher: lane i word 0 takes lane i-1's top high half (wave_shr:1, lane 0 reads 0)
    const uint32_t dn = __builtin_amdgcn_update_dpp(0u, hi[WPL - 1], 0x138, 0xf, 0xf, true);
    uint32_t hs[WPL];
#pragma unroll
    for (int k = 0; k < WPL; k++) hs[k] = (k > 0) ? hi[k - 1] : dn;
    add(x, hs, false, 0);
    // drop the zero word: lane i word k takes word k+1, the top word of a lane comes from lane i+1 (wave_shl:1, lane 63 reads 0)
    const uint32_t up = __builtin_amdgcn_update_dpp(0u, x[0], 0x130, 0xf, 0xf, true);
#pragma unroll
    for (int k = 0; k < WPL; k++) x[k] = (k + 1 < WPL) ? x[k + 1] : up;
  }
};

// ---------------------------------------------------------------------------------------------
// Inversion by batched division steps (Bernstein-Yang "safegcd": decisions depend on the LOW bits only, so 30 steps can be
// simulated on two scalar registers and applied to the multiword numbers as ONE 2x2 matrix with 31-bit entries).
//   invariants:  d x = f,  e x = g  (mod n);  start (f, g, d, e) = (n, x, 0, 1);  after enough steps g = 0 and f = +-gcd.
//   one round:   (u v; q r) <- 30 division steps on the low words of f, g (scalar unit, wave-uniform)
//                (f, g) <- (u f + v g, q f + r g) / 2^30                       (exact)
//                (d, e) <- (u d + v e + md n, q d + r e + me n) / 2^30         (md, me in (-2^30, 0] make the sums divisible)
// The multiword numbers are kept REDUNDANTLY: lane L holds a signed value x_L of WPL words plus a signed overflow word, the
// number is sum x_L B^L with B = 2^(32 WPL).  A round is then lane-local (the matrix is applied to every lane's value on its own;
// only the 30 bits shifted out of a lane travel to the lane below, one DPP move per number) and there is no carry propagation
// across lanes until the very end.
// PROVEN bound on the lane values (oracle/xgcd_model.py restates the kernel and tests/test_xgcd_model_cpu.py drives it with
// adversarial matrices and inputs).  The transition matrix of k division steps is a product of the per-step matrices
// (0 2; -1 1), (2 0; 1 1), (2 0; 0 1), whose rows have absolute sums <= 2, so |u| + |v| <= 2^30 and |q| + |r| <= 2^30 for a
// round.  With M_t = max_L |x_L| after t rounds:  |u f_L + v g_L| <= 2^30 M_t, its quotient by 2^30 is at most M_t + 1 in
// magnitude, and the 30 bits arriving from the lane above add a value in [0, B):   M_(t+1) <= M_t + B + 1  for f, g.
// For d, e the multiple m n_L (|m| < 2^30, 0 <= n_L < B) adds at most B more:       M_(t+1) <= M_t + 2 B + 1.
// Growth is ADDITIVE, not a bit per round: from M_0 < B,  M_t < (t + 1)(2 B + 1) -- below 2^11 B for the 788 rounds of the
// largest operand (8192 bits), against the 2^31 B the signed overflow word can hold.  The 64-bit partial sums of apply() stay
// below 2^63 for the same reason ((2^31 - 1)(2^32 - 1) + 2^31 < 2^63).  The kernel checks the bound it relies on (|overflow
// word| < 2^24 in every lane after the last round) and reports status 2 instead of a result if it were ever violated.
// The value-level bound |d| <= (rounds + 1) n holds as well.
// The round count is the proven bound for the operand size (no zero test on a redundant g is needed: once g = 0 further steps
// leave f unchanged and d congruent).  4096-bit residues: 394 rounds of ~250 vector + ~350 scalar instructions instead of
// ~8200 iterations of multiword shift / add / compare with a ballot carry look-ahead each.
// ---------------------------------------------------------------------------------------------
template <int WPL>
struct DS {
  static constexpr int NW = WPL + 1;   // words per lane value: WPL unsigned words + the signed overflow word
  // y = a X + b Y + m N  (lane-local, exact), then x' = y >> 30 with the 30 low bits of the lane above entering at the top
  static __device__ __forceinline__ void apply(uint32_t (&out)[NW], int32_t a, const uint32_t (&X)[NW], int32_t b, const uint32_t (&Y)[NW],
                                               int32_t m, const uint32_t (&N)[WPL]) {
    uint32_t y[NW + 1];
    int64_t carry = 0;
    const uint32_t am = (a < 0) ? ~0u : 0u, bm = (b < 0) ? ~0u : 0u, mm = (m < 0) ? ~0u : 0u;
#pragma unroll
    for (int k = 0; k < WPL; k++) {
      // signed scalar x unsigned word: (scalar mod 2^32) * word - (scalar < 0 ? word << 32 : 0); |sum| < 2^63 (see header)
      uint64_t t = (uint64_t)(uint32_t)a * X[k] + (uint64_t)(uint32_t)b * Y[k] + (uint64_t)(uint32_t)m * N[k];
      t -= ((uint64_t)(X[k] & am) + (uint64_t)(Y[k] & bm) + (uint64_t)(N[k] & mm)) << 32;
      const int64_t v = (int64_t)t + carry;
      y[k] = (uint32_t)v;
      carry = v >> 32;
    }
    {
      const int64_t v = (int64_t)a * (int64_t)(int32_t)X[WPL] + (int64_t)b * (int64_t)(int32_t)Y[WPL] + carry;
      y[WPL] = (uint32_t)v;
      y[WPL + 1] = (uint32_t)(v >> 32);
    }
    // the 30 bits this lane's value loses go to the lane below; lane 0 loses zeros (the sums are divisible by 2^30)
    const uint32_t from_above = __builtin_amdgcn_update_dpp(0u, y[0] & 0x3fffffffu, 0x130, 0xf, 0xf, true);   // wave_shl:1
#pragma unroll
    for (int k = 0; k < NW; k++) out[k] = (y[k] >> 30) | (y[k + 1] << 2);
    const uint64_t top = (uint64_t)out[WPL - 1] + ((uint64_t)from_above << 2);
    out[WPL - 1] = (uint32_t)top;
    out[WPL] += (uint32_t)(top >> 32);
  }
  // bring a redundant number into exact two's complement words (K = 64 WPL words): lane by lane, the overflow word of lane L
  // is added (sign-extended) to the value of lane L + 1
  static __device__ __forceinline__ void normalize(uint32_t (&x)[NW]) {
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int L = 0; L < 63; L++) {
      const int32_t c = (int32_t)__builtin_amdgcn_readlane(x[WPL], L);
      if (lane == L) x[WPL] = 0;
      if (lane == L + 1) {
        int64_t v = (int64_t)(uint64_t)x[0] + (int64_t)c;
        x[0] = (uint32_t)v;
        int64_t cc = v >> 32;
#pragma unroll
        for (int k = 1; k < WPL; k++) { v = (int64_t)(uint64_t)x[k] + cc; x[k] = (uint32_t)v; cc = v >> 32; }
        x[WPL] = (uint32_t)((int64_t)(int32_t)x[WPL] + cc);
      }
    }
  }
};

// 30 division steps on the low 32 bits of f and g (two's complement) on the scalar unit; returns the transition matrix and
// updates eta = -delta (start -1).  One step:  g odd and delta > 0: (f, g) <- (g, (g - f) / 2), delta <- 1 - delta;
// g odd: g <- (g + f) / 2;  else g <- g / 2;  delta += 1.  Several steps per iteration: a run of zero bits of g is
// skipped with one count-trailing-zeros, and up to 8 low bits of g are cancelled at once by adding the multiple
// w = -g / f mod 2^k of f (as long as neither the 30 steps nor the sign of eta run out) -- about 5.5 iterations per 30 steps.
__device__ __forceinline__ void divsteps30(int32_t& eta, uint32_t f, uint32_t g, int32_t& u_, int32_t& v_, int32_t& q_, int32_t& r_) {
  uint32_t u = 1, v = 0, q = 0, r = 1;
  int i = 30;
#pragma unroll 1
  for (;;) {
    const int zeros = __builtin_ctz(g | (0xffffffffu << i));   // i <= 30: the sentinel bit limits the count to the steps left
    g >>= zeros; u <<= zeros; v <<= zeros; eta -= zeros; i -= zeros;
    if (i == 0) break;
    if (eta < 0) {          // delta > 0 and g odd: exchange
      eta = -eta;
      uint32_t t = f; f = g; g = 0u - t;
      t = u; u = q; q = 0u - t;
      t = v; v = r; r = 0u - t;
    }
    const int limit = (eta + 1 > i) ? i : eta + 1;
    const uint32_t m = (0xffffffffu >> (32 - limit)) & 255u;
    uint32_t x = f;                       // f^-1 mod 2^12 by Newton (f odd: f f = 1 mod 8)
    x *= 2u - f * x;
    x *= 2u - f * x;
    const uint32_t w = (g * (0u - x)) & m;
    g += f * w; q += u * w; r += v * w;
  }
  u_ = (int32_t)u; v_ = (int32_t)v; q_ = (int32_t)q; r_ = (int32_t)r;
}

template <int WPL>
__global__ void __launch_bounds__(64) k_xgcd(const uint32_t* __restrict__ xin, uint32_t* __restrict__ out,
                                             const uint32_t* __restrict__ nwords_dev, int nw, int* __restrict__ status) {
  using M = MW<WPL>;
  using D = DS<WPL>;
  constexpr int NW = WPL + 1;
  const int lane = threadIdx.x & 63;
  const uint64_t item = blockIdx.x;
  uint32_t n[WPL], x[WPL];
#pragma unroll
  for (int k = 0; k < WPL; k++) {
    const int i = lane * WPL + k;
    n[k] = (i < nw) ? nwords_dev[i] : 0u;
    x[k] = (i < nw) ? xin[item * nw + i] : 0u;
  }
  while (M::cmp(x, n) >= 0) M::add(x, n, true, 1);   // inputs are canonical residues: this is a guard
  uint32_t f[NW], g[NW], d[NW], e[NW];
#pragma unroll
  for (int k = 0; k < WPL; k++) { f[k] = n[k]; g[k] = x[k]; d[k] = 0; e[k] = (lane == 0 && k == 0) ? 1u : 0u; }
  f[WPL] = g[WPL] = d[WPL] = e[WPL] = 0;
  // n^-1 mod 2^30 (Newton on the low word)
  const uint32_t n0 = __builtin_amdgcn_readfirstlane(n[0]);
  uint32_t ninv = 1;
  for (int i = 0; i < 5; i++) ninv *= 2u - n0 * ninv;
  // rounds: the proven bound on division steps for inputs below 2^bits, (49 bits + 57) / 17 (safegcd paper, theorem 11.2)
  const int bits = 32 * nw;
  const int rounds = ((49 * bits + 57) / 17 + 1 + 29) / 30;
  int32_t eta = -1;
#pragma unroll 1
  for (int it = 0; it < rounds; it++) {
    // g = 0 for certain when every lane's value is zero: the remaining rounds would change nothing (the converse does not
    // hold for a redundant zero, which simply runs to the bound)
    {
      uint32_t any = 0;
#pragma unroll
      for (int k = 0; k < NW; k++) any |= g[k];
      if (__ballot(any != 0) == 0) break;
    }
    const uint32_t f0 = __builtin_amdgcn_readfirstlane(f[0]), g0 = __builtin_amdgcn_readfirstlane(g[0]);
    const uint32_t d0 = __builtin_amdgcn_readfirstlane(d[0]), e0 = __builtin_amdgcn_readfirstlane(e[0]);
    int32_t u, v, q, r;
    divsteps30(eta, f0, g0, u, v, q, r);
    const uint32_t cd = (uint32_t)u * d0 + (uint32_t)v * e0, ce = (uint32_t)q * d0 + (uint32_t)r * e0;   // low 30 bits matter
    const int32_t md = -(int32_t)((ninv * cd) & 0x3fffffffu), me = -(int32_t)((ninv * ce) & 0x3fffffffu);
    uint32_t nf[NW], ng[NW], nd[NW], ne[NW];
    D::apply(nf, u, f, v, g, 0, n);
    D::apply(ng, q, f, r, g, 0, n);
    D::apply(nd, u, d, v, e, md, n);
    D::apply(ne, q, d, r, e, me, n);
#pragma unroll
    for (int k = 0; k < NW; k++) { f[k] = nf[k]; g[k] = ng[k]; d[k] = nd[k]; e[k] = ne[k]; }
  }
  // the bound the redundant form relies on (see the header): overflow words far below 2^31
  bool bound_ok;
  {
    const int32_t lim = 1 << 24;
    const int32_t of = (int32_t)f[WPL], og = (int32_t)g[WPL], od = (int32_t)d[WPL], oe = (int32_t)e[WPL];
    const bool bad = of >= lim || of <= -lim || og >= lim || og <= -lim || od >= lim || od <= -lim || oe >= lim || oe <= -lim;
    bound_ok = __ballot(bad) == 0;
  }
  D::normalize(f);
  D::normalize(g);
  D::normalize(d);
  // exact K-word two's complement numbers from here on (|d| <= (rounds + 1) n < 2^11 n, far below 2^(32 K - 1))
  uint32_t fw[WPL], gw[WPL], dw[WPL], zero[WPL];
#pragma unroll
  for (int k = 0; k < WPL; k++) { fw[k] = f[k]; gw[k] = g[k]; dw[k] = d[k]; zero[k] = 0; }
  uint32_t allones = 0xffffffffu;
#pragma unroll
  for (int k = 0; k < WPL; k++) allones &= fw[k];
  const bool f_is_one = M::is_one(fw), f_is_minus_one = __ballot(allones != 0xffffffffu) == 0;
  const bool ok = bound_ok && M::is_zero(gw) && (f_is_one || f_is_minus_one);
  if (f_is_minus_one) {   // x^-1 = -d
    uint32_t t[WPL];
#pragma unroll
    for (int k = 0; k < WPL; k++) t[k] = 0;
    M::add(t, dw, true, 1);
#pragma unroll
    for (int k = 0; k < WPL; k++) dw[k] = t[k];
  }
  // canonical residue: d + 2^11 n is positive and below 2^12 n; subtract n 2^j for j = 11 .. 0 where it fits
  uint32_t ns[WPL];
#pragma unroll
  for (int k = 0; k < WPL; k++) ns[k] = n[k];
  for (int j = 0; j < 11; j++) M::shl1(ns);
  M::add(dw, ns, false, 0);
  for (int j = 11; j >= 0; j--) {
    if (M::cmp(dw, ns) >= 0) M::add(dw, ns, true, 1);
    M::shr1(ns);
  }
  if (ok) {
#pragma unroll
    for (int k = 0; k < WPL; k++) {
      const int i = lane * WPL + k;
      if (i < nw) out[item * nw + i] = dw[k];
    }
  }
  if (lane == 0) status[item] = ok ? 1 : (bound_ok ? 0 : 2);      // 2: internal bound violated (never expected; reported as an error)
  (void)zero;
}

// host launcher: d_n = modulus as canonical device words
inline int launch_xgcd(hipStream_t stream, const uint32_t* x, uint32_t* out, const uint32_t* d_n, int nw, uint64_t count,
                       int* d_status) {
  const int need = nw + 2;  // headroom: |d| < 2^11 n and the sign
  if (need <= 64) hipLaunchKernelGGL(k_xgcd<1>, dim3((unsigned)count), dim3(64), 0, stream, x, out, d_n, nw, d_status);
  else if (need <= 128) hipLaunchKernelGGL(k_xgcd<2>, dim3((unsigned)count), dim3(64), 0, stream, x, out, d_n, nw, d_status);
  else if (need <= 256) hipLaunchKernelGGL(k_xgcd<4>, dim3((unsigned)count), dim3(64), 0, stream, x, out, d_n, nw, d_status);
  else if (need <= 512) hipLaunchKernelGGL(k_xgcd<8>, dim3((unsigned)count), dim3(64), 0, stream, x, out, d_n, nw, d_status);
  else return -2;
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// v_mad_u64_u32 issue-rate probe: 8 independent accumulator chains per lane, nothing else in the loop
// (same kernel as tools/microbench/probe.hip::k_mad64, which measured 3.1e13 lane-MAC/s at 8 waves/SIMD).
__global__ void k_peak_probe(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
  uint64_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "s20", "s21");
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

}  // namespace sc
