// Modular inversion of a few residues on the device: one 64-lane wave per residue, canonical 32-bit
// words spread over the lanes (WPL consecutive words per lane).  Kaliski's "almost Montgomery
// inverse" (binary extended GCD using only shifts, adds, subtracts and compares) followed by the
// 2^-k correction by modular halving.  Used only at the top of the simultaneous-inversion tree
// (sc_modinv), i.e. for at most a few dozen residues per call, so it is latency- not throughput-tuned.
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

template <int WPL>
__global__ void __launch_bounds__(64) k_xgcd(const uint32_t* __restrict__ xin, uint32_t* __restrict__ out,
                                             const uint32_t* __restrict__ nwords_dev, int nw, int* __restrict__ status) {
  using M = MW<WPL>;
  const int lane = threadIdx.x & 63;
  const uint64_t item = blockIdx.x;
  uint32_t u[WPL], v[WPL], r[WPL], s[WPL], n[WPL];
#pragma unroll
  for (int k = 0; k < WPL; k++) {
    const int i = lane * WPL + k;
    n[k] = (i < nw) ? nwords_dev[i] : 0u;
    u[k] = n[k];
    v[k] = (i < nw) ? xin[item * nw + i] : 0u;
    r[k] = 0;
    s[k] = (i == 0) ? 1u : 0u;
  }
  // reduce the input below n first (inputs are canonical residues, this is a guard)
  while (M::cmp(v, n) >= 0) M::add(v, n, true, 1);
  int kk = 0;
  const int max_iter = 64 * nw + 8;
  while (!M::is_zero(v) && kk < max_iter) {
    if (!M::odd(u)) {
      M::shr1(u); M::shl1(s);
    } else if (!M::odd(v)) {
      M::shr1(v); M::shl1(r);
    } else if (M::cmp(u, v) > 0) {
      M::add(u, v, true, 1); M::shr1(u);
      M::add(r, s, false, 0); M::shl1(s);
    } else {
      M::add(v, u, true, 1); M::shr1(v);
      M::add(s, r, false, 0); M::shl1(r);
    }
    kk++;
  }
  const bool ok = M::is_one(u) && M::is_zero(v);
  if (M::cmp(r, n) >= 0) M::add(r, n, true, 1);
  // x = n - r  (= a^-1 * 2^kk mod n); r may be 0 only when not invertible
  uint32_t x[WPL];
#pragma unroll
  for (int k = 0; k < WPL; k++) x[k] = n[k];
  M::add(x, r, true, 1);
  if (M::cmp(x, n) >= 0) M::add(x, n, true, 1);
  // x * 2^-kk mod n: whole words by word-level Montgomery steps, the remaining kk mod 32 bits by modular halving
  uint32_t n0inv = 1;
  {
    const uint32_t n0 = __builtin_amdgcn_readfirstlane(n[0]);
    for (int i = 0; i < 5; i++) n0inv *= 2u - n0 * n0inv;   // n^-1 mod 2^32 (Newton)
    n0inv = 0u - n0inv;
  }
  int t = 0;
  for (; t + 32 <= kk; t += 32) M::halve32(x, n, n0inv);
  for (; t < kk; t++) {
    if (M::odd(x)) M::add(x, n, false, 0);
    M::shr1(x);
  }
  if (M::cmp(x, n) >= 0) M::add(x, n, true, 1);
  if (ok) {
#pragma unroll
    for (int k = 0; k < WPL; k++) {
      const int i = lane * WPL + k;
      if (i < nw) out[item * nw + i] = x[k];
    }
  }
  if (lane == 0) status[item] = ok ? 1 : 0;
}

// host launcher: d_n = modulus as canonical device words
inline int launch_xgcd(hipStream_t stream, const uint32_t* x, uint32_t* out, const uint32_t* d_n, int nw, uint64_t count,
                       int* d_status) {
  const int need = nw + 1;  // one spare word: r, s < 2n
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
