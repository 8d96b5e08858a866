// Prototype / microbenchmark: one-lane-per-number products (G = 1, L = 37, 28-bit limbs) against the two-lane form (2, 18, 29)
// for a 1024-bit modulus: N squarings (single modulus) and N pair squarings (arithmetic modulo p^2) per item.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 tools/proto/g1_proto.hip -o tools/proto/g1_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "../../protocols/secure_comparison_amd/csrc/sc_device.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);}}while(0)
using namespace sc;

#ifndef WAVES
#define WAVES 2
#endif

// ---- single-modulus squarings
template <int G, int L, int W>
__global__ void __launch_bounds__(64, WAVES) k_sq(const uint32_t* __restrict__ modctx, uint32_t n0inv, const uint32_t* __restrict__ x, uint32_t* out, uint64_t count, int nsq) {
  using GT = Grp<G, L, W>;
  constexpr int S = GT::S, NG = GT::NG, SP = GT::SP;
  __shared__ uint32_t s_a[G == 1 ? 1 : NG * SP];
  __shared__ uint32_t s_a2[G == 1 ? 1 : NG * SP];
  GT gp; gp.init(modctx, n0inv);
  uint32_t* my_a = s_a + (G == 1 ? 0 : gp.g * SP);
  uint32_t* my_a2 = s_a2 + (G == 1 ? 0 : gp.g * SP);
  for (uint64_t base = (uint64_t)blockIdx.x * NG; base < count; base += (uint64_t)gridDim.x * NG) {
    const uint64_t idx = (base + gp.g < count) ? base + gp.g : count - 1;
    uint32_t acc[L];
    gp.load_limbs(acc, x + idx * S);
#pragma unroll 1
    for (int it = 0; it < nsq; it++) {
      uint32_t r[L];
      if constexpr (G == 1) {
        uint32_t dummy[L];
        gp.template mont_r<3>(r, acc, acc, dummy, dummy);
      } else {
        __syncthreads();
        gp.stage(my_a, acc); gp.stage_doubled(my_a2, acc);
        __syncthreads();
        gp.sqr(r, my_a, my_a2, acc);
      }
#pragma unroll
      for (int l = 0; l < L; l++) acc[l] = r[l];
    }
    gp.canonical(acc);
    if (base + gp.g < count) gp.store_limbs(out + idx * S, acc);
  }
}

// ---- pair squarings
template <int G, int L, int W>
__global__ void __launch_bounds__(64, WAVES) k_psq(const uint32_t* __restrict__ modctx, uint32_t n0inv, const uint32_t* __restrict__ x, uint32_t* out, uint64_t count, int nsq) {
  using GT = Grp<G, L, W>;
  constexpr int S = GT::S, NG = GT::NG, SP = GT::SP;
  __shared__ uint32_t s_a[G == 1 ? 1 : NG * SP];
  __shared__ uint32_t s_a2[G == 1 ? 1 : NG * SP];
  GT gp; gp.init(modctx, n0inv);
  uint32_t* my_a = s_a + (G == 1 ? 0 : gp.g * SP);
  uint32_t* my_a2 = s_a2 + (G == 1 ? 0 : gp.g * SP);
  for (uint64_t base = (uint64_t)blockIdx.x * NG; base < count; base += (uint64_t)gridDim.x * NG) {
    const uint64_t idx = (base + gp.g < count) ? base + gp.g : count - 1;
    uint32_t x0[L], x1[L];
    gp.load_limbs(x0, x + idx * S);
#pragma unroll
    for (int l = 0; l < L; l++) x1[l] = x0[(l + 5) % L] >> 1;
#pragma unroll 1
    for (int it = 0; it < nsq; it++) {
      if constexpr (G == 1) {
        uint32_t t[L], q[L];
        gp.template mont_r<3, true>(t, x0, x0, q, q);
        gp.neg_quot_init(q);
        gp.template mont_r<0, false, true, true, true>(x1, x0, x1, q, q);
#pragma unroll
        for (int l = 0; l < L; l++) x0[l] = t[l];
      } else {
        __syncthreads();
        gp.stage(my_a, x0); gp.stage_doubled(my_a2, x0);
        __syncthreads();
        gp.pair_sqr(x0, x1, my_a, my_a2);
      }
    }
    gp.canonical(x0); gp.canonical(x1);
    if (base + gp.g < count) { gp.store_limbs(out + idx * 2 * S, x0); gp.store_limbs(out + idx * 2 * S + S, x1); }
  }
}

// ---- host helpers (tiny big integers on 32-bit words)
typedef std::vector<uint32_t> Big;
static int cmp(const Big& a, const Big& b) { for (int i = (int)a.size() - 1; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i] ? 1 : -1; return 0; }
static void sub(Big& a, const Big& b) { uint64_t br = 0; for (size_t i = 0; i < a.size(); i++) { uint64_t v = (uint64_t)a[i] - b[i] - br; a[i] = (uint32_t)v; br = (v >> 32) & 1; } }
static void dblmod(Big& a, const Big& n) { uint32_t c = 0; for (size_t i = 0; i < a.size(); i++) { uint32_t nc = a[i] >> 31; a[i] = (a[i] << 1) | c; c = nc; } if (cmp(a, n) >= 0) sub(a, n); }
static Big shlmod(Big v, const Big& n, int k) { for (int i = 0; i < k; i++) dblmod(v, n); return v; }
static std::vector<uint32_t> limbs(const Big& x, int S, int W) { std::vector<uint32_t> o(S); for (int i = 0; i < S; i++) { int bit = W * i, w0 = bit >> 5, sh = bit & 31; uint64_t v = w0 < (int)x.size() ? x[w0] : 0; if (w0 + 1 < (int)x.size()) v |= (uint64_t)x[w0 + 1] << 32; o[i] = (uint32_t)(v >> sh) & ((1u << W) - 1); } return o; }
static Big words(const uint32_t* l, int S, int W, int nw) { Big o(nw + 2, 0); for (int i = 0; i < S; i++) { int bit = W * i, w0 = bit >> 5, sh = bit & 31; uint64_t v = (uint64_t)l[i] << sh; if (w0 < (int)o.size()) o[w0] += (uint32_t)v; /* no overlap: limbs exact */ if (w0 + 1 < (int)o.size()) o[w0 + 1] += (uint32_t)(v >> 32); } o.resize(nw); return o; }

template <int G, int L, int W>
double run(const char* name, const Big& n, const std::vector<Big>& xs, uint64_t count, int nsq, bool pair, std::vector<Big>* results) {
  constexpr int S = G * L;
  const int nw = (int)n.size() - 1;
  uint32_t n0 = n[0], inv = 1; for (int i = 0; i < 6; i++) inv *= 2 - n0 * inv;
  uint32_t n0inv = (0u - inv) & ((1u << W) - 1);
  Big one(n.size(), 0); one[0] = 1;
  Big r1 = shlmod(one, n, W * S);
  std::vector<uint32_t> ctx = limbs(n, S, W);
  std::vector<uint32_t> hx((size_t)count * S);
  for (uint64_t i = 0; i < count; i++) { Big xm = shlmod(xs[i % xs.size()], n, W * S); auto l = limbs(xm, S, W); for (int k = 0; k < S; k++) hx[i * S + k] = l[k]; }
  uint32_t *dctx, *dx, *dout;
  const size_t outw = pair ? 2 * S : S;
  CK(hipMalloc(&dctx, ctx.size() * 4)); CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dout, (size_t)count * outw * 4));
  CK(hipMemcpy(dctx, ctx.data(), ctx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int occ = 0;
  if (pair) { CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_psq<G, L, W>, 64, 0)); } else { CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_sq<G, L, W>, 64, 0)); }
  constexpr int NG = 64 / G;
  uint64_t need = (count + NG - 1) / NG;
  unsigned grid = (unsigned)std::min<uint64_t>(need, (uint64_t)prop.multiProcessorCount * occ);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0));
    if (pair) hipLaunchKernelGGL((k_psq<G, L, W>), dim3(grid), dim3(64), 0, 0, dctx, n0inv, dx, dout, count, nsq);
    else hipLaunchKernelGGL((k_sq<G, L, W>), dim3(grid), dim3(64), 0, 0, dctx, n0inv, dx, dout, count, nsq);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  std::vector<uint32_t> ho(outw * 4);
  CK(hipMemcpy(ho.data(), dout, outw * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));   // first four items
  if (results) { results->clear(); for (int i = 0; i < 4; i++) for (size_t h = 0; h < outw / S; h++) results->push_back(words(ho.data() + i * outw + h * S, S, W, nw)); }
  const double macs_sq = (double)S * S + (double)G * G * L * (L + 1) / 2.0;
  const double macs = pair ? macs_sq + 3.0 * S * S : macs_sq;
  printf("%-28s (%d,%d,%d) occ %d grid %5u  %8.2f ms  %6.2f T executed MAC/s  (%.0f ns per %s per item-lane-group)\n", name, G, L, W, occ, grid, best,
         count * (double)nsq * macs / (best * 1e-3) / 1e12, best * 1e6 / nsq / ((double)(need + grid - 1) / grid), pair ? "pair squaring" : "squaring");
  (void)r1;
  CK(hipFree(dctx)); CK(hipFree(dx)); CK(hipFree(dout));
  return best;
}

int main(int argc, char** argv) {
  const uint64_t count = argc > 1 ? strtoull(argv[1], 0, 10) : 196608;
  const int nsq = argc > 2 ? atoi(argv[2]) : 512;
  srand(7);
  Big n(33, 0);   // 1024-bit odd modulus with one spare word
  for (int i = 0; i < 32; i++) n[i] = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
  n[0] |= 1; n[31] |= 0x80000000u;
  std::vector<Big> xs(64, Big(33, 0));
  for (auto& x : xs) { for (int i = 0; i < 32; i++) x[i] = ((uint32_t)rand() << 16) ^ (uint32_t)rand(); x[31] &= 0x7fffffffu; }
  std::vector<Big> ra, rb, rc, rd;
  run<2, 18, 29>("squarings two-lane", n, xs, count, nsq, false, &ra);
  run<1, 37, 28>("squarings one-lane", n, xs, count, nsq, false, &rb);
  run<2, 18, 29>("pair squarings two-lane", n, xs, count, nsq, true, &rc);
  run<1, 37, 28>("pair squarings one-lane", n, xs, count, nsq, true, &rd);
  // The Montgomery radix differs (R = 2^1044 vs 2^1036), so results are compared after n squarings only through x^(2^k) R-power
  // bookkeeping in the product's own tests; here: report whether each form is self-consistent across identical inputs.
  printf("self-consistency: two-lane item0==item64? n/a; one-lane vs two-lane differ by a power of 2 (checked in the library tests)\n");
  return 0;
}
