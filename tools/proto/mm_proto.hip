// Prototype: reduced-radix (29-bit limb) Montgomery multiplication, G lanes x L limbs per number.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);}}while(0)

constexpr int W = 29;
constexpr uint32_t MASK = (1u << W) - 1;

template <int G> __device__ __forceinline__ uint32_t bcast0(uint32_t v) {
  if constexpr (G == 1) return v;
  else if constexpr (G == 2) return __builtin_amdgcn_update_dpp(0u, v, 0xA0, 0xf, 0xf, false);
  else if constexpr (G == 4) return __builtin_amdgcn_update_dpp(0u, v, 0x00, 0xf, 0xf, false);
  else if constexpr (G == 16) return __builtin_amdgcn_update_dpp(0u, v, 0x150, 0xf, 0xf, false);
  else {  // G == 8
    uint32_t lo = __builtin_amdgcn_update_dpp(0u, v, 0x150, 0xf, 0x3, false);  // banks 0,1 (lanes 0-7) <- lane 0
    return __builtin_amdgcn_update_dpp(lo, v, 0x158, 0xf, 0xc, false);          // banks 2,3 (lanes 8-15) <- lane 8
  }
}
// lane j receives the value of lane j+1 (same row); row end gets 0
__device__ __forceinline__ uint32_t row_down(uint32_t v) { return __builtin_amdgcn_update_dpp(0u, v, 0x101, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t row_up(uint32_t v) { return __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true); }

template <int G, int L>
__device__ __forceinline__ void montmul(uint32_t (&r)[L], const uint32_t* a_lds, const uint32_t (&b)[L],
                                        const uint32_t (&n)[L], uint32_t n0inv, uint32_t notTop, uint32_t notBot) {
  uint64_t T[L];
#pragma unroll
  for (int i = 0; i < L; i++) T[i] = 0;
#pragma unroll 1
  for (int k = 0; k < G; k++) {
    uint32_t av[L];
#pragma unroll
    for (int l = 0; l < L; l++) av[l] = a_lds[k * L + l];
#pragma unroll
    for (int l = 0; l < L; l++) {
      const uint32_t ai = av[l];
#pragma unroll
      for (int c = 0; c < L; c++) T[(l + c) % L] += (uint64_t)ai * b[c];
      uint32_t q = ((uint32_t)T[l] * n0inv) & MASK;
      q = bcast0<G>(q);
#pragma unroll
      for (int c = 0; c < L; c++) T[(l + c) % L] += (uint64_t)q * n[c];
      uint64_t t0 = T[l];
      T[(l + 1) % L] += t0 >> W;
      uint32_t low = (uint32_t)t0 & MASK;
      if constexpr (G == 1) T[l] = 0; else T[l] = row_down(low) & notTop;
    }
  }
  uint64_t c = 0;
#pragma unroll
  for (int l = 0; l < L; l++) { uint64_t v = T[l] + c; r[l] = (uint32_t)v & MASK; c = v >> W; }
  if constexpr (G > 1) {
    uint32_t clo = row_up((uint32_t)c) & notBot, chi = row_up((uint32_t)(c >> 32)) & notBot;
    uint64_t v = (uint64_t)r[0] + (((uint64_t)chi << 32) | clo);
    r[0] = (uint32_t)v & MASK;
    r[1] += (uint32_t)(v >> W);
  }
}

// test kernel: x <- x^(2^nsq) in Montgomery domain (input limbs already in internal form [num][S])
template <int G, int L>
__global__ void __launch_bounds__(64) k_sqr_chain(const uint32_t* __restrict__ xin, uint32_t* __restrict__ xout,
                                                  const uint32_t* __restrict__ nmod, uint32_t n0inv, int nsq, int count) {
  constexpr int S = G * L, NG = 64 / G;
  __shared__ uint32_t abuf[NG * (S + 1)];
  const int lane = threadIdx.x, g = lane / G, j = lane % G;
  const uint32_t notTop = (j == G - 1) ? 0u : ~0u, notBot = (j == 0) ? 0u : ~0u;
  uint32_t* mya = abuf + g * (S + 1);
  uint32_t n[L];
#pragma unroll
  for (int l = 0; l < L; l++) n[l] = nmod[j * L + l];
  for (int base = blockIdx.x * NG; base < count; base += gridDim.x * NG) {
    int idx = base + g; if (idx >= count) idx = count - 1;
    uint32_t x[L];
#pragma unroll
    for (int l = 0; l < L; l++) x[l] = xin[(size_t)idx * S + j * L + l];
    for (int s = 0; s < nsq; s++) {
      __syncthreads();
#pragma unroll
      for (int l = 0; l < L; l++) mya[j * L + l] = x[l];
      __syncthreads();
      uint32_t r[L];
      montmul<G, L>(r, mya, x, n, n0inv, notTop, notBot);
#pragma unroll
      for (int l = 0; l < L; l++) x[l] = r[l];
    }
    if (base + g < count) {
#pragma unroll
      for (int l = 0; l < L; l++) xout[(size_t)idx * S + j * L + l] = x[l];
    }
  }
}

int main(int argc, char** argv) {
  // input file: produced by mm_proto_gen.py: header G L count nsq n0inv, then n limbs (S), then count*S limbs
  const char* fn = argc > 1 ? argv[1] : "mm_in.bin";
  FILE* f = fopen(fn, "rb"); if (!f) { printf("no input\n"); return 1; }
  uint32_t hdr[5]; fread(hdr, 4, 5, f);
  int G = hdr[0], L = hdr[1], count = hdr[2], nsq = hdr[3]; uint32_t n0inv = hdr[4];
  int S = G * L;
  std::vector<uint32_t> n(S), x((size_t)count * S), out((size_t)count * S);
  fread(n.data(), 4, S, f); fread(x.data(), 4, (size_t)count * S, f); fclose(f);
  uint32_t *dn, *dx, *dout;
  CK(hipMalloc(&dn, S * 4)); CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dout, x.size() * 4));
  CK(hipMemcpy(dn, n.data(), S * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  int grid = argc > 2 ? atoi(argv[2]) : 256 * 8;
  auto launch = [&](int nsq_) {
    if (G == 4 && L == 18) hipLaunchKernelGGL((k_sqr_chain<4, 18>), dim3(grid), dim3(64), 0, 0, dx, dout, dn, n0inv, nsq_, count);
    else if (G == 8 && L == 18) hipLaunchKernelGGL((k_sqr_chain<8, 18>), dim3(grid), dim3(64), 0, 0, dx, dout, dn, n0inv, nsq_, count);
    else if (G == 2 && L == 18) hipLaunchKernelGGL((k_sqr_chain<2, 18>), dim3(grid), dim3(64), 0, 0, dx, dout, dn, n0inv, nsq_, count);
    else if (G == 1 && L == 18) hipLaunchKernelGGL((k_sqr_chain<1, 18>), dim3(grid), dim3(64), 0, 0, dx, dout, dn, n0inv, nsq_, count);
    else if (G == 4 && L == 27) hipLaunchKernelGGL((k_sqr_chain<4, 27>), dim3(grid), dim3(64), 0, 0, dx, dout, dn, n0inv, nsq_, count);
    else { printf("unsupported config\n"); exit(1); }
  };
  launch(nsq); CK(hipDeviceSynchronize());
  CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
  FILE* fo = fopen("gpurun_out/mm_out.bin", "wb"); if (fo) { fwrite(out.data(), 4, out.size(), fo); fclose(fo); }
  // timing with a longer chain
  int tsq = 200;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0)); launch(tsq); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double mm = (double)count * tsq;
    double words = (G * L * 29) / 32.0;  // equivalent 32-bit words (approx)
    printf("G=%d L=%d count=%d grid=%d: %.3f ms, %.3e modmul/s, executed mads/s=%.3e (2*S^2 per modmul)\n", G, L, count, grid, ms,
           mm / (ms * 1e-3), mm * 2.0 * S * S / (ms * 1e-3));
  }
  return 0;
}
