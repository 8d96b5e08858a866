// Instruction-rate probe for gfx950: measures issue rates of the integer multiply
// family used by the big-integer kernels.  Build: hipcc --offload-arch=gfx950 -O3 probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

constexpr int ITER = 4096;

// 8 independent v_mad_u64_u32 chains per lane
__global__ void k_mad64(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
  uint64_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "s20", "s21");
  }
  uint64_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
// mad64 + addc of carry (the realistic MAC primitive)
__global__ void k_mad64_addc(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
  uint64_t acc[8]; uint32_t cc[8];
  for (int i = 0; i < 8; i++) { acc[i] = i + threadIdx.x; cc[i] = 0; }
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(cc[i]) : "v"(a), "v"(b) : "vcc");
  }
  uint64_t s = 0; for (int i = 0; i < 8; i++) s += acc[i] + cc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
__global__ void k_mullo(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t a = a0 + threadIdx.x;
  uint32_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + b0;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t a = a0 + threadIdx.x;
  uint32_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + b0;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t a = a0 + threadIdx.x;
  uint32_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + b0;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_addco(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t a = a0 + threadIdx.x;
  uint32_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + b0;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(acc[i]) : "v"(a) : "vcc");
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad24(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
  uint32_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi24(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t a = a0 + threadIdx.x;
  uint32_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + b0;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma64(uint32_t* out, uint32_t a0, uint32_t b0) {
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9 * b0;
  double acc[8];
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + a0;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  double s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s;
}
__global__ void k_dpp_wshr(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + b0 + a0;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]));
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_dpp_add_wshl(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t acc[8];
  uint32_t src = threadIdx.x * 7 + a0;
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + b0;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      asm volatile("v_add_u32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(src));
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_readlane(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + b0 + a0;
  uint32_t tot = 0;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint32_t s;
      asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s) : "v"(acc[i]), "s"(it & 63));
      tot += s;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = tot + acc[0];
}
__global__ void k_bpermute(uint32_t* out, uint32_t a0, uint32_t b0) {
  int acc[8];
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + b0 + a0;
  int addr = ((threadIdx.x + 1) & 63) * 4;
  for (int it = 0; it < ITER / 8; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_ds_bpermute(addr, acc[i]);
  }
  int s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
int run(const char* name, K kern, double ops_per_thread, int wavesPerSimd, uint32_t* dout) {
  int cus = 256;
  int block = 256;                       // 4 waves = 1 per SIMD
  int grid = cus * wavesPerSimd;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, dout, 12345u, 67890u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, dout, 12345u, 67890u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  double total = ops_per_thread * (double)grid * block;
  double rate = total / (best * 1e-3);
  // cycles per wave-instruction per SIMD assuming 2.4 GHz
  double waveinstr_per_simd = ops_per_thread * wavesPerSimd;
  double cyc = best * 1e-3 * 2.4e9 / waveinstr_per_simd;
  printf("%-14s waves/SIMD=%d  %.3f ms  %.3e lane-ops/s  ~%.2f cyc/wave-instr/SIMD (at 2.4GHz)\n", name, wavesPerSimd, best, rate, cyc);
  return 0;
}

int main() {
  uint32_t* dout; CK(hipMalloc(&dout, 256 * 8 * 256 * 4 * 2));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  double n = (double)ITER * 8;
  for (int w : {1, 2, 4, 8}) {
    run("mad_u64_u32", k_mad64, n, w, dout);
    run("mad64+addc", k_mad64_addc, n, w, dout);
    run("mul_lo_u32", k_mullo, n, w, dout);
    run("mul_hi_u32", k_mulhi, n, w, dout);
    run("add_u32", k_add, n, w, dout);
    run("addc_co_u32", k_addco, n, w, dout);
    run("mad_u32_u24", k_mad24, n, w, dout);
    run("mul_hi_u24", k_mulhi24, n, w, dout);
    run("fma_f64", k_fma64, n, w, dout);
    run("mov_dpp_wshr", k_dpp_wshr, n, w, dout);
    run("add_dpp_wshl", k_dpp_add_wshl, n, w, dout);
    run("readlane", k_readlane, n, w, dout);
    run("ds_bpermute", k_bpermute, n / 8, w, dout);
  }
  return 0;
}
