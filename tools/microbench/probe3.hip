// Third instruction-cost probe (round 3): the MARGINAL cost of each bookkeeping instruction of a limb step when it sits in a
// stream of v_mad_u64_u32, at the two waves per SIMD the product kernels run with.  Body = 8 independent multiply-adds + one
// candidate instruction; cost = (time(body) - time(8 multiply-adds)) per iteration, in ns and in cycles at the clock the
// multiply-add-only stream implies for 4.75 cycles per instruction (probe2).  Build: hipcc --offload-arch=gfx950 -O2 probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
constexpr int ITER = 40000;

#define MACS8 _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "s20","s21");

#define KERNEL(NAME, EXTRA)                                                                     \
__global__ void NAME(uint32_t* out, uint32_t a0, uint32_t b0) {                                 \
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;                                          \
  uint64_t r[8]; for (int i = 0; i < 8; i++) r[i] = i + threadIdx.x + b0;                       \
  uint32_t x = a0 * 3 + threadIdx.x, y = b0 + 7, z = threadIdx.x * 5, qv = z + 1, o2 = 0; uint64_t w = r[3] * 5;   \
  asm volatile("v_cmp_eq_u32 vcc, %0, %1\n\tv_cmp_eq_u32 s[22:23], %0, %1" :: "v"(threadIdx.x & 3), "v"(b0 & 3) : "vcc", "s22", "s23"); \
  for (int it = 0; it < ITER; it++) { MACS8 EXTRA }                                             \
  uint64_t s = w; for (int i = 0; i < 8; i++) s += r[i];                                        \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ x ^ y ^ z ^ qv ^ o2;  \
}

KERNEL(k_base, )
KERNEL(k_mac9, asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(w) : "v"(a), "v"(b) : "s20","s21");)
KERNEL(k_mac_const8, asm volatile("v_mad_u64_u32 %0, s[20:21], %1, 8, %0" : "+v"(w) : "v"(a) : "s20","s21");)
KERNEL(k_cndmask_vcc, asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(y));)
KERNEL(k_cndmask_sgpr, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(x) : "v"(y));)
KERNEL(k_and_or, asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));)
KERNEL(k_bfi, asm volatile("v_bfi_b32 %0, %2, %1, %0" : "+v"(x) : "v"(y), "v"(z));)
KERNEL(k_dpp_bankmov, asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x2" : "+v"(x) : "v"(y));)
KERNEL(k_and_dpp, asm volatile("v_and_b32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(x) : "v"(y), "v"(z));)
KERNEL(k_and_dpp_bcast, asm volatile("v_and_b32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(x) : "v"(y), "v"(z));)
KERNEL(k_lshr64, asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(w));)
KERNEL(k_lshladd64, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w) : "v"(r[0]));)
KERNEL(k_mullo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(y));)
KERNEL(k_and, asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(y));)
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));)
KERNEL(k_alignbit, asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(x) : "v"(y));)
KERNEL(k_lshr32, asm volatile("v_lshrrev_b32 %0, 29, %1" : "=v"(x) : "v"(y));)
KERNEL(k_dsread, asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"((threadIdx.x & 63) * 4));)
KERNEL(k_dsread_nowait, asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"((threadIdx.x & 63) * 4));)
#define LIMBSTEP(SELECT) asm volatile("v_lshrrev_b64 %0, 29, %5\n\tv_lshl_add_u64 %1, %1, 0, %0\n\tv_and_b32_dpp %2, %6, %7 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" SELECT "\n\tv_and_b32_dpp %4, %6, %7 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" \
                              : "+v"(w), "+v"(r[1]), "=&v"(x), "+v"(qv), "=&v"(o2) : "v"(r[0]), "v"(y), "v"(z));
KERNEL(k_limbstep_now, LIMBSTEP("v_cndmask_b32_e32 %3, %3, %6, vcc"))
KERNEL(k_limbstep_andor, LIMBSTEP("v_and_or_b32 %3, %6, %7, %3"))
KERNEL(k_limbstep_nosel, LIMBSTEP("s_nop 0"))

template <typename K>
int run(const char* name, K kern, uint32_t* dout, double* base_ms, int wps) {
  int grid = 256 * wps, block = 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, dout, 12345u, 67890u);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, dout, 12345u, 67890u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  if (*base_ms == 0) *base_ms = best;
  // per SIMD: wps waves x ITER iterations; the base body is 8 multiply-adds per wave and iteration
  double ns_iter = best * 1e6 / ((double)ITER * wps), base_iter = *base_ms * 1e6 / ((double)ITER * wps);
  double cyc_per_ns = 8 * 4.75 / base_iter;   // clock implied by 4.75 cycles per multiply-add in the base stream
  printf("%-20s w/SIMD=%d  %.3f ms  per iteration and wave %.2f ns  marginal %.2f ns = %.2f cycles (at %.2f GHz implied)\n", name, wps, best,
         ns_iter, ns_iter - base_iter, (ns_iter - base_iter) * cyc_per_ns, cyc_per_ns);
  return 0;
}

int main() {
  uint32_t* dout; CK(hipMalloc(&dout, 256 * 4 * 256 * 4));
  for (int wps : {2, 1}) {
    double base = 0;
#define R(k) run(#k, k, dout, &base, wps)
    R(k_base); R(k_mac9); R(k_mac_const8); R(k_cndmask_vcc); R(k_cndmask_sgpr); R(k_and_or); R(k_bfi); R(k_dpp_bankmov); R(k_and_dpp); R(k_and_dpp_bcast);
    R(k_lshr64); R(k_lshladd64); R(k_mullo); R(k_and); R(k_mov); R(k_alignbit); R(k_lshr32); R(k_dsread); R(k_dsread_nowait);
    R(k_limbstep_now); R(k_limbstep_andor); R(k_limbstep_nosel);
  }
  return 0;
}
