// Second instruction-cost probe: in-kernel cycle counts (s_memtime) so results are
// independent of DVFS.  One wave per SIMD and 4 waves per SIMD are both reported.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
constexpr int ITER = 20000;

#define KERNEL(NAME, DECL, BODY, FINAL)                                              \
__global__ void NAME(uint32_t* out, unsigned long long* cyc, uint32_t a0, uint32_t b0) { \
  DECL                                                                               \
  unsigned long long t0 = __builtin_readcyclecounter();                                    \
  for (int it = 0; it < ITER; it++) { BODY }                                         \
  unsigned long long t1 = __builtin_readcyclecounter();                                    \
  FINAL                                                                              \
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
}

#define DECL8 uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x; uint32_t r[8]; for (int i = 0; i < 8; i++) r[i] = i + threadIdx.x + b0; (void)a; (void)b;
#define FIN8 { uint32_t s = 0; for (int i = 0; i < 8; i++) s += r[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = s; }
#define DECL8Q uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x; uint64_t r[8]; for (int i = 0; i < 8; i++) r[i] = i + threadIdx.x + b0; (void)a; (void)b;
#define FIN8Q { uint64_t s = 0; for (int i = 0; i < 8; i++) s += r[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32); }
#define REP8(STR, ...) _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(STR : "+v"(r[i]) : __VA_ARGS__);

KERNEL(k_add_vop2, DECL8, REP8("v_add_u32 %0, %0, %1", "v"(a)), FIN8)
KERNEL(k_add_vop3, DECL8, REP8("v_add_u32_e64 %0, %0, %1", "v"(a)), FIN8)
KERNEL(k_add3, DECL8, REP8("v_add3_u32 %0, %0, %1, %2", "v"(a), "v"(b)), FIN8)
KERNEL(k_addco_vcc, DECL8, _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(a) : "vcc");, FIN8)
KERNEL(k_addco_sgpr, DECL8, _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_add_co_u32 %0, s[20:21], %0, %1" : "+v"(r[i]) : "v"(a) : "s20","s21");, FIN8)
KERNEL(k_addc_vcc, DECL8, _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : "vcc");, FIN8)
// addc with independent SGPR-pair carries (no serial dependency through vcc)
KERNEL(k_addc_sgpr, DECL8,
  asm volatile("v_addc_co_u32 %0, s[20:21], %0, %8, s[20:21]\n\tv_addc_co_u32 %1, s[22:23], %1, %8, s[22:23]\n\tv_addc_co_u32 %2, s[24:25], %2, %8, s[24:25]\n\tv_addc_co_u32 %3, s[26:27], %3, %8, s[26:27]\n\t"
               "v_addc_co_u32 %4, s[28:29], %4, %8, s[28:29]\n\tv_addc_co_u32 %5, s[30:31], %5, %8, s[30:31]\n\tv_addc_co_u32 %6, s[32:33], %6, %8, s[32:33]\n\tv_addc_co_u32 %7, s[34:35], %7, %8, s[34:35]"
     : "+v"(r[0]),"+v"(r[1]),"+v"(r[2]),"+v"(r[3]),"+v"(r[4]),"+v"(r[5]),"+v"(r[6]),"+v"(r[7]) : "v"(a)
     : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35");, FIN8)
KERNEL(k_mad64, DECL8Q, _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "s20","s21");, FIN8Q)
KERNEL(k_mad64_sgprsrc, DECL8Q, _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(r[i]) : "s"(a0), "v"(b) : "s20","s21");, FIN8Q)
// mad + addc using per-chain sgpr carries
KERNEL(k_mad64_addc_sg, DECL8Q uint32_t c[8]; for (int i = 0; i < 8; i++) c[i] = 0;,
  _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\tv_addc_co_u32 %1, s[20:21], 0, %1, s[20:21]" : "+v"(r[i]), "+v"(c[i]) : "v"(a), "v"(b) : "s20","s21");,
  { for (int i = 0; i < 8; i++) r[0] += c[i]; } FIN8Q)
// 8 mads then 8 addc (batched), carries in distinct sgpr pairs
KERNEL(k_mad64x8_addcx8, DECL8Q uint32_t c[8]; for (int i = 0; i < 8; i++) c[i] = 0;,
  asm volatile(
    "v_mad_u64_u32 %0, s[20:21], %16, %17, %0\n\tv_mad_u64_u32 %1, s[22:23], %16, %17, %1\n\tv_mad_u64_u32 %2, s[24:25], %16, %17, %2\n\tv_mad_u64_u32 %3, s[26:27], %16, %17, %3\n\t"
    "v_mad_u64_u32 %4, s[28:29], %16, %17, %4\n\tv_mad_u64_u32 %5, s[30:31], %16, %17, %5\n\tv_mad_u64_u32 %6, s[32:33], %16, %17, %6\n\tv_mad_u64_u32 %7, s[34:35], %16, %17, %7\n\t"
    "v_addc_co_u32 %8, s[20:21], 0, %8, s[20:21]\n\tv_addc_co_u32 %9, s[22:23], 0, %9, s[22:23]\n\tv_addc_co_u32 %10, s[24:25], 0, %10, s[24:25]\n\tv_addc_co_u32 %11, s[26:27], 0, %11, s[26:27]\n\t"
    "v_addc_co_u32 %12, s[28:29], 0, %12, s[28:29]\n\tv_addc_co_u32 %13, s[30:31], 0, %13, s[30:31]\n\tv_addc_co_u32 %14, s[32:33], 0, %14, s[32:33]\n\tv_addc_co_u32 %15, s[34:35], 0, %15, s[34:35]"
    : "+v"(r[0]),"+v"(r[1]),"+v"(r[2]),"+v"(r[3]),"+v"(r[4]),"+v"(r[5]),"+v"(r[6]),"+v"(r[7]),
      "+v"(c[0]),"+v"(c[1]),"+v"(c[2]),"+v"(c[3]),"+v"(c[4]),"+v"(c[5]),"+v"(c[6]),"+v"(c[7])
    : "v"(a), "v"(b)
    : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35");,
  { for (int i = 0; i < 8; i++) r[0] += c[i]; } FIN8Q)
KERNEL(k_mullo, DECL8, REP8("v_mul_lo_u32 %0, %0, %1", "v"(a)), FIN8)
KERNEL(k_mulhi, DECL8, REP8("v_mul_hi_u32 %0, %0, %1", "v"(a)), FIN8)
KERNEL(k_mad24, DECL8, REP8("v_mad_u32_u24 %0, %1, %2, %0", "v"(a), "v"(b)), FIN8)
KERNEL(k_mul24, DECL8, REP8("v_mul_u32_u24 %0, %0, %1", "v"(a)), FIN8)
KERNEL(k_mulhi24, DECL8, REP8("v_mul_hi_u32_u24 %0, %0, %1", "v"(a)), FIN8)
KERNEL(k_and, DECL8, REP8("v_and_b32 %0, %0, %1", "v"(a)), FIN8)
KERNEL(k_lshr32, DECL8, REP8("v_lshrrev_b32 %0, 3, %0", "v"(a)), FIN8)
KERNEL(k_lshr64, DECL8Q, _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(r[i]));, FIN8Q)
KERNEL(k_alignbit, DECL8, REP8("v_alignbit_b32 %0, %0, %1, 29", "v"(a)), FIN8)
KERNEL(k_cndmask, DECL8, REP8("v_cndmask_b32 %0, %0, %1, vcc", "v"(a)), FIN8)
KERNEL(k_mov, DECL8, REP8("v_mov_b32 %0, %1", "v"(a)), FIN8)
KERNEL(k_lshladd, DECL8, REP8("v_lshl_add_u32 %0, %0, 3, %1", "v"(a)), FIN8)
KERNEL(k_dpp_rowshr, DECL8, REP8("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v"(a)), FIN8)
KERNEL(k_dpp_waveshr, DECL8, REP8("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf", "v"(a)), FIN8)
KERNEL(k_dpp_newbcast, DECL8, REP8("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf", "v"(a)), FIN8)
KERNEL(k_dpp_add_rowshr, DECL8, REP8("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v"(a)), FIN8)
KERNEL(k_fma64, double a = 1.0 + 1e-9 * threadIdx.x; double b = 1e-9 * b0; double r[8]; for (int i = 0; i < 8; i++) r[i] = i + threadIdx.x + a0;,
  _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));,
  { double s = 0; for (int i = 0; i < 8; i++) s += r[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s; })
KERNEL(k_fma32, float a = 1.0f + 1e-6f * threadIdx.x; float b = 1e-6f * b0; float r[8]; for (int i = 0; i < 8; i++) r[i] = i + threadIdx.x + a0;,
  _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));,
  { float s = 0; for (int i = 0; i < 8; i++) s += r[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s; })
// readlane with constant lane and s_mul chain (SALU + VALU mix): 8 x (readlane -> s_mul_i32 -> v_add with sgpr)
KERNEL(k_readlane_const, DECL8, _Pragma("unroll") for (int i = 0; i < 8; i++) { uint32_t s; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(r[i])); asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[(i+1)&7]) : "s"(s)); }, FIN8)
KERNEL(k_readfirstlane, DECL8, _Pragma("unroll") for (int i = 0; i < 8; i++) { uint32_t s; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(r[i])); asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[(i+1)&7]) : "s"(s)); }, FIN8)
KERNEL(k_smul, uint32_t s0 = a0; uint32_t s1 = b0; uint32_t s2 = a0 ^ 5; uint32_t s3 = b0 + 9; uint32_t r0 = threadIdx.x;,
  asm volatile("s_mul_i32 %0, %0, %4\n\ts_mul_i32 %1, %1, %4\n\ts_mul_i32 %2, %2, %4\n\ts_mul_i32 %3, %3, %4\n\ts_mul_hi_u32 %0, %0, %4\n\ts_mul_hi_u32 %1, %1, %4\n\ts_mul_hi_u32 %2, %2, %4\n\ts_mul_hi_u32 %3, %3, %4" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(b0));,
  out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1 + s2 + s3 + r0;)

struct Res { double med, mn; };
template <typename K>
int run(const char* name, K kern, int instr_per_iter, uint32_t* dout, unsigned long long* dcyc) {
  for (int w : {1, 2, 4, 8}) {
    int grid = 256 * w, block = 256;
    int nw = grid * 4;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, dout, dcyc, 12345u, 67890u);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, dout, dcyc, 12345u, 67890u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nw);
    CK(hipMemcpy(h.data(), dcyc, nw * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    double med = (double)h[nw / 2] / ((double)ITER * instr_per_iter);
    // per-SIMD issue cost = wave cycles per instr / waves per SIMD
    printf("%-18s w/SIMD=%d  wave-cyc/instr=%.2f  =>SIMD-cyc/instr=%.2f  wall=%.3f ms  (eff clock via wall: %.2f GHz)\n",
           name, w, med, med / w, ms, (double)h[nw/2] / (ms * 1e-3) / 1e9);
  }
  return 0;
}

int main() {
  uint32_t* dout; CK(hipMalloc(&dout, 256 * 8 * 256 * 4));
  unsigned long long* dcyc; CK(hipMalloc(&dcyc, 256 * 8 * 4 * 8));
#define R(k, n) run(#k, k, n, dout, dcyc)
  R(k_add_vop2, 8); R(k_add_vop3, 8); R(k_add3, 8); R(k_addco_vcc, 8); R(k_addco_sgpr, 8); R(k_addc_vcc, 8); R(k_addc_sgpr, 8);
  R(k_mad64, 8); R(k_mad64_sgprsrc, 8); R(k_mad64_addc_sg, 16); R(k_mad64x8_addcx8, 16);
  R(k_mullo, 8); R(k_mulhi, 8); R(k_mad24, 8); R(k_mul24, 8); R(k_mulhi24, 8);
  R(k_and, 8); R(k_lshr32, 8); R(k_lshr64, 8); R(k_alignbit, 8); R(k_cndmask, 8); R(k_mov, 8); R(k_lshladd, 8);
  R(k_dpp_rowshr, 8); R(k_dpp_waveshr, 8); R(k_dpp_newbcast, 8); R(k_dpp_add_rowshr, 8);
  R(k_fma64, 8); R(k_fma32, 8); R(k_readlane_const, 16); R(k_readfirstlane, 16); R(k_smul, 8);
  return 0;
}
