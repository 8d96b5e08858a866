// Where does the dispatcher put the waves of an under-filling launch (768 single-wave workgroups) when the chip is empty, and
// when another stream's launch (2048 single-wave workgroups, 128 VGPRs, 12 KB of LDS each: two waves on every SIMD) is already
// resident?  And the same 768 waves as 192 workgroups of four waves.  Every wave records its HW_ID / XCC_ID and then spins.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/placement.hip -o /tmp/placement && /tmp/placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int LDSW>
__device__ __forceinline__ void body(uint32_t* out, int spin, uint32_t* lds) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  lds[threadIdx.x % LDSW] = hw;
  uint64_t acc = threadIdx.x;
  uint32_t a = hw | 1u;
  for (int i = 0; i < spin; i++) { acc = (uint64_t)a * (uint32_t)acc + acc; a += lds[(threadIdx.x + i) % LDSW] & 1u; }
  if ((threadIdx.x & 63) == 0) { out[2 * wave] = hw; out[2 * wave + 1] = xcc; }
  if (acc == 0x1234567ull) out[0] = 0;
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(128))) k_filler(uint32_t* out, int spin) {
  __shared__ uint32_t lds[3072];
  body<3072>(out, spin, lds);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(80))) k_small1(uint32_t* out, int spin) {
  __shared__ uint32_t lds[1664];
  body<1664>(out, spin, lds);
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(80))) k_small4(uint32_t* out, int spin) {
  __shared__ uint32_t lds[4 * 1664];
  body<4 * 1664>(out, spin, lds);
}

static void report(const char* name, const std::vector<uint32_t>& h, int waves, float ms) {
  // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] ; XCC_ID [3:0]
  std::map<uint32_t, int> per_simd, per_cu;
  for (int w = 0; w < waves; w++) {
    const uint32_t hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
    const uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const uint32_t cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    per_cu[cukey]++; per_simd[(cukey << 2) | simd]++;
  }
  std::map<int, int> hist_simd, hist_cu;
  for (auto& kv : per_simd) hist_simd[kv.second]++;
  for (auto& kv : per_cu) hist_cu[kv.second]++;
  printf("%-70s %7.2f ms | CUs used %3zu, SIMDs used %4zu | waves per used SIMD:", name, ms, per_cu.size(), per_simd.size());
  for (auto& kv : hist_simd) printf(" %dx%d", kv.first, kv.second);
  printf(" | per used CU:");
  for (auto& kv : hist_cu) printf(" %dx%d", kv.first, kv.second);
  printf("\n");
}

int main() {
  uint32_t *o1, *o2;
  CK(hipMalloc(&o1, 8 * 4096 * 4)); CK(hipMalloc(&o2, 8 * 4096 * 4));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<uint32_t> h(2 * 4096);
  const int SPIN_SMALL = 60000, SPIN_FILL = 400000;
  for (int scen = 0; scen < 6; scen++) {
    const bool filler = scen >= 2 && scen < 4, filler_after = scen >= 4, four = scen & 1;
    CK(hipDeviceSynchronize());
    if (filler) { hipLaunchKernelGGL(k_filler, dim3(2048), dim3(64), 0, s2, o2, SPIN_FILL); CK(hipStreamQuery(s2) == hipErrorNotReady ? hipSuccess : hipSuccess); 
      // give the filler time to become resident
      for (volatile int spin = 0; spin < 3000000; spin++) {} }
    CK(hipEventRecord(e0, s1));
    if (four) hipLaunchKernelGGL(k_small4, dim3(192), dim3(256), 0, s1, o1, SPIN_SMALL);
    else hipLaunchKernelGGL(k_small1, dim3(768), dim3(64), 0, s1, o1, SPIN_SMALL);
    CK(hipEventRecord(e1, s1));
    if (filler_after) hipLaunchKernelGGL(k_filler, dim3(2048), dim3(64), 0, s2, o2, SPIN_FILL / 4);
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), o1, 2 * 768 * 4, hipMemcpyDeviceToHost));
    char name[128];
    snprintf(name, sizeof name, "768 waves as %s, %s", four ? "192 workgroups x 4 waves" : "768 workgroups x 1 wave", filler ? "placed on a chip holding 2048 filler waves" : (filler_after ? "placed first, filler launched right after" : "alone"));
    report(name, h, 768, ms);
    if (filler) { CK(hipMemcpy(h.data(), o2, 2 * 2048 * 4, hipMemcpyDeviceToHost)); report("   (the filler's own 2048 waves)", h, 2048, 0.f); }
  }
  return 0;
}
