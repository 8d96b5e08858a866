// Instances of the single-modulus interpreter k_vm and their launcher.  Compiled three times (-DSC_PART=0/1/2), each part holding a
// third of the configurations, so that the parts build in parallel and a kernel change recompiles only what it touches.
#include "sc_internal.h"
#include "sc_kernel_vm.h"

#ifndef SC_PART
#error "compile with -DSC_PART=0, 1 or 2"
#endif

using namespace sc;

namespace {

template <int G, int L, int WB, bool NEG1 = false>
int launch_vm_cfg(sc_ctx* ctx, const VmArgs& a) {
  const int key = 10000 * G + 10 * L + (NEG1 ? 1 : 0) + 1000000 * WB;
  auto it = ctx->occ_cache.find(key);
  int occ;
  if (it == ctx->occ_cache.end()) {
    int nb = 0;
    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_vm<G, L, WB, NEG1>, 64, 0));
    occ = std::max(1, std::min(nb, 16));
    ctx->occ_cache[key] = occ;
  } else {
    occ = it->second;
  }
  constexpr int NG = 64 / G;
  uint64_t need = (a.count + NG - 1) / NG;
  uint64_t maxb = (uint64_t)ctx->num_cu * occ;
  uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min(need, maxb));
  size_t scratch_bytes = (size_t)grid * NG * a.nscratch * (G * L) * 4;
  VmArgs args = a;
  int rc = sc_host::ensure_scratch(ctx, scratch_bytes, &args.scratch);
  if (rc) return rc;
  hipLaunchKernelGGL((k_vm<G, L, WB, NEG1>), dim3(grid), dim3(64), 0, ctx->stream, args);
  HIPCHK(ctx, hipGetLastError());
  return SC_OK;
}

}  // namespace

#define SC_CAT_(a, b) a##b
#define SC_CAT(a, b) SC_CAT_(a, b)
#define SC_CASE(GG, LL, WW) if (G == GG && L == LL && W == WW && !neg1) return launch_vm_cfg<GG, LL, WW>(ctx, a);

int sc_host::SC_CAT(launch_vm_part, SC_PART)(sc_ctx* ctx, int G, int L, int W, bool neg1, const sc::VmArgs& a) {
#if SC_PART == 0
  if (G == 4 && L == 18 && W == 29 && neg1) return launch_vm_cfg<4, 18, 29, true>(ctx, a);     // modulus = -1 (mod 2^29): no quotient multiply
  SC_CASE(4, 18, 29) SC_CASE(2, 18, 29) SC_CASE(1, 18, 29) SC_CASE(2, 9, 29) SC_CASE(4, 9, 29)
#elif SC_PART == 1
  SC_CASE(1, 37, 28) SC_CASE(8, 18, 29) SC_CASE(16, 18, 29) SC_CASE(8, 9, 29) SC_CASE(16, 9, 29) SC_CASE(2, 27, 29)
#else
  SC_CASE(4, 27, 29) SC_CASE(8, 27, 29) SC_CASE(4, 14, 29) SC_CASE(8, 14, 29) SC_CASE(16, 14, 29)
#endif
  (void)ctx; (void)a;
  return SC_ERR_UNSUPPORTED;
}
