// Instances of the pair interpreter k_pvm and their launcher.  Compiled three times (-DSC_PART=0/1/2), see sc_launch_vm.hip.
#include "sc_internal.h"
#include "sc_kernel_pvm.h"

#ifndef SC_PART
#error "compile with -DSC_PART=0, 1 or 2"
#endif

using namespace sc;

namespace {

template <int G, int L, int WB = 29, bool NEG1 = false, bool STAMP = false>
int pvm_occupancy(sc_ctx* ctx) {
  const int key = 1000 + 100 * L + G + (NEG1 ? 100000 : 0) + (STAMP ? 200000 : 0);
  auto it = ctx->occ_cache.find(key);
  if (it != ctx->occ_cache.end()) return it->second;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (k_pvm<G, L, WB, NEG1, STAMP>), 64, 0) != hipSuccess) return 0;
  const int occ = std::max(1, std::min(nb, 16));
  ctx->occ_cache[key] = occ;
  return occ;
}

template <int G, int L, int WB = 29, bool NEG1 = false, bool STAMP = false>
int launch_pvm_cfg(sc_ctx* ctx, const VmArgs& a) {
  constexpr int NG = 64 / G;
  const int occ = pvm_occupancy<G, L, WB, NEG1, STAMP>(ctx);
  if (occ <= 0) return sc_host::fail(ctx, SC_ERR_HIP, "occupancy query failed for k_pvm<%d,%d>", G, L);
  uint64_t need = (a.count + NG - 1) / NG;
  // slot_per_item (segmented launches of more than one round): one wave and one table slot per group of items, so that what a
  // segment parks in the slot's table is still there for the next one; otherwise a grid-stride loop of the resident waves
  uint32_t grid = (uint32_t)std::max<uint64_t>(1, ctx->slot_per_item ? need : std::min<uint64_t>(need, (uint64_t)ctx->num_cu * occ));
  VmArgs args = a;
  int rc = sc_host::ensure_scratch(ctx, (size_t)grid * NG * a.nscratch * (G * L) * 4, &args.scratch);
  if (rc) return rc;
  if constexpr (STAMP) { args.stamps = ctx->stamps; ctx->stamp_grid = grid; }
  hipLaunchKernelGGL((k_pvm<G, L, WB, NEG1, STAMP>), dim3(grid), dim3(64), 0, ctx->stream, args);
  HIPCHK(ctx, hipGetLastError());
  return SC_OK;
}

}  // namespace

#define SC_CAT_(a, b) a##b
#define SC_CAT(a, b) SC_CAT_(a, b)
#define SC_CASE(GG, LL) if (G == GG && L == LL && !neg1) return launch_pvm_cfg<GG, LL>(ctx, a);
#define SC_CASE_NEG1(GG, LL) if (G == GG && L == LL && neg1 && !stamp) return launch_pvm_cfg<GG, LL, 29, true>(ctx, a);

int sc_host::SC_CAT(launch_pvm_part, SC_PART)(sc_ctx* ctx, int G, int L, bool neg1, bool stamp, const sc::VmArgs& a) {
#if SC_PART == 0
  if (G == 4 && L == 18 && neg1 && stamp) return launch_pvm_cfg<4, 18, 29, true, true>(ctx, a);   // sc_clock_probe's diagnostic twin
  SC_CASE_NEG1(4, 18) SC_CASE(4, 18) SC_CASE(2, 18) SC_CASE(1, 18)
#elif SC_PART == 1
  SC_CASE(8, 18) SC_CASE(16, 18) SC_CASE(2, 9) SC_CASE(4, 9) SC_CASE(8, 9) SC_CASE(4, 5) SC_CASE(8, 5) SC_CASE(16, 5)
#else
  SC_CASE_NEG1(4, 14) SC_CASE(4, 14) SC_CASE_NEG1(8, 14) SC_CASE(8, 14)
#endif
  (void)ctx; (void)a; (void)stamp;
  return SC_ERR_UNSUPPORTED;
}

#undef SC_CASE
#undef SC_CASE_NEG1
#define SC_CASE(GG, LL) if (G == GG && L == LL && !neg1) return pvm_occupancy<GG, LL>(ctx);
#define SC_CASE_NEG1(GG, LL) if (G == GG && L == LL && neg1) return pvm_occupancy<GG, LL, 29, true>(ctx);

int sc_host::SC_CAT(pvm_occupancy_part, SC_PART)(sc_ctx* ctx, int G, int L, bool neg1) {
#if SC_PART == 0
  SC_CASE_NEG1(4, 18) SC_CASE(4, 18) SC_CASE(2, 18) SC_CASE(1, 18)
#elif SC_PART == 1
  SC_CASE(8, 18) SC_CASE(16, 18) SC_CASE(2, 9) SC_CASE(4, 9) SC_CASE(8, 9) SC_CASE(4, 5) SC_CASE(8, 5) SC_CASE(16, 5)
#else
  SC_CASE_NEG1(4, 14) SC_CASE(4, 14) SC_CASE_NEG1(8, 14) SC_CASE(8, 14)
#endif
  (void)ctx;
  return -1;
}
