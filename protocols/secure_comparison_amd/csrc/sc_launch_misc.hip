// The remaining kernels of the library -- the division-step inversion kernel, the device generator, the plain-word kernels of the
// protocol steps and the multiply-add probe -- behind plain launch functions (sc_internal.h).
#include "sc_internal.h"
#include "sc_kernel_plain.h"
#include "sc_rng.h"
#include "sc_xgcd.h"

using namespace sc;

namespace {
inline int launched() { return hipGetLastError() == hipSuccess ? 0 : -1; }
}  // namespace

int sc_host::launch_xgcd(hipStream_t stream, const uint32_t* x, uint32_t* out, const uint32_t* d_n, int nw, uint64_t count, int* d_status) {
  return sc::launch_xgcd(stream, x, out, d_n, nw, count, d_status);
}
int sc_host::launch_plain_alice(hipStream_t stream, const uint32_t* r, const uint32_t* nmod, const uint32_t* halfn, int nw, int l, uint64_t count,
                                uint32_t* m1, uint64_t* alpha, uint64_t* alpha_tilde, uint64_t* rsmall, uint32_t* rshift) {
  hipLaunchKernelGGL(k_plain_alice, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, r, nmod, halfn, nw, l, count, m1, alpha, alpha_tilde, rsmall, rshift);
  return launched();
}
int sc_host::launch_plain_bob(hipStream_t stream, const uint32_t* z, const uint32_t* nmod, const uint32_t* halfn, int nw, int l, uint64_t count,
                              uint64_t* beta, uint64_t* dbit, uint32_t* zeta1, uint32_t* zeta2, uint8_t* bits) {
  hipLaunchKernelGGL(k_plain_bob, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, z, nmod, halfn, nw, l, count, beta, dbit, zeta1, zeta2, bits);
  return launched();
}
int sc_host::launch_rng_bits(hipStream_t stream, const RngKey& key, uint64_t call, int bits, int nw, uint32_t* out, uint64_t count) {
  hipLaunchKernelGGL(k_rng_bits, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, key, call, bits, nw, out, count);
  return launched();
}
int sc_host::launch_rng_below(hipStream_t stream, const RngKey& key, uint64_t call, const uint32_t* d_n, int nbits, int nw, int nonzero, uint32_t* out,
                              uint64_t count) {
  hipLaunchKernelGGL(k_rng_below, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, key, call, d_n, nbits, nw, nonzero, out, count);
  return launched();
}
int sc_host::launch_rng_coins(hipStream_t stream, const RngKey& key, uint64_t call, uint64_t* out, uint64_t count) {
  const uint64_t threads = (count + 511) / 512;          // one thread per 512 coins (a keystream block holds 512 bits)
  hipLaunchKernelGGL(k_rng_coins, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, key, call, out, count);
  return launched();
}
int sc_host::launch_rng_perm(hipStream_t stream, const RngKey& key, uint64_t call, int k, int64_t* out, uint64_t count) {
  hipLaunchKernelGGL(k_rng_perm, dim3((unsigned)((count + 63) / 64)), dim3(64), (size_t)64 * k, stream, key, call, k, out, count);
  return launched();
}
int sc_host::launch_peak_probe(hipStream_t stream, int grid, uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  hipLaunchKernelGGL(k_peak_probe, dim3(grid), dim3(256), 0, stream, out, a0, b0, iters);
  return launched();
}
