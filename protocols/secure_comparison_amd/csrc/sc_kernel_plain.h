// Plain-word helper kernels of the protocol steps (gfx950).  Included by sc_launch_misc.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sc {

// ---------------------------------------------------------------------------------------------
// Plain-integer helper kernels on canonical 32-bit words (HBM-bound, one thread per item).
// ---------------------------------------------------------------------------------------------
// Alice's plaintext-side values derived from r (SC/initiator.py:250-256, :270, :289, :373, :558-562):
//   m1 = 2^l + r (as nw+1 words), alpha = r mod 2^l, alpha_tilde = (r - N) mod 2^l,
//   rsmall = [r < (N-1)/2], rshift = r >> l.
__global__ void k_plain_alice(const uint32_t* __restrict__ r, const uint32_t* __restrict__ nmod,
                              const uint32_t* __restrict__ halfn /* (N-1)/2 */, int nw, int l, uint64_t count,
                              uint32_t* __restrict__ m1, uint64_t* __restrict__ alpha,
                              uint64_t* __restrict__ alpha_tilde, uint64_t* __restrict__ rsmall,
                              uint32_t* __restrict__ rshift) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t* ri = r + i * nw;
  const uint64_t lmask = (l >= 64) ? ~0ull : ((1ull << l) - 1);
  const uint64_t rlow = (uint64_t)ri[0] | ((nw > 1) ? ((uint64_t)ri[1] << 32) : 0ull);
  const uint64_t nlow = (uint64_t)nmod[0] | ((nw > 1) ? ((uint64_t)nmod[1] << 32) : 0ull);
  alpha[i] = rlow & lmask;
  alpha_tilde[i] = (rlow - nlow) & lmask;
  int cmp = 0;  // r ? halfn
  for (int k = nw - 1; k >= 0 && cmp == 0; k--) cmp = (ri[k] > halfn[k]) ? 1 : ((ri[k] < halfn[k]) ? -1 : 0);
  rsmall[i] = (cmp < 0) ? 1ull : 0ull;
  // m1 = r + 2^l  (nw + 1 words)
  uint64_t carry = 0;
  for (int k = 0; k <= nw; k++) {
    uint64_t v = (k < nw ? (uint64_t)ri[k] : 0ull) + carry + ((k == (l >> 5)) ? (1ull << (l & 31)) : 0ull);
    m1[i * (nw + 1) + k] = (uint32_t)v;
    carry = v >> 32;
  }
  // rshift = r >> l
  const int ws = l >> 5, bs = l & 31;
  for (int k = 0; k < nw; k++) {
    const uint64_t lo = (k + ws < nw) ? ri[k + ws] : 0u, hi = (k + ws + 1 < nw) ? ri[k + ws + 1] : 0u;
    rshift[i * nw + k] = (uint32_t)(((hi << 32) | lo) >> bs);
  }
}

// Bob's plaintext-side values derived from z (SC/keyholder.py:196, :213, :274-282):
//   beta = z mod 2^l, dbit = [z < (N-1)/2], zeta1 = z >> l, zeta2 = (z + N) >> l if dbit else z >> l.
//   bits (nullable): the plaintext bits of steps 4a / 4b as bytes, bit-major [l+1][count]: plane 0 = d, plane 1 + i = bit i of beta
//   (SC/keyholder.py:213, 230-233) -- what the g^bit selection of the DGK encryption launch reads.
__global__ void k_plain_bob(const uint32_t* __restrict__ z, const uint32_t* __restrict__ nmod,
                            const uint32_t* __restrict__ halfn, int nw, int l, uint64_t count,
                            uint64_t* __restrict__ beta, uint64_t* __restrict__ dbit, uint32_t* __restrict__ zeta1,
                            uint32_t* __restrict__ zeta2, uint8_t* __restrict__ bits) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t* zi = z + i * nw;
  const uint64_t lmask = (l >= 64) ? ~0ull : ((1ull << l) - 1);
  const uint64_t zlow = (uint64_t)zi[0] | ((nw > 1) ? ((uint64_t)zi[1] << 32) : 0ull);
  beta[i] = zlow & lmask;
  int cmp = 0;
  for (int k = nw - 1; k >= 0 && cmp == 0; k--) cmp = (zi[k] > halfn[k]) ? 1 : ((zi[k] < halfn[k]) ? -1 : 0);
  const bool d = cmp < 0;
  dbit[i] = d ? 1ull : 0ull;
  if (bits) {
    bits[i] = d ? 1 : 0;
    for (int k = 0; k < l; k++) bits[(uint64_t)(k + 1) * count + i] = (uint8_t)((zlow >> k) & 1);
  }
  const int ws = l >> 5, bs = l & 31;
  // zeta2: first the sum z + (d ? N : 0) (nw words + a carry word), then an in-place
  // ascending funnel shift (word o only reads words >= o).
  uint32_t* z2 = zeta2 + i * nw;
  uint64_t carry = 0;
  for (int k = 0; k < nw; k++) {
    const uint64_t v = (uint64_t)zi[k] + (d ? nmod[k] : 0u) + carry;
    z2[k] = (uint32_t)v;
    carry = v >> 32;
  }
  const uint32_t top = (uint32_t)carry;  // word nw of the sum (z + N may exceed nw words)
  for (int k = 0; k < nw; k++) {
    const int a = k + ws, b = k + ws + 1;
    const uint64_t lo = (a < nw) ? z2[a] : ((a == nw) ? top : 0u), hi = (b < nw) ? z2[b] : ((b == nw) ? top : 0u);
    z2[k] = (uint32_t)(((hi << 32) | lo) >> bs);
  }
  for (int k = 0; k < nw; k++) {
    const uint64_t lo = (k + ws < nw) ? zi[k + ws] : 0u, hi = (k + ws + 1 < nw) ? zi[k + ws + 1] : 0u;
    zeta1[i * nw + k] = (uint32_t)(((hi << 32) | lo) >> bs);
  }
}

}  // namespace sc
