// The pair interpreter k_pvm (gfx950): arithmetic modulo n^2 with products modulo n.  Included by sc_launch_pvm.hip.
#pragma once
#include "sc_device.h"
#include "sc_vm.h"

namespace sc {
#ifndef SC_PAIR_SQR_INTERLEAVED
#define SC_PAIR_SQR_INTERLEAVED 1   // measurement switch: 0 = the two passes of a small-batch pair squaring one after the other
#endif
#ifndef SC_PVM_WAVES
#define SC_PVM_WAVES 2    // waves per SIMD the pair interpreter is compiled for
#endif

// ---------------------------------------------------------------------------------------------
// The pair interpreter: exponentiation modulo n^2 carried out with Montgomery products modulo n only (sc_device.h,
// "pair arithmetic").  Same launch geometry and argument block as k_vm; compiled for the L = 18 configurations.
// ---------------------------------------------------------------------------------------------
// STAMP: diagnostic twin (sc_clock_probe, never a timed launch): every wave records s_memtime (shader clock) and s_memrealtime
// (constant-rate clock) at entry and exit in args.stamps[4 * blockIdx.x ..], from which the host derives the engine clock the
// kernel actually held.
template <int G, int L, int WB, bool NEG1 = false, bool STAMP = false>
__global__ void __launch_bounds__(64, ((G == 16 && L > 9) ? 1 : SC_PVM_WAVES)) k_pvm(const VmArgs args) {
  using GT = Grp<G, L, WB, NEG1>;
  uint64_t stamp_c0 = 0, stamp_r0 = 0;
  if constexpr (STAMP) { stamp_c0 = __builtin_amdgcn_s_memtime(); stamp_r0 = __builtin_amdgcn_s_memrealtime(); }
  constexpr int S = GT::S, NG = GT::NG, SP = GT::SP, WP = GT::WP;
  __shared__ uint32_t s_a[NG * SP];            // first LDS-side operand  (x0, or y0)
  __shared__ uint32_t s_a2[G == 1 ? 1 : NG * SP];  // 2 * x0 (squarings) or y1 (products); also the word scratch of PV_LOADU
                                               // (one-lane numbers: squarings out of registers, products restage one area)
  __shared__ uint32_t s_c[VM_MAX_CONST * SP];
  static_assert(WP <= SP, "word scratch must fit the staging area it aliases");

  GT gp;
  gp.init(args.modctx, args.n0inv);
  uint32_t* const my_a = s_a + gp.g * SP;
  uint32_t* const my_a2 = (G == 1) ? my_a : s_a2 + gp.g * SP;
  uint32_t* const my_w = my_a2;
  for (int t = threadIdx.x; t < 2 * S; t += 64) s_c[(t / S) * SP + (t % S)] = args.modctx[S + t];
  for (int t = threadIdx.x; t < (int)args.nconst_extra * S; t += 64)
    s_c[(2 + t / S) * SP + (t % S)] = args.consts[t];
  SC_WAVE_SYNC();

  constexpr int TS = (G == 1) ? 64 : 1;      // element stride of the slot's table rows (k_vm: one-lane rows interleave by lane)
  const uint64_t slot = (uint64_t)blockIdx.x * NG + gp.g;
  uint32_t* const my_tbl = (G == 1) ? args.scratch + (uint64_t)blockIdx.x * NG * args.nscratch * S + gp.g
                                    : args.scratch + slot * (uint64_t)args.nscratch * S;

  for (uint64_t base = (uint64_t)blockIdx.x * NG; base < args.count; base += (uint64_t)gridDim.x * NG) {
    const bool live = base + gp.g < args.count;
    const uint64_t idx = live ? base + gp.g : args.count - 1;
    uint32_t x0[L], x1[L];
#pragma unroll
    for (int l = 0; l < L; l++) { x0[l] = 0; x1[l] = 0; }

    // G <= 2 (the half-size moduli: a pair squaring is half as long as at G = 4): the next micro-op is fetched -- a scalar load --
    // while this one runs, its latency hides behind the staging of the operands.  Measured on the MI355X, alternating builds:
    // x^p mod p^2 on k_pvm<2,18> 43.4 -> 43.0 ms, the key holder's decryption 29.5 -> 29.3 ms; k_pvm<4,18> 104.25 -> 104.5 ms, hence not there.
    constexpr bool PREFETCH_OP = (G <= 2);
    VmOp nxt = args.prog[0];
#pragma unroll 1
    for (uint32_t pc = 0; pc < args.nops; pc++) {
      VmOp op;
      if constexpr (PREFETCH_OP) { op = nxt; nxt = args.prog[pc + 1 < args.nops ? pc + 1 : pc]; } else { op = args.prog[pc]; }
      const uint32_t opc = op.w0 & 0xff;
      switch (opc) {
        case PV_LOADU: {
          const VmExt& e = args.ext[op.w1 & 0xf];
          const uint64_t flat = (uint64_t)op.w2 * args.count + idx;
          const uint32_t woff = op.w3 >> 16;
          const uint32_t nw = (op.w3 & 0xffff) ? (op.w3 & 0xffff) : e.nwords;
          gp.load_words(x0, (const uint32_t*)e.ptr + flat * e.stride + woff, nw, my_w);
#pragma unroll
          for (int l = 0; l < L; l++) x1[l] = 0;
          break;
        }
        case PV_MULC:
        case PV_MULT: {
          if constexpr (G == 1) {
            // one code path for both (the product pass exists once): the operand pair is copied into the staging area pass by pass
            const bool cst = opc == PV_MULC;
            const uint32_t* src = cst ? (const uint32_t*)(s_c + op.w1 * SP) : (const uint32_t*)(my_tbl + (uint64_t)(2 * op.w1) * S * TS);
            gp.template pair_mul_seq<TS>(x0, x1, src, src + (cst ? SP : S * TS), cst ? 1 : TS, my_a, my_tbl + (uint64_t)(args.nscratch - 1) * S * TS);
          } else if (opc == PV_MULC) {
            gp.pair_mul(x0, x1, s_c + op.w1 * SP, s_c + (op.w1 + 1) * SP);
          } else {
            const uint32_t* src = my_tbl + (uint64_t)(2 * op.w1) * S;
            SC_WAVE_SYNC();
#pragma unroll
            for (int l = 0; l < L; l++) {
              my_a[gp.j * L + l] = src[gp.j * L + l];
              my_a2[gp.j * L + l] = src[S + gp.j * L + l];
            }
            SC_WAVE_SYNC();
            gp.pair_mul(x0, x1, my_a, my_a2);
          }
          break;
        }
        case PV_SQR: {
          if constexpr (G == 1) {
            gp.pair_sqr_r(x0, x1, my_a);
          } else {
            SC_WAVE_SYNC();
            gp.stage(my_a, x0);
            gp.stage_doubled(my_a2, x0);
            SC_WAVE_SYNC();
            // small-batch configurations: the two passes interleaved -- the chain's latency is what a squaring costs there
            if constexpr (L <= 9 && SC_PAIR_SQR_INTERLEAVED) gp.pair_sqr_il(x0, x1, my_a, my_a2); else
            gp.pair_sqr(x0, x1, my_a, my_a2);
          }
          break;
        }
        case PV_STT: {
          uint32_t* dst = my_tbl + (uint64_t)(2 * op.w1) * S * TS;
          gp.template store_limbs<TS>(dst, x0);
          gp.template store_limbs<TS>(dst + S * TS, x1);
          break;
        }
        case PV_LOADT: {
          const uint32_t* src = my_tbl + (uint64_t)(2 * op.w1) * S * TS;
          gp.template load_limbs<TS>(x0, src);
          gp.template load_limbs<TS>(x1, src + S * TS);
          break;
        }
        case PV_ADDT: {
          const uint32_t* src = my_tbl + (uint64_t)(2 * op.w1) * S * TS;
          uint32_t t0[L], t1[L];
          gp.template load_limbs<TS>(t0, src);
          gp.template load_limbs<TS>(t1, src + S * TS);
#pragma unroll
          for (int l = 0; l < L; l++) { x0[l] += t0[l]; x1[l] += t1[l]; }
          gp.renorm(x0);
          gp.renorm(x1);
          break;
        }
        case PV_OUT: {
          gp.pair_redc(x0, x1);
          uint32_t zero[L];
#pragma unroll
          for (int l = 0; l < L; l++) zero[l] = 0;
          gp.normalize(x0, zero);
          gp.normalize(x1, zero);
          const VmExt& e0 = args.ext[op.w1 & 0xf];
          const VmExt& e1 = args.ext[op.w2 & 0xf];
          gp.store_words((uint32_t*)e0.ptr + idx * e0.stride, e0.nwords, x0, my_a, live);
          gp.store_words((uint32_t*)e1.ptr + idx * e1.stride, e1.nwords, x1, my_a, live);
          break;
        }
        default: break;
      }
    }
  }
  if constexpr (STAMP) {
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
      uint64_t* o = args.stamps + 4 * (uint64_t)blockIdx.x;
      o[0] = stamp_c0; o[1] = stamp_r0; o[2] = c1; o[3] = r1;
    }
  }
}

}  // namespace sc
