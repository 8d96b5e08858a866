// The single-modulus micro-op interpreter k_vm (gfx950).  Included by sc_launch_vm.hip, which instantiates its configurations.
#pragma once
#include "sc_device.h"
#include "sc_vm.h"

namespace sc {
#ifndef SC_VM_WAVES
#define SC_VM_WAVES 2     // waves per SIMD the single-modulus interpreter is compiled for (L <= 18 configurations)
#endif

// ---------------------------------------------------------------------------------------------
// The micro-op interpreter.  One 64-lane workgroup (= one wave) processes 64/G items per pass of
// the program and grid-strides over the batch.
// ---------------------------------------------------------------------------------------------
#ifndef SC_L27_WAVES
#define SC_L27_WAVES 2    // waves per SIMD of the L = 27 configurations (1536 / 3072 / 6144-bit moduli): 256 VGPRs, the 120 spill
                          // instructions all outside the product loops; +2 % on configs[4] with 3072-bit DGK over one wave + AGPR copies
#endif
template <int G, int L, int WB, bool NEG1 = false>
__global__ void __launch_bounds__(64, (G == 1 && L == 37) ? 2 : (L == 27 ? SC_L27_WAVES : ((L > 18 || (G == 16 && L > 14)) ? 1 : SC_VM_WAVES))) k_vm(const VmArgs args) {
  using GT = Grp<G, L, WB, NEG1>;
  constexpr int S = GT::S, NG = GT::NG, SP = GT::SP, WP = GT::WP;
  __shared__ uint32_t s_a[NG * SP];            // per-group staging area for the LDS-side operand
  __shared__ uint32_t s_a2[G == 1 ? 1 : NG * SP];  // the same operand doubled (squarings of the multi-lane forms); also the
                                               // 32-bit-word scratch of the format conversions (WP <= SP; never live at the same
                                               // time).  One-lane numbers square out of registers: no second area
  __shared__ uint32_t s_c[VM_MAX_CONST * SP];  // modulus constants shared by all groups
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

  // the slot's scratch table: rows of S limbs.  Multi-lane forms: one number's rows are contiguous (element stride 1).  One-lane
  // form: the 64 numbers of the wave interleave, [row][limb][lane] (element stride 64), so that the wave's access to a limb is
  // one contiguous 256-byte request instead of 64 scattered words
  constexpr int TS = (G == 1) ? 64 : 1;
  const uint64_t slot = (uint64_t)blockIdx.x * NG + gp.g;
  uint32_t* const my_tbl = (G == 1) ? args.scratch + (uint64_t)blockIdx.x * NG * args.nscratch * S + gp.g
                                    : args.scratch + slot * (uint64_t)args.nscratch * S;

  for (uint64_t base = (uint64_t)blockIdx.x * NG; base < args.count; base += (uint64_t)gridDim.x * NG) {
    const bool live = base + gp.g < args.count;
    const uint64_t idx = live ? base + gp.g : args.count - 1;
    uint32_t acc[L];
#pragma unroll
    for (int l = 0; l < L; l++) acc[l] = 0;

#pragma unroll 1
    for (uint32_t pc = 0; pc < args.nops; pc++) {
      const VmOp op = args.prog[pc];
      const uint32_t opc = op.w0 & 0xff, akind = (op.w0 >> 8) & 0xf, imm = op.w0 >> 16;

      // ---- resolve a limb-form / word-form source described by akind (used by MUL and LOADT)
      const uint32_t* a_ptr = my_a;          // LDS pointer handed to the multiplier
      const uint32_t* src_limbs = nullptr;   // global limb-form source (copied to registers / LDS)
      int src_ts = 1;                        // its element stride (TS for rows of the slot's table)
      bool src_is_one = false;
      if (opc == OP_MUL || opc == OP_LOADT) {
        if (akind == AK_CONST) {
          a_ptr = s_c + op.w1 * SP;
        } else if (akind == AK_CONSTSEL) {   // one of two LDS constants, chosen per item by a byte flag (control flow stays uniform)
          const VmExt& fe = args.ext[op.w1 & 0xf];                                            // stride: BYTES between items (0 = 1; 8 = the low byte of u64 flags)
          const uint8_t f = ((const uint8_t*)fe.ptr)[idx * (fe.stride ? fe.stride : 1u)];
          a_ptr = s_c + ((f ? (op.w2 >> 8) : op.w2) & 0xff) * SP;
        } else if (akind == AK_TBL) {
          src_limbs = my_tbl + (uint64_t)op.w1 * S * TS;
          src_ts = TS;
        } else if (akind == AK_TBLSEL) {
          const VmExt& ea = args.ext[op.w1 & 0xf];
          const VmExt& eb = args.ext[(op.w1 >> 12) & 0xf];
          const uint64_t fa = ((const uint64_t*)ea.ptr)[idx] >> ((op.w1 >> 4) & 0xff);
          const uint64_t fb = ((const uint64_t*)eb.ptr)[idx] >> ((op.w1 >> 16) & 0xff);
          const uint32_t sel = (uint32_t)((fa & 1) * 2 + (fb & 1));
          src_limbs = my_tbl + (uint64_t)((op.w2 >> (8 * sel)) & 0xff) * S * TS;
          src_ts = TS;
        } else if (akind == AK_TBLDIG || akind == AK_FBT) {
          const VmExt& e = args.ext[op.w1 & 0xf];
          const uint32_t bitpos = (op.w1 >> 4) & 0xfffff, width = op.w1 >> 24;
          const uint32_t* ew = (const uint32_t*)e.ptr + idx * e.stride;
          const uint32_t w0i = bitpos >> 5, sh = bitpos & 31;
          uint64_t v = (w0i < e.nwords) ? ew[w0i] : 0u;
          if (w0i + 1 < e.nwords) v |= (uint64_t)ew[w0i + 1] << 32;
          const uint32_t digit = (uint32_t)(v >> sh) & ((1u << width) - 1);
          src_limbs = (akind == AK_TBLDIG) ? my_tbl + (uint64_t)(op.w2 + digit) * S * TS
                                           : args.fbt + (((uint64_t)op.w2 << width) + digit) * S;
          src_ts = (akind == AK_TBLDIG) ? TS : 1;
        } else if (akind == AK_EXTL) {
          const VmExt& e = args.ext[op.w1 & 0xf];
          const uint64_t flat = (uint64_t)op.w2 * args.count + idx;
          if (flat < e.limit) src_limbs = (const uint32_t*)e.ptr + flat * e.stride; else src_is_one = true;
        }
      }

      switch (opc) {
        case OP_MUL: {
          if (akind == AK_ACC) {
            // imm = number of consecutive squarings (the host merges runs): the loop stays inside this case, so a run pays
            // the fetch / decode of one micro-op
#pragma unroll 1
            for (uint32_t rep = (imm ? imm : 1); rep > 0; rep--) {
              uint32_t r[L];
              if constexpr (G == 1) {
                gp.template mont_r<3>(r, acc, acc, r, r);     // both operands in registers
              } else {
                SC_WAVE_SYNC();
                gp.stage(my_a, acc);
                gp.stage_doubled(my_a2, acc);
                SC_WAVE_SYNC();
                gp.sqr(r, my_a, my_a2, acc);
              }
#pragma unroll
              for (int l = 0; l < L; l++) acc[l] = r[l];
            }
            break;
          }
          SC_WAVE_SYNC();
          if (akind == AK_EXTW) {
            const VmExt& e = args.ext[op.w1 & 0xf];
            const uint64_t flat = (uint64_t)op.w2 * args.count + idx;
            uint32_t t[L];
            const bool ok = flat < e.limit;  // may differ per group: keep the barriers in load_words uniform
            gp.load_words(t, (const uint32_t*)e.ptr + (ok ? flat : 0) * e.stride, e.nwords, my_w);
#pragma unroll
            for (int l = 0; l < L; l++) t[l] = ok ? t[l] : ((gp.j == 0 && l == 0) ? 1u : 0u);
            gp.stage(my_a, t);
          } else if (src_limbs != nullptr) {
#pragma unroll
            for (int l = 0; l < L; l++) my_a[gp.j * L + l] = src_limbs[(gp.j * L + l) * src_ts];
          } else if (src_is_one) {
            a_ptr = s_c + 1 * SP;  // Montgomery one: multiplying by it is the identity
          }
          SC_WAVE_SYNC();
          uint32_t r[L];
          gp.mul(r, a_ptr, acc);
#pragma unroll
          for (int l = 0; l < L; l++) acc[l] = r[l];
          break;
        }
        case OP_LOADT: {
          if (akind == AK_CONST || akind == AK_CONSTSEL || src_is_one) {
            const uint32_t* c = src_is_one ? s_c + 1 * SP : a_ptr;
#pragma unroll
            for (int l = 0; l < L; l++) acc[l] = c[gp.j * L + l];
          } else {
#pragma unroll
            for (int l = 0; l < L; l++) acc[l] = src_limbs[(gp.j * L + l) * src_ts];
          }
          break;
        }
        case OP_LOADW:
        case OP_ADDW: {
          // imm != 0: one of two arrays, chosen per item by bit (w1 >> 12) of the u64 flag ext (w1 >> 8) & 15: set -> ext w1 & 15,
          // clear -> ext (w1 >> 4) & 15 (same shape; the step formulas' "a if flag else b" without a select pass over the batch)
          const bool pick = imm && ((((const uint64_t*)args.ext[(op.w1 >> 8) & 0xf].ptr)[idx] >> ((op.w1 >> 12) & 0x3f)) & 1) == 0;
          const VmExt& e = args.ext[(pick ? (op.w1 >> 4) : op.w1) & 0xf];
          const uint64_t flat = (uint64_t)op.w2 * args.count + idx;
          const uint32_t woff = op.w3 >> 16;
          const uint32_t nw = (op.w3 & 0xffff) ? (op.w3 & 0xffff) : e.nwords;
          uint32_t t[L];
          const bool ok = flat < e.limit;
          gp.load_words(t, (const uint32_t*)e.ptr + (ok ? flat : 0) * e.stride + woff, nw, my_w);
#pragma unroll
          for (int l = 0; l < L; l++) t[l] = ok ? t[l] : ((gp.j == 0 && l == 0) ? 1u : 0u);
#pragma unroll
          for (int l = 0; l < L; l++) acc[l] = (opc == OP_LOADW) ? t[l] : acc[l] + t[l];
          if (opc == OP_ADDW) gp.renorm(acc);
          break;
        }
        case OP_REDC: {
          uint32_t r[L];
          // imm = 1 (contexts of a modulus multiple M = c n only): times the small factor c on the way out, see redc_scaled
          if (NEG1 && imm) gp.redc_scaled(r, acc, args.small_c); else
          gp.redc(r, acc);
#pragma unroll
          for (int l = 0; l < L; l++) acc[l] = r[l];
          break;
        }
        case OP_CANON: {
          gp.canonical(acc);
          break;
        }
        case OP_STOREW: {
          gp.canonical(acc);
          if (NEG1 && (imm & 1)) gp.exact_div_small(acc, args.small_c, args.small_cinv);   // c (a mod n) -> a mod n, canonical modulo n
          const VmExt& e = args.ext[op.w1 & 0xf];
          uint64_t flat = (uint64_t)op.w2 * args.count + idx;
          if (op.w3 && (imm & 2)) {
            // the step-4i shuffle from the permutation itself (SC/initiator.py:212-226, :516): ext[w3-1] = int64 [inner][planes]
            // (stride = planes, limit = inner); item (plane j, comparison b) = j * inner + b goes to output plane k with
            // perm[b][k] == j.  A row that is not a permutation of 0 .. planes-1 is treated as the identity, so every output row
            // is written exactly once whatever the row holds (Initiator.permutation_is_valid tells the caller before anything is
            // sent).  One row read per item (planes * 8 bytes, every lane of the group redundantly: a few hundred instructions
            // against the ~250 000 of the item's products) replaces a separate launch that built a destination array.
            const VmExt& pe = args.ext[(op.w3 - 1) & 0xf];
            const uint64_t inner = pe.limit;
            const uint32_t planes = pe.stride;
            const uint64_t b = idx % inner;
            const uint32_t jp = (uint32_t)(idx / inner);
            const int64_t* __restrict__ prow = (const int64_t*)pe.ptr + b * planes;
            uint64_t seen0 = 0, seen1 = 0;
            bool ok = true;
            uint32_t found = jp;
#pragma unroll 4
            for (uint32_t k = 0; k < planes; k++) {
              const uint64_t v = (uint64_t)prow[k];
              const bool inr = v < (uint64_t)planes;
              const uint64_t bit = 1ull << (v & 63);
              const uint64_t word = (v & 64) ? seen1 : seen0;
              ok = ok && inr && !(word & bit);
              if (inr) { if (v & 64) seen1 |= bit; else seen0 |= bit; }
              found = (inr && (uint32_t)v == jp) ? k : found;
            }
            flat = ok ? (uint64_t)found * inner + b : idx;
          } else if (op.w3) flat = ((const uint64_t*)args.ext[(op.w3 - 1) & 0xf].ptr)[idx];   // scatter to explicit rows; guarded by e.limit
          gp.store_words((uint32_t*)e.ptr + flat * e.stride, e.nwords, acc, my_a, live && flat < e.limit);
          break;
        }
        case OP_STOREFLAG: {
          gp.canonical(acc);
          uint32_t ref[L];
#pragma unroll
          for (int l = 0; l < L; l++) ref[l] = s_c[op.w3 * SP + gp.j * L + l];
          const bool eq = gp.equal(acc, ref);
          const VmExt& e = args.ext[op.w1 & 0xf];
          const uint64_t flat = (uint64_t)op.w2 * args.count + idx;
          if (imm) {
            // item (plane, b) = plane * inner + b: the flags of all planes of b are OR-ed into one u64 (zeroed by the host);
            // hits are rare (at most one plane per comparison), so the atomic costs nothing
            if (live && gp.j == 0 && eq) atomicOr((unsigned long long*)e.ptr + (idx % e.limit), 1ull);
          } else if (live && gp.j == 0 && flat < e.limit) {
            ((uint8_t*)e.ptr)[flat] = eq ? 1 : 0;
          }
          break;
        }
        case OP_STT: {
          gp.template store_limbs<TS>(my_tbl + (uint64_t)imm * S * TS, acc);
          break;
        }
        case OP_STOREL: {
          const VmExt& e = args.ext[op.w1 & 0xf];
          const uint64_t flat = (uint64_t)op.w2 * args.count + idx;
          if (live && flat < e.limit) gp.store_limbs((uint32_t*)e.ptr + flat * e.stride, acc);
          break;
        }
        case OP_NEG: {
          gp.canonical(acc);
          uint32_t t[L];
#pragma unroll
          for (int l = 0; l < L; l++) t[l] = gp.n[l];
          gp.normalize(t, acc);   // n - ACC in (0, n]
          gp.canonical(t);        // n -> 0
#pragma unroll
          for (int l = 0; l < L; l++) acc[l] = t[l];
          break;
        }
        case OP_ADDT: {
          uint32_t t[L];
          gp.template load_limbs<TS>(t, my_tbl + (uint64_t)imm * S * TS);
#pragma unroll
          for (int l = 0; l < L; l++) acc[l] += t[l];
          gp.renorm(acc);
          break;
        }
        case OP_ADD1: {
          // imm != 0: add bit (w1 >> 4) & 63 of the u64 flag ext w1 & 15, inverted when w1 >> 12 is set, instead of 1
          uint32_t one = 1u;
          if (imm) one = (uint32_t)((((const uint64_t*)args.ext[op.w1 & 0xf].ptr)[idx] >> ((op.w1 >> 4) & 0x3f)) & 1) ^ ((op.w1 >> 12) & 1);
          acc[0] += (gp.j == 0) ? one : 0u;
          gp.renorm(acc);
          break;
        }
        case OP_SUB1: {
          uint32_t sub[L];
#pragma unroll
          for (int l = 0; l < L; l++) sub[l] = (gp.j == 0 && l == 0) ? 1u : 0u;
          gp.normalize(acc, sub);
          break;
        }
        case OP_TAKEFLAG: {
          // one lane of the group reads, resets and hands on the flag word (same thread, same address: program order holds)
          uint32_t v = 0;
          if (gp.j == 0 && live) {
            unsigned long long* accb = (unsigned long long*)args.ext[op.w1 & 0xf].ptr + idx;
            const unsigned long long w = *accb;
            *accb = 0ull;
            ((unsigned long long*)args.ext[op.w2 & 0xf].ptr)[idx] = w;
            v = (uint32_t)w & 1u;
          }
#pragma unroll
          for (int l = 0; l < L; l++) acc[l] = 0;
          acc[0] = v;                       // lanes j > 0 hold the higher limbs: zero
          break;
        }
        case OP_QUOT: {
          uint32_t r[L], quot[L], zero[L];
#pragma unroll
          for (int l = 0; l < L; l++) { quot[l] = 0; zero[l] = 0; }
          gp.template mont<2>(r, nullptr, acc, quot);
          // quot = -ACC / n mod R  ->  ACC / n = (R - quot) mod R
#pragma unroll
          for (int l = 0; l < L; l++) acc[l] = 0;
          gp.normalize(acc, quot);
          (void)zero;
          break;
        }
        default: break;
      }
    }
  }
}

}  // namespace sc
