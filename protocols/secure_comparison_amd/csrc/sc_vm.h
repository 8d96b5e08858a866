// Micro-op "VM" shared between the host program builder (sc_host.cpp part of sc_lib.hip) and the
// device interpreter kernel (sc_kernels part).  Every batched big-integer operation of the library
// (modular product, the three exponentiation shapes, Paillier/DGK step arithmetic, batch inversion
// sweeps) is a short wave-uniform program over ONE accumulator ACC (limb form, in VGPRs) and a
// per-slot scratch table in HBM; the interpreter has a single Montgomery-multiply call site so the
// hot loop stays resident in the instruction cache.
#pragma once
#include <stdint.h>

namespace sc {

enum VmOpcode : uint32_t {
  OP_END = 0,
  OP_MUL = 1,        // ACC = mont(A, ACC)                     A given by akind
  OP_LOADW = 2,      // ACC = words  ext[w1] at (off = w2), word offset w3>>16, nwords w3&0xffff (0 = ext default)
                     //   imm != 0: w1 = extA | extB<<4 | flag ext<<8 | bit<<12: ext A where the item's u64 flag bit is set, else ext B
  OP_ADDW = 3,       // ACC += words (same addressing as LOADW), lazy limb-wise add
  OP_LOADT = 4,      // ACC = limb-form operand given by akind (table / const / fbt / ext-limbs)
  OP_REDC = 5,       // ACC = ACC / R mod n;  imm = 1 (modulus-multiple contexts): ACC = ACC c / R mod M
  OP_STOREW = 6,     // canonical(ACC) -> words ext[w1] at off w2; w3 != 0: at the flat item index ext[w3-1][item] (u64) instead;
                     //   imm bit 0 (modulus-multiple contexts): canonical(ACC) / c exactly, i.e. the residue modulo n
                     //   imm bit 1 (with w3): ext[w3-1] is a batch of permutations int64 [inner][planes] (stride = planes,
                     //   limit = inner) and item (j, b) = j * inner + b lands in row k * inner + b with perm[b][k] == j
  OP_STOREFLAG = 7,  // (canonical(ACC) == const[w3]) -> u8 ext[w1] at off w2;  imm != 0: OR the flag into the u64 ext[w1][item mod ext.stride64]
                     //   instead (ext.limit = inner count): one flag per group of items, e.g. delta_B = OR over the l+1 zero tests
  OP_STT = 8,        // scratch[imm] = ACC
  OP_ADD1 = 9,       // ACC += 1 (lazy);  imm != 0: ACC += bit (w1>>4)&63 of the item's u64 flag in ext w1&15, inverted if w1>>12
  OP_SUB1 = 10,      // ACC = (ACC - 1) mod R, exact limbs
  OP_QUOT = 11,      // ACC = ACC / n exactly (ACC must be an exact multiple of n, value < R)
  OP_STOREL = 12,    // limb-form store of ACC to ext[w1] at off w2 (ext stride = S)
  OP_CANON = 13,     // ACC = canonical(ACC)
  OP_ADDT = 14,      // ACC += scratch[imm]  (lazy limb-wise add)
  OP_NEG = 15,       // ACC = (n - ACC) mod n, exact limbs
  OP_TAKEFLAG = 16,  // ACC = v = ext[w1][item] (u64 accumulator of OP_STOREFLAG's OR form); the accumulator is reset to 0 (so that it is
                     //   zero again for its next use: no clearing pass) and v is written to ext[w2][item] (u64, the caller's copy)
};

// Opcodes of the pair interpreter k_pvm (arithmetic modulo n^2 with products modulo n, sc_device.h "pair arithmetic").
// State: the pair (ACC0, ACC1); scratch pair entry e occupies limb-form entries 2e and 2e+1 of the slot's table.
enum PvOpcode : uint32_t {
  PV_END = 0,
  PV_LOADU = 1,   // ACC0 = words ext[w1] at off w2 (word offset w3>>16, nwords w3&0xffff), ACC1 = 0   (pair of u / R)
  PV_MULC = 2,    // pair *= constant pair held in LDS constants w1, w1+1
  PV_MULT = 3,    // pair *= scratch pair entry w1
  PV_SQR = 4,     // pair = pair^2
  PV_STT = 5,     // scratch pair entry w1 = pair
  PV_LOADT = 6,   // pair = scratch pair entry w1
  PV_ADDT = 7,    // pair += scratch pair entry w1 (component-wise, lazy)
  PV_OUT = 8,     // leave the pair form and store: ACC0 -> ext[w1], ACC1 -> ext[w2] as exact words (not reduced mod n)
};

enum VmAKind : uint32_t {
  AK_CONST = 0,   // w1 = LDS constant index (0 = R^2 mod n, 1 = R mod n, 2.. = extra constants)
  AK_ACC = 1,     // the accumulator itself (squaring)
  AK_TBL = 2,     // w1 = scratch entry
  AK_TBLSEL = 3,  // w1 = flag descriptor (extA | bitA<<4 | extB<<12 | bitB<<16), w2 = 4 scratch entries (bytes) indexed by fa*2+fb
  AK_TBLDIG = 4,  // w1 = ext | bitpos<<4 | width<<24 ; w2 = base scratch entry ; entry = base + digit
  AK_FBT = 5,     // w1 = ext | bitpos<<4 | width<<24 ; w2 = window index ; row = fbt[(win << width) + digit]
  AK_EXTW = 6,    // w1 = ext, w2 = off : plain words operand (staged through LDS)
  AK_EXTL = 7,    // w1 = ext, w2 = off : limb-form operand in an ext array (ext stride: S, or 0 = broadcast)
  AK_CONSTSEL = 8,  // w1 = ext (one byte per item; ext.stride = bytes between items, 0 = 1), w2 = LDS constant index for byte 0 | index for byte != 0 << 8
};

struct VmOp {
  uint32_t w0;  // opcode[7:0] | akind[11:8] | imm[31:16]
  uint32_t w1, w2, w3;
};

struct VmExt {
  const void* ptr;
  uint32_t stride;   // u32 words between consecutive items (0 = broadcast one item)
  uint32_t nwords;   // words per item for word-form operands
  uint64_t limit;    // flat item indices >= limit read as the integer 1 and are not written
};

struct RngKey { uint32_t k[8]; };   // ChaCha20 key of a context's generator (sc_rng.h)

constexpr int VM_MAX_EXT = 8;
constexpr int VM_MAX_CONST = 8;  // including R^2 and R

struct VmArgs {
  const uint32_t* modctx;   // limb form: n | R^2 mod n | R mod n   (3*S words)
  const uint32_t* consts;   // extra limb-form constants, nconst_extra * S words
  const VmOp* prog;
  uint32_t* scratch;        // nslots * nscratch * S words
  const uint32_t* fbt;      // fixed-base table rows, limb form
  uint64_t count;           // items
  uint32_t n0inv;
  uint32_t nops;
  uint32_t nconst_extra;
  uint32_t nscratch;
  uint32_t small_c, small_cinv;   // contexts of a modulus multiple M = c n: c and c^-1 mod 2^W (OP_REDC / OP_STOREW with imm = 1)
  uint64_t* stamps;               // clock stamps of the diagnostic twin of the pair interpreter (k_pvm<.., STAMP>), else unused
  VmExt ext[VM_MAX_EXT];
};

}  // namespace sc
