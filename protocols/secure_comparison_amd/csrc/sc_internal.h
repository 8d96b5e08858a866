// Host-side internals shared by the translation units of libsc_amd.so: set-up-time integers, the registered objects of a context
// (moduli, exponents, constants, tables, programs), the context itself, and the launch entry points of the kernel translation
// units.  The kernels are instantiated in sc_launch_vm.hip / sc_launch_pvm.hip (three parts each, compiled in parallel) and
// sc_launch_misc.hip; sc_lib.hip (+ sc_schemes.h) holds no device code, so a change of host logic or policy rebuilds in seconds.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sys/random.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../../include/sc_amd_dev.h"
#include "sc_vm.h"

#ifndef SC_INV_TOP
#define SC_INV_TOP 2048
#endif
#ifndef SC_PVM_WAVES
#define SC_PVM_WAVES 2    // waves per SIMD the pair interpreter is compiled for (sc_kernel_pvm.h)
#endif

namespace {
using namespace sc;

// ------------------------------------------------------------------------------------------------
// tiny host big-integer helpers on little-endian uint32 word vectors (setup-time only)
// ------------------------------------------------------------------------------------------------
typedef std::vector<uint32_t> Big;

int big_bits(const Big& a) {
  for (int i = (int)a.size() - 1; i >= 0; i--)
    if (a[i]) return 32 * i + (32 - __builtin_clz(a[i]));
  return 0;
}
int big_cmp(const Big& a, const Big& b) {  // same length
  for (int i = (int)a.size() - 1; i >= 0; i--)
    if (a[i] != b[i]) return a[i] > b[i] ? 1 : -1;
  return 0;
}
void big_sub(Big& a, const Big& b) {  // a -= b, same length
  uint64_t borrow = 0;
  for (size_t i = 0; i < a.size(); i++) {
    uint64_t v = (uint64_t)a[i] - b[i] - borrow;
    a[i] = (uint32_t)v;
    borrow = (v >> 32) & 1;
  }
}
// a = 2a mod n  (a < n, a and n have the same length with one spare top word)
void big_dbl_mod(Big& a, const Big& n) {
  uint32_t carry = 0;
  for (size_t i = 0; i < a.size(); i++) {
    uint32_t nc = a[i] >> 31;
    a[i] = (a[i] << 1) | carry;
    carry = nc;
  }
  if (big_cmp(a, n) >= 0) big_sub(a, n);
}
// x * 2^k mod n
Big big_shl_mod(const Big& x, const Big& n, int k) {
  Big nn = n; nn.push_back(0);
  Big v = x; v.resize(nn.size(), 0);
  while (big_cmp(v, nn) >= 0) big_sub(v, nn);
  for (int i = 0; i < k; i++) big_dbl_mod(v, nn);
  v.resize(n.size());
  return v;
}
Big big_trimmed_words(Big a) { while (a.size() > 1 && a.back() == 0) a.pop_back(); return a; }
Big big_mul(const Big& a, const Big& b) {  // schoolbook product (set-up time only)
  Big r(a.size() + b.size(), 0);
  for (size_t i = 0; i < a.size(); i++) {
    uint64_t carry = 0;
    for (size_t j = 0; j < b.size(); j++) {
      uint64_t v = (uint64_t)a[i] * b[j] + r[i + j] + carry;
      r[i + j] = (uint32_t)v;
      carry = v >> 32;
    }
    r[i + b.size()] = (uint32_t)carry;
  }
  return r;
}
// x = q * m + rem by restoring division, bit by bit (set-up time only); q has x.size() words, rem m.size() words
void big_divmod(const Big& x, const Big& m, Big* q, Big* rem) {
  Big r(m.size() + 1, 0), mm = m; mm.push_back(0);
  q->assign(x.size(), 0);
  for (int bit = 32 * (int)x.size() - 1; bit >= 0; bit--) {
    uint32_t carry = (x[bit >> 5] >> (bit & 31)) & 1;
    for (size_t i = 0; i < r.size(); i++) { uint32_t nc = r[i] >> 31; r[i] = (r[i] << 1) | carry; carry = nc; }
    if (big_cmp(r, mm) >= 0) { big_sub(r, mm); (*q)[bit >> 5] |= 1u << (bit & 31); }
  }
  r.resize(m.size());
  *rem = r;
}
std::vector<uint32_t> to_limbs(const Big& x, int S, int W) {
  std::vector<uint32_t> out(S, 0);
  const uint32_t mask = (1u << W) - 1;
  for (int i = 0; i < S; i++) {
    int bit = W * i, w0 = bit >> 5, sh = bit & 31;
    uint64_t v = (w0 < (int)x.size()) ? x[w0] : 0;
    if (w0 + 1 < (int)x.size()) v |= (uint64_t)x[w0 + 1] << 32;
    out[i] = (uint32_t)(v >> sh) & mask;
  }
  return out;
}

struct Config { int G, L, W; bool primary; };   // primary: eligible as a modulus's own configuration (sc_mod_create)
// ordered by capacity W*G*L; sc_mod_create takes the first one that fits.  A 28-bit-limb L = 37 family ((2,37), (4,37)) was
// built and measured in round 1: it needs > 256 registers (one wave per SIMD plus AGPR copies) and came out 2-3 % slower
// than (4,18) / (8,18), so it is not compiled in; the limb width stays a template parameter for such experiments.
const Config kConfigs[] = {{1, 18, 29, true}, {2, 18, 29, true}, {2, 27, 29, true}, {4, 14, 29, true}, {4, 18, 29, true}, {4, 27, 29, true},
                           {8, 14, 29, true}, {8, 18, 29, true}, {8, 27, 29, true}, {16, 14, 29, true}, {16, 18, 29, true}};
// The one-lane configuration for moduli up to 1028 bits (the primes of 2048-bit Paillier / DGK keys): (1, 37) with 28-bit limbs.
// A number lives in ONE lane, so the per-limb-step bookkeeping is paid once per number instead of once per lane of a group, the
// operand of a squaring never leaves the registers and the modulus sits in scalar registers: 1.15 - 1.25x the (2, 18) rate
// per number.  It needs 64 numbers per wave, i.e. large batches, and is therefore never a modulus's own configuration: the
// shared-exponent entry points switch to an internal twin context of the same modulus when the batch fills the chip
// (onelane_for, sc_ctx_set_onelane_mode).
const Config kOneLane = {1, 37, 28, false};
// configurations with a pair kernel (k_pvm): every L = 18 one, and (4,14) / (8,14) for the 1536 / 3072-bit sizes whose direct
// configuration is L = 27 (the pair arithmetic needs the L <= 18 column bound)
// (the one-lane (1, 37, 28) configuration has no pair kernel: measured on the MI355X its pair squarings run 3 % faster than the
// (2, 18) ones but its pair products -- three passes over a single LDS staging area, the second area would cost the eighth wave
// of the CU -- 2.6x a squaring instead of 1.4x, a net loss of 12 % on x^p mod p^2; the one-lane form is used where it wins:
// the single-modulus exponentiations)
// (8,5) / (16,5): the SMALL-BATCH pair configurations of 1024 / 2048-bit moduli (kLatencyPair below)
inline bool pair_capable(int G, int L, int W) { return W == 29 && (L == 18 || (L == 14 && (G == 4 || G == 8)) || (L == 5 && (G == 4 || G == 8 || G == 16))); }
// Small batches of pair exponentiations (Alice's rho^N mod N^2, the key holder's c^(p-1) mod p^2 at B = 4096) are one dependent
// chain of ~2400 pair squarings per item, and a wave's time per squaring is its own instruction count: S limb steps of
// (L + L/2) multiply-adds + ~7 bookkeeping instructions each, whatever the number of lanes.  When even the (2G, 9) form
// leaves half of the SIMDs without a wave, 4x the lanes with 5 limbs each -- (16,5) for 2048-bit, (8,5) for 1024-bit moduli,
// S = 80 / 40 limbs -- shorten every limb step from ~22 to ~15 instructions at a multiply-add density (45 %) that would be
// wasteful on a full chip but costs nothing on an empty one.  A twin context of the same modulus, like the other twins.
const Config kLatencyPair16 = {16, 5, 29, false}, kLatencyPair8 = {8, 5, 29, false}, kLatencyPair4 = {4, 5, 29, false};   // (4,5): 512-bit primes of 1024-bit keys

struct Mod {
  int G = 0, L = 0, W = 29, S = 0, nwords = 0, nbits = 0;
  Big n;
  uint32_t n0inv = 0;
  uint32_t small_c = 0, small_cinv = 0;   // a modulus multiple M = c n (neg1_twin): c and c^-1 mod 2^W, else 0
  uint32_t* d_ctx = nullptr;  // n | R^2 | R  limb form
};
struct Exp { Big e; int bits = 0; };
struct Const { int mod = -1; uint32_t* d_limbs = nullptr; };
// the rows of a fixed-base table are shared between contexts (sc_fbt_import): freed when the last table that uses them goes
struct FbtRows {
  int device = 0; uint32_t* d = nullptr; size_t bytes = 0;
  ~FbtRows() { if (d) { (void)hipSetDevice(device); (void)hipFree(d); } }
};
struct Fbt { int mod = -1, window = 0, nwin = 0, exp_bits = 0; uint32_t* d_rows = nullptr; std::shared_ptr<FbtRows> rows; };
struct Prog {
  uint32_t nops = 0, nscratch = 1, nconst = 0;
  VmOp* d_ops = nullptr;
  uint32_t* d_consts = nullptr;
  double muls_per_item = 0;   // Montgomery products (full) per item
  double redcs_per_item = 0;  // reduction-only passes per item
  double sqrs_per_item = 0;   // squarings (a*a part costs L(L+1)/2 per block instead of L^2)
  std::shared_ptr<std::vector<VmOp>> host_ops;   // the micro-ops on the host (pair programs: cut into segments on demand)
  uint32_t pair_sqrs = 0, pair_muls = 0;         // pair programs: pair squarings / pair products per item (the hold-time model)
};

}  // namespace

struct sc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t switch_event = nullptr;   // orders the work of the previous stream before the next one (sc_ctx_set_stream)
  int num_cu = 256;
  std::string err;
  int64_t last_bad_index = -1;                              // the element named by the last SC_ERR_NOT_INVERTIBLE (sc_last_bad_index)
  std::vector<Mod> mods;
  std::vector<Exp> exps;
  std::vector<Const> consts;
  std::vector<Fbt> fbts;
  std::map<std::string, Prog> progs;
  std::map<std::string, std::vector<Prog>> seg_progs;   // pair programs cut into segments (sc_modexp_shared_sq)
  uint32_t* scratch = nullptr;
  size_t scratch_bytes = 0;
  std::vector<void*> owned;
  double mac_counter = 0;
  std::map<int, int> occ_cache;  // config index -> blocks per CU
  std::map<int, std::pair<void*, size_t>> tmp;          // grow-only temporaries, reused across calls (same stream => ordered)
  std::map<std::vector<uint32_t>, uint32_t*> nwords_cache;  // device copy of {n, (n-1)/2} for the plain-word kernels
  std::map<int, int> kred_cache;                            // mod -> constant id of 2^(32 nwords) (wide-operand reduction)
  std::map<std::pair<int, std::vector<uint32_t>>, int> const_by_value;  // (mod, residue) -> constant id
  int latency_mode = 1;                                     // sc_ctx_set_latency_mode: 0 never, 1 automatic, 2 whenever available
  int onelane_mode = 1;                                     // sc_ctx_set_onelane_mode: 0 never, 1 automatic, 2 whenever available
  bool slot_per_item = false;                               // pair launches: a table slot per item instead of per resident wave (segments)
  // Segment policy of long pair launches on a shared chip (sc_modexp_shared_sq, sc_ctx_set_pair_policy): a resident wave should not
  // hold its slot for much longer than pair_hold_ms; launches of more than pair_max_rounds rounds are left whole
  double pair_hold_ms = 5.0, pair_max_rounds = 2.5;
  uint64_t stat_segmented_launches = 0, stat_segments = 0, stat_pair_calibrations = 0;   // sc_ctx_stats
  int chip_share = 1;                                       // sc_ctx_set_chip_share: contexts working on this GPU at the same time
  void* comm = nullptr;                                     // RCCL communicator of this rank (sc_comm_init), one context per GPU
  int comm_rank = 0, comm_nranks = 0;
  std::map<int, int> onelane_twins;                         // mod -> context of the same modulus in the one-lane configuration
  std::map<int, int> pair_twins;                            // mod -> context of the same modulus in a pair-capable configuration
  std::map<int, int> neg1_twins;                            // (4,18) mod n -> context of the multiple M = c n = -1 (mod 2^29)
  std::map<int, int> latency_pair_twins;                    // mod -> context of the same modulus in the (16,5) / (8,5) small-batch pair configuration
  std::map<int, uint32_t*> pair_consts;                     // mod -> 4 limb arrays: pair(R^2), pair(B R) for the pair arithmetic
  RngKey rng_key;                                           // ChaCha20 key of the context's generator (sc_rng_seed)
  // fork / join inside one library call (AuxFork): independent halves of a small batch -- the p- and q-side of the key holder's CRT
  // -- run on a second stream of the context with its own scratch arena and temporaries
  hipStream_t aux_stream = nullptr;
  hipEvent_t aux_fork = nullptr, aux_join = nullptr;
  uint32_t* scratch_aux = nullptr;
  size_t scratch_aux_bytes = 0;
  bool in_aux = false;
  // verdict words of the inversion kernel: pinned host memory the kernel writes directly (one buffer per stream of the context), so
  // the host reads them after a stream synchronisation with no copy in between -- a device-to-host copy of a few words from pageable
  // memory is a runtime blit kernel (__amd_rocclr_copyBuffer) that queues behind other contexts' chip-filling launches
  int* status_host[2] = {nullptr, nullptr};
  size_t status_cap[2] = {0, 0};
  int fork_mode = 1;                                        // sc_ctx_set_fork_mode: 0 never fork inside a call, 1 automatic (small batches), 2 always
  void* scheme_keys = nullptr;                              // Paillier / DGK key objects of the scheme-level entry points (sc_schemes.h)
  bool rng_seeded = false;
  std::atomic<uint64_t> rng_call{0};                        // generator calls since seeding: part of every keystream's nonce (atomic: two
                                                            // host threads that ever share a context must never draw one (key, call) twice)
  std::mutex rng_seed_mutex;                                // the lazy first seeding happens once
  uint64_t* stamps = nullptr;                               // sc_clock_probe: the next (4,18,neg1) pair launch runs its stamping twin
  uint32_t stamp_grid = 0;                                  // ... and reports its grid size here
};

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return sc_host::fail(ctx, SC_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

// ---- what the other translation units provide --------------------------------------------------------------------------------
namespace sc_host {
// sc_lib.hip
int fail(sc_ctx* ctx, int code, const char* fmt, ...);
int ensure_scratch(sc_ctx* ctx, size_t bytes, uint32_t** out);
// sc_launch_vm.hip / sc_launch_pvm.hip, parts 0 .. 2: launch the instance (G, L, W, NEG1[, STAMP]) of the interpreter on the context's
// stream (grid, scratch arena and occupancy handled there); SC_ERR_UNSUPPORTED when the instance lives in another part
int launch_vm_part0(sc_ctx* ctx, int G, int L, int W, bool neg1, const sc::VmArgs& a);
int launch_vm_part1(sc_ctx* ctx, int G, int L, int W, bool neg1, const sc::VmArgs& a);
int launch_vm_part2(sc_ctx* ctx, int G, int L, int W, bool neg1, const sc::VmArgs& a);
int launch_pvm_part0(sc_ctx* ctx, int G, int L, bool neg1, bool stamp, const sc::VmArgs& a);
int launch_pvm_part1(sc_ctx* ctx, int G, int L, bool neg1, bool stamp, const sc::VmArgs& a);
int launch_pvm_part2(sc_ctx* ctx, int G, int L, bool neg1, bool stamp, const sc::VmArgs& a);
// resident waves per CU of that pair kernel instance (occupancy query, cached in the context); <= 0 when the instance lives elsewhere
int pvm_occupancy_part0(sc_ctx* ctx, int G, int L, bool neg1);
int pvm_occupancy_part1(sc_ctx* ctx, int G, int L, bool neg1);
int pvm_occupancy_part2(sc_ctx* ctx, int G, int L, bool neg1);
// sc_launch_misc.hip: 0 on success, a negative number when the launch itself failed
int launch_xgcd(hipStream_t stream, const uint32_t* x, uint32_t* out, const uint32_t* d_n, int nw, uint64_t count, int* d_status);
int launch_plain_alice(hipStream_t stream, const uint32_t* r, const uint32_t* nmod, const uint32_t* halfn, int nw, int l, uint64_t count, uint32_t* m1,
                       uint64_t* alpha, uint64_t* alpha_tilde, uint64_t* rsmall, uint32_t* rshift);
int launch_plain_bob(hipStream_t stream, const uint32_t* z, const uint32_t* nmod, const uint32_t* halfn, int nw, int l, uint64_t count, uint64_t* beta,
                     uint64_t* dbit, uint32_t* zeta1, uint32_t* zeta2, uint8_t* bits);
int launch_rng_bits(hipStream_t stream, const sc::RngKey& key, uint64_t call, int bits, int nw, uint32_t* out, uint64_t count);
int launch_rng_below(hipStream_t stream, const sc::RngKey& key, uint64_t call, const uint32_t* d_n, int nbits, int nw, int nonzero, uint32_t* out, uint64_t count);
int launch_rng_coins(hipStream_t stream, const sc::RngKey& key, uint64_t call, uint64_t* out, uint64_t count);
int launch_rng_perm(hipStream_t stream, const sc::RngKey& key, uint64_t call, int k, int64_t* out, uint64_t count);
int launch_peak_probe(hipStream_t stream, int grid, uint32_t* out, uint32_t a0, uint32_t b0, int iters);
}  // namespace sc_host
