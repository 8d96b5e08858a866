// libsc_amd.so -- host side of the C ABI declared in include/sc_amd.h and include/sc_amd_dev.h.
// Builds the micro-programs (sc_vm.h) for each batched operation and queues the launches; the kernels themselves are instantiated in
// the sc_launch_*.hip translation units (sc_internal.h), so this file holds no device code.
#include "sc_internal.h"

using namespace sc;
using namespace sc_host;

namespace sc_host {
int fail(sc_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}
}  // namespace sc_host

namespace {

void free_scheme_keys(void* p);   // sc_schemes.h

// every allocation selects the context's device first: the caller may have switched the thread's current device
int dev_alloc(sc_ctx* ctx, size_t bytes, void** out) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMalloc(out, bytes ? bytes : 4));
  ctx->owned.push_back(*out);
  return SC_OK;
}
int upload(sc_ctx* ctx, const void* h, size_t bytes, void** out) {
  int rc = dev_alloc(ctx, bytes, out);
  if (rc) return rc;
  // on the context's stream, then waited for: a plain hipMemcpy runs on the null stream, which does not order with the
  // non-blocking streams a caller may hand in (torch side streams), and the source is a short-lived host buffer
  HIPCHK(ctx, hipMemcpyAsync(*out, h, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}
}  // namespace
// the scratch arena of the stream the context currently launches on (its own, or the forked one's: AuxFork)
int sc_host::ensure_scratch(sc_ctx* ctx, size_t bytes, uint32_t** out) {
  uint32_t*& arena = ctx->in_aux ? ctx->scratch_aux : ctx->scratch;
  size_t& have = ctx->in_aux ? ctx->scratch_aux_bytes : ctx->scratch_bytes;
  if (bytes > have) {
    if (arena) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(arena)); arena = nullptr; have = 0; }
    size_t want = bytes + bytes / 4;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMalloc((void**)&arena, want));
    have = want;
  }
  *out = arena;
  return SC_OK;
}
namespace {

// Temporary device buffer `slot`, at least `bytes` large.  Buffers are reused by later calls: every kernel of a context
// runs on one stream, so a later call cannot overtake an earlier one that still reads the buffer.
int tmp_buf(sc_ctx* ctx, int slot, size_t bytes, void** out) {
  auto& e = ctx->tmp[slot + (ctx->in_aux ? 1000 : 0)];       // the forked stream's calls never share a temporary with the main one's
  if (e.second < bytes) {
    if (e.first) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(e.first)); e.first = nullptr; e.second = 0; }
    size_t want = bytes + bytes / 8 + 256;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMalloc(&e.first, want));
    e.second = want;
  }
  *out = e.first;
  return SC_OK;
}
// Fork / join inside one library call.  begin(): everything queued so far on the context's stream happens before the forked work;
// until suspend() the context launches on its second stream (own scratch arena, own temporaries); after suspend() it is back on
// its own stream, whose later launches run BESIDE the forked work; join(): the context's stream waits for the forked work.
// (The forked half is queued first: an event recorded after the other half's launches would wait for them.)
// Used for the q-side of the key holder's CRT when a launch of the batch leaves most of the chip idle.
struct AuxFork {
  sc_ctx* ctx = nullptr;
  hipStream_t main = nullptr;
  bool active = false, pending = false;
  int begin(sc_ctx* c) {
    ctx = c;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->aux_stream) {
      HIPCHK(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
      HIPCHK(c, hipEventCreateWithFlags(&c->aux_fork, hipEventDisableTiming));
      HIPCHK(c, hipEventCreateWithFlags(&c->aux_join, hipEventDisableTiming));
    }
    HIPCHK(c, hipEventRecord(c->aux_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->aux_stream, c->aux_fork, 0));
    main = c->stream;
    c->stream = c->aux_stream;
    c->in_aux = true;
    active = true;
    return SC_OK;
  }
  int suspend() {
    if (!active) return SC_OK;
    active = false;
    sc_ctx* c = ctx;
    const hipError_t e1 = hipEventRecord(c->aux_join, c->aux_stream);
    c->stream = main;
    c->in_aux = false;
    if (e1 != hipSuccess) return fail(c, SC_ERR_HIP, "hipEventRecord: %s", hipGetErrorString(e1));
    pending = true;
    return SC_OK;
  }
  int join() {
    int rc = suspend();
    if (rc) return rc;
    if (pending) { pending = false; HIPCHK(ctx, hipStreamWaitEvent(main, ctx->aux_join, 0)); }
    return SC_OK;
  }
  ~AuxFork() { (void)join(); }      // error paths: never leave the context on its second stream, nor forked work unordered
};
// Should the two CRT halves of a call run side by side on two streams?  Small batches: when a second, independent launch fits
// beside the first (a wave per SIMD for both: half of the chip's 4 x CUs SIMDs each; off with latency mode 0).  Large ones: when
// this context has the chip to itself.
inline bool small_enough_to_fork(const sc_ctx* ctx, const Mod& m, uint64_t count) {
  if (ctx->fork_mode == 0 || ctx->in_aux) return false;
  if (ctx->fork_mode == 2) return true;                     // always: the two halves' launches beside each other whatever the batch size
  // a context that has the chip to itself: the halves' launches side by side pack their partial rounds (three launches of 1.5
  // rounds each are 2 rounds long one after the other) -- single-stream step 381.1 -> 377.2 ms at B = 65536; with concurrent
  // shards the other shard's launches fill those rounds already and forking costs 1 % (profiles/r04_fork_modes.txt)
  if (ctx->chip_share == 1 && ctx->latency_mode != 2) {
    const uint64_t per_wave_full = std::max(1, 64 / m.G);
    if ((count + per_wave_full - 1) / per_wave_full > (uint64_t)ctx->num_cu * 4) return true;
  }
  if (ctx->latency_mode == 0) return false;
  if (ctx->latency_mode == 2) return true;
  const uint64_t per_wave = std::max(1, 64 / (2 * m.G));     // in the small-batch configuration this batch would take
  return (count + per_wave - 1) / per_wave <= (uint64_t)ctx->num_cu * 4 / (uint64_t)ctx->chip_share;
}

enum TmpSlot { TMP_PARK = 1, TMP_ANYFLAG = 2, TMP_CRT = 3, TMP_PAIR = 4, TMP_INV_MEMBERS = 5, TMP_INV_BASE = 16 /* + 2*depth, + 2*depth+1 */ };

// A u64 accumulator array of the context that is ZERO whenever no step is using it: cleared when it is (re)allocated, reset by its
// reader afterwards (OP_TAKEFLAG).  The zero tests of step 4j OR their verdicts into it; no clearing launch inside a step.
int zero_kept_flags(sc_ctx* ctx, uint64_t count, uint64_t** out) {
  auto& e = ctx->tmp[TMP_ANYFLAG + (ctx->in_aux ? 1000 : 0)];
  const size_t bytes = (size_t)count * sizeof(uint64_t);
  if (e.second < bytes) {
    if (e.first) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(e.first)); e.first = nullptr; e.second = 0; }
    const size_t want = bytes + bytes / 8 + 256;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMalloc(&e.first, want));
    e.second = want;
    HIPCHK(ctx, hipMemsetAsync(e.first, 0, want, ctx->stream));
  }
  *out = (uint64_t*)e.first;
  return SC_OK;
}

void drop_zero_kept_flags(sc_ctx* ctx) {      // forget the accumulator: the next use allocates and clears a new one
  auto& e = ctx->tmp[TMP_ANYFLAG + (ctx->in_aux ? 1000 : 0)];
  if (e.first) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(e.first); e.first = nullptr; e.second = 0; }
}

// `count` verdict words in pinned, device-visible host memory (grow-only, per stream of the context)
int status_words(sc_ctx* ctx, size_t count, int** out) {
  const int s = ctx->in_aux ? 1 : 0;
  if (ctx->status_cap[s] < count) {
    if (ctx->status_host[s]) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipHostFree(ctx->status_host[s])); ctx->status_host[s] = nullptr; ctx->status_cap[s] = 0; }
    const size_t want = std::max<size_t>(count, SC_INV_TOP);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostMalloc((void**)&ctx->status_host[s], want * sizeof(int), hipHostMallocMapped));
    ctx->status_cap[s] = want;
  }
  *out = ctx->status_host[s];
  return SC_OK;
}

// device copy of n | (n-1)/2 as canonical words (plain-word kernels)
int device_n_half(sc_ctx* ctx, const uint32_t* n_hptr, int nw, uint32_t** out) {
  std::vector<uint32_t> n(n_hptr, n_hptr + nw);
  auto it = ctx->nwords_cache.find(n);
  if (it != ctx->nwords_cache.end()) { *out = it->second; return SC_OK; }
  std::vector<uint32_t> both(2 * nw);
  for (int k = 0; k < nw; k++) { both[k] = n[k]; both[nw + k] = (n[k] >> 1) | ((k + 1 < nw) ? (n[k + 1] << 31) : 0u); }
  uint32_t* d = nullptr;
  int rc = upload(ctx, both.data(), both.size() * 4, (void**)&d);
  if (rc) return rc;
  ctx->nwords_cache[n] = d;
  *out = d;
  return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// program builder
// ------------------------------------------------------------------------------------------------
struct Builder {
  std::vector<VmOp> ops;
  std::vector<int> consts;  // extra constant ids (LDS index = 2 + position)
  uint32_t nscratch = 1;
  double muls = 0, redcs = 0, sqrs = 0;
  void touch(uint32_t e) { nscratch = std::max(nscratch, e + 1); }
  void emit(uint32_t opc, uint32_t ak = 0, uint32_t imm = 0, uint32_t w1 = 0, uint32_t w2 = 0, uint32_t w3 = 0) {
    ops.push_back(VmOp{opc | (ak << 8) | (imm << 16), w1, w2, w3});
  }
  int use_const(int cid) {
    for (size_t i = 0; i < consts.size(); i++) if (consts[i] == cid) return 2 + (int)i;
    consts.push_back(cid);
    return 2 + (int)consts.size() - 1;
  }
  void mul_const(int lds_idx) { emit(OP_MUL, AK_CONST, 0, lds_idx); muls++; }
  void sqr() {   // consecutive squarings merge into one micro-op with a repeat count (imm, 16 bits)
    sqrs++;
    if (!ops.empty()) {
      VmOp& last = ops.back();
      const uint32_t imm = last.w0 >> 16;
      if ((last.w0 & 0xffff) == (OP_MUL | (AK_ACC << 8)) && imm >= 1 && imm < 0xffff) { last.w0 += 1u << 16; return; }
    }
    emit(OP_MUL, AK_ACC, 1);
  }
  void mul_tbl(uint32_t e) { touch(e); emit(OP_MUL, AK_TBL, 0, e); muls++; }
  void mul_tblsel(int extA, int bitA, int extB, int bitB, uint32_t e00, uint32_t e01, uint32_t e10, uint32_t e11) {
    touch(std::max(std::max(e00, e01), std::max(e10, e11)));
    emit(OP_MUL, AK_TBLSEL, 0, extA | (bitA << 4) | (extB << 12) | (bitB << 16), e00 | (e01 << 8) | (e10 << 16) | (e11 << 24));
    muls++;
  }
  void mul_tbldig(int ext, uint32_t bitpos, uint32_t width, uint32_t base) {
    touch(base + (1u << width) - 1);
    emit(OP_MUL, AK_TBLDIG, 0, ext | (bitpos << 4) | (width << 24), base); muls++;
  }
  void mul_fbt(int ext, uint32_t bitpos, uint32_t width, uint32_t win) { emit(OP_MUL, AK_FBT, 0, ext | (bitpos << 4) | (width << 24), win); muls++; }
  void mul_extw(int ext, uint32_t off = 0) { emit(OP_MUL, AK_EXTW, 0, ext, off); muls++; }
  void mul_extl(int ext, uint32_t off = 0) { emit(OP_MUL, AK_EXTL, 0, ext, off); muls++; }
  void loadw(int ext, uint32_t off = 0, uint32_t woff = 0, uint32_t nw = 0) { emit(OP_LOADW, 0, 0, ext, off, (woff << 16) | nw); }
  // ext_set where bit `bit` of the item's u64 flag (ext flag_ext) is set, else ext_clear (sc_vm.h)
  void loadw_sel(int ext_set, int ext_clear, int flag_ext, int bit, uint32_t off = 0) { emit(OP_LOADW, 0, 1, ext_set | (ext_clear << 4) | (flag_ext << 8) | (bit << 12), off); }
  void add_flag(int flag_ext, int bit, bool invert) { emit(OP_ADD1, 0, 1, flag_ext | (bit << 4) | ((invert ? 1 : 0) << 12)); }
  void addw(int ext, uint32_t off = 0, uint32_t woff = 0, uint32_t nw = 0) { emit(OP_ADDW, 0, 0, ext, off, (woff << 16) | nw); }
  void loadt_const(int lds_idx) { emit(OP_LOADT, AK_CONST, 0, lds_idx); }
  void loadt_tbl(uint32_t e) { touch(e); emit(OP_LOADT, AK_TBL, 0, e); }
  void loadt_tbldig(int ext, uint32_t bitpos, uint32_t width, uint32_t base) {
    touch(base + (1u << width) - 1);
    emit(OP_LOADT, AK_TBLDIG, 0, ext | (bitpos << 4) | (width << 24), base);
  }
  void loadt_tblsel(int extA, int bitA, int extB, int bitB, uint32_t e00, uint32_t e01, uint32_t e10, uint32_t e11) {
    touch(std::max(std::max(e00, e01), std::max(e10, e11)));
    emit(OP_LOADT, AK_TBLSEL, 0, extA | (bitA << 4) | (extB << 12) | (bitB << 16), e00 | (e01 << 8) | (e10 << 16) | (e11 << 24));
  }
  void loadt_fbt(int ext, uint32_t bitpos, uint32_t width, uint32_t win) { emit(OP_LOADT, AK_FBT, 0, ext | (bitpos << 4) | (width << 24), win); }
  void loadt_extl(int ext, uint32_t off = 0) { emit(OP_LOADT, AK_EXTL, 0, ext, off); }
  void stt(uint32_t e) { touch(e); emit(OP_STT, 0, e); }
  void addt(uint32_t e) { touch(e); emit(OP_ADDT, 0, e); }
  void redc(bool times_c = false) { emit(OP_REDC, 0, times_c ? 1 : 0); redcs++; }     // times_c / over_c: leaving a modulus-multiple context (sc_vm.h)
  void storew(int ext, uint32_t off = 0, bool over_c = false) { emit(OP_STOREW, 0, over_c ? 1 : 0, ext, off); }
  void storew_at(int ext, int ext_index, bool over_c = false, bool by_perm = false) {   // row = ext_index[item] (u64), or from a batch of permutations (sc_vm.h)
    emit(OP_STOREW, 0, (over_c ? 1 : 0) | (by_perm ? 2 : 0), ext, 0, 1 + ext_index);
  }
  void storel(int ext, uint32_t off = 0) { emit(OP_STOREL, 0, 0, ext, off); }
  void storeflag(int ext, uint32_t off, int lds_const) { emit(OP_STOREFLAG, 0, 0, ext, off, lds_const); }
  void end() { emit(OP_END); }
};

int finalize_prog(sc_ctx* ctx, const Mod& m, Builder& b, Prog* out) {
  if (b.consts.size() + 2 > VM_MAX_CONST) return fail(ctx, SC_ERR_ARG, "too many constants in program");
  Prog p;
  p.nops = (uint32_t)b.ops.size();
  p.nscratch = b.nscratch;
  p.nconst = (uint32_t)b.consts.size();
  p.muls_per_item = b.muls;
  p.redcs_per_item = b.redcs;
  p.sqrs_per_item = b.sqrs;
  int rc = upload(ctx, b.ops.data(), b.ops.size() * sizeof(VmOp), (void**)&p.d_ops);
  if (rc) return rc;
  if (p.nconst) {
    rc = dev_alloc(ctx, (size_t)p.nconst * m.S * 4, (void**)&p.d_consts);
    if (rc) return rc;
    for (uint32_t i = 0; i < p.nconst; i++)
      HIPCHK(ctx, hipMemcpyAsync(p.d_consts + (size_t)i * m.S, ctx->consts[b.consts[i]].d_limbs, (size_t)m.S * 4,
                                 hipMemcpyDeviceToDevice, ctx->stream));   // stream-ordered before the program's first launch
  }
  *out = p;
  return SC_OK;
}

// Small batches: an L = 18 configuration with count * G lanes fills only part of the chip, and the run time is the latency of
// one wave's chain of products.  The same limb arrays (S = G L limbs, identical layout in memory) can be worked on by twice
// the lanes with half the limbs each -- (2G, 9) -- which doubles the waves and shortens the chain by 1.8x; the multiply-add
// density drops from 88 % to 78 % of the instruction stream, so this is used only while the L = 18 launch would leave at
// least half of the SIMDs without a wave.
inline bool use_latency_config(const sc_ctx* ctx, const Mod& m, uint64_t count) {
  if (ctx->latency_mode == 0 || m.L != 18 || m.W != 29 || m.G > 8) return false;
  if (ctx->latency_mode == 2) return true;
  const uint64_t waves = (count + (64 / m.G) - 1) / (64 / m.G);
  return waves <= (uint64_t)ctx->num_cu * 2;
}

int run_vm(sc_ctx* ctx, int mod, const Prog& p, const VmExt* exts, int next, uint64_t count, const uint32_t* fbt_rows = nullptr) {
  if (count == 0) return SC_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));  // the caller may have switched the current device since sc_ctx_create
  const Mod& m = ctx->mods[mod];
  VmArgs a;
  memset(&a, 0, sizeof a);
  a.modctx = m.d_ctx;
  a.consts = p.d_consts;
  a.prog = p.d_ops;
  a.fbt = fbt_rows;
  a.count = count;
  a.n0inv = m.n0inv;
  a.nops = p.nops;
  a.nconst_extra = p.nconst;
  a.nscratch = p.nscratch;
  a.small_c = m.small_c; a.small_cinv = m.small_cinv;
  for (int i = 0; i < next; i++) a.ext[i] = exts[i];
  // multiply-adds issued per item: a product or a reduction pass is S^2 (S/G limb steps x L per lane x G lanes); the a*a part
  // of a squaring is L(L+1)/2 per (lane, block) pair, G^2 pairs
  const bool lat = use_latency_config(ctx, m, count);
  const int G = lat ? 2 * m.G : m.G, L = lat ? 9 : m.L;
  ctx->mac_counter += (double)count * ((p.muls_per_item * 2.0 + p.redcs_per_item + p.sqrs_per_item) * (double)m.S * m.S +
                                        p.sqrs_per_item * (double)G * G * L * (L + 1) / 2.0);
  // a (4,18) modulus = -1 (mod 2^29) -- in practice the multiple M = c n of neg1_twin -- runs the instance without the quotient multiply
  const bool neg1 = G == 4 && L == 18 && m.W == 29 && m.n0inv == 1;
  int rc = launch_vm_part0(ctx, G, L, m.W, neg1, a);
  if (rc == SC_ERR_UNSUPPORTED) rc = launch_vm_part1(ctx, G, L, m.W, neg1, a);
  if (rc == SC_ERR_UNSUPPORTED) rc = launch_vm_part2(ctx, G, L, m.W, neg1, a);
  if (rc == SC_ERR_UNSUPPORTED) return fail(ctx, rc, "no kernel configuration for G=%d L=%d", G, L);
  return rc;
}

// which instance of the pair kernel a launch of `count` items of modulus m takes
void pvm_instance(const sc_ctx* ctx, const Mod& m, uint64_t count, int* G, int* L, bool* neg1) {
  *G = m.G; *L = m.L;
  if (use_latency_config(ctx, m, count) && (m.G == 1 || m.G == 2 || m.G == 4)) { *G = 2 * m.G; *L = 9; }   // (2,9): the 512-bit primes of 1024-bit keys (BASELINE configs[0])
  // n = -1 (mod 2^29): the instances without the quotient multiply exist for (4,18), (4,14), (8,14)
  *neg1 = m.n0inv == 1 && ((*G == 4 && *L == 18) || (*L == 14 && (*G == 4 || *G == 8)));
}
// resident waves per CU of that instance (the runtime's occupancy answer, not a compiled-in assumption)
int pvm_occupancy(sc_ctx* ctx, int G, int L, bool neg1) {
  int occ = pvm_occupancy_part0(ctx, G, L, neg1);
  if (occ < 0) occ = pvm_occupancy_part1(ctx, G, L, neg1);
  if (occ < 0) occ = pvm_occupancy_part2(ctx, G, L, neg1);
  return occ;
}

// pair programs: nscratch counts limb-form entries (2 per pair entry); macs = multiply-adds per item
int run_pvm(sc_ctx* ctx, int mod, const Prog& p, const VmExt* exts, int next, uint64_t count) {
  if (count == 0) return SC_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const Mod& m = ctx->mods[mod];
  VmArgs a;
  memset(&a, 0, sizeof a);
  a.modctx = m.d_ctx; a.consts = p.d_consts; a.prog = p.d_ops; a.count = count; a.n0inv = m.n0inv;
  a.nops = p.nops; a.nconst_extra = p.nconst; a.nscratch = p.nscratch;
  for (int i = 0; i < next; i++) a.ext[i] = exts[i];
  ctx->mac_counter += (double)count * p.muls_per_item;   // pair programs carry their exact multiply-add count here
  if (!pair_capable(m.G, m.L, m.W)) return fail(ctx, SC_ERR_UNSUPPORTED, "no pair kernel for G=%d L=%d", m.G, m.L);
  int G, L; bool neg1;
  pvm_instance(ctx, m, count, &G, &L, &neg1);
  const bool stamp = neg1 && G == 4 && L == 18 && ctx->stamps != nullptr;     // the stamping twin of (4,18,neg1) is sc_clock_probe's diagnostic launch
  int rc = launch_pvm_part0(ctx, G, L, neg1, stamp, a);
  if (rc == SC_ERR_UNSUPPORTED) rc = launch_pvm_part1(ctx, G, L, neg1, stamp, a);
  if (rc == SC_ERR_UNSUPPORTED) rc = launch_pvm_part2(ctx, G, L, neg1, stamp, a);
  if (rc == SC_ERR_UNSUPPORTED) return fail(ctx, rc, "no pair kernel for G=%d L=%d", G, L);
  return rc;
}

VmExt mk_ext(const void* p, uint32_t stride, uint32_t nwords, uint64_t limit = ~0ull) {
  VmExt e; e.ptr = p; e.stride = stride; e.nwords = nwords; e.limit = limit; return e;
}

bool valid_mod(sc_ctx* ctx, int mod) { return ctx && mod >= 0 && mod < (int)ctx->mods.size(); }

// emit: ACC = (ext value reduced mod n), for an operand of x_words >= mod words (Horner over chunks)
void emit_load_reduced(sc_ctx* ctx, const Mod& m, Builder& b, int ext, int x_words, int kred_lds) {
  const int cw = m.nwords;
  const int nch = (x_words + cw - 1) / cw;
  for (int t = nch - 1; t >= 0; t--) {
    const int nw = std::min(cw, x_words - t * cw);
    if (t == nch - 1) {
      b.loadw(ext, 0, t * cw, nw);
    } else {
      b.mul_const(kred_lds);          // ACC *= 2^(32 cw)   (kred is Montgomery form of that power)
      b.addw(ext, 0, t * cw, nw);
    }
  }
  (void)ctx;
}

int best_window(int bits) {
  int best = 1; double bc = 1e30;
  for (int w = 1; w <= 6; w++) {
    double c = (double)(1 << (w - 1)) + (double)bits / (w + 1);
    if (c < bc) { bc = c; best = w; }
  }
  return best;
}
inline int ebit(const Big& e, int i) { return (e[i >> 5] >> (i & 31)) & 1; }

// emit sliding-window exponentiation of the Montgomery-form value currently in ACC; leaves ACC = x^e (Montgomery form)
void emit_pow_shared(Builder& b, const Exp& ex) {
  const int bits = ex.bits;
  if (bits == 0) { b.loadt_const(1); return; }
  const int w = best_window(bits);
  const int NT = 1 << (w - 1);            // odd powers x^1, x^3, ..
  b.stt(0);
  if (NT > 1) {
    b.sqr(); b.stt(NT);                   // x^2
    for (int k = 1; k < NT; k++) { b.loadt_tbl(k - 1); b.mul_tbl(NT); b.stt(k); }
  }
  bool first = true;
  int i = bits - 1;
  while (i >= 0) {
    if (!ebit(ex.e, i)) { b.sqr(); i--; continue; }
    int j = std::max(0, i - w + 1);
    while (!ebit(ex.e, j)) j++;
    int v = 0;
    for (int k = i; k >= j; k--) v = (v << 1) | ebit(ex.e, k);
    if (first) { b.loadt_tbl((v - 1) / 2); first = false; }
    else { for (int k = 0; k < i - j + 1; k++) b.sqr(); b.mul_tbl((v - 1) / 2); }
    i = j - 1;
  }
}

int get_const_kred(sc_ctx* ctx, int mod, int* out_cid);

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int sc_abi_version(void) { return SC_ABI_VERSION; }

int sc_ctx_create(int device_id, sc_ctx** out_ctx) {
  if (!out_ctx) return SC_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return SC_ERR_HIP;
  if (hipSetDevice(device_id) != hipSuccess) return SC_ERR_HIP;
  sc_ctx* c = new sc_ctx();
  c->device = device_id;
  // developer switches, read per context: SC_PAIR_HOLD_MS (target hold time of a resident wave of a long pair launch on a shared chip;
  // 0 = never cut such launches), SC_PAIR_SEGMENTS=1 (the same "never", kept from round 4)
  if (const char* e = getenv("SC_PAIR_HOLD_MS")) c->pair_hold_ms = std::max(0.0, atof(e));
  if (const char* e = getenv("SC_PAIR_SEGMENTS")) if (atoi(e) == 1) c->pair_hold_ms = 0.0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->num_cu = prop.multiProcessorCount;
  *out_ctx = c;
  return SC_OK;
}

void sc_ctx_destroy(sc_ctx* ctx) {
  if (!ctx) return;
  // teardown: errors are not reportable any more, results deliberately ignored
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (void* p : ctx->owned) (void)hipFree(p);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->scratch_aux) (void)hipFree(ctx->scratch_aux);
  if (ctx->aux_fork) (void)hipEventDestroy(ctx->aux_fork);
  if (ctx->aux_join) (void)hipEventDestroy(ctx->aux_join);
  if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
  for (auto& kv : ctx->tmp) if (kv.second.first) (void)hipFree(kv.second.first);
  for (int* h : ctx->status_host) if (h) (void)hipHostFree(h);
  if (ctx->switch_event) (void)hipEventDestroy(ctx->switch_event);
  if (ctx->comm) (void)sc_comm_destroy(ctx);
  if (ctx->scheme_keys) free_scheme_keys(ctx->scheme_keys);
  memset(&ctx->rng_key, 0, sizeof ctx->rng_key);            // the generator's key does not outlive the context
  delete ctx;
}

int sc_ctx_set_latency_mode(sc_ctx* ctx, int mode) {
  if (!ctx || mode < 0 || mode > 2) return SC_ERR_ARG;
  ctx->latency_mode = mode;
  return SC_OK;
}
// The context reuses its scratch arena, temporaries and parked tables from call to call, so work queued on the previous stream
// must be ordered before anything the next stream does with them: an event recorded on the old stream, waited for on the new.
int sc_ctx_set_stream(sc_ctx* ctx, void* hip_stream) {
  if (!ctx) return SC_ERR_ARG;
  hipStream_t s = (hipStream_t)hip_stream;
  if (s == ctx->stream) return SC_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (!ctx->switch_event) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->switch_event, hipEventDisableTiming));
  if (hipEventRecord(ctx->switch_event, ctx->stream) == hipSuccess) {
    HIPCHK(ctx, hipStreamWaitEvent(s, ctx->switch_event, 0));
  } else {
    (void)hipGetLastError();                       // the previous stream no longer exists: its work was completed when it was destroyed
  }
  ctx->stream = s;
  return SC_OK;
}
int sc_ctx_synchronize(sc_ctx* ctx) { if (!ctx) return SC_ERR_ARG; HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); return SC_OK; }
const char* sc_last_error(sc_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int64_t sc_last_bad_index(sc_ctx* ctx) { return ctx ? ctx->last_bad_index : -1; }

int sc_malloc(sc_ctx* ctx, size_t bytes, void** out_dptr) {
  if (!ctx || !out_dptr) return SC_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMalloc(out_dptr, bytes ? bytes : 4));
  return SC_OK;
}
int sc_free(sc_ctx* ctx, void* dptr) { if (!ctx) return SC_ERR_ARG; HIPCHK(ctx, hipFree(dptr)); return SC_OK; }
int sc_memcpy_h2d(sc_ctx* ctx, void* dptr, const void* hptr, size_t bytes) {
  if (!ctx) return SC_ERR_ARG;
  HIPCHK(ctx, hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}
int sc_memcpy_d2h(sc_ctx* ctx, void* hptr, const void* dptr, size_t bytes) {
  if (!ctx) return SC_ERR_ARG;
  HIPCHK(ctx, hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

static int create_mod(sc_ctx* ctx, const uint32_t* n_hptr, int nwords, bool for_pairs, int* out_mod, const Config* forced = nullptr);

int sc_mod_create(sc_ctx* ctx, const uint32_t* n_hptr, int nwords, int* out_mod) {
  return create_mod(ctx, n_hptr, nwords, false, out_mod);
}

// for_pairs: take the first configuration that has a pair kernel (used for the internal twin context of sc_modexp_shared_sq);
// forced: this configuration or failure (the one-lane twin)
static int create_mod(sc_ctx* ctx, const uint32_t* n_hptr, int nwords, bool for_pairs, int* out_mod, const Config* forced) {
  if (!ctx || !n_hptr || nwords <= 0 || !out_mod) return fail(ctx, SC_ERR_ARG, "sc_mod_create: bad argument");
  Mod m;
  m.n.assign(n_hptr, n_hptr + nwords);
  m.nwords = nwords;
  m.nbits = big_bits(m.n);
  if (m.nbits < 2 || !(m.n[0] & 1)) return fail(ctx, SC_ERR_ARG, "sc_mod_create: modulus must be odd and > 1");
  auto fits = [&](const Config& c, bool need_words) {
    const int cap = c.W * c.G * c.L;
    return cap >= m.nbits + 8 && (!need_words || cap >= 32 * nwords);
  };
  if (forced) {
    if (fits(*forced, false)) { m.G = forced->G; m.L = forced->L; m.W = forced->W; }
  } else {
    for (int pass = 0; pass < 2 && !m.G; pass++)
      for (const Config& c : kConfigs) {
        if (!c.primary || (for_pairs && !pair_capable(c.G, c.L, c.W))) continue;
        if (fits(c, pass == 0)) { m.G = c.G; m.L = c.L; m.W = c.W; break; }
      }
  }
  if (!m.G) return fail(ctx, SC_ERR_UNSUPPORTED, "sc_mod_create: %d-bit modulus exceeds the largest configuration", m.nbits);
  m.S = m.G * m.L;
  const int W = m.W;
  const uint32_t LMASK = (1u << W) - 1;
  // the residue arrays must be representable below R = 2^(W S)
  if (32 * nwords > W * m.S + 31) return fail(ctx, SC_ERR_ARG, "sc_mod_create: nwords=%d too wide for a %d-bit modulus", nwords, m.nbits);
  // n0inv = -n^-1 mod 2^W (Newton)
  uint32_t n0 = m.n[0], inv = 1;
  for (int i = 0; i < 6; i++) inv *= 2 - n0 * inv;
  m.n0inv = (0u - inv) & LMASK;
  Big one(nwords, 0); one[0] = 1;
  Big r1 = big_shl_mod(one, m.n, W * m.S);
  Big r2 = big_shl_mod(r1, m.n, W * m.S);
  std::vector<uint32_t> ctxv;
  auto app = [&](const Big& x) { auto l = to_limbs(x, m.S, W); ctxv.insert(ctxv.end(), l.begin(), l.end()); };
  app(m.n); app(r2); app(r1);
  int rc = upload(ctx, ctxv.data(), ctxv.size() * 4, (void**)&m.d_ctx);
  if (rc) return rc;
  ctx->mods.push_back(m);
  *out_mod = (int)ctx->mods.size() - 1;
  return SC_OK;
}

int sc_mod_words(sc_ctx* ctx, int mod) { return valid_mod(ctx, mod) ? ctx->mods[mod].nwords : SC_ERR_ARG; }

int sc_exp_create(sc_ctx* ctx, const uint32_t* e_hptr, int ewords, int* out_exp) {
  if (!ctx || !e_hptr || ewords <= 0 || !out_exp) return fail(ctx, SC_ERR_ARG, "sc_exp_create: bad argument");
  Exp e; e.e.assign(e_hptr, e_hptr + ewords); e.bits = big_bits(e.e);
  ctx->exps.push_back(e);
  *out_exp = (int)ctx->exps.size() - 1;
  return SC_OK;
}

int sc_const_create(sc_ctx* ctx, int mod, const uint32_t* v_hptr, int nwords, int* out_const) {
  if (!valid_mod(ctx, mod) || !v_hptr || !out_const) return fail(ctx, SC_ERR_ARG, "sc_const_create: bad argument");
  const Mod& m = ctx->mods[mod];
  Big v(v_hptr, v_hptr + nwords);
  v.resize(std::max(nwords, m.nwords), 0);
  Big nn = m.n; nn.resize(v.size(), 0);
  Big vm = big_shl_mod(v, nn, m.W * m.S);  // Montgomery form
  auto l = to_limbs(vm, m.S, m.W);
  Const c; c.mod = mod;
  int rc = upload(ctx, l.data(), l.size() * 4, (void**)&c.d_limbs);
  if (rc) return rc;
  ctx->consts.push_back(c);
  *out_const = (int)ctx->consts.size() - 1;
  return SC_OK;
}

}  // extern "C"

namespace {
// registers the residue v (host words) as a constant of `mod` once per (mod, value)
int sc_const_create_cached(sc_ctx* ctx, int mod, const Big& v, int* out_cid) {
  auto key = std::make_pair(mod, v);
  auto it = ctx->const_by_value.find(key);
  if (it != ctx->const_by_value.end()) { *out_cid = it->second; return SC_OK; }
  int cid;
  int rc = sc_const_create(ctx, mod, v.data(), (int)v.size(), &cid);
  if (rc) return rc;
  ctx->const_by_value[key] = cid;
  *out_cid = cid;
  return SC_OK;
}
// Montgomery form of 2^(32 * nwords): multiplying by it shifts a residue up by one operand width
int get_const_kred(sc_ctx* ctx, int mod, int* out_cid) {
  const Mod& m = ctx->mods[mod];
  auto it = ctx->kred_cache.find(mod);
  if (it != ctx->kred_cache.end()) { *out_cid = it->second; return SC_OK; }
  Big one(m.nwords, 0); one[0] = 1;
  Big v = big_shl_mod(one, m.n, 32 * m.nwords);
  int cid;
  int rc = sc_const_create(ctx, mod, v.data(), m.nwords, &cid);
  if (rc) return rc;
  ctx->kred_cache[mod] = cid;
  *out_cid = cid;
  return SC_OK;
}
}  // namespace

extern "C" {

int sc_modmul(sc_ctx* ctx, int mod, const uint32_t* a, int a_stride, const uint32_t* b, int b_stride, uint32_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;  // empty batch: nothing to do (pointers may be null)
  if (!valid_mod(ctx, mod) || !a || !b || !out) return fail(ctx, SC_ERR_ARG, "sc_modmul: bad argument");
  const Mod& m = ctx->mods[mod];
  std::string key = "modmul:" + std::to_string(mod);
  auto it = ctx->progs.find(key);
  if (it == ctx->progs.end()) {
    Builder bd;
    bd.loadw(0); bd.mul_const(0);      // a * R
    bd.mul_extw(1);                    // (aR) * b / R = a b
    bd.storew(2); bd.end();
    Prog p; int rc = finalize_prog(ctx, m, bd, &p); if (rc) return rc;
    it = ctx->progs.emplace(key, p).first;
  }
  VmExt ex[3] = {mk_ext(a, a_stride, m.nwords), mk_ext(b, b_stride, m.nwords), mk_ext(out, m.nwords, m.nwords)};
  return run_vm(ctx, mod, it->second, ex, 3, count);
}

int sc_modmul_const(sc_ctx* ctx, int mod, const uint32_t* a, int cst, uint32_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;  // empty batch: nothing to do (pointers may be null)
  if (!valid_mod(ctx, mod) || !a || !out || cst < 0 || cst >= (int)ctx->consts.size() || ctx->consts[cst].mod != mod)
    return fail(ctx, SC_ERR_ARG, "sc_modmul_const: bad argument");
  const Mod& m = ctx->mods[mod];
  std::string key = "modmulc:" + std::to_string(mod) + ":" + std::to_string(cst);
  auto it = ctx->progs.find(key);
  if (it == ctx->progs.end()) {
    Builder bd;
    int c = bd.use_const(cst);
    bd.loadw(0); bd.mul_const(c);      // a * (cR) / R = a c
    bd.storew(1); bd.end();
    Prog p; int rc = finalize_prog(ctx, m, bd, &p); if (rc) return rc;
    it = ctx->progs.emplace(key, p).first;
  }
  VmExt ex[2] = {mk_ext(a, m.nwords, m.nwords), mk_ext(out, m.nwords, m.nwords)};
  return run_vm(ctx, mod, it->second, ex, 2, count);
}

int sc_modmul_const_sel(sc_ctx* ctx, int mod, const uint32_t* a, int cst0, int cst1, const uint8_t* flags, uint32_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  if (!valid_mod(ctx, mod) || !a || !flags || !out) return fail(ctx, SC_ERR_ARG, "sc_modmul_const_sel: bad argument");
  for (int c : {cst0, cst1})
    if (c != -1 && (c < 0 || c >= (int)ctx->consts.size() || ctx->consts[c].mod != mod)) return fail(ctx, SC_ERR_ARG, "sc_modmul_const_sel: bad constant");
  const Mod& m = ctx->mods[mod];
  std::string key = "mmsel:" + std::to_string(mod) + ":" + std::to_string(cst0) + ":" + std::to_string(cst1);
  auto it = ctx->progs.find(key);
  if (it == ctx->progs.end()) {
    Builder bd;
    const int l0 = cst0 < 0 ? 1 : bd.use_const(cst0), l1 = cst1 < 0 ? 1 : bd.use_const(cst1);   // LDS constant 1 = R mod n = the residue 1
    bd.loadw(0);
    bd.emit(OP_MUL, AK_CONSTSEL, 0, 2, (uint32_t)l0 | ((uint32_t)l1 << 8)); bd.muls++;           // a * (c R) / R = a c
    bd.storew(1); bd.end();
    Prog p; int rc = finalize_prog(ctx, m, bd, &p); if (rc) return rc;
    it = ctx->progs.emplace(key, p).first;
  }
  VmExt ex[3] = {mk_ext(a, m.nwords, m.nwords), mk_ext(out, m.nwords, m.nwords), mk_ext(flags, 0, 0)};
  return run_vm(ctx, mod, it->second, ex, 3, count);
}

static int onelane_for(sc_ctx* ctx, int mod, uint64_t count);

// any_flags_clean: the caller guarantees that any_flags[0 .. inner) is zero already (a context-owned accumulator that its reader
// resets, OP_TAKEFLAG): no clearing launch
static int modexp_shared_impl(sc_ctx* ctx, int mod, int exp, const uint32_t* x, int x_words, const uint32_t* mul_into,
                              uint32_t* out, uint8_t* flags, uint64_t count, uint64_t* any_flags, uint64_t inner, bool any_flags_clean = false) {
  if (ctx && count == 0) return SC_OK;  // empty batch: nothing to do (pointers may be null)
  if (!valid_mod(ctx, mod) || exp < 0 || exp >= (int)ctx->exps.size() || !x || (!out && !flags && !any_flags))
    return fail(ctx, SC_ERR_ARG, "sc_modexp_shared: bad argument");
  if (any_flags) {
    if (inner == 0 || count % inner != 0) return fail(ctx, SC_ERR_ARG, "sc_modexp_shared_isone_any: count must be a multiple of the inner count");
    if (!any_flags_clean) HIPCHK(ctx, hipMemsetAsync(any_flags, 0, inner * sizeof(uint64_t), ctx->stream));
  }
  if (ctx->exps[exp].bits > 64) mod = onelane_for(ctx, mod, count);   // long exponentiations of a chip-filling batch: one-lane twin
  const Mod& m = ctx->mods[mod];
  if (x_words <= 0) x_words = m.nwords;
  const int mode = any_flags ? 3 : (flags ? 2 : (mul_into ? 1 : 0));
  std::string key = "mexp:" + std::to_string(mod) + ":" + std::to_string(exp) + ":" + std::to_string(x_words) + ":" + std::to_string(mode);
  auto it = ctx->progs.find(key);
  if (it == ctx->progs.end()) {
    Builder bd;
    if (x_words > m.nwords) {
      int kc; int rc = get_const_kred(ctx, mod, &kc); if (rc) return rc;
      emit_load_reduced(ctx, m, bd, 0, x_words, bd.use_const(kc));
    } else {
      bd.loadw(0, 0, 0, x_words);
    }
    bd.mul_const(0);                   // to Montgomery form
    emit_pow_shared(bd, ctx->exps[exp]);
    if (mode == 0) { bd.redc(); bd.storew(1); }
    else if (mode == 1) { bd.mul_extw(2); bd.storew(1); }
    else if (mode == 2) { bd.storeflag(1, 0, 1); }    // compare with R mod n (Montgomery one)
    else { bd.emit(OP_STOREFLAG, 0, 1, 1, 0, 1); }    // ... and OR the verdict into the item's group flag
    bd.end();
    Prog p; int rc = finalize_prog(ctx, m, bd, &p); if (rc) return rc;
    it = ctx->progs.emplace(key, p).first;
  }
  VmExt ex[3] = {mk_ext(x, x_words, x_words),
                 any_flags ? mk_ext(any_flags, 0, 0, inner) : (flags ? mk_ext(flags, 0, 0) : mk_ext(out, m.nwords, m.nwords)),
                 mk_ext(mul_into, m.nwords, m.nwords)};
  return run_vm(ctx, mod, it->second, ex, 3, count);
}

int sc_modexp_shared(sc_ctx* ctx, int mod, int exp, const uint32_t* x, int x_words, const uint32_t* mul_into, uint32_t* out, uint64_t count) {
  return modexp_shared_impl(ctx, mod, exp, x, x_words, mul_into, out, nullptr, count, nullptr, 0);
}
int sc_modexp_shared_isone(sc_ctx* ctx, int mod, int exp, const uint32_t* x, int x_words, uint8_t* flags, uint64_t count) {
  return modexp_shared_impl(ctx, mod, exp, x, x_words, nullptr, nullptr, flags, count, nullptr, 0);
}
int sc_modexp_shared_isone_any(sc_ctx* ctx, int mod, int exp, const uint32_t* x, int x_words, uint64_t inner, uint64_t* any_flags,
                               uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  if (!any_flags) return fail(ctx, SC_ERR_ARG, "sc_modexp_shared_isone_any: no output");
  return modexp_shared_impl(ctx, mod, exp, x, x_words, nullptr, nullptr, nullptr, count, any_flags, inner);
}

int sc_fbt_create(sc_ctx* ctx, int mod, const uint32_t* base_hptr, int exp_bits, int window, int* out_fbt) {
  if (!valid_mod(ctx, mod) || !base_hptr || exp_bits <= 0 || window < 1 || window > 24 || !out_fbt)
    return fail(ctx, SC_ERR_ARG, "sc_fbt_create: bad argument");
  const Mod& m = ctx->mods[mod];
  Fbt f; f.mod = mod; f.window = window; f.exp_bits = exp_bits; f.nwin = (exp_bits + window - 1) / window;
  const uint64_t rows = (uint64_t)f.nwin << window;
  f.rows = std::make_shared<FbtRows>();
  f.rows->device = ctx->device; f.rows->bytes = rows * m.S * 4;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMalloc((void**)&f.rows->d, f.rows->bytes));
  f.d_rows = f.rows->d;
  int rc;
  int cbase; rc = sc_const_create(ctx, mod, base_hptr, m.nwords, &cbase); if (rc) return rc;
  // stage A: B_j = base^(2^(window j)), limb form, one item
  uint32_t* d_B; rc = dev_alloc(ctx, (size_t)f.nwin * m.S * 4, (void**)&d_B); if (rc) return rc;
  {
    Builder bd; int c = bd.use_const(cbase);
    bd.loadt_const(c); bd.storel(0, 0);
    for (int j = 1; j < f.nwin; j++) { for (int k = 0; k < window; k++) bd.sqr(); bd.storel(0, j); }
    bd.end();
    Prog p; rc = finalize_prog(ctx, m, bd, &p); if (rc) return rc;
    VmExt ex[1] = {mk_ext(d_B, m.S, 0)};
    rc = run_vm(ctx, mod, p, ex, 1, 1); if (rc) return rc;
  }
  // stage B: per window, item ch computes rows d = ch, ch + NCH, ... (start B^ch, step B^NCH)
  const int lg = std::min(window, 10);
  const uint32_t NCH = 1u << lg, CH = (1u << window) / NCH;
  std::vector<uint32_t> idx(NCH); for (uint32_t i = 0; i < NCH; i++) idx[i] = i;
  uint32_t* d_idx; rc = upload(ctx, idx.data(), NCH * 4, (void**)&d_idx); if (rc) return rc;
  {
    Builder bd;
    bd.loadt_extl(0); bd.stt(1);            // t[1] = B_j
    bd.loadt_const(1); bd.stt(0);           // t[0] = one
    bd.loadt_tbl(1); for (int k = 0; k < lg; k++) bd.sqr(); bd.stt(2);   // t[2] = B_j^NCH
    bd.loadt_tbldig(1, lg - 1, 1, 0);       // B_j^ch by square-and-multiply over the bits of ch
    for (int k = lg - 2; k >= 0; k--) { bd.sqr(); bd.mul_tbldig(1, k, 1, 0); }
    bd.storel(2, 0);
    for (uint32_t i = 1; i < CH; i++) { bd.mul_tbl(2); bd.storel(2, i); }
    bd.end();
    Prog p; rc = finalize_prog(ctx, m, bd, &p); if (rc) return rc;
    for (int j = 0; j < f.nwin; j++) {
      VmExt ex[3] = {mk_ext(d_B + (size_t)j * m.S, 0, 0), mk_ext(d_idx, 1, 1),
                     mk_ext(f.d_rows + ((uint64_t)j << window) * m.S, m.S, 0)};
      rc = run_vm(ctx, mod, p, ex, 3, NCH); if (rc) return rc;
    }
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->fbts.push_back(f);
  *out_fbt = (int)ctx->fbts.size() - 1;
  return SC_OK;
}

int sc_fbt_import(sc_ctx* ctx, int mod, sc_ctx* src_ctx, int src_fbt, int* out_fbt) {
  if (!valid_mod(ctx, mod) || !src_ctx || src_fbt < 0 || src_fbt >= (int)src_ctx->fbts.size() || !out_fbt)
    return fail(ctx, SC_ERR_ARG, "sc_fbt_import: bad argument");
  const Fbt& sf = src_ctx->fbts[src_fbt];
  const Mod& sm = src_ctx->mods[sf.mod];
  const Mod& m = ctx->mods[mod];
  if (src_ctx->device != ctx->device) return fail(ctx, SC_ERR_ARG, "sc_fbt_import: the table lives on another device");
  if (sm.G != m.G || sm.L != m.L || sm.W != m.W || sm.n != m.n)
    return fail(ctx, SC_ERR_ARG, "sc_fbt_import: the table was built for another modulus");
  Fbt f = sf;            // shares the rows (reference-counted); read-only from here on
  f.mod = mod;
  ctx->fbts.push_back(f);
  *out_fbt = (int)ctx->fbts.size() - 1;
  return SC_OK;
}

int sc_fbt_bytes(sc_ctx* ctx, int fbt, uint64_t* out_bytes) {
  if (!ctx || fbt < 0 || fbt >= (int)ctx->fbts.size() || !out_bytes) return fail(ctx, SC_ERR_ARG, "sc_fbt_bytes: bad argument");
  *out_bytes = ctx->fbts[fbt].rows ? (uint64_t)ctx->fbts[fbt].rows->bytes : 0;
  return SC_OK;
}

int sc_fixedbase_pow(sc_ctx* ctx, int fbt, const uint32_t* e, int ewords, const uint32_t* mul_into, uint32_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;  // empty batch: nothing to do (pointers may be null)
  if (!ctx || fbt < 0 || fbt >= (int)ctx->fbts.size() || !e || !out || ewords <= 0) return fail(ctx, SC_ERR_ARG, "sc_fixedbase_pow: bad argument");
  const Fbt& f = ctx->fbts[fbt];
  const Mod& m = ctx->mods[f.mod];
  std::string key = "fbp:" + std::to_string(fbt) + ":" + std::to_string(mul_into ? 1 : 0);
  auto it = ctx->progs.find(key);
  if (it == ctx->progs.end()) {
    Builder bd;
    bd.loadt_fbt(0, 0, f.window, 0);
    for (int j = 1; j < f.nwin; j++) bd.mul_fbt(0, j * f.window, f.window, j);
    if (mul_into) bd.mul_extw(2); else bd.redc();
    bd.storew(1); bd.end();
    Prog p; int rc = finalize_prog(ctx, m, bd, &p); if (rc) return rc;
    it = ctx->progs.emplace(key, p).first;
  }
  VmExt ex[3] = {mk_ext(e, ewords, ewords), mk_ext(out, m.nwords, m.nwords), mk_ext(mul_into, m.nwords, m.nwords)};
  return run_vm(ctx, f.mod, it->second, ex, 3, count, f.d_rows);
}

static int neg1_vm_twin(sc_ctx* ctx, int mod, uint64_t count);
// premul (nullable): a finished factor per item (e.g. a randomizer h^r computed ahead of time on another stream) multiplied in
// before the store -- the alternative to the fixed-base tail (fbt / e2)
// perm (nullable, instead of dest): int64 [count / planes][planes], one permutation per comparison of a bit-major vector
// [planes][inner]: item (j, b) is stored at plane k with perm[b][k] == j (the step-4i shuffle inside the store)
static int modexp_var_impl(sc_ctx* ctx, int mod, const uint32_t* x, const uint32_t* e, int ewords, int ebits, int fbt,
                           const uint32_t* e2, int e2words, const uint64_t* dest, uint32_t* out, uint64_t count,
                           const uint32_t* premul = nullptr, const int64_t* perm = nullptr, uint32_t planes = 0) {
  if (ctx && count == 0) return SC_OK;  // empty batch: nothing to do (pointers may be null)
  if (!valid_mod(ctx, mod) || !x || !e || !out || ewords <= 0 || ebits <= 0 || ebits > 32 * ewords)
    return fail(ctx, SC_ERR_ARG, "sc_modexp_var: bad argument");
  if (perm && (dest || planes == 0 || planes > 128 || count % planes != 0)) return fail(ctx, SC_ERR_ARG, "sc_modexp_var: bad permutation batch");
  const Fbt* f = nullptr;
  if (fbt >= 0) {
    if (fbt >= (int)ctx->fbts.size() || ctx->fbts[fbt].mod != mod || !e2 || e2words <= 0) return fail(ctx, SC_ERR_ARG, "sc_modexp_var: bad fixed-base table");
    f = &ctx->fbts[fbt];
  }
  // chip-filling batches of a (4,18) modulus run in the context of its multiple M = c n = -1 (mod 2^29) (no quotient multiply per
  // limb step): inputs, constants and table rows are residues modulo n and therefore valid modulo M as they are; the program
  // leaves through the reduction pass times c and the exact division by c (sc_device.h).  Not with premul: that variant leaves
  // Montgomery form through a product, which has no room for the factor c.
  const int tmod = premul ? -1 : neg1_vm_twin(ctx, mod, count);
  const bool twin = tmod >= 0;
  const int rmod = twin ? tmod : mod;
  const Mod& m = ctx->mods[mod];        // (taken after the twin exists: creating it may move the table of moduli)
  std::string key = "mvar:" + std::to_string(rmod) + ":" + std::to_string(ebits) + ":" + std::to_string(fbt) + (dest ? ":s" : "") + (perm ? ":q" : "") + (premul ? ":p" : "");
  auto it = ctx->progs.find(key);
  if (it == ctx->progs.end()) {
    const int w = ebits <= 4 ? 1 : (ebits <= 12 ? 2 : 3);
    const int nd = (ebits + w - 1) / w;
    Builder bd;
    bd.loadw(0); bd.mul_const(0); bd.stt(1);          // t[1] = x (Montgomery)
    bd.loadt_const(1); bd.stt(0);                      // t[0] = 1
    for (int k = 2; k < (1 << w); k++) { bd.loadt_tbl(k - 1); bd.mul_tbl(1); bd.stt(k); }
    bd.loadt_tbldig(1, (nd - 1) * w, w, 0);
    for (int d = nd - 2; d >= 0; d--) { for (int k = 0; k < w; k++) bd.sqr(); bd.mul_tbldig(1, d * w, w, 0); }
    if (f) for (int j = 0; j < f->nwin; j++) bd.mul_fbt(3, j * f->window, f->window, j);
    if (premul) bd.mul_extw(5); else bd.redc(twin);      // (x^e R) * premul / R = x^e premul: leaves Montgomery form by itself
    if (dest || perm) bd.storew_at(2, 4, twin, perm != nullptr); else bd.storew(2, 0, twin);
    bd.end();
    Prog p; int rc = finalize_prog(ctx, ctx->mods[rmod], bd, &p); if (rc) return rc;
    it = ctx->progs.emplace(key, p).first;
  }
  // a scattered store never leaves the output array: rows >= count are dropped by the limit of the output operand
  VmExt ex[6] = {mk_ext(x, m.nwords, m.nwords), mk_ext(e, ewords, ewords), mk_ext(out, m.nwords, m.nwords, (dest || perm) ? count : ~0ull),
                 mk_ext(e2, e2words, e2words), perm ? mk_ext(perm, planes, 0, count / planes) : mk_ext(dest, 2, 2), mk_ext(premul, m.nwords, m.nwords)};
  return run_vm(ctx, rmod, it->second, ex, 6, count, f ? f->d_rows : nullptr);
}

int sc_modexp_var(sc_ctx* ctx, int mod, const uint32_t* x, const uint32_t* e, int ewords, int ebits, int fbt,
                  const uint32_t* e2, int e2words, uint32_t* out, uint64_t count) {
  return modexp_var_impl(ctx, mod, x, e, ewords, ebits, fbt, e2, e2words, nullptr, out, count);
}

int sc_modexp_var_scatter(sc_ctx* ctx, int mod, const uint32_t* x, const uint32_t* e, int ewords, int ebits, int fbt,
                          const uint32_t* e2, int e2words, const uint64_t* dest_index, uint32_t* out, uint64_t count) {
  if (ctx && count != 0 && !dest_index) return fail(ctx, SC_ERR_ARG, "sc_modexp_var_scatter: no destination index");
  return modexp_var_impl(ctx, mod, x, e, ewords, ebits, fbt, e2, e2words, dest_index, out, count);
}

static int paillier_encrypt_raw_impl(sc_ctx* ctx, int mod_n2, int cst_n, const uint32_t* mwords, int m_words, uint32_t* out, uint64_t count, bool negate);

int sc_paillier_encrypt_raw(sc_ctx* ctx, int mod_n2, int cst_n, const uint32_t* mwords, int m_words, uint32_t* out, uint64_t count) {
  return paillier_encrypt_raw_impl(ctx, mod_n2, cst_n, mwords, m_words, out, count, false);
}
int sc_paillier_encrypt_raw_neg(sc_ctx* ctx, int mod_n2, int cst_n, const uint32_t* mwords, int m_words, uint32_t* out, uint64_t count) {
  return paillier_encrypt_raw_impl(ctx, mod_n2, cst_n, mwords, m_words, out, count, true);
}

static int paillier_encrypt_raw_impl(sc_ctx* ctx, int mod_n2, int cst_n, const uint32_t* mwords, int m_words, uint32_t* out, uint64_t count, bool negate) {
  if (ctx && count == 0) return SC_OK;  // empty batch: nothing to do (pointers may be null)
  if (!valid_mod(ctx, mod_n2) || cst_n < 0 || cst_n >= (int)ctx->consts.size() || ctx->consts[cst_n].mod != mod_n2 || !mwords || !out || m_words <= 0)
    return fail(ctx, SC_ERR_ARG, "sc_paillier_encrypt_raw: bad argument");
  const Mod& m = ctx->mods[mod_n2];
  if (m_words > m.nwords) return fail(ctx, SC_ERR_ARG, "sc_paillier_encrypt_raw: plaintext wider than N^2");
  std::string key = std::string(negate ? "pencn:" : "penc:") + std::to_string(mod_n2) + ":" + std::to_string(cst_n) + ":" + std::to_string(m_words);
  auto it = ctx->progs.find(key);
  if (it == ctx->progs.end()) {
    Builder bd; int c = bd.use_const(cst_n);
    bd.loadw(0, 0, 0, m_words); bd.mul_const(c);   // m * (N R) / R = m N  (mod N^2)
    if (negate) bd.emit(OP_NEG);                   // -m N: the inverse ciphertext (1 + mN)^-1 = 1 - mN (mod N^2)
    bd.emit(OP_ADD1); bd.storew(1); bd.end();
    Prog p; int rc = finalize_prog(ctx, m, bd, &p); if (rc) return rc;
    it = ctx->progs.emplace(key, p).first;
  }
  VmExt ex[2] = {mk_ext(mwords, m_words, m_words), mk_ext(out, m.nwords, m.nwords)};
  return run_vm(ctx, mod_n2, it->second, ex, 2, count);
}

int sc_paillier_l_mul(sc_ctx* ctx, int mod, int cst_k, const uint32_t* x, int x_words, uint32_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;  // empty batch: nothing to do (pointers may be null)
  if (!valid_mod(ctx, mod) || cst_k < 0 || cst_k >= (int)ctx->consts.size() || ctx->consts[cst_k].mod != mod || !x || !out || x_words <= 0)
    return fail(ctx, SC_ERR_ARG, "sc_paillier_l_mul: bad argument");
  const Mod& m = ctx->mods[mod];
  std::string key = "plmul:" + std::to_string(mod) + ":" + std::to_string(cst_k) + ":" + std::to_string(x_words);
  auto it = ctx->progs.find(key);
  if (it == ctx->progs.end()) {
    Builder bd; int c = bd.use_const(cst_k);
    const int lw = std::min(x_words, (m.W * m.S + 31) / 32);   // only x mod R matters for the exact quotient
    bd.loadw(0, 0, 0, lw);
    bd.emit(OP_SUB1);                 // y = (x - 1) mod R, exact limbs
    bd.emit(OP_QUOT); bd.redcs++;     // y / n  (< n because x < n^2)
    bd.mul_const(c);                  // * k
    bd.storew(1); bd.end();
    Prog p; int rc = finalize_prog(ctx, m, bd, &p); if (rc) return rc;
    it = ctx->progs.emplace(key, p).first;
  }
  VmExt ex[2] = {mk_ext(x, x_words, x_words), mk_ext(out, m.nwords, m.nwords)};
  return run_vm(ctx, mod, it->second, ex, 2, count);
}

int sc_plain_alice(sc_ctx* ctx, const uint32_t* r, const uint32_t* n_hptr, int nw, int l, uint64_t count, uint32_t* m1,
                   uint64_t* alpha, uint64_t* alpha_tilde, uint64_t* rsmall, uint32_t* rshift) {
  if (ctx && count == 0) return SC_OK;  // empty batch: nothing to do (pointers may be null)
  if (!ctx || !r || !n_hptr || nw <= 0 || l <= 0 || l > 64 || !m1 || !alpha || !alpha_tilde || !rsmall || !rshift)
    return fail(ctx, SC_ERR_ARG, "sc_plain_alice: bad argument");
  if (count == 0) return SC_OK;
  uint32_t* d_n = nullptr;
  { int rc = device_n_half(ctx, n_hptr, nw, &d_n); if (rc) return rc; }
  if (launch_plain_alice(ctx->stream, r, d_n, d_n + nw, nw, l, count, m1, alpha, alpha_tilde, rsmall, rshift)) return fail(ctx, SC_ERR_HIP, "sc_plain_alice: launch failed");
  return SC_OK;
}

static int plain_bob_impl(sc_ctx* ctx, const uint32_t* z, const uint32_t* n_hptr, int nw, int l, uint64_t count, uint64_t* beta,
                          uint64_t* dbit, uint32_t* zeta1, uint32_t* zeta2, uint8_t* bits);
int sc_plain_bob(sc_ctx* ctx, const uint32_t* z, const uint32_t* n_hptr, int nw, int l, uint64_t count, uint64_t* beta,
                 uint64_t* dbit, uint32_t* zeta1, uint32_t* zeta2) {
  return plain_bob_impl(ctx, z, n_hptr, nw, l, count, beta, dbit, zeta1, zeta2, nullptr);
}
// bits (nullable): additionally the bytes [l+1][count] of d and the bits of beta (k_plain_bob)
static int plain_bob_impl(sc_ctx* ctx, const uint32_t* z, const uint32_t* n_hptr, int nw, int l, uint64_t count, uint64_t* beta,
                          uint64_t* dbit, uint32_t* zeta1, uint32_t* zeta2, uint8_t* bits) {
  if (ctx && count == 0) return SC_OK;  // empty batch: nothing to do (pointers may be null)
  if (!ctx || !z || !n_hptr || nw <= 0 || l <= 0 || l > 64 || !beta || !dbit || !zeta1 || !zeta2)
    return fail(ctx, SC_ERR_ARG, "sc_plain_bob: bad argument");
  if (count == 0) return SC_OK;
  uint32_t* d_n = nullptr;
  { int rc = device_n_half(ctx, n_hptr, nw, &d_n); if (rc) return rc; }
  if (launch_plain_bob(ctx->stream, z, d_n, d_n + nw, nw, l, count, beta, dbit, zeta1, zeta2, bits)) return fail(ctx, SC_ERR_HIP, "sc_plain_bob: launch failed");
  return SC_OK;
}

// ------------------------------------------------------------------------------------------------
// batch inversion: Montgomery's trick over strided chunks + on-device binary extended GCD at the top
// ------------------------------------------------------------------------------------------------
// One level of the tree, kept for the error path: its operand array, chunk length and chunk count
struct InvLevel { const uint32_t* x; uint64_t count; uint32_t K; uint64_t C; };
struct InvPending { std::vector<InvLevel> levels; int* d_status = nullptr; uint64_t top_count = 0; };

// Queues every launch of the inversion (up-sweeps, the division-step kernel at the top, down-sweeps) WITHOUT waiting for the top
// kernel's verdict: the host reads the status words once, after everything is in the stream (sc_modinv), so the GPU never idles
// on a host round trip in the middle of the tree.  A non-invertible element makes the results garbage; the caller reports it.
static int modinv_rec(sc_ctx* ctx, int mod, const uint32_t* x, uint32_t* out, uint64_t count, InvPending* pend, int depth) {
  const Mod& m = ctx->mods[mod];
  // residues inverted directly by the division-step kernel (one wave each, all of them resident at once: a few thousand cost
  // the latency of one); below that the tree's levels -- two latency-bound launches each -- cost more than they save
  const uint64_t TOP = SC_INV_TOP;
  if (count <= TOP) {
    int* d_status;
    { int rc0 = status_words(ctx, count, &d_status); if (rc0) return rc0; }
    uint32_t* d_nw = nullptr;
    { int rc0 = device_n_half(ctx, m.n.data(), m.nwords, &d_nw); if (rc0) return rc0; }
    int rc = launch_xgcd(ctx->stream, x, out, d_nw, m.nwords, count, d_status);
    if (rc != 0) return fail(ctx, SC_ERR_HIP, "xgcd launch failed");
    pend->d_status = d_status;
    pend->top_count = count;
    return SC_OK;
  }
  // Chunk length: a chunk is one sequential chain of 3 (K - 1) products, so the tree's latency is the sum of the chunk lengths
  // over its levels while its work (3 products per element) does not depend on K.  Take the shortest chunks that still fill
  // every resident group slot of the chip at this level (large batches: up to 24), and never fewer than 4 (fewer, wider levels).
  const uint64_t resident_groups = (uint64_t)ctx->num_cu * 8 * (64 / m.G);
  const uint32_t K = (uint32_t)std::min<uint64_t>(24, std::max<uint64_t>(4, (count + resident_groups - 1) / resident_groups));
  const uint64_t C = (count + K - 1) / K;
  uint32_t *d_P = nullptr, *d_tot = nullptr, *d_totinv = nullptr;
  { int rc0 = tmp_buf(ctx, TMP_INV_BASE + 2 * depth, (size_t)K * C * m.S * 4, (void**)&d_P); if (rc0) return rc0; }
  { int rc0 = tmp_buf(ctx, TMP_INV_BASE + 2 * depth + 1, (size_t)C * m.nwords * 4 * 2, (void**)&d_tot); if (rc0) return rc0; }
  d_totinv = d_tot + (size_t)C * m.nwords;
  // No operand is converted to Montgomery form and no result reduced out of it: the prefix products simply drift by one factor
  // 1/R per product,  Q_i = x_0 .. x_i / R^i  (Q_0 = x_0), the chunk total Q_{K-1} goes up the tree as it is (a residue like any
  // other), and its inverse u_{K-1} = (x_0 .. x_{K-1})^-1 R^(K-1) comes back carrying exactly the powers of R that the way down
  // divides out again:   x_i^-1 = u_i Q_{i-1} / R   and   u_{i-1} = u_i x_i / R = (x_0 .. x_{i-1})^-1 R^(i-1),   u_0 = x_0^-1.
  // Three products per element (round 3: six -- two conversions, a reduction pass and the same three).  Elements past the end
  // of the batch read as 1 and take part like any other factor.
  std::string k1 = "inv1d:" + std::to_string(mod) + ":" + std::to_string(K), k2 = "inv2d:" + std::to_string(mod) + ":" + std::to_string(K);
  auto it1 = ctx->progs.find(k1);
  if (it1 == ctx->progs.end()) {
    Builder bd;
    for (uint32_t i = 0; i < K; i++) {
      if (i == 0) bd.loadw(0, 0); else bd.mul_extw(0, i);
      bd.storel(1, i);
    }
    bd.storew(2); bd.end();
    Prog p; int rc = finalize_prog(ctx, m, bd, &p); if (rc) return rc;
    it1 = ctx->progs.emplace(k1, p).first;
  }
  auto it2 = ctx->progs.find(k2);
  if (it2 == ctx->progs.end()) {
    Builder bd;
    bd.loadw(3);                                           // u_{K-1}: the inverse of the chunk total
    for (uint32_t i = K - 1; i >= 1; i--) {
      bd.stt(0);
      bd.mul_extl(1, i - 1); bd.storew(4, i);              // x_i^-1 = u_i Q_{i-1} / R
      bd.loadt_tbl(0); bd.mul_extw(0, i);                  // u_{i-1} = u_i x_i / R
    }
    bd.storew(4, 0); bd.end();
    Prog p; int rc = finalize_prog(ctx, m, bd, &p); if (rc) return rc;
    it2 = ctx->progs.emplace(k2, p).first;
  }
  pend->levels.push_back(InvLevel{x, count, K, C});
  int rc;
  {
    VmExt ex[3] = {mk_ext(x, m.nwords, m.nwords, count), mk_ext(d_P, m.S, 0, (uint64_t)K * C), mk_ext(d_tot, m.nwords, m.nwords)};
    rc = run_vm(ctx, mod, it1->second, ex, 3, C);
  }
  if (!rc) rc = modinv_rec(ctx, mod, d_tot, d_totinv, C, pend, depth + 1);
  if (!rc) {
    VmExt ex[5] = {mk_ext(x, m.nwords, m.nwords, count), mk_ext(d_P, m.S, 0, (uint64_t)K * C), mk_ext(nullptr, 0, 0),
                   mk_ext(d_totinv, m.nwords, m.nwords), mk_ext(out, m.nwords, m.nwords, count)};
    rc = run_vm(ctx, mod, it2->second, ex, 5, C);
  }
  return rc;
}

// The product of chunk `chunk` of a level (members x[i C + chunk], i < K) is not invertible: test its members one by one with the
// division-step kernel and return the first that fails, like the reference's pow / gmpy2.invert name the operand.
static int modinv_find_member(sc_ctx* ctx, const Mod& m, const InvLevel& lv, int64_t chunk, int64_t* out_index) {
  uint64_t members = 0;
  while (members < lv.K && members * lv.C + (uint64_t)chunk < lv.count) members++;
  uint32_t* d_m = nullptr; int* d_status = nullptr; uint32_t* d_nw = nullptr;
  { int rc0 = tmp_buf(ctx, TMP_INV_MEMBERS, (size_t)members * m.nwords * 4 * 2, (void**)&d_m); if (rc0) return rc0; }
  { int rc0 = status_words(ctx, members, &d_status); if (rc0) return rc0; }
  { int rc0 = device_n_half(ctx, m.n.data(), m.nwords, &d_nw); if (rc0) return rc0; }
  HIPCHK(ctx, hipMemcpy2DAsync(d_m, (size_t)m.nwords * 4, lv.x + (size_t)chunk * m.nwords, (size_t)lv.C * m.nwords * 4,
                               (size_t)m.nwords * 4, members, hipMemcpyDeviceToDevice, ctx->stream));
  if (launch_xgcd(ctx->stream, d_m, d_m + (size_t)members * m.nwords, d_nw, m.nwords, members, d_status) != 0)
    return fail(ctx, SC_ERR_HIP, "xgcd launch failed");
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  const volatile int* st = d_status;                    // written by the kernel, in host memory
  for (uint64_t i = 0; i < members; i++)
    if (st[i] != 1) { *out_index = (int64_t)(i * lv.C + (uint64_t)chunk); return SC_OK; }
  return fail(ctx, SC_ERR_HIP, "sc_modinv: chunk %lld is not invertible but all of its members are", (long long)chunk);
}

int sc_modinv(sc_ctx* ctx, int mod, const uint32_t* x, uint32_t* out, uint64_t count, int64_t* bad_index) {
  if (ctx && count == 0) return SC_OK;  // empty batch: nothing to do (pointers may be null)
  if (!valid_mod(ctx, mod) || !x || !out) return fail(ctx, SC_ERR_ARG, "sc_modinv: bad argument");
  {
    // The down-sweeps are queued before the verdicts are read, and the error path re-reads the operands to name the bad
    // element: `out` must not overlap `x` (an in-place call would have overwritten them with garbage by then).
    const size_t bytes = (size_t)count * ctx->mods[mod].nwords * 4;
    const char *xb = (const char*)x, *ob = (const char*)out;
    if (xb < ob + bytes && ob < xb + bytes) return fail(ctx, SC_ERR_ARG, "sc_modinv: out overlaps x (in-place inversion is not supported)");
  }
  if (bad_index) *bad_index = -1;
  InvPending pend;
  int rc = modinv_rec(ctx, mod, x, out, count, &pend, 0);
  if (rc) return rc;
  // the one host round trip of the call: the verdicts of the top kernel (in pinned host memory, written by the kernel itself),
  // read after every launch has been queued
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  const volatile int* st = pend.d_status;
  int64_t bad = -1;
  for (uint64_t i = 0; i < pend.top_count && bad < 0; i++) if (st[i] != 1) bad = (int64_t)i;
  if (bad < 0) return SC_OK;
  if (st[bad] == 2) return fail(ctx, SC_ERR_HIP, "sc_modinv: the inversion kernel left its proven value range (internal error)");
  // error path: `bad` indexes the deepest level's chunk products; walk back up, one member test per level
  const Mod& m = ctx->mods[mod];
  for (int lv = (int)pend.levels.size() - 1; lv >= 0; lv--) {
    int64_t member = -1;
    rc = modinv_find_member(ctx, m, pend.levels[lv], bad, &member);
    if (rc) return rc;
    bad = member;
  }
  if (bad_index) *bad_index = bad;
  ctx->last_bad_index = bad;
  return fail(ctx, SC_ERR_NOT_INVERTIBLE, "element %lld is not invertible", (long long)bad);
}

int sc_dgk_step4(sc_ctx* ctx, int mod, int cst_g, int cst_ginv, int l, const uint32_t* beta, const uint32_t* beta_inv,
                 const uint32_t* d, const uint32_t* d_inv, const uint64_t* alpha, const uint64_t* alpha_tilde,
                 const uint64_t* rsmall, const uint64_t* delta_a, uint32_t* c_out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;  // empty batch: nothing to do (pointers may be null)
  if (!valid_mod(ctx, mod) || l <= 0 || l > 64 || !beta || !beta_inv || !d || !d_inv || !alpha || !alpha_tilde || !rsmall || !delta_a || !c_out)
    return fail(ctx, SC_ERR_ARG, "sc_dgk_step4: bad argument");
  if (cst_g < 0 || cst_ginv < 0 || cst_g >= (int)ctx->consts.size() || cst_ginv >= (int)ctx->consts.size() ||
      ctx->consts[cst_g].mod != mod || ctx->consts[cst_ginv].mod != mod)
    return fail(ctx, SC_ERR_ARG, "sc_dgk_step4: bad constant");
  const Mod& m = ctx->mods[mod];
  std::string key = "step4:" + std::to_string(mod) + ":" + std::to_string(cst_g) + ":" + std::to_string(cst_ginv) + ":" + std::to_string(l);
  // Per comparison and bit the step formulas (SC/initiator.py:317-320, 368-371, 471-482) multiply by factors that depend only on the
  // comparison and on the two flag bits (alpha_i, alpha~_i):
  //   w_i = base_i * K[alpha_i][alpha~_i],   base_i = [beta_i] (alpha_i = 0) or [beta_i]^-1 (alpha_i = 1),
  //         K = 1, d'^-1, g d'^-1, g   for (0,0), (0,1), (1,0), (1,1)      (4d: [1] - [beta_i] = g [beta_i]^-1; 4e: - [d] where the bits differ)
  //   c_i = [beta_i]^-1 * C[alpha_i][alpha~_i] * (w_sum)^3,
  //         C = g^s, g^s d', g^s g d'^-1, g^s g                            (4h: [s] + alpha_i + [d] (alpha~_i - alpha_i))
  // so launch (a) computes the six factors that are not constants once per comparison and parks them, and the bit loop of launch
  // (b) spends 6 + i products per bit (round 3: 9 + i and a reduction pass; it multiplied g, d'^-1, D and g^s g^alpha in one after
  // the other, by the residue 1 where a factor did not apply, and converted [beta_i], [beta_i]^-1 to Montgomery form first).
  // No conversion and no reduction pass inside the loop: [beta_i] and [beta_i]^-1 enter as the plain residues they are,
  //   * the K factors are parked with a second factor R (K R^2): plain base_i times K R^2 / R = w_i in Montgomery form, which is
  //     what the squarings need;
  //   * the C factors in Montgomery form (C R): plain [beta_i]^-1 times C R / R is plain, times (w_sum)^3 R / R is plain -- the
  //     stored value; likewise g^delta_a is parked plain for c_-1.
  // Same factors, another order: the canonical residues are the same.
  //   park entries: 0 = K01 = d'^-1, 1 = K10 = g d'^-1 (both times R^2), 2 = C00 = g^s, 3 = C01 = g^s d', 4 = C10 = g^s g d'^-1,
  //                 5 = C11 = g^s g (times R), 6 = g^delta_a (plain)
  const int NP = 7;
  uint32_t* d_park;
  { int rc0 = tmp_buf(ctx, TMP_PARK, (size_t)NP * count * m.S * 4, (void**)&d_park); if (rc0) return rc0; }
  std::string ka = key + ":a3", kb = key + ":b3";
  auto ita = ctx->progs.find(ka);
  if (ita == ctx->progs.end()) {
    Builder bd; const int cg = bd.use_const(cst_g), cgi = bd.use_const(cst_ginv);
    // scratch (Montgomery form): 0 one, 1 g, 2 ginv, 3 d, 4 dinv, 5 d', 6 d'^-1, 7 g^s, 8 g d'^-1
    bd.loadt_const(1); bd.stt(0);
    bd.loadt_const(cg); bd.stt(1);
    bd.loadt_const(cgi); bd.stt(2);
    bd.loadw(0); bd.mul_const(0); bd.stt(3);
    bd.loadw(1); bd.mul_const(0); bd.stt(4);
    // d' = rsmall ? one : d ; d'^-1 likewise   (SC/initiator.py:289-290: [d] <- [0] = g^0 = 1)
    bd.loadt_tblsel(2, 0, 2, 0, 3, 3, 0, 0); bd.stt(5);
    bd.loadt_tblsel(2, 0, 2, 0, 4, 4, 0, 0); bd.stt(6);
    bd.mul_const(0); bd.storel(4, 0);                                              // K01 = d'^-1          (R^2)
    bd.loadt_tbl(6); bd.mul_tbl(1); bd.stt(8); bd.mul_const(0); bd.storel(4, 1);   // K10 = g d'^-1        (R^2)
    // g^s, s = 1 - 2 delta_a  (delta_a = 1 -> g^-1, SC/initiator.py:459-461)
    bd.loadt_tblsel(3, 0, 3, 0, 1, 1, 2, 2); bd.stt(7); bd.storel(4, 2);          // C00 = g^s
    bd.mul_tbl(5); bd.storel(4, 3);                                                // C01 = g^s d'
    bd.loadt_tbl(7); bd.mul_tbl(1); bd.storel(4, 5);                               // C11 = g^s g            (alpha_i = 1, :476)
    bd.loadt_tbl(7); bd.mul_tbl(8); bd.storel(4, 4);                               // C10 = g^s g d'^-1
    bd.loadt_tblsel(3, 0, 3, 0, 0, 0, 1, 1); bd.redc(); bd.storel(4, 6);           // g^delta_a (:484), plain
    bd.end();
    Prog p; int rc = finalize_prog(ctx, m, bd, &p); if (rc) return rc;
    ita = ctx->progs.emplace(ka, p).first;
  }
  {
    VmExt ex[5] = {mk_ext(d, m.nwords, m.nwords), mk_ext(d_inv, m.nwords, m.nwords), mk_ext(rsmall, 2, 2), mk_ext(delta_a, 2, 2),
                   mk_ext(d_park, m.S, 0)};
    int rc = run_vm(ctx, mod, ita->second, ex, 5, count); if (rc) return rc;
  }
  // ---- launch (b): the bit loop i = l-1 .. 0 (SC/initiator.py:471-482) then c_-1 (:484)
  auto itb = ctx->progs.find(kb);
  if (itb == ctx->progs.end()) {
    Builder bd; const int cg = bd.use_const(cst_g);
    // ext: 0 beta, 1 beta_inv, 2 park, 3 alpha, 4 alpha_tilde, 5 out
    // scratch: 0 K00 = R^2, 1 K01, 2 K10, 3 K11 = g R^2, 4 C00, 5 C01, 6 C10, 7 C11, 8 w_sum, 9 w_sum^3
    bd.loadt_const(0); bd.stt(0);
    bd.loadt_extl(2, 0); bd.stt(1);
    bd.loadt_extl(2, 1); bd.stt(2);
    bd.loadt_const(cg); bd.mul_const(0); bd.stt(3);
    for (int e = 0; e < 4; e++) { bd.loadt_extl(2, 2 + e); bd.stt(4 + e); }
    for (int i = l - 1; i >= 0; i--) {
      bd.loadw(1, i);                                                     // [beta_i]^-1, plain
      bd.mul_tblsel(3, i, 4, i, 4, 5, 6, 7);                              // times C                (4h, :471-478)   flags: fa = alpha_i, fb = alpha~_i
      if (i != l - 1) bd.mul_tbl(9);                                      // times (w_sum)^3 ; first iteration: `3 * 0` is the int 0 -> [0] = 1
      bd.storew(5, i + 1);                                                // c_i
      bd.loadw_sel(1, 0, 3, i, i);                                        // base_i, plain          (4d, :317-320)
      bd.mul_tblsel(3, i, 4, i, 0, 1, 2, 3);                              // w_i = base_i K         (4d + 4e, :368-371), Montgomery form
      for (int k = 0; k < i; k++) bd.sqr();                               // w_i^(2^i)              (4f, :406)
      if (i != l - 1) bd.mul_tbl(8);                                      // w_sum *= w_i^(2^i)
      bd.stt(8);
      if (i > 0) { bd.sqr(); bd.mul_tbl(8); bd.stt(9); }                  // its cube for the next bit
    }
    bd.loadt_extl(2, 6); bd.mul_tbl(8); bd.storew(5, 0);                  // c_-1 = g^delta_a * w_sum (:484)
    bd.end();
    Prog p; int rc = finalize_prog(ctx, m, bd, &p); if (rc) return rc;
    itb = ctx->progs.emplace(kb, p).first;
  }
  int rc;
  {
    VmExt ex[6] = {mk_ext(beta, m.nwords, m.nwords), mk_ext(beta_inv, m.nwords, m.nwords), mk_ext(d_park, m.S, 0),
                   mk_ext(alpha, 2, 2), mk_ext(alpha_tilde, 2, 2), mk_ext(c_out, m.nwords, m.nwords)};
    rc = run_vm(ctx, mod, itb->second, ex, 6, count);
  }
  return rc;
}

int sc_crt_combine(sc_ctx* ctx, int mod_p, int mod_full, int cst_k, int cst_negk, int cst_mq, const uint32_t* a_p, int a_p_words,
                   const uint32_t* a_q, int a_q_words, uint32_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  if (!valid_mod(ctx, mod_p) || !valid_mod(ctx, mod_full) || !a_p || !a_q || !out || a_p_words <= 0 || a_q_words <= 0)
    return fail(ctx, SC_ERR_ARG, "sc_crt_combine: bad argument");
  for (int c : {cst_k, cst_negk, cst_mq})
    if (c < 0 || c >= (int)ctx->consts.size()) return fail(ctx, SC_ERR_ARG, "sc_crt_combine: bad constant");
  if (ctx->consts[cst_k].mod != mod_p || ctx->consts[cst_negk].mod != mod_p || ctx->consts[cst_mq].mod != mod_full)
    return fail(ctx, SC_ERR_ARG, "sc_crt_combine: constant registered for another modulus");
  const Mod& mp = ctx->mods[mod_p];
  const Mod& mf = ctx->mods[mod_full];
  uint32_t* d_t;
  { int rc0 = tmp_buf(ctx, TMP_CRT, (size_t)count * mp.nwords * 4, (void**)&d_t); if (rc0) return rc0; }
  std::string k1 = "crt1:" + std::to_string(mod_p) + ":" + std::to_string(cst_k) + ":" + std::to_string(cst_negk) + ":" + std::to_string(a_p_words) + ":" + std::to_string(a_q_words);
  auto it1 = ctx->progs.find(k1);
  if (it1 == ctx->progs.end()) {
    Builder bd; const int ck = bd.use_const(cst_k), cn = bd.use_const(cst_negk);
    int kc = -1;
    if (a_p_words > mp.nwords || a_q_words > mp.nwords) { int cid; int rc = get_const_kred(ctx, mod_p, &cid); if (rc) return rc; kc = bd.use_const(cid); }
    if (a_p_words > mp.nwords) emit_load_reduced(ctx, mp, bd, 0, a_p_words, kc); else bd.loadw(0, 0, 0, a_p_words);
    bd.mul_const(ck); bd.stt(0);                                     // a_p * k
    if (a_q_words > mp.nwords) emit_load_reduced(ctx, mp, bd, 1, a_q_words, kc); else bd.loadw(1, 0, 0, a_q_words);
    bd.mul_const(cn); bd.addt(0);                                    // + a_q * (m_p - k)  = (a_p - a_q) * m_q^-1  (mod m_p)
    bd.storew(2); bd.end();
    Prog p; int rc = finalize_prog(ctx, mp, bd, &p); if (rc) return rc;
    it1 = ctx->progs.emplace(k1, p).first;
  }
  {
    VmExt ex[3] = {mk_ext(a_p, a_p_words, a_p_words), mk_ext(a_q, a_q_words, a_q_words), mk_ext(d_t, mp.nwords, mp.nwords)};
    int rc = run_vm(ctx, mod_p, it1->second, ex, 3, count); if (rc) return rc;
  }
  std::string k2 = "crt2:" + std::to_string(mod_full) + ":" + std::to_string(cst_mq) + ":" + std::to_string(mp.nwords) + ":" + std::to_string(a_q_words);
  auto it2 = ctx->progs.find(k2);
  if (it2 == ctx->progs.end()) {
    Builder bd; const int cm = bd.use_const(cst_mq);
    bd.loadw(0, 0, 0, mp.nwords); bd.mul_const(cm);                  // m_q * t   (< m_p m_q: exact)
    bd.addw(1, 0, 0, a_q_words);                                     // + a_q
    bd.storew(2); bd.end();
    Prog p; int rc = finalize_prog(ctx, mf, bd, &p); if (rc) return rc;
    it2 = ctx->progs.emplace(k2, p).first;
  }
  VmExt ex[3] = {mk_ext(d_t, mp.nwords, mp.nwords), mk_ext(a_q, a_q_words, a_q_words), mk_ext(out, mf.nwords, mf.nwords)};
  return run_vm(ctx, mod_full, it2->second, ex, 3, count);
}

// Limb-form constant pairs of a modulus m for the pair arithmetic: pair(R^2) embeds an integer (u,0) -> u; pair(B R) is
// the radix B = 2^(32 nwords) of the operand chunks.  Each pair (c0, c1) satisfies c0 + c1 m = value (mod m^2), c0, c1 < m.
static int get_pair_consts(sc_ctx* ctx, int mod, uint32_t** out) {
  auto it = ctx->pair_consts.find(mod);
  if (it != ctx->pair_consts.end()) { *out = it->second; return SC_OK; }
  const Mod& m = ctx->mods[mod];
  Big m2 = big_mul(m.n, m.n);
  Big one(m2.size(), 0); one[0] = 1;
  Big r2 = big_shl_mod(one, m2, 2 * m.W * m.S);                       // R^2 mod m^2
  Big br = big_shl_mod(one, m2, m.W * m.S + 32 * m.nwords);           // B R mod m^2
  std::vector<uint32_t> limbs;
  for (const Big* v : {&r2, &br}) {
    Big q, rem;
    big_divmod(*v, m.n, &q, &rem);
    q.resize(m.nwords);
    for (const Big* part : {&rem, &q}) { auto l = to_limbs(*part, m.S, m.W); limbs.insert(limbs.end(), l.begin(), l.end()); }
  }
  uint32_t* d = nullptr;
  int rc = upload(ctx, limbs.data(), limbs.size() * 4, (void**)&d);
  if (rc) return rc;
  ctx->pair_consts[mod] = d;
  *out = d;
  return SC_OK;
}

// the context the pair kernel runs in for modulus `mod`: itself when its configuration has a pair kernel, otherwise a twin
// context of the same modulus in a pair-capable configuration (created on first use); -1 if no configuration fits
static int pair_twin(sc_ctx* ctx, int mod) {
  {
    const Mod& m = ctx->mods[mod];
    if (pair_capable(m.G, m.L, m.W)) return mod;
  }
  auto it = ctx->pair_twins.find(mod);
  if (it != ctx->pair_twins.end()) return it->second;
  const Big n = ctx->mods[mod].n;
  int twin = -1;
  if (create_mod(ctx, n.data(), (int)n.size(), true, &twin) != SC_OK) twin = -1;
  ctx->pair_twins[mod] = twin;
  return twin;
}

// The small-batch pair twin of `mod` ((16,5) for a (4,18) modulus, (8,5) for a (2,18) one, (4,5) for a (1,18) one) when this batch should run on it, else
// -1.  Automatic policy: the (2G, 9) launch would still leave at least half of the SIMDs without a wave.
static int latency_pair_twin(sc_ctx* ctx, int mod, uint64_t count) {
  if (ctx->latency_mode == 0) return -1;
  const Config* cfg = nullptr;
  {
    const Mod& m = ctx->mods[mod];
    if (m.W != 29 || m.L != 18 || (m.G != 1 && m.G != 2 && m.G != 4)) return -1;
    cfg = (m.G == 4) ? &kLatencyPair16 : (m.G == 2 ? &kLatencyPair8 : &kLatencyPair4);
    const uint64_t per_wave = 64 / (2 * m.G), waves = (count + per_wave - 1) / per_wave;
    if (ctx->latency_mode == 1 && waves > (uint64_t)ctx->num_cu * 2 / (uint64_t)ctx->chip_share) return -1;
    if (m.nbits + 8 > cfg->W * cfg->G * cfg->L || 32 * m.nwords > cfg->W * cfg->G * cfg->L) return -1;
  }
  auto it = ctx->latency_pair_twins.find(mod);
  if (it != ctx->latency_pair_twins.end()) return it->second;
  const Big n = ctx->mods[mod].n;
  int twin = -1;
  if (create_mod(ctx, n.data(), (int)n.size(), false, &twin, cfg) != SC_OK) twin = -1;
  ctx->latency_pair_twins[mod] = twin;
  return twin;
}

// The twin of a (4,18) / (4,14) / (8,14) modulus n whose own modulus is the multiple M = c n, c = -n^-1 mod 2^29, so that M = -1
// (mod 2^29): in its context the Montgomery quotient digit needs no multiplication (Grp::NEG1).  M has at most 29 more bits than n;
// the (4,18) configuration holds 2088 bits, so 2048-bit moduli fit (R / M >= 2^11), and the L = 14 pair twins of 1536 / 3072-bit
// moduli (1624 / 3248 bits) have the room as well.  Residues modulo M (and pairs modulo M^2) reduce to
// residues modulo n (n^2): the caller finishes in a context of the original modulus.  -1: no such twin (other configurations,
// moduli too long, or n already = -1).
static int neg1_twin(sc_ctx* ctx, int mod) {
  auto it = ctx->neg1_twins.find(mod);
  if (it != ctx->neg1_twins.end()) return it->second;
  int twin = -1;
  {
    const Mod m = ctx->mods[mod];
    static const bool enabled = []{ const char* e = getenv("SC_NEG1"); return !(e && e[0] == '0'); }();      // A/B switch (dev): SC_NEG1=0
    const bool has_kernel = m.W == 29 && ((m.L == 18 && m.G == 4) || (m.L == 14 && (m.G == 4 || m.G == 8)));   // k_pvm<.., true> instances
    if (enabled && has_kernel && m.n0inv != 1 && m.nbits + 29 + 8 <= m.W * m.S) {
      Big c(1, m.n0inv);
      Big M = big_trimmed_words(big_mul(m.n, c));
      const Config same = {m.G, m.L, m.W, false};
      if (create_mod(ctx, M.data(), (int)M.size(), false, &twin, &same) != SC_OK) twin = -1;
      if (twin >= 0 && ctx->mods[twin].n0inv != 1) twin = -1;     // (cannot happen: M = -1 mod 2^29 by construction)
      if (twin >= 0) {           // what a single-modulus program needs to leave the context with a residue modulo n (sc_vm.h)
        uint32_t inv = m.n0inv;                                    // c is odd: c c = 1 (mod 8); Newton doubles the good bits
        for (int it = 0; it < 5; it++) inv *= 2u - m.n0inv * inv;
        ctx->mods[twin].small_c = m.n0inv;
        ctx->mods[twin].small_cinv = inv & ((1u << m.W) - 1);
      }
    }
  }
  ctx->neg1_twins[mod] = twin;
  return twin;
}

// The same twin for the single-modulus interpreter: only the (4,18) instance k_vm<4,18,29,true> exists, and only programs whose
// outputs are word stores can use it (modexp_var_impl: the blinding launch of step 4i) -- limb-form outputs would carry residues
// up to 2M into launches of the original context.
static int neg1_vm_twin(sc_ctx* ctx, int mod, uint64_t count) {
  const Mod& m = ctx->mods[mod];
  if (m.W != 29 || m.L != 18 || m.G != 4 || use_latency_config(ctx, m, count)) return -1;
  const int t = neg1_twin(ctx, mod);
  return (t >= 0 && ctx->mods[t].small_c != 0) ? t : -1;
}

// ---- the constants of the one-lane policy, measured per device ---------------------------------------------------------------
// Both forms run in rounds of their resident waves: 2 waves per SIMD, 64 numbers per one-lane wave and 32 per two-lane wave.  What a
// batch costs in either form is a sum of full rounds plus one partial round, and a partial round that leaves every SIMD at most one
// wave costs about half -- how much exactly depends on the part (issue rate of a lone wave, clocks), so it is measured, once per
// device and process, on a fixed 1024-bit odd modulus and a 1024-bit exponent -- the shape of the key holder's first CRT stage; a
// 256-bit exponent gave other RATIOS (two-lane round 0.53 instead of 0.60 of a one-lane round: the window tables weigh differently)
// and mis-ranked the forms at half a round.  About 80 ms of launches.  Round 3 had these as constants fitted on one box (0.6 / 0.27 /
// 0.38 of a one-lane round).
struct OneLaneCal { bool ok = false; double one_full = 1.0, one_half = 0.5, two_full = 0.6, two_half = 0.27, two_only_half = 0.38; int simds = 0; };
static std::mutex g_cal_mutex;
static std::map<int, OneLaneCal> g_cal;           // by device

static int modexp_shared_impl(sc_ctx* ctx, int mod, int exp, const uint32_t* x, int x_words, const uint32_t* mul_into,
                              uint32_t* out, uint8_t* flags, uint64_t count, uint64_t* any_flags, uint64_t inner, bool any_flags_clean);

static int onelane_calibrate(sc_ctx* ctx, OneLaneCal* out) {
  OneLaneCal c;
  c.simds = ctx->num_cu * 4;
  const int nw = 32;
  Big n(nw, 0xffffffffu); n[0] = 0xffffff61u; n[nw - 1] = 0xfffffff1u;          // any odd 1024-bit number serves Montgomery arithmetic
  Big e(32, 0xa5c3965au); e[31] = 0x96a5c35au;                                    // 1024 bits, half of them set
  int mod = -1, exp = -1;
  int rc = create_mod(ctx, n.data(), nw, false, &mod); if (rc) return rc;
  rc = sc_exp_create(ctx, e.data(), (int)e.size(), &exp); if (rc) return rc;
  const uint64_t n1 = (uint64_t)c.simds * 2 * 64, n2 = n1 / 2;                    // numbers of a full one-lane / two-lane round
  uint32_t *x = nullptr, *y = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  struct Cleanup {       // every exit path frees the operands and the events
    uint32_t **x, **y; hipEvent_t *e0, *e1;
    ~Cleanup() { if (*x) (void)hipFree(*x); if (*y) (void)hipFree(*y); if (*e0) (void)hipEventDestroy(*e0); if (*e1) (void)hipEventDestroy(*e1); }
  } cleanup{&x, &y, &e0, &e1};
  HIPCHK(ctx, hipMalloc((void**)&x, n1 * nw * 4));
  HIPCHK(ctx, hipMalloc((void**)&y, n1 * nw * 4));
  HIPCHK(ctx, hipMemsetAsync(x, 0x5a, n1 * nw * 4, ctx->stream));
  HIPCHK(ctx, hipEventCreate(&e0)); HIPCHK(ctx, hipEventCreate(&e1));
  const int saved_lat = ctx->latency_mode, saved_one = ctx->onelane_mode, saved_share = ctx->chip_share;
  const double saved_macs = ctx->mac_counter;
  ctx->latency_mode = 0; ctx->chip_share = 1;
  auto timed = [&](int form, uint64_t count, double* ms) -> int {
    ctx->onelane_mode = form;                       // 0: the modulus's own two-lane configuration, 2: the one-lane twin
    double best = 1e30;
    for (int rep = 0; rep < 2; rep++) {
      HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
      int r = modexp_shared_impl(ctx, mod, exp, x, nw, nullptr, y, nullptr, count, nullptr, 0, false); if (r) return r;
      HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
      HIPCHK(ctx, hipEventSynchronize(e1));
      float t = 0; HIPCHK(ctx, hipEventElapsedTime(&t, e0, e1));
      if (t < best) best = t;
    }
    *ms = best;
    return SC_OK;
  };
  double two_15 = 0, warm = 0;
  rc = timed(2, 64, &warm);                         // program build, code load
  if (!rc) rc = timed(0, 64, &warm);
  if (!rc) rc = timed(2, n1, &c.one_full);
  if (!rc) rc = timed(2, n1 / 2, &c.one_half);
  if (!rc) rc = timed(0, n2, &c.two_full);
  if (!rc) rc = timed(0, n2 / 2, &c.two_only_half);
  if (!rc) rc = timed(0, n2 + n2 / 2, &two_15);
  ctx->latency_mode = saved_lat; ctx->onelane_mode = saved_one; ctx->chip_share = saved_share; ctx->mac_counter = saved_macs;
  if (rc) return rc;
  c.two_half = std::max(0.0, two_15 - c.two_full);
  // A calibration taken while something else used the chip (another context's launches, another process) gives ratios no idle
  // chip produces; the policy then keeps the built-in constants rather than rank the forms by a skewed measurement (results are
  // the same residues either way: only throughput depends on it).  Plausible: a half round costs between a third of and a whole
  // full round; a two-lane round -- half the numbers -- between 0.3 and 0.9 of a one-lane round.
  const double r_half = c.one_half / c.one_full, r_two = c.two_full / c.one_full, r_only = c.two_only_half / c.two_full;
  c.ok = c.one_full > 0 && c.two_full > 0 && r_half > 0.33 && r_half <= 1.0 && r_two > 0.3 && r_two < 0.9 && r_only > 0.3 && r_only <= 1.05 &&
         c.two_half <= c.two_full * 1.05;
  if (!c.ok) { OneLaneCal d; d.simds = c.simds; const double scale = c.one_full > 0 ? c.one_full : 1.0;
               d.one_full *= scale; d.one_half *= scale; d.two_full *= scale; d.two_half *= scale; d.two_only_half *= scale; c = d; }
  *out = c;
  return SC_OK;
}

// the calibration of this context's device (measured by the first context that asks; a failed measurement leaves round 3's constants)
static OneLaneCal onelane_cal(sc_ctx* ctx) {
  std::lock_guard<std::mutex> lock(g_cal_mutex);
  auto it = g_cal.find(ctx->device);
  if (it != g_cal.end()) return it->second;
  OneLaneCal c;
  if (onelane_calibrate(ctx, &c) != SC_OK) { c = OneLaneCal(); c.simds = ctx->num_cu * 4; }
  g_cal[ctx->device] = c;
  return c;
}

int sc_ctx_policy(sc_ctx* ctx, double* out6) {
  if (!ctx || !out6) return SC_ERR_ARG;
  const OneLaneCal c = onelane_cal(ctx);
  out6[0] = c.one_full; out6[1] = c.one_half; out6[2] = c.two_full; out6[3] = c.two_half; out6[4] = c.two_only_half; out6[5] = (double)c.simds;
  return c.ok ? SC_OK : fail(ctx, SC_ERR_HIP, "sc_ctx_policy: the calibration launches failed (the built-in constants are in use)");
}

// The one-lane twin of `mod` when this batch should run on it, else `mod` itself.  Automatic policy: the modulus fits the one-lane
// configuration and the model above -- full rounds plus one partial round per form, with this device's measured round times -- says
// the one-lane form is faster.  Shape of the trade (MI355X, 1024-bit modulus and exponent, profiles/r03_onelane_policy_sweep.txt):
// 196608 numbers 19.8 -> 16.6 ms, 2.1 M numbers (zero tests) 37.3 -> 33.0 ms, but 98304 numbers 9.5 -> 11.6 ms -- three quarters of a
// round leaves a quarter of the SIMDs with one wave and nobody to hide its latencies, where the two-lane form still runs 1.5 rounds.
// When several contexts work on the GPU at once (concurrent shards: sc_ctx_set_chip_share) a launch owns its share of the chip
// only, and the idle SIMDs of an under-filled launch are taken by the other contexts' kernels: rounds are counted on that share
// (two shards of 32768 comparisons: the 98304-number launches run one-lane, +1.5 % on the whole step).
static int onelane_for(sc_ctx* ctx, int mod, uint64_t count) {
  if (ctx->onelane_mode == 0) return mod;
  {
    const Mod& m = ctx->mods[mod];
    // the residue arrays (and the raw chunks a wide operand is read in) must fit below R = 2^(28 * 37): at most 32 words
    if (m.W == kOneLane.W || m.nbits + 8 > kOneLane.W * kOneLane.G * kOneLane.L || 32 * m.nwords > kOneLane.W * kOneLane.G * kOneLane.L) return mod;
  }
  if (ctx->onelane_mode == 1) {
    const OneLaneCal c = onelane_cal(ctx);          // (may register moduli: no reference into ctx->mods is held across it)
    const double share = (double)ctx->chip_share / ((double)ctx->num_cu * 4 * 2 * 64);
    const double r1 = (double)count * share, r2 = 2.0 * r1;
    const double f1 = r1 - std::floor(r1), f2 = r2 - std::floor(r2);
    const double one = c.one_full * std::floor(r1) + (f1 <= 0.0 ? 0.0 : (f1 <= 0.5 ? c.one_half : c.one_full));
    const double two = c.two_full * std::floor(r2) + (f2 <= 0.0 ? 0.0 : (f2 <= 0.5 ? (r2 < 1.0 ? c.two_only_half : c.two_half) : c.two_full));
    if (one >= two) return mod;
  }
  auto it = ctx->onelane_twins.find(mod);
  if (it != ctx->onelane_twins.end()) return it->second < 0 ? mod : it->second;
  const Big n = ctx->mods[mod].n;
  int twin = -1;
  if (create_mod(ctx, n.data(), (int)n.size(), false, &twin, &kOneLane) != SC_OK) twin = -1;
  ctx->onelane_twins[mod] = twin;
  return twin < 0 ? mod : twin;
}

// ---- measured op times of the pair kernel instances (the segment policy of sc_modexp_shared_sq) -------------------------------------
// One pair squaring and one pair product of a resident wave of the instance a launch takes, in milliseconds, with the chip as full as the
// launch itself makes it (up to one round of resident waves: two waves share a SIMD's issue slots, a lone wave runs its program almost
// twice as fast): timed as the difference of two short programs (16 / 48 squarings; 8 / 24 products) on the caller's own operands, best
// of three (other contexts may be using the chip), once per process, device and instance.  A few milliseconds.
struct PairOpTimes { double sqr_ms = 0, mul_ms = 0; };
static std::mutex g_pair_cal_mutex;
static std::map<std::tuple<int, int, int, bool>, PairOpTimes> g_pair_cal;     // (device, G, L, neg1)

static int pair_op_times(sc_ctx* ctx, int mod_m, uint64_t count, const uint32_t* x, int x_words, uint32_t* d_w, uint32_t* d_w1, int wm,
                         double* sqr_ms, double* mul_ms) {
  int G, L; bool neg1;
  pvm_instance(ctx, ctx->mods[mod_m], count, &G, &L, &neg1);
  const auto key = std::make_tuple(ctx->device, G, L, neg1);
  std::lock_guard<std::mutex> lock(g_pair_cal_mutex);
  auto it = g_pair_cal.find(key);
  if (it != g_pair_cal.end()) { *sqr_ms = it->second.sqr_ms; *mul_ms = it->second.mul_ms; return SC_OK; }
  const Mod m = ctx->mods[mod_m];
  const uint32_t nw = (uint32_t)std::min(x_words, m.nwords);
  auto build = [&](int nsq, int nmul, Prog* out) -> int {
    std::vector<VmOp> ops;
    ops.push_back(VmOp{PV_LOADU, 0, 0, nw});
    ops.push_back(VmOp{PV_MULC, 2, 0, 0});
    ops.push_back(VmOp{PV_STT, 1, 0, 0});
    for (int i = 0; i < nsq; i++) ops.push_back(VmOp{PV_SQR, 0, 0, 0});
    for (int i = 0; i < nmul; i++) ops.push_back(VmOp{PV_MULT, 1, 0, 0});
    ops.push_back(VmOp{PV_OUT, 1, 2, 0});
    ops.push_back(VmOp{PV_END, 0, 0, 0});
    Prog p;
    p.nops = (uint32_t)ops.size(); p.nscratch = 4 + (m.G == 1 ? 1 : 0); p.nconst = 4;
    int rc = upload(ctx, ops.data(), ops.size() * sizeof(VmOp), (void**)&p.d_ops); if (rc) return rc;
    rc = get_pair_consts(ctx, mod_m, &p.d_consts); if (rc) return rc;
    *out = p;
    return SC_OK;
  };
  Prog progs[4];
  const int shapes[4][2] = {{16, 0}, {48, 0}, {0, 8}, {0, 24}};
  for (int i = 0; i < 4; i++) { int rc = build(shapes[i][0], shapes[i][1], &progs[i]); if (rc) return rc; }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return fail(ctx, SC_ERR_HIP, "pair_op_times: no event"); }
  const int saved_lat = ctx->latency_mode;
  const double saved_macs = ctx->mac_counter;
  // the wave must run on the SAME instance as the launch it stands for: the batch-size policy is told the real batch size by keeping
  // the latency mode out of it (a tiny batch would otherwise pick the small-batch twin)
  ctx->latency_mode = use_latency_config(ctx, m, count) ? 2 : 0;
  const int occ = pvm_occupancy(ctx, G, L, neg1);
  const uint64_t one_wave = std::min<uint64_t>(count, (uint64_t)(64 / G) * (uint64_t)ctx->num_cu * (uint64_t)std::max(1, occ));   // (at most one round)
  VmExt ex3[3] = {mk_ext(x, x_words, x_words), mk_ext(d_w, wm, wm), mk_ext(d_w1, wm, wm)};
  double ms[4] = {1e30, 1e30, 1e30, 1e30};
  int rc = SC_OK;
  for (int rep = 0; rep < 4 && !rc; rep++)          // the first pass loads the code
    for (int i = 0; i < 4 && !rc; i++) {
      if (hipEventRecord(e0, ctx->stream) != hipSuccess) { rc = fail(ctx, SC_ERR_HIP, "pair_op_times: event"); break; }
      rc = run_pvm(ctx, mod_m, progs[i], ex3, 3, one_wave);
      if (!rc && (hipEventRecord(e1, ctx->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = fail(ctx, SC_ERR_HIP, "pair_op_times: launch failed");
      float t = 0;
      if (!rc && hipEventElapsedTime(&t, e0, e1) == hipSuccess && rep > 0 && t < ms[i]) ms[i] = t;
    }
  ctx->latency_mode = saved_lat; ctx->mac_counter = saved_macs;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc) return rc;
  PairOpTimes t;
  t.sqr_ms = std::max(1e-6, (ms[1] - ms[0]) / 32.0);
  t.mul_ms = std::max(1e-6, (ms[3] - ms[2]) / 16.0);
  g_pair_cal[key] = t;
  ctx->stat_pair_calibrations++;
  *sqr_ms = t.sqr_ms; *mul_ms = t.mul_ms;
  return SC_OK;
}

int sc_ctx_set_pair_policy(sc_ctx* ctx, double hold_ms, double max_rounds) {
  if (!ctx || hold_ms < 0 || !(max_rounds > 0)) return SC_ERR_ARG;
  ctx->pair_hold_ms = hold_ms;
  ctx->pair_max_rounds = max_rounds;
  return SC_OK;
}

int sc_ctx_stats(sc_ctx* ctx, uint64_t* out, int n) {
  if (!ctx || !out || n < 0) return SC_ERR_ARG;
  const uint64_t v[3] = {ctx->stat_segmented_launches, ctx->stat_segments, ctx->stat_pair_calibrations};
  for (int i = 0; i < n; i++) out[i] = i < 3 ? v[i] : 0;
  return SC_OK;
}

int sc_ctx_set_chip_share(sc_ctx* ctx, int contexts) {
  if (!ctx || contexts < 1 || contexts > 64) return SC_ERR_ARG;
  ctx->chip_share = contexts;
  return SC_OK;
}

int sc_ctx_set_fork_mode(sc_ctx* ctx, int mode) {
  if (!ctx || mode < 0 || mode > 2) return SC_ERR_ARG;
  ctx->fork_mode = mode;
  return SC_OK;
}

int sc_ctx_set_onelane_mode(sc_ctx* ctx, int mode) {
  if (!ctx || mode < 0 || mode > 2) return SC_ERR_ARG;
  ctx->onelane_mode = mode;
  return SC_OK;
}

int sc_mod_supports_sq(sc_ctx* ctx, int mod) {
  if (!valid_mod(ctx, mod)) return SC_ERR_ARG;
  return pair_twin(ctx, mod) >= 0 ? 1 : 0;
}

int sc_modexp_shared_sq(sc_ctx* ctx, int mod_m, int mod_m2, int exp, const uint32_t* x, int x_words, const uint32_t* mul_into,
                        uint32_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  if (!valid_mod(ctx, mod_m) || !valid_mod(ctx, mod_m2) || exp < 0 || exp >= (int)ctx->exps.size() || !x || !out || x_words <= 0)
    return fail(ctx, SC_ERR_ARG, "sc_modexp_shared_sq: bad argument");
  {
    const int lat = latency_pair_twin(ctx, mod_m, count);
    mod_m = lat >= 0 ? lat : pair_twin(ctx, mod_m);
  }
  if (mod_m < 0) return fail(ctx, SC_ERR_UNSUPPORTED, "sc_modexp_shared_sq: no pair configuration fits this modulus");
  {
    const Mod& m0 = ctx->mods[mod_m];
    const Mod& m2 = ctx->mods[mod_m2];
    Big sq = big_mul(m0.n, m0.n);
    sq.resize(std::max(sq.size(), m2.n.size()), 0);
    Big other = m2.n; other.resize(sq.size(), 0);
    if (big_cmp(sq, other) != 0) return fail(ctx, SC_ERR_ARG, "sc_modexp_shared_sq: mod_m2 is not the square of mod_m");
    if (x_words > 4 * m0.nwords) return fail(ctx, SC_ERR_ARG, "sc_modexp_shared_sq: operand wider than 4 chunks");
  }
  // chip-filling batches of a (4,18) / (4,14) / (8,14) modulus: the pair launch runs modulo the multiple M = c m = -1 (mod 2^29),
  // which needs no quotient multiply; the assembly launch below reduces w0 + w1 M modulo m^2 (m | M)
  if (!use_latency_config(ctx, ctx->mods[mod_m], count)) { const int t = neg1_twin(ctx, mod_m); if (t >= 0) mod_m = t; }
  const Mod& m = ctx->mods[mod_m];
  const Mod& m2 = ctx->mods[mod_m2];
  const Exp& ex = ctx->exps[exp];
  const int wm = m.nwords + 1;                                           // words of the raw pair halves (< 2m + 1)
  uint32_t* d_w;
  { int rc0 = tmp_buf(ctx, TMP_PAIR, (size_t)count * wm * 4 * 2, (void**)&d_w); if (rc0) return rc0; }
  uint32_t* d_w1 = d_w + (size_t)count * wm;
  // ---- launch 1: pair exponentiation in the m context
  std::string k1 = "psq:" + std::to_string(mod_m) + ":" + std::to_string(exp) + ":" + std::to_string(x_words);
  auto it1 = ctx->progs.find(k1);
  if (it1 == ctx->progs.end()) {
    std::vector<VmOp> ops;
    uint32_t nsc = 2;
    double macs = 0;
    // multiply-adds: a pair squaring is (a*a part) + 3 S^2; a pair product 5 S^2 with the two-row pass, 6 S^2 in the one-lane
    // form (three single passes over one staging area)
    const double S2 = (double)m.S * m.S, SQ = S2 + (double)m.G * m.G * m.L * (m.L + 1) / 2.0 + 2.0 * S2, MU = (m.G == 1 ? 6.0 : 5.0) * S2;
    auto emit = [&](uint32_t opc, uint32_t w1 = 0, uint32_t w2 = 0, uint32_t w3 = 0) { ops.push_back(VmOp{opc, w1, w2, w3}); };
    auto touch = [&](uint32_t e) { nsc = std::max(nsc, 2 * e + 2); };
    // embed the operand: Horner over chunks of nwords words; constants: LDS 2,3 = pair(R^2), 4,5 = pair(B R)
    const int nch = (x_words + m.nwords - 1) / m.nwords;
    const uint32_t TMP_E = 0;                                            // scratch pair entry used while embedding
    for (int t = nch - 1; t >= 0; t--) {
      const int nw = std::min(m.nwords, x_words - t * m.nwords);
      if (t != nch - 1) { emit(PV_MULC, 4); macs += MU; emit(PV_STT, TMP_E); touch(TMP_E); }
      emit(PV_LOADU, 0, 0, ((uint32_t)(t * m.nwords) << 16) | (uint32_t)nw);
      emit(PV_MULC, 2); macs += MU;
      if (t != nch - 1) emit(PV_ADDT, TMP_E);
    }
    // sliding-window exponentiation on pairs (same schedule as emit_pow_shared)
    const int bits = ex.bits;
    if (bits == 0) return fail(ctx, SC_ERR_ARG, "sc_modexp_shared_sq: zero exponent");
    const int w = best_window(bits), NT = 1 << (w - 1);
    const uint32_t T0 = 1;                                               // table entries T0 .. T0+NT-1, x^2 at T0+NT
    emit(PV_STT, T0); touch(T0);
    if (NT > 1) {
      emit(PV_SQR); macs += SQ; emit(PV_STT, T0 + NT); touch(T0 + NT);
      for (int k = 1; k < NT; k++) { emit(PV_LOADT, T0 + k - 1); emit(PV_MULT, T0 + NT); macs += MU; emit(PV_STT, T0 + k); touch(T0 + k); }
    }
    bool first = true;
    int i = bits - 1;
    while (i >= 0) {
      if (!ebit(ex.e, i)) { emit(PV_SQR); macs += SQ; i--; continue; }
      int jj = std::max(0, i - w + 1);
      while (!ebit(ex.e, jj)) jj++;
      int v = 0;
      for (int k = i; k >= jj; k--) v = (v << 1) | ebit(ex.e, k);
      if (first) { emit(PV_LOADT, T0 + (v - 1) / 2); first = false; }
      else { for (int k = 0; k < i - jj + 1; k++) { emit(PV_SQR); macs += SQ; } emit(PV_MULT, T0 + (v - 1) / 2); macs += MU; }
      i = jj - 1;
    }
    emit(PV_OUT, 1, 2); macs += 2.0 * S2;
    emit(PV_END);
    Prog p;
    if (m.G == 1) nsc += 1;   // one-lane pair products park an intermediate in a spare row (the last one) of the slot's table
    p.nops = (uint32_t)ops.size(); p.nscratch = nsc; p.nconst = 4; p.muls_per_item = macs;
    for (const VmOp& o : ops) {
      const uint32_t oc = o.w0 & 0xff;
      if (oc == PV_SQR) p.pair_sqrs++; else if (oc == PV_MULT || oc == PV_MULC) p.pair_muls++;
    }
    p.host_ops = std::make_shared<std::vector<VmOp>>(ops);
    int rc = upload(ctx, ops.data(), ops.size() * sizeof(VmOp), (void**)&p.d_ops); if (rc) return rc;
    rc = get_pair_consts(ctx, mod_m, &p.d_consts); if (rc) return rc;
    it1 = ctx->progs.emplace(k1, p).first;
  }
  {
    VmExt ex3[3] = {mk_ext(x, x_words, x_words), mk_ext(d_w, wm, wm), mk_ext(d_w1, wm, wm)};
    // A context that shares the chip (concurrent shards, sc_ctx_set_chip_share) runs a SINGLE-ROUND pair launch -- every resident
    // wave holds its one group of items for the whole exponentiation: Alice's rho^N for a shard of 32768 is 2048 waves for 54 ms --
    // in SEGMENTS: the same micro-program cut at window boundaries into a few launches, the pair parked in a row of the slot's
    // table in between (slot = item while there is one round).  Waves of a launch retire together, so nothing another context
    // queues behind such a launch gets a wave slot before it ends: the other shard's short, latency-bound launches (the inversion
    // sweeps of steps 1 and 6 / 7, the assembly launch after ITS pair launch) were seen waiting 26 .. 32 ms each.  With segments
    // they wait for a quarter of that.  Multi-round launches need none of this (their waves retire a round apart).
    // Whether, and into how many segments: from MEASUREMENTS, not from the shape of the launch.  A resident wave holds its slot for
    // hold = (pair squarings x t_sqr + pair products x t_mul) of this program, with the two op times of this kernel instance measured on
    // the device (pair_op_times, once per process and instance: a lone wave, like each of two waves sharing a SIMD, issues one
    // multiply-add per ~9.5 cycles, so its time through the program is the time a round of resident waves takes).  The launch is cut when
    // it would take more than half of the chip's wave slots for that long on a SHARED chip (sc_ctx_set_chip_share), into
    // round(hold / pair_hold_ms) segments; launches of more than pair_max_rounds rounds stay whole (their waves retire a round apart
    // anyway, and every segment boundary costs a drain of the chip).  Both knobs: sc_ctx_set_pair_policy.
    int iG, iL; bool ineg1;
    pvm_instance(ctx, m, count, &iG, &iL, &ineg1);
    const int occ = pvm_occupancy(ctx, iG, iL, ineg1);
    const uint64_t wave_items = (count + (uint64_t)(64 / iG) - 1) / (uint64_t)(64 / iG);
    const uint64_t resident = (uint64_t)ctx->num_cu * (uint64_t)std::max(1, occ);
    const double rounds = (double)wave_items / (double)resident;
    int K = 1;
    if (ctx->pair_hold_ms > 0 && ctx->chip_share > 1 && !ctx->stamps && occ > 0 && wave_items * 2 > resident && rounds <= ctx->pair_max_rounds) {
      double t_sqr = 0, t_mul = 0;
      int rcc = pair_op_times(ctx, mod_m, count, x, x_words, d_w, d_w1, wm, &t_sqr, &t_mul); if (rcc) return rcc;
      const double hold_ms = it1->second.pair_sqrs * t_sqr + it1->second.pair_muls * t_mul;
      K = (int)std::min(16.0, std::max(1.0, std::floor(hold_ms / ctx->pair_hold_ms + 0.5)));
    }
    const bool segmented = K > 1;
    if (!segmented) {
      int rc = run_pvm(ctx, mod_m, it1->second, ex3, 3, count); if (rc) return rc;
    } else {
      std::string ks = k1 + ":seg" + std::to_string(K);
      auto its = ctx->seg_progs.find(ks);
      if (its == ctx->seg_progs.end()) {
        const Prog& full = it1->second;
        if (!full.host_ops) return fail(ctx, SC_ERR_HIP, "sc_modexp_shared_sq: pair program without its host copy (internal error)");
        const std::vector<VmOp>& ops = *full.host_ops;
        // cut after a PV_MULT (the end of a window) nearest to each k / K of the op list past the table build
        std::vector<size_t> mults;
        size_t body = 0;
        for (size_t i = 0; i < ops.size(); i++) if ((ops[i].w0 & 0xff) == PV_MULT) mults.push_back(i);
        for (size_t i = 0; i < ops.size(); i++) if ((ops[i].w0 & 0xff) == PV_SQR && i > 0 && (ops[i - 1].w0 & 0xff) == PV_LOADT) body = i;   // first squaring run after the first window's load
        std::vector<size_t> cuts;      // index of the first op of segments 1 .. K-1
        for (int k = 1; k < K; k++) {
          const size_t target = body + (ops.size() - body) * (size_t)k / (size_t)K;
          size_t best = 0;
          for (size_t mi : mults) if (mi + 1 > body && (best == 0 || (mi + 1 > target ? mi + 1 - target : target - mi - 1) < (best > target ? best - target : target - best))) best = mi + 1;
          if (best > (cuts.empty() ? body : cuts.back()) && best + 1 < ops.size()) cuts.push_back(best);
        }
        const uint32_t E = full.nscratch / 2;            // one more pair row of the slot's table: the parked state
        std::vector<Prog> segs;
        size_t from = 0;
        for (size_t c = 0; c <= cuts.size(); c++) {
          const size_t to = c < cuts.size() ? cuts[c] : ops.size();     // the last segment ends with the program's own PV_OUT, PV_END
          std::vector<VmOp> so;
          if (c > 0) so.push_back(VmOp{PV_LOADT, E, 0, 0});
          so.insert(so.end(), ops.begin() + from, ops.begin() + to);
          if (c < cuts.size()) { so.push_back(VmOp{PV_STT, E, 0, 0}); so.push_back(VmOp{PV_END, 0, 0, 0}); }
          Prog sp = full;
          sp.host_ops.reset();
          sp.nops = (uint32_t)so.size(); sp.nscratch = full.nscratch + 2;
          sp.muls_per_item = full.muls_per_item * (double)(to - from) / (double)ops.size();
          int rc = upload(ctx, so.data(), so.size() * sizeof(VmOp), (void**)&sp.d_ops); if (rc) return rc;
          segs.push_back(sp);
          from = to;
        }
        its = ctx->seg_progs.emplace(ks, segs).first;
      }
      ctx->slot_per_item = true;        // a table slot per item: what a segment parks is still there for the next one, whatever the rounds
      ctx->stat_segmented_launches++; ctx->stat_segments += its->second.size();
      int rcs = SC_OK;
      for (const Prog& sp : its->second) { rcs = run_pvm(ctx, mod_m, sp, ex3, 3, count); if (rcs) break; }
      ctx->slot_per_item = false;
      if (rcs) return rcs;
    }
  }
  // ---- launch 2 (m^2 context): out = (w0 + w1 m) [* mul_into] mod m^2
  int cst_m;
  { int rc = sc_const_create_cached(ctx, mod_m2, m.n, &cst_m); if (rc) return rc; }
  std::string k2 = "psq2:" + std::to_string(mod_m2) + ":" + std::to_string(cst_m) + ":" + std::to_string(wm) + ":" + std::to_string(mul_into ? 1 : 0);
  auto it2 = ctx->progs.find(k2);
  if (it2 == ctx->progs.end()) {
    Builder bd; const int cm = bd.use_const(cst_m);
    bd.loadw(1, 0, 0, wm); bd.mul_const(cm);           // w1 * m  (mod m^2)
    bd.addw(0, 0, 0, wm);                              // + w0
    if (mul_into) { bd.mul_const(0); bd.mul_extw(2); } // to Montgomery form, times the ciphertext
    bd.storew(3); bd.end();
    Prog p; int rc = finalize_prog(ctx, m2, bd, &p); if (rc) return rc;
    it2 = ctx->progs.emplace(k2, p).first;
  }
  VmExt ex4[4] = {mk_ext(d_w, wm, wm), mk_ext(d_w1, wm, wm), mk_ext(mul_into, m2.nwords, m2.nwords), mk_ext(out, m2.nwords, m2.nwords)};
  return run_vm(ctx, mod_m2, it2->second, ex4, 4, count);
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU: comparisons are independent, so the only exchange is the reassembly of per-rank result blocks -- one
// RCCL all-gather over xGMI on the context's stream.  RCCL is bound at run time (dlopen): a single-GPU user never loads it, and
// a process that already runs torch.distributed gets the very library instance torch loaded.
// ------------------------------------------------------------------------------------------------
namespace {
struct RcclId { char internal[128]; };
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
struct RcclLoad { RcclApi api; std::string err; };
void rccl_load(RcclLoad* out) {
  RcclLoad& r = *out;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
    r.api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.api.handle) break;
  }
  if (!r.api.handle) {
    const char* e = dlerror();
    r.err = std::string("cannot load librccl: ") + (e ? e : "not found");
    return;
  }
  r.api.GetUniqueId = (int (*)(RcclId*))dlsym(r.api.handle, "ncclGetUniqueId");
  r.api.CommInitRank = (int (*)(void**, int, RcclId, int))dlsym(r.api.handle, "ncclCommInitRank");
  r.api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(r.api.handle, "ncclAllGather");
  r.api.CommDestroy = (int (*)(void*))dlsym(r.api.handle, "ncclCommDestroy");
  r.api.GetErrorString = (const char* (*)(int))dlsym(r.api.handle, "ncclGetErrorString");
  if (!r.api.GetUniqueId || !r.api.CommInitRank || !r.api.AllGather || !r.api.CommDestroy) { r.err = "librccl lacks an expected symbol"; r.api.handle = nullptr; }
}
RcclApi* rccl_api(std::string* why) {
  static RcclLoad loaded = [] { RcclLoad r; rccl_load(&r); return r; }();   // once per process; initialisation of a local static is thread-safe
  if (!loaded.api.handle) { if (why) *why = loaded.err; return nullptr; }
  return &loaded.api;
}
int rccl_fail(sc_ctx* ctx, RcclApi* api, const char* what, int code) {
  return fail(ctx, SC_ERR_HIP, "%s: %s", what, api->GetErrorString ? api->GetErrorString(code) : "RCCL error");
}
}  // namespace

int sc_comm_unique_id(sc_ctx* ctx, void* id_hptr) {
  if (!ctx || !id_hptr) return fail(ctx, SC_ERR_ARG, "sc_comm_unique_id: bad argument");
  std::string why;
  RcclApi* api = rccl_api(&why);
  if (!api) return fail(ctx, SC_ERR_UNSUPPORTED, "%s", why.c_str());
  RcclId id;
  const int rc = api->GetUniqueId(&id);
  if (rc) return rccl_fail(ctx, api, "ncclGetUniqueId", rc);
  memcpy(id_hptr, id.internal, sizeof id.internal);
  return SC_OK;
}

int sc_comm_init(sc_ctx* ctx, const void* id_hptr, int rank, int nranks) {
  if (!ctx || !id_hptr || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, SC_ERR_ARG, "sc_comm_init: bad argument");
  if (ctx->comm) return fail(ctx, SC_ERR_ARG, "sc_comm_init: the context already has a communicator");
  std::string why;
  RcclApi* api = rccl_api(&why);
  if (!api) return fail(ctx, SC_ERR_UNSUPPORTED, "%s", why.c_str());
  HIPCHK(ctx, hipSetDevice(ctx->device));
  RcclId id;
  memcpy(id.internal, id_hptr, sizeof id.internal);
  void* comm = nullptr;
  const int rc = api->CommInitRank(&comm, nranks, id, rank);
  if (rc) return rccl_fail(ctx, api, "ncclCommInitRank", rc);
  ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_nranks = nranks;
  return SC_OK;
}

int sc_allgather(sc_ctx* ctx, const uint32_t* send, uint32_t* recv, uint64_t words_per_rank) {
  if (!ctx || !ctx->comm) return fail(ctx, SC_ERR_ARG, "sc_allgather: no communicator (sc_comm_init)");
  if (words_per_rank == 0) return SC_OK;
  if (!send || !recv) return fail(ctx, SC_ERR_ARG, "sc_allgather: bad argument");
  RcclApi* api = rccl_api(nullptr);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int rc = api->AllGather(send, recv, (size_t)words_per_rank, 3 /* ncclUint32 */, ctx->comm, ctx->stream);
  if (rc) return rccl_fail(ctx, api, "ncclAllGather", rc);
  return SC_OK;
}

int sc_comm_destroy(sc_ctx* ctx) {
  if (!ctx) return SC_ERR_ARG;
  if (!ctx->comm) return SC_OK;
  RcclApi* api = rccl_api(nullptr);
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  const int rc = api ? api->CommDestroy(ctx->comm) : 0;
  ctx->comm = nullptr; ctx->comm_nranks = 0;
  return rc ? rccl_fail(ctx, api, "ncclCommDestroy", rc) : SC_OK;
}

// ------------------------------------------------------------------------------------------------
// Device-side CSPRNG (sc_rng.h): the random draws of a batch are generated where they are consumed.
// ------------------------------------------------------------------------------------------------
int sc_rng_seed(sc_ctx* ctx, const uint8_t* key32_hptr) {
  if (!ctx) return SC_ERR_ARG;
  uint8_t buf[32];
  if (key32_hptr) {
    memcpy(buf, key32_hptr, 32);
  } else {
    size_t got = 0;
    while (got < 32) {
      const ssize_t r = getrandom(buf + got, 32 - got, 0);
      if (r <= 0) return fail(ctx, SC_ERR_HIP, "sc_rng_seed: the operating system's random source failed");
      got += (size_t)r;
    }
  }
  for (int i = 0; i < 8; i++)
    ctx->rng_key.k[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) | ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
  memset(buf, 0, sizeof buf);
  ctx->rng_seeded = true;
  ctx->rng_call = 0;
  return SC_OK;
}

// every generator call: seeded context (from the OS on first use), at most 2^32 items (the item index is one nonce word), and a
// call number of its own
static int rng_begin(sc_ctx* ctx, uint64_t count, const void* out, const char* who, uint64_t* call) {
  if (!ctx) return SC_ERR_ARG;
  if (!out) return fail(ctx, SC_ERR_ARG, "%s: no output array", who);
  if (count > 0xffffffffull) return fail(ctx, SC_ERR_ARG, "%s: at most 2^32 - 1 items per call", who);
  {
    std::lock_guard<std::mutex> lock(ctx->rng_seed_mutex);
    if (!ctx->rng_seeded) { int rc = sc_rng_seed(ctx, nullptr); if (rc) return rc; }
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  *call = ctx->rng_call.fetch_add(1);
  return SC_OK;
}

int sc_rng_bits(sc_ctx* ctx, int bits, uint32_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  if (ctx && bits <= 0) return fail(ctx, SC_ERR_ARG, "sc_rng_bits: bits must be positive");
  uint64_t call;
  int rc = rng_begin(ctx, count, out, "sc_rng_bits", &call); if (rc) return rc;
  if (launch_rng_bits(ctx->stream, ctx->rng_key, call, bits, (bits + 31) / 32, out, count)) return fail(ctx, SC_ERR_HIP, "sc_rng_bits: launch failed");
  return SC_OK;
}

int sc_rng_below(sc_ctx* ctx, const uint32_t* n_hptr, int nwords, int nonzero, uint32_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  if (ctx && (!n_hptr || nwords <= 0)) return fail(ctx, SC_ERR_ARG, "sc_rng_below: bad bound");
  if (!ctx) return SC_ERR_ARG;
  Big n(n_hptr, n_hptr + nwords);
  const int nbits = big_bits(n);
  if (nbits == 0 || (nonzero && nbits == 1)) return fail(ctx, SC_ERR_ARG, "sc_rng_below: empty range");
  const int nw = (nbits + 31) / 32;
  if (nw != nwords) return fail(ctx, SC_ERR_ARG, "sc_rng_below: the bound must fill its top word (nwords = %d, bound has %d words)", nwords, nw);
  uint64_t call;
  int rc = rng_begin(ctx, count, out, "sc_rng_below", &call); if (rc) return rc;
  uint32_t* d_n = nullptr;
  rc = device_n_half(ctx, n_hptr, nwords, &d_n); if (rc) return rc;
  if (launch_rng_below(ctx->stream, ctx->rng_key, call, d_n, nbits, nw, nonzero ? 1 : 0, out, count)) return fail(ctx, SC_ERR_HIP, "sc_rng_below: launch failed");
  return SC_OK;
}

int sc_rng_coins(sc_ctx* ctx, uint64_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  uint64_t call;
  int rc = rng_begin(ctx, (count + 511) / 512, out, "sc_rng_coins", &call); if (rc) return rc;
  if (launch_rng_coins(ctx->stream, ctx->rng_key, call, out, count)) return fail(ctx, SC_ERR_HIP, "sc_rng_coins: launch failed");
  return SC_OK;
}

int sc_rng_permutations(sc_ctx* ctx, int k, int64_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  if (ctx && (k < 1 || k > 256)) return fail(ctx, SC_ERR_ARG, "sc_rng_permutations: 1 <= k <= 256");
  uint64_t call;
  int rc = rng_begin(ctx, count, out, "sc_rng_permutations", &call); if (rc) return rc;
  if (launch_rng_perm(ctx->stream, ctx->rng_key, call, k, out, count)) return fail(ctx, SC_ERR_HIP, "sc_rng_permutations: launch failed");
  return SC_OK;
}

int sc_peak_probe(sc_ctx* ctx, double* out_mac_per_s) {
  if (!ctx || !out_mac_per_s) return SC_ERR_ARG;
  uint32_t* d_out;
  const int grid = ctx->num_cu * 8;
  HIPCHK(ctx, hipMalloc((void**)&d_out, (size_t)grid * 256 * 4));
  hipEvent_t e0, e1; HIPCHK(ctx, hipEventCreate(&e0)); HIPCHK(ctx, hipEventCreate(&e1));
  const int iters = 40000;
  double best = 0;
  for (int rep = 0; rep < 4; rep++) {
    HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
    if (launch_peak_probe(ctx->stream, grid, d_out, 12345u, 67890u, iters)) return fail(ctx, SC_ERR_HIP, "sc_peak_probe: launch failed");
    HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(e1));
    float ms = 0; HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
    double rate = (double)grid * 256 * (double)iters * 8 / (ms * 1e-3);
    if (rep > 0 && rate > best) best = rate;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d_out);
  *out_mac_per_s = best;
  return SC_OK;
}

int sc_table_traffic_probe(sc_ctx* ctx, int mod, const uint32_t* x, uint32_t* out, uint64_t count, int entries, int reads,
                           int* out_row_limbs) {
  if (!valid_mod(ctx, mod) || !x || !out || entries < 1 || entries > 64 || reads < 1 || reads > 4096)
    return fail(ctx, SC_ERR_ARG, "sc_table_traffic_probe: bad argument");
  const Mod& m = ctx->mods[mod];
  if (out_row_limbs) *out_row_limbs = m.S;
  std::string key = "tprobe:" + std::to_string(mod) + ":" + std::to_string(entries) + ":" + std::to_string(reads);
  auto it = ctx->progs.find(key);
  if (it == ctx->progs.end()) {
    Builder bd;
    bd.loadw(0); bd.mul_const(0);                                   // x in Montgomery form
    for (int e = 0; e < entries; e++) bd.stt((uint32_t)e);          // `entries` rows written
    for (int r = 0; r < reads; r++) bd.loadt_tbl((uint32_t)((r * 7 + 3) % entries));   // `reads` rows read
    bd.redc(); bd.storew(1); bd.end();
    Prog p; int rc = finalize_prog(ctx, m, bd, &p); if (rc) return rc;
    it = ctx->progs.emplace(key, p).first;
  }
  VmExt ex[2] = {mk_ext(x, m.nwords, m.nwords), mk_ext(out, m.nwords, m.nwords)};
  return run_vm(ctx, mod, it->second, ex, 2, count);
}

int sc_mac_counter(sc_ctx* ctx, int reset, double* out_macs) {
  if (!ctx) return SC_ERR_ARG;
  if (out_macs) *out_macs = ctx->mac_counter;
  if (reset) ctx->mac_counter = 0;
  return SC_OK;
}

}  // extern "C"

#include "sc_schemes.h"
