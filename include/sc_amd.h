/*
 * sc_amd.h -- C ABI of libsc_amd.so: batched Paillier / DGK big-integer arithmetic on MI355X (gfx950).
 *
 * The reference (TNO-MPC/protocols.secure_comparison 4.4.0) has NO native/FFI boundary: its hot path
 * is reached through the Python object API of the un-vendored scheme packages
 * (tno.mpc.encryption_schemes.{paillier,dgk,templates,utils}, pyproject.toml:32-38) by operator
 * overloading from Initiator.step_* / KeyHolder.step_*.  This header is therefore the boundary a
 * maintainer would bind with ctypes underneath those scheme objects; every entry point names the
 * reference call sites (file:line under /root/reference/src/tno/mpc/protocols/secure_comparison/,
 * "SC/") whose arithmetic it replaces.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - Big integers are canonical residues stored as little-endian arrays of uint32_t words; a batch is
 *     a dense row-major array [count][nwords] in DEVICE memory (hipMalloc / sc_malloc / a torch tensor's
 *     data_ptr()).  "dptr" parameters are device pointers; "hptr" parameters are host pointers.
 *   - All functions return 0 on success or a negative sc_status; sc_last_error() gives the message.
 *   - Calls are asynchronous on the context's stream (sc_ctx_set_stream) unless stated otherwise.
 *   - No function falls back to host arithmetic: without a gfx950 device every call fails.
 *   - A context belongs to one host thread at a time and orders all its work on one stream; its temporary device buffers are
 *     reused from call to call.  sc_ctx_set_stream orders the work already queued on the previous stream before anything
 *     queued on the new one (event wait, no host synchronisation), so alternating streams on one context is safe; the
 *     caller's own input / output buffers follow the usual stream rules.  Use one context per GPU (one process per GPU under
 *     torch.distributed), or one per concurrent shard of a GPU.
 */
#ifndef SC_AMD_H
#define SC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sc_ctx sc_ctx;

enum sc_status {
  SC_OK = 0,
  SC_ERR_ARG = -1,          /* bad argument (AssertionError / ValueError at the Python layer) */
  SC_ERR_HIP = -2,          /* HIP runtime error */
  SC_ERR_NOT_INVERTIBLE = -3, /* an element has no modular inverse (gmpy2/pow raise in the reference) */
  SC_ERR_UNSUPPORTED = -4   /* modulus too large for the compiled configurations */
};

/* ---- context ------------------------------------------------------------------------------------ */
int sc_ctx_create(int device_id, sc_ctx** out_ctx);
void sc_ctx_destroy(sc_ctx* ctx);
int sc_ctx_set_stream(sc_ctx* ctx, void* hip_stream);   /* hipStream_t; NULL = default stream */
int sc_ctx_synchronize(sc_ctx* ctx);
const char* sc_last_error(sc_ctx* ctx);
/* The element named by the most recent SC_ERR_NOT_INVERTIBLE of this context (-1 before the first): the step-level entry points
 * have no bad_index parameter of their own. */
int64_t sc_last_bad_index(sc_ctx* ctx);
/* Bumped whenever an entry point is added or changes meaning; the binding checks it (round 1: 1, round 2 shipped 1 by mistake, round 3: 3). */
#define SC_ABI_VERSION 3
int sc_abi_version(void);
/* Small-batch policy.  A modulus in an L = 18 configuration can be worked on by twice the lanes with 9 limbs each (same limb
 * arrays in memory): twice the waves, 1.8x shorter dependent chains, lower multiply-add density.  mode 0: never; 1 (default):
 * when the L = 18 launch would leave at least half of the SIMDs without a wave; 2: whenever such a kernel exists (tests). */
int sc_ctx_set_latency_mode(sc_ctx* ctx, int mode);
/* Large-batch policy for moduli of at most 1028 bits (the primes p, q of 2048-bit keys: key-holder CRT, DGK zero test).  The
 * shared-exponent entry points sc_modexp_shared and sc_modexp_shared_isone can run such a modulus in the one-lane
 * configuration -- one number per lane, 37 limbs of 28 bits, operands of a squaring in registers, modulus in scalar
 * registers: 1.25x the rate per number, but 64 numbers per wave (exponents of more than 64 bits only).  mode 0: never; 1 (default): from one and a half rounds of the
 * chip's resident waves (196608 numbers on 256 CUs); 2: whenever the modulus fits (tests).  Results are the same canonical residues in every mode. */
int sc_ctx_set_onelane_mode(sc_ctx* ctx, int mode);
/* Tell the context that `contexts` library contexts (this one included) work on its GPU at the same time -- the concurrent
 * shards of one batch, each on its own stream.  Batch-size policies then count rounds of 1/contexts of the chip (a launch that
 * under-fills the whole chip is not alone on it).  Default 1. */
int sc_ctx_set_chip_share(sc_ctx* ctx, int contexts);
/* Fork / join inside one call.  The p- and q-side of the key holder's CRT (sc_paillier_decrypt, sc_paillier_randomize with a secret
 * key) are independent; when a launch of the batch leaves room for a second one beside it (small batches) the q-side is queued on
 * a second stream of the context, with its own scratch arena and temporaries, and joined before the recombination.  mode 1
 * (default): automatic; 0: never (a context that already shares the GPU with another busy one, e.g. the second context that
 * computes randomizers ahead of time).  Off as well when the latency mode is 0; forced on by latency mode 2 (tests). */
int sc_ctx_set_fork_mode(sc_ctx* ctx, int mode);

/* device memory helpers for callers that do not bring their own allocator */
int sc_malloc(sc_ctx* ctx, size_t bytes, void** out_dptr);
int sc_free(sc_ctx* ctx, void* dptr);
int sc_memcpy_h2d(sc_ctx* ctx, void* dptr, const void* hptr, size_t bytes);
int sc_memcpy_d2h(sc_ctx* ctx, void* hptr, const void* dptr, size_t bytes);

/* ---- per-key setup (replaces what the scheme constructors precompute; [ext] Paillier/DGK __init__) ---- */
/* Register an odd modulus (Paillier N, N^2, p^2, q^2; DGK n, p).  nwords = words of every residue array. */
int sc_mod_create(sc_ctx* ctx, const uint32_t* n_hptr, int nwords, int* out_mod);
int sc_mod_words(sc_ctx* ctx, int mod);
/* Register an exponent shared by a whole batch (Paillier N or lambda, DGK v_p, 2^i, 3, ...). */
int sc_exp_create(sc_ctx* ctx, const uint32_t* e_hptr, int ewords, int* out_exp);
/* Register a constant residue (kept in Montgomery form on the device): g, g^-1, mu, N (mod N^2) ... */
int sc_const_create(sc_ctx* ctx, int mod, const uint32_t* v_hptr, int nwords, int* out_const);
/* Build the fixed-base table base^(d * 2^(window*j)) for exponents below 2^exp_bits
 * (DGK h and g: the randomizers h^r of SC/initiator.py:153-154 and SC/keyholder.py:106-108). */
int sc_fbt_create(sc_ctx* ctx, int mod, const uint32_t* base_hptr, int exp_bits, int window, int* out_fbt);
/* Use a table another context of the same device built for the same modulus (a window-20 table for h is 6 GB: concurrent
 * shard contexts of one GPU read one copy).  The rows are read-only and reference-counted: they are freed when the last
 * context holding the table is destroyed.  `mod` must be `ctx`'s registration of the modulus the table was built for.  Call it
 * while `src_ctx` is idle (its table list is read without a lock); the imported table itself is safe to use from `ctx`'s thread
 * while `src_ctx` works. */
int sc_fbt_import(sc_ctx* ctx, int mod, sc_ctx* src_ctx, int src_fbt, int* out_fbt);
/* Device bytes of a table's rows (reported by bench.py next to the throughput that depends on them). */
int sc_fbt_bytes(sc_ctx* ctx, int fbt, uint64_t* out_bytes);

/* ---- batched residue arithmetic (the ciphertext operator algebra) ---------------- */
/* out[i] = a[i] * b[i] mod n.  Stride 0 broadcasts a single residue.  ct + ct (SC/initiator.py:254,
 * 476-483, 563), ct * randomizer (every .randomize()). */
int sc_modmul(sc_ctx* ctx, int mod, const uint32_t* a_dptr, int a_stride_words, const uint32_t* b_dptr,
              int b_stride_words, uint32_t* out_dptr, uint64_t count);
/* out[i] = a[i] * c mod n for a registered constant c (ct + int with int in {0,1}: SC/initiator.py:320,
 * 476, 484, 531). */
int sc_modmul_const(sc_ctx* ctx, int mod, const uint32_t* a_dptr, int cst, uint32_t* out_dptr, uint64_t count);
/* out[i] = a[i] * (flags[i] ? c1 : c0) mod n for registered constants (-1 = the residue 1), one byte flag per item: the
 * unrandomized DGK encryption of a bit folded into its randomizer, g^bit * h^r (SC/keyholder.py:213, 231 with :106-108) --
 * no array of g^bit values is ever materialised. */
int sc_modmul_const_sel(sc_ctx* ctx, int mod, const uint32_t* a_dptr, int cst0, int cst1, const uint8_t* flags_dptr,
                        uint32_t* out_dptr, uint64_t count);
/* out[i] = x[i]^e mod n [* mul_into[i]], e shared: Paillier rho^N mod N^2 (SC/initiator.py:109,
 * SC/keyholder.py:126-128), c^lambda (SC/keyholder.py:195), w^(2^i) (SC/initiator.py:406), w_sum^3 (:480).
 * x may be wider than the modulus (x_words > nwords): it is reduced first (DGK zero test works mod p). */
int sc_modexp_shared(sc_ctx* ctx, int mod, int exp, const uint32_t* x_dptr, int x_words,
                     const uint32_t* mul_into_dptr /* nullable */, uint32_t* out_dptr, uint64_t count);
/* out[i] = x[i]^e mod m^2 [* mul_into[i]] for a modulus that is a perfect square m^2 (Paillier N^2, and p^2 / q^2 in the key
 * holder's CRT), computed with Montgomery products modulo m only: elements are held as pairs (x0, x1), X = (x0 + x1 m)/R, and
 * the recorded Montgomery quotient of x0 y0 carries the overflow into the m-part -- 3.5 S^2 multiply-adds per squaring
 * instead of 6 S^2 (S = limbs of m), identical residues.  mod_m2 must be registered for m^2 with 2 * words(mod_m) words;
 * x: [count][x_words], x_words <= 4 * words(mod_m) (wider operands are reduced mod m^2 implicitly); out / mul_into:
 * [count][2 * words(mod_m)].  Available when sc_mod_supports_sq(mod_m) returns 1 (moduli up to 2080 bits), else
 * SC_ERR_UNSUPPORTED -- use sc_modexp_shared on mod_m2 then.  Same reference call sites as sc_modexp_shared. */
int sc_modexp_shared_sq(sc_ctx* ctx, int mod_m, int mod_m2, int exp, const uint32_t* x_dptr, int x_words,
                        const uint32_t* mul_into_dptr /* nullable */, uint32_t* out_dptr, uint64_t count);
int sc_mod_supports_sq(sc_ctx* ctx, int mod);
/* flags[i] = (x[i]^e mod n == 1): DGK.is_zero, SC/keyholder.py:249 (e = v_p, n = p). */
int sc_modexp_shared_isone(sc_ctx* ctx, int mod, int exp, const uint32_t* x_dptr, int x_words,
                           uint8_t* flags_dptr, uint64_t count);
/* any_flags[b] = OR over the planes i of (x[i * inner + b]^e mod n == 1), b < inner, count = planes * inner items (bit-major):
 * the whole of KeyHolder.step_4j -- delta_B = any(is_zero(c_i)) (SC/keyholder.py:246-253) -- in the zero-test launch itself
 * (uint64 per comparison, 0 or 1; the array is cleared first). */
int sc_modexp_shared_isone_any(sc_ctx* ctx, int mod, int exp, const uint32_t* x_dptr, int x_words, uint64_t inner,
                               uint64_t* any_flags_dptr, uint64_t count);
/* out[i] = base^e[i] mod n [* mul_into[i]] with the fixed-base table: DGK randomize / encrypt
 * g^m h^r (SC/keyholder.py:106-108, 213, 231; SC/initiator.py:153-154). e: [count][ewords]. */
int sc_fixedbase_pow(sc_ctx* ctx, int fbt, const uint32_t* e_dptr, int ewords,
                     const uint32_t* mul_into_dptr /* nullable */, uint32_t* out_dptr, uint64_t count);
/* out[i] = x[i]^e[i] mod n [* base^e2[i]] with per-element exponents of at most ebits bits:
 * the blinding c_i^rho_i of SC/initiator.py:512 optionally fused with the randomizer h^r_i of :153-154. */
int sc_modexp_var(sc_ctx* ctx, int mod, const uint32_t* x_dptr, const uint32_t* e_dptr, int ewords, int ebits,
                  int fbt /* -1 = none */, const uint32_t* e2_dptr, int e2words, uint32_t* out_dptr,
                  uint64_t count);
/* The same with the result of item i stored at row dest_index[i] of out[count][nwords] (uint64 per item): the per-comparison
 * shuffle of the blinded c-vector (SC/initiator.py:212-226, :516) happens in the store of the blinding launch instead of a
 * separate gather pass over the 0.5 GB vector.  dest_index should be a permutation of 0 .. count-1; rows >= count are dropped
 * (never written out of bounds), rows named twice keep one of the values. */
int sc_modexp_var_scatter(sc_ctx* ctx, int mod, const uint32_t* x_dptr, const uint32_t* e_dptr, int ewords, int ebits,
                          int fbt /* -1 = none */, const uint32_t* e2_dptr, int e2words, const uint64_t* dest_index_dptr,
                          uint32_t* out_dptr, uint64_t count);
/* out[i] = x[i]^-1 mod n (Montgomery's simultaneous inversion + an on-device binary extended GCD):
 * ct * -1 / int - ct / ct - ct (SC/initiator.py:254, 320, 371, 466, 478, 531, 559).
 * Synchronous.  On SC_ERR_NOT_INVERTIBLE *bad_index (nullable) is the index of a non-invertible element (found by testing
 * the members of the failing chunk individually) and sc_last_error() names it; `out` is unspecified then.
 * `out` must not overlap `x` (SC_ERR_ARG): the operands are re-read on the error path. */
int sc_modinv(sc_ctx* ctx, int mod, const uint32_t* x_dptr, uint32_t* out_dptr, uint64_t count,
              int64_t* bad_index);

/* ---- Paillier pieces that are not plain residue products --------------------------------------- */
/* out[i] = 1 + m[i] * N mod N^2 (g = N+1 encryption without randomness: unsafe_encrypt(..) of
 * SC/initiator.py:256, 562; SC/keyholder.py:274-286).  mod_n2 = N^2, cst_n = sc_const_create(mod_n2, N).
 * m: [count][m_words], any m < 2^(32 m_words) (reduction mod N is implicit). */
int sc_paillier_encrypt_raw(sc_ctx* ctx, int mod_n2, int cst_n, const uint32_t* m_dptr, int m_words,
                            uint32_t* out_dptr, uint64_t count);
/* out[i] = 1 - m[i] * N mod N^2 = [[-m]] = [[m]]^-1: lets the Initiator form ([[r div 2^l]])^-1 of SC/initiator.py:559-563 without
 * a modular inversion. */
int sc_paillier_encrypt_raw_neg(sc_ctx* ctx, int mod_n2, int cst_n, const uint32_t* m_dptr, int m_words,
                                uint32_t* out_dptr, uint64_t count);
/* out[i] = ((x[i] - 1) / n) * k mod n for x[i] = 1 (mod n), x: [count][x_words]: the L function and the
 * mu multiplication of Paillier.decrypt (SC/keyholder.py:195).  mod = N (or p, q for CRT), cst_k = mu. */
int sc_paillier_l_mul(sc_ctx* ctx, int mod, int cst_k, const uint32_t* x_dptr, int x_words, uint32_t* out_dptr,
                      uint64_t count);

/* out[i] = the x in [0, m_p m_q) with x = a_p[i] (mod m_p), x = a_q[i] (mod m_q):  a_q + m_q ((a_p - a_q) m_q^-1 mod m_p).
 * Used by the key holder to recombine the CRT halves of decrypt (m_p = p, m_q = q; SC/keyholder.py:195) and of rho^N (m_p = p^2,
 * m_q = q^2; SC/keyholder.py:126-128); identical integers to the reference's single-modulus pow_mod.  Constants: cst_k = m_q^-1 mod m_p and cst_negk =
 * m_p - cst_k registered for mod_p, cst_mq = m_q registered for mod_full (= m_p m_q). */
int sc_crt_combine(sc_ctx* ctx, int mod_p, int mod_full, int cst_k, int cst_negk, int cst_mq, const uint32_t* a_p_dptr,
                   int a_p_words, const uint32_t* a_q_dptr, int a_q_words, uint32_t* out_dptr, uint64_t count);

/* ---- plaintext-side word arithmetic of the two parties (HBM-bound helpers) ------------------------ */
/* From r[count][nw] and the Paillier N: m1 = 2^l + r ([count][nw+1], SC/initiator.py:256), alpha = r mod 2^l
 * (:270), alpha_tilde = (r - N) mod 2^l (:373), rsmall = [r < (N-1)/2] (:289, :559), rshift = r >> l (:562).
 * alpha / alpha_tilde / rsmall are uint64 per item. l <= 64. */
int sc_plain_alice(sc_ctx* ctx, const uint32_t* r_dptr, const uint32_t* n_hptr, int nw, int l, uint64_t count,
                   uint32_t* m1_dptr, uint64_t* alpha_dptr, uint64_t* alpha_tilde_dptr, uint64_t* rsmall_dptr,
                   uint32_t* rshift_dptr);
/* From z[count][nw]: beta = z mod 2^l (SC/keyholder.py:196), dbit = [z < (N-1)/2] (:213), zeta1 = z >> l,
 * zeta2 = (z + N) >> l if dbit else z >> l (:274-282). */
int sc_plain_bob(sc_ctx* ctx, const uint32_t* z_dptr, const uint32_t* n_hptr, int nw, int l, uint64_t count,
                 uint64_t* beta_dptr, uint64_t* dbit_dptr, uint32_t* zeta1_dptr, uint32_t* zeta2_dptr);

/* ---- fused Initiator steps 4c-4h (SC/initiator.py:272-485) --------------------------------------- */
/* Inputs, all bit-major: beta[l][count][nw], beta_inv[l][count][nw], d[count][nw], d_inv[count][nw]
 * (DGK ciphertexts mod n and their inverses), alpha / alpha_tilde / rsmall / delta_a uint64 per comparison,
 * cst_g / cst_ginv = registered g and g^-1.  Output c[l+1][count][nw] in the order c_-1, c_0 .. c_{l-1}
 * (SC/initiator.py:484), not blinded. */
int sc_dgk_step4(sc_ctx* ctx, int mod, int cst_g, int cst_ginv, int l, const uint32_t* beta_dptr,
                 const uint32_t* beta_inv_dptr, const uint32_t* d_dptr, const uint32_t* d_inv_dptr,
                 const uint64_t* alpha_dptr, const uint64_t* alpha_tilde_dptr, const uint64_t* rsmall_dptr,
                 const uint64_t* delta_a_dptr, uint32_t* c_out_dptr, uint64_t count);

/* ---- scheme-level entry points: what the reference's scheme objects and protocol steps do, one call each ------------- */
/* A key object holds everything the scheme constructors derive ([ext] Paillier / DGK __init__; SC/keyholder.py:155-166): the
 * moduli N, N^2 (and p, p^2, q, q^2 for the key holder), the exponents N, lambda, p-1, q mod (p-1) .., mu / h_p / the CRT
 * recombination constants, g^-1, and the fixed-base tables for h.  With p / q (and v_p / v_q) the key is the key holder's and
 * -- unless SC_KEY_NO_CRT -- every exponentiation runs through CRT (identical integers, ~3.3x fewer limb products); without
 * them it is the public copy Alice receives (SC/initiator.py:177-203).  SC_KEY_NO_PAIRS forces exponentiations modulo N^2 to
 * use products modulo N^2 instead of the pair arithmetic modulo N (measurement / tests).  All arrays of the calls below are
 * device arrays of canonical words; N has `nwords` words, ciphertexts 2 * nwords; DGK residues `nwords` of its key. */
#define SC_KEY_NO_CRT 1
#define SC_KEY_NO_PAIRS 2
/* `flags` of the step entry points below.  SC_STEP_RANDOMIZERS_READY: the randomizer argument (rho_z / r_rand / rho3) holds the
 * FINISHED randomizers -- rho^N mod N^2 as [..][2 nwords(N)], h^r mod n as [..][nwords(n)] -- computed ahead of time by
 * sc_paillier_randomize / sc_dgk_randomize with c = NULL, e.g. on a second context and stream while the protocol's critical
 * path runs (the reference pre-generates its randomizers in background workers: boot_randomness_generation,
 * SC/initiator.py:205-210, SC/keyholder.py:174-179).  The step then applies them with one modular product each. */
#define SC_STEP_RANDOMIZERS_READY 1
/* SC_STEP_DEFER_CHECKS: the step's modular inversion (sc_initiator_step1 / _step4 / _step67) does not wait for its verdict --
 * no host round trip in the middle of a protocol step, the whole step is queued ahead of the GPU.  The caller must call
 * sc_ctx_check before trusting the outputs (one synchronisation for all pending inversions; a non-invertible element is named as
 * by sc_modinv: the failing inversion is repeated with its verdicts read at once, so the step's INPUT arrays must be intact until
 * then).  Without the flag a step reports a non-invertible element itself (SC_ERR_NOT_INVERTIBLE).  For drivers that run both
 * parties in one process; a party that is about to SEND a step's output checks first. */
#define SC_STEP_DEFER_CHECKS 2
int sc_ctx_check(sc_ctx* ctx, int64_t* bad_index /* nullable */);
/* Paillier key: the public modulus N and optionally the secret primes p, q (key holder) -- what `Paillier.from_security_parameter`
 * produces at SC/keyholder.py:155-158 and what the initiator receives as the public scheme (SC/initiator.py:177-203).  Every derived
 * modulus (N, N^2, p, q, p^2, q^2), exponent (N, lambda, p - 1, q - 1, q mod p - 1, ..), CRT constant and the pair contexts are
 * registered once; flags: SC_KEY_NO_CRT / SC_KEY_NO_PAIRS keep the literal single-modulus forms (tests, A/B). */
int sc_paillier_key_create(sc_ctx* ctx, const uint32_t* n_hptr, int nwords, const uint32_t* p_hptr /* nullable */,
                           const uint32_t* q_hptr /* nullable */, int pwords, int flags, int* out_key);
/* the primitive handles behind a key (for callers that mix scheme-level and primitive calls) */
int sc_paillier_key_mods(sc_ctx* ctx, int key, int* out_mod_n, int* out_mod_n2);
/* out[i] = 1 + m[i] N (negate: 1 - m[i] N) mod N^2: unsafe_encrypt(m, apply_encoding=False), SC/initiator.py:256, 562;
 * SC/keyholder.py:274-286. */
int sc_paillier_encrypt(sc_ctx* ctx, int key, const uint32_t* m_dptr, int m_words, int negate, uint32_t* out_dptr, uint64_t count);
/* ct.randomize() for a batch: out[i] = c[i] * rho[i]^N mod N^2 (c = NULL: the randomizers alone), rho: [count][nwords].
 * SC/initiator.py:109; SC/keyholder.py:126-128 (the key holder's goes through CRT over p^2, q^2). */
int sc_paillier_randomize(sc_ctx* ctx, int key, const uint32_t* c_dptr /* nullable */, const uint32_t* rho_dptr, uint32_t* out_dptr,
                          uint64_t count);
/* Paillier.decrypt(ct, apply_encoding=False): out[i] = L(c[i]^lambda mod N^2) mu mod N, [count][nwords] (SC/keyholder.py:195). */
int sc_paillier_decrypt(sc_ctx* ctx, int key, const uint32_t* c_dptr, uint32_t* out_dptr, uint64_t count);
/* DGK key (`DGK.from_security_parameter`, SC/keyholder.py:161-166): public (n, g, h, u, t) and optionally secret (p, q, v_p, v_q); randomizer_bits = width of the exponent r of h^r ([ext]
 * ~2.5 t), window = fixed-base window of the tables for h (2^window rows per window).  table_src_ctx / table_src_key (nullable /
 * ignored): another context of the same GPU whose key of the same modulus, h, window and width already built the tables -- they
 * are shared read-only instead of built again (concurrent shard contexts; see sc_fbt_import). */
int sc_dgk_key_create(sc_ctx* ctx, const uint32_t* n_hptr, const uint32_t* g_hptr, const uint32_t* h_hptr, int nwords,
                      const uint32_t* u_hptr, int uwords, int t_bits, const uint32_t* p_hptr /* nullable */, const uint32_t* q_hptr,
                      int pwords, const uint32_t* vp_hptr, const uint32_t* vq_hptr, int vwords, int randomizer_bits, int window,
                      int flags, sc_ctx* table_src_ctx /* nullable */, int table_src_key, int* out_key);
int sc_dgk_key_info(sc_ctx* ctx, int key, int* out_mod_n, int* out_mod_p /* -1 without secret key */, uint64_t* out_table_bytes);
/* ct.randomize() for DGK: out[i] = c[i] * h^r[i] mod n (c = NULL: the randomizers alone); r: [count][ewords].
 * SC/keyholder.py:106-108 (CRT with exponents reduced modulo v_p, v_q), SC/initiator.py:153-154. */
int sc_dgk_randomize(sc_ctx* ctx, int key, const uint32_t* c_dptr /* nullable */, const uint32_t* r_dptr, int ewords,
                     uint32_t* out_dptr, uint64_t count);
/* unsafe_encrypt(bit) + .randomize() in one go: out[i] = g^bits[i] * h^r[i] mod n, bits: one byte per item
 * (SC/keyholder.py:213, 231 with :106-108). */
int sc_dgk_encrypt_bits_randomized(sc_ctx* ctx, int key, const uint8_t* bits_dptr, const uint32_t* r_dptr, int ewords,
                                   uint32_t* out_dptr, uint64_t count);
/* DGK.is_zero per ciphertext (SC/keyholder.py:249), and the whole of step 4j: any_flags[b] = OR over the planes of the bit-major
 * vector c[planes][inner][nwords] (:246-253). */
int sc_dgk_is_zero(sc_ctx* ctx, int key, const uint32_t* c_dptr, uint8_t* flags_dptr, uint64_t count);
int sc_dgk_any_zero(sc_ctx* ctx, int key, const uint32_t* c_dptr, int planes, uint64_t inner, uint64_t* any_flags_dptr);

/* Initiator.step_1 + step_3 + the plaintext side of 4c / 4e / 7 for a batch (SC/initiator.py:228-270, :289, :373, :558-562):
 * z = [[y]] [[x]]^-1 [[2^l + r]] mod N^2, randomized with rho_z^N when rho_z is given (:109); alpha = r mod 2^l,
 * alpha_tilde = (r - N) mod 2^l, rsmall = [r < (N-1)/2] (uint64 each), rshift = r div 2^l ([count][nwords]). */
int sc_initiator_step1(sc_ctx* ctx, int paillier_key, int l, const uint32_t* x_enc_dptr, const uint32_t* y_enc_dptr,
                       const uint32_t* r_dptr, const uint32_t* rho_z_dptr /* nullable */, int flags, uint32_t* z_out_dptr, uint64_t* alpha_dptr,
                       uint64_t* alpha_tilde_dptr, uint64_t* rsmall_dptr, uint32_t* rshift_dptr, uint64_t count);
/* KeyHolder.step_2 + step_4a + step_4b (+ the l + 1 .randomize() of SC/keyholder.py:106-108 when r_rand is given): decrypt z,
 * derive beta / d / zeta_1 / zeta_2, and encrypt d and the bits of beta bit-major: d_beta_out[l+1][count][nwords(n)], plane 0 =
 * [d], plane 1 + i = [beta_i].  z_out: [count][nwords(N)]; beta / dbit uint64. */
int sc_keyholder_step2_4b(sc_ctx* ctx, int paillier_key, int dgk_key, int l, const uint32_t* z_enc_dptr,
                          const uint32_t* r_rand_dptr /* nullable */, int r_words, int flags, uint32_t* z_out_dptr, uint64_t* beta_dptr,
                          uint64_t* dbit_dptr, uint32_t* zeta1_dptr, uint32_t* zeta2_dptr, uint32_t* d_beta_out_dptr, uint64_t count);
/* Initiator.step_4c .. step_4i for a batch (SC/initiator.py:272-516): one inversion pass over [d], [beta_i], the fused steps
 * 4c-4h (sc_dgk_step4), then -- when rhos is given -- the blinding c_i^rho_i (:512), the re-randomization * h^r_i when r_rand is
 * given (:153-154) and the shuffle when permutation ([count][l+1] int64, output k takes blinded c at index permutation[b][k]) is
 * given (:516; a row that is not a permutation of 0 .. l is replaced by the identity), the last three in ONE launch whose store
 * is the shuffle.  beta: [l][count][nwords] bit-major; rhos / r_rand:
 * [l+1][count][words].  c_unblinded_out (nullable) receives the output of step 4h.  c_out: [l+1][count][nwords]. */
int sc_initiator_step4(sc_ctx* ctx, int dgk_key, int l, const uint32_t* d_enc_dptr, const uint32_t* beta_enc_dptr,
                       const uint64_t* alpha_dptr, const uint64_t* alpha_tilde_dptr, const uint64_t* rsmall_dptr,
                       const uint64_t* delta_a_dptr, const uint32_t* rhos_dptr /* nullable */, int rho_words,
                       const int64_t* permutation_dptr /* nullable */, const uint32_t* r_rand_dptr /* nullable */, int r_words, int flags,
                       uint32_t* c_unblinded_out_dptr /* nullable */, uint32_t* c_out_dptr, uint64_t count);
/* Step 4i on its own (blinding, optional re-randomization, optional shuffle) for a vector c_in[l+1][count][nwords] that is already
 * there (SC/initiator.py:487-516, :153-154); c_out must not be c_in when a permutation is given. */
int sc_initiator_step4i(sc_ctx* ctx, int dgk_key, int l, const uint32_t* c_in_dptr, const uint32_t* rhos_dptr, int rho_words,
                        const int64_t* permutation_dptr /* nullable */, const uint32_t* r_rand_dptr /* nullable */, int r_words, int flags,
                        uint32_t* c_out_dptr, uint64_t count);
/* KeyHolder.step_4j + step_5 (+ the 3 .randomize() of SC/keyholder.py:126-128 when rho3 is given): delta_B per comparison,
 * then out3[3][count][2 nwords] = [[zeta_1]], [[zeta_2]], [[delta_B]]; rho3: [3][count][nwords] in the same order. */
int sc_keyholder_step4j_5(sc_ctx* ctx, int paillier_key, int dgk_key, int l, const uint32_t* c_enc_dptr, const uint32_t* zeta1_dptr,
                          const uint32_t* zeta2_dptr, const uint32_t* rho3_dptr /* nullable */, int flags, uint64_t* delta_b_out_dptr,
                          uint32_t* out3_dptr, uint64_t count);
/* Initiator.step_6 + step_7 (SC/initiator.py:518-564) with one inversion pass: out = [[x <= y]], not randomized. */
int sc_initiator_step67(sc_ctx* ctx, int paillier_key, const uint64_t* delta_a_dptr, const uint32_t* delta_b_enc_dptr,
                        const uint32_t* zeta1_enc_dptr, const uint32_t* zeta2_enc_dptr, const uint64_t* rsmall_dptr,
                        const uint32_t* rshift_dptr, int flags, uint32_t* out_dptr, uint64_t count);

/* ---- device-side CSPRNG: the random draws of a batch, generated where they are consumed ---------------- */
/* The reference draws from Python's `secrets` (SC/initiator.py:223 permutation, :250 r, :420 delta_A, :512 rho_i) and the
 * scheme packages draw the randomizers behind every .randomize() ([ext]).  A batch of 65536 comparisons needs ~0.3 GB of such
 * draws per step; these entry points produce them on the device from a counter-mode generator: the ChaCha20 block function
 * (RFC 8439 2.3), keystream(call, item) = ChaCha20_block(key, counter = 0, 1, .., nonce = (item, call_lo, call_hi)) read as
 * little-endian words, `call` = number of generator calls on this context since it was seeded.  Asynchronous on the stream.
 *   sc_rng_seed:         32-byte key from the caller (reproducible tests, known-answer vectors) or, with NULL, from the OS
 *                        (getrandom); resets the call counter.  An unseeded context seeds itself from the OS on first use.
 *   sc_rng_bits:         out[count][ceil(bits/32)]: uniform below 2^bits (DGK randomizer exponents).
 *   sc_rng_below:        out[count][nwords]: uniform in [0, n) or, nonzero != 0, in [1, n), by rejection sampling on the device
 *                        (r below N, rho_i in [1, u), Paillier randomizer bases in [1, N)); n must fill its top word.
 *   sc_rng_coins:        out[count] uint64, each 0 or 1 (delta_A).
 *   sc_rng_permutations: out[count][k] int64: one uniform permutation of 0 .. k-1 per item (Fisher-Yates with rejection-sampled
 *                        indices; the step-4i shuffle, consumed by sc_modexp_var_scatter as a destination index). */
int sc_rng_seed(sc_ctx* ctx, const uint8_t* key32_hptr /* nullable */);
int sc_rng_bits(sc_ctx* ctx, int bits, uint32_t* out_dptr, uint64_t count);
int sc_rng_below(sc_ctx* ctx, const uint32_t* n_hptr, int nwords, int nonzero, uint32_t* out_dptr, uint64_t count);
int sc_rng_coins(sc_ctx* ctx, uint64_t* out_dptr, uint64_t count);
int sc_rng_permutations(sc_ctx* ctx, int k, int64_t* out_dptr, uint64_t count);

/* ---- multi-GPU ---------------------------------------------------------------------- */
/* The comparisons of a batch are independent: every rank (one process and one context per GPU) runs all steps on its own block
 * with no traffic during compute; the only exchange is the reassembly of the ranks' result blocks -- [[x <= y]], or the per-bit
 * vectors in the blocked layout [rank][l+1][B/ranks][words] -- by ONE all-gather (RCCL over xGMI) on the context's stream.  The
 * reference has no counterpart (its only concurrency is session_id namespacing, SC/initiator.py:86-87).  RCCL is loaded on first
 * use (dlopen), so single-GPU users never need it.
 *   sc_comm_unique_id: rank 0 creates the 128-byte rendezvous id (SC_COMM_ID_BYTES); the caller ships it to the other ranks.
 *   sc_comm_init:      every rank joins with the same id (collective call).
 *   sc_allgather:      recv[r * words_per_rank ..] = rank r's send[0 .. words_per_rank) for every r; asynchronous on the stream.
 *   sc_comm_destroy:   also done by sc_ctx_destroy. */
#define SC_COMM_ID_BYTES 128
int sc_comm_unique_id(sc_ctx* ctx, void* id_hptr);
int sc_comm_init(sc_ctx* ctx, const void* id_hptr, int rank, int nranks);
int sc_allgather(sc_ctx* ctx, const uint32_t* send_dptr, uint32_t* recv_dptr, uint64_t words_per_rank);
int sc_comm_destroy(sc_ctx* ctx);

/* ---- measurement -------------------------------------------------------------------------------- */
/* Runs an on-device v_mad_u64_u32 issue-rate probe; returns lane-MACs per second (the VALU-integer peak
 * used as the roofline denominator).  Synchronous. */
int sc_peak_probe(sc_ctx* ctx, double* out_mac_per_s);
/* Number of Montgomery limb-products (v_mad_u64_u32 lane operations) issued by library calls since the
 * last reset -- counted on the host from the micro-programs, used for the roofline numerator. */
int sc_mac_counter(sc_ctx* ctx, int reset, double* out_macs);
/* Measurement aid (bench / profiles only, no reference counterpart): per item, write `entries` rows of the per-slot scratch
 * table and read rows back `reads` times with the kernels' own limb-form access pattern, then store the last row read
 * (= x) to out[count][nwords].  Known HBM bytes per item: entries * S * 4 written, reads * S * 4 read (S = *out_row_limbs),
 * plus one operand in and out -- the calibration point for the FETCH_SIZE / WRITE_SIZE counters of this access pattern. */
int sc_table_traffic_probe(sc_ctx* ctx, int mod, const uint32_t* x, uint32_t* out, uint64_t count, int entries, int reads,
                           int* out_row_limbs);

#ifdef __cplusplus
}
#endif
#endif /* SC_AMD_H */
