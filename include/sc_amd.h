/*
 * sc_amd.h -- C ABI of libsc_amd.so: batched Paillier / DGK big-integer arithmetic on MI355X (gfx950).
 *
 * THIS header is what a maintainer binds: the context, key objects, the scheme operations (encrypt / randomize / decrypt /
 * is_zero), the protocol steps as five calls per batch, the random draws and the multi-GPU reassembly.  The primitives those are
 * composed of (modular products, the exponentiation shapes, inversion ...), the kernel-policy switches and the measurement
 * probes live in sc_amd_dev.h -- the toolbox of this repository's own tests, tools and bench, same library.
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
/* Bumped whenever an entry point is added, removed or changes meaning; the binding checks it (round 3: 3; round 4: 4 -- the header
 * split into sc_amd.h / sc_amd_dev.h, SC_STEP_DEFER_CHECKS and sc_ctx_check removed, sc_clock_probe and sc_ctx_policy added; round 5: 5
 * -- sc_ctx_set_pair_policy and sc_ctx_stats added in sc_amd_dev.h). */
#define SC_ABI_VERSION 5
int sc_abi_version(void);
/* device memory helpers for callers that do not bring their own allocator */
int sc_malloc(sc_ctx* ctx, size_t bytes, void** out_dptr);
int sc_free(sc_ctx* ctx, void* dptr);
int sc_memcpy_h2d(sc_ctx* ctx, void* dptr, const void* hptr, size_t bytes);
int sc_memcpy_d2h(sc_ctx* ctx, void* hptr, const void* dptr, size_t bytes);

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
/* Paillier key: the public modulus N and optionally the secret primes p, q (key holder) -- what `Paillier.from_security_parameter`
 * produces at SC/keyholder.py:155-158 and what the initiator receives as the public scheme (SC/initiator.py:177-203).  Every derived
 * modulus (N, N^2, p, q, p^2, q^2), exponent (N, lambda, p - 1, q - 1, q mod p - 1, ..), CRT constant and the pair contexts are
 * registered once; flags: SC_KEY_NO_CRT / SC_KEY_NO_PAIRS keep the literal single-modulus forms (tests, A/B). */
int sc_paillier_key_create(sc_ctx* ctx, const uint32_t* n_hptr, int nwords, const uint32_t* p_hptr /* nullable */,
                           const uint32_t* q_hptr /* nullable */, int pwords, int flags, int* out_key);
/* the primitive handles behind a key (for callers that mix scheme-level calls and the primitives of sc_amd_dev.h) */
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
 * ~2.5 t), window = fixed-base window of the tables for h, 1 .. 24 (2^window rows of the modulus's limb size per window: 6 GB
 * at 20, 82 GB at 24 for a 2048-bit n and 400-bit r; the key holder's half-size tables stop at 20).  table_src_ctx / table_src_key (nullable /
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
 * given (:516; a row that is not a permutation of 0 .. l acts as the identity), the last three in ONE launch whose store
 * is the shuffle (each item finds its output plane in the permutation row itself: no destination array, no extra launch).  beta: [l][count][nwords] bit-major; rhos / r_rand:
 * [l+1][count][words].  c_unblinded_out (nullable) receives the output of step 4h.  c_out: [l+1][count][nwords].
 * FLAG WORDS (here and in sc_initiator_step67): rsmall and delta_a hold 0 or 1 per comparison and only BIT 0 is read -- alpha and
 * alpha_tilde are bit fields (bit i = the i-th bit of r mod 2^l, of (r - N) mod 2^l), so every flag of a step is taken by position;
 * a caller's own "true" must be the integer 1 (sc_initiator_step1 and sc_rng_coins produce exactly that). */
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
 *   sc_rng_seed:         32-byte key from the OS (NULL: getrandom) or from the caller -- the latter for TESTS and known-answer
 *                        vectors only: seeding resets the call counter, so the same key replays the same streams.  An unseeded
 *                        context seeds itself from the OS on first use.  A context belongs to one host thread at a time (see the
 *                        conventions above); the call counter is atomic all the same, so two threads that did share a context
 *                        would never draw one (key, call) pair twice.
 *   sc_rng_bits:         out[count][ceil(bits/32)]: uniform below 2^bits (DGK randomizer exponents).
 *   sc_rng_below:        out[count][nwords]: uniform in [0, n) or, nonzero != 0, in [1, n), by rejection sampling on the device
 *                        (r below N, rho_i in [1, u), Paillier randomizer bases in [1, N)); n must fill its top word.
 *   sc_rng_coins:        out[count] uint64, each 0 or 1 (delta_A).
 *   sc_rng_permutations: out[count][k] int64: one uniform permutation of 0 .. k-1 per item (Fisher-Yates with rejection-sampled
 *                        indices; the step-4i shuffle, handed to sc_initiator_step4 / _step4i as `permutation`). */
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

#ifdef __cplusplus
}
#endif
#endif /* SC_AMD_H */
