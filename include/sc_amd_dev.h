/*
 * sc_amd_dev.h -- the rest of libsc_amd.so's C ABI: the primitives the scheme-level entry points of sc_amd.h are composed of, the
 * kernel-policy switches and the measurement probes.  A maintainer binding the reference's scheme packages needs none of this
 * (sc_amd.h is complete for that); it is the toolbox of this repository's generic ciphertext operator algebra (schemes.py),
 * tests, tools/ and bench.py.  Same conventions as sc_amd.h; handles (`mod`, `exp`, `cst`, `fbt`) are small integers valid for
 * the context that created them.
 */
#ifndef SC_AMD_DEV_H
#define SC_AMD_DEV_H

#include "sc_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- kernel policies (defaults are right for production; tests and A/B runs force them) ------------------------------------ */
/* Small-batch policy.  A modulus in an L = 18 configuration can be worked on by twice the lanes with 9 limbs each (same limb
 * arrays in memory): twice the waves, 1.8x shorter dependent chains, lower multiply-add density.  mode 0: never; 1 (default):
 * when the L = 18 launch would leave at least half of the SIMDs without a wave; 2: whenever such a kernel exists (tests). */
int sc_ctx_set_latency_mode(sc_ctx* ctx, int mode);
/* Large-batch policy for moduli of at most 1028 bits (the primes p, q of 2048-bit keys: key-holder CRT, DGK zero test).  The
 * shared-exponent entry points sc_modexp_shared / _isone / _isone_any can run such a modulus in the one-lane configuration -- one
 * number per lane, 37 limbs of 28 bits, operands of a squaring in registers, modulus in scalar registers: 1.25x the rate per
 * number, but 64 numbers per wave (exponents of more than 64 bits only).  mode 0: never; 1 (default): whichever form a model of
 * both forms' rounds of resident waves says is faster for this batch size, with the per-round times MEASURED on this device
 * (sc_ctx_policy); 2: whenever the modulus fits (tests).  Results are the same canonical residues in every mode. */
int sc_ctx_set_onelane_mode(sc_ctx* ctx, int mode);
/* Tell the context that `contexts` library contexts (this one included) work on its GPU at the same time -- the concurrent
 * shards of one batch, each on its own stream.  Batch-size policies then count rounds of 1/contexts of the chip (a launch that
 * under-fills the whole chip is not alone on it).  Default 1. */
int sc_ctx_set_chip_share(sc_ctx* ctx, int contexts);
/* Fork / join inside one call.  The p- and q-side of the key holder's CRT (sc_paillier_decrypt, sc_paillier_randomize with a secret
 * key) are independent; when a launch of the batch leaves room for a second one beside it (small batches) the q-side is queued on
 * a second stream of the context, with its own scratch arena and temporaries, and joined before the recombination.  mode 1
 * (default): automatic -- small batches as described (off when the latency mode is 0, forced on by latency mode 2: tests), and
 * large ones when the context has the chip to itself (chip share 1: launches of 1.5 rounds pack when they overlap); 0: never;
 * 2: always. */
int sc_ctx_set_fork_mode(sc_ctx* ctx, int mode);
/* Long pair launches on a shared chip.  The waves of a launch of few rounds retire together, so nothing another context queues
 * behind it gets a wave slot before a round ends: Alice's rho^N of a shard of 32768 is 2048 waves for 54 ms, and the other shard's
 * sub-millisecond launches were seen waiting 26 .. 33 ms each behind it.  A context that shares its GPU (sc_ctx_set_chip_share > 1)
 * therefore cuts such a launch (sc_modexp_shared_sq) into segments -- the same micro-program cut at window boundaries, the pair
 * parked in the item's table slot in between; same residues.  The decision uses MEASURED quantities only: the time a resident wave
 * holds its slot = the program's pair squarings and products times the per-op times of the kernel instance, measured on the device
 * once per process (a lone wave of the caller's operands, ~2 ms); the launch is cut when it would occupy more than half of the
 * chip's wave slots (by the runtime's occupancy answer) for at most `max_rounds` rounds, into round(hold / hold_ms) segments (at
 * most 16).  hold_ms: how long a wave may hold its slot, default 5 ms (environment SC_PAIR_HOLD_MS at context creation; 0 or
 * SC_PAIR_SEGMENTS=1: never cut); max_rounds default 2.5 (longer launches retire their waves a round apart already). */
int sc_ctx_set_pair_policy(sc_ctx* ctx, double hold_ms, double max_rounds);
/* Counters of the context since its creation: out[0] = pair launches that ran in segments, out[1] = segments queued for them,
 * out[2] = op-time calibrations this context performed (the others are zero).  For tests and tools. */
int sc_ctx_stats(sc_ctx* ctx, uint64_t* out, int n);
/* The constants behind the automatic policies, measured once per device and process when the first secret key is created (about
 * 80 ms: full, half and one-and-a-half rounds of x^e mod p, 1024 bits, on the two-lane and on the one-lane kernel) instead of fitted
 * on one box: out[0..5] =
 * one-lane full round, one-lane half round, two-lane full round, two-lane half round, two-lane single half round (all in ms), and
 * the number of SIMDs the rounds were counted on.  Forces the calibration if it has not happened yet. */
int sc_ctx_policy(sc_ctx* ctx, double* out6);

/* ---- per-key setup (replaces what the scheme constructors precompute; [ext] Paillier/DGK __init__) ---- */
/* Register an odd modulus (Paillier N, N^2, p^2, q^2; DGK n, p).  nwords = words of every residue array. */
int sc_mod_create(sc_ctx* ctx, const uint32_t* n_hptr, int nwords, int* out_mod);
int sc_mod_words(sc_ctx* ctx, int mod);
/* Register an exponent shared by a whole batch (Paillier N or lambda, DGK v_p, 2^i, 3, ...). */
int sc_exp_create(sc_ctx* ctx, const uint32_t* e_hptr, int ewords, int* out_exp);
/* Register a constant residue (kept in Montgomery form on the device): g, g^-1, mu, N (mod N^2) ... */
int sc_const_create(sc_ctx* ctx, int mod, const uint32_t* v_hptr, int nwords, int* out_const);
/* Build the fixed-base table base^(d * 2^(window*j)) for exponents below 2^exp_bits, window 1 .. 24
 * (DGK h and g: the randomizers h^r of SC/initiator.py:153-154 and SC/keyholder.py:106-108). */
int sc_fbt_create(sc_ctx* ctx, int mod, const uint32_t* base_hptr, int exp_bits, int window, int* out_fbt);
/* Use a table another context of the same device built for the same modulus (a window-20 table for h is 6 GB, a window-24 one 82 GB: concurrent
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
/* In-kernel clock of the dominant launch (diagnostic, never a timed kernel): runs rho^N mod N^2 for `count` items of `paillier_key`
 * (a public key: the pair launch k_pvm<4,18,neg1>) on a twin of the kernel whose waves stamp s_memtime (shader clock) and
 * s_memrealtime (constant-rate clock) at entry and exit.  *out_ghz = mean over waves of d(memtime) / d(realtime) x the
 * constant clock's rate -- the engine clock the kernel actually held; *out_ms = the launch's duration by HIP events.  bench.py
 * calls it before the warm-up and after the timed loop (roofline.clock_ghz_before / _after).  Synchronous. */
int sc_clock_probe(sc_ctx* ctx, int paillier_key, const uint32_t* rho_dptr, uint64_t count, double* out_ghz, double* out_ms);

#ifdef __cplusplus
}
#endif
#endif /* SC_AMD_DEV_H */
