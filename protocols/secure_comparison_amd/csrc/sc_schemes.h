// Scheme-level entry points of the C ABI: what the reference's scheme objects and protocol steps do, ONE call each.
// Included at the end of sc_lib.hip (same translation unit: uses its context, program builder and launch helpers).
//
// The reference reaches its arithmetic through `ct.randomize()` (SC/keyholder.py:106-108, 126-128; SC/initiator.py:109,
// 153-154), `Paillier.decrypt` (SC/keyholder.py:195), `DGK.is_zero` (:249) and the operator algebra inside step_1 / step_4* /
// step_6 / step_7.  Rounds 1-2 exposed the primitives those decompose into; a maintainer binding them had to re-implement the
// key holder's CRT, the pair arithmetic's plumbing and the exponent reductions.  Here that composition lives in the library:
// a key object holds every derived modulus, exponent, constant and fixed-base table, and each entry point queues the same
// launches the Python layer used to compose -- identical residues.
#pragma once

namespace {

// ---- more set-up-time host integers (little-endian uint32 words) -------------------------------------------------------------
Big big_trimmed(Big a) { while (a.size() > 1 && a.back() == 0) a.pop_back(); return a; }
Big big_fit(Big a, size_t words) { a.resize(words, 0); return a; }
bool big_is_zero(const Big& a) { for (uint32_t w : a) if (w) return false; return true; }
bool big_is_one(const Big& a) { if (a.empty() || a[0] != 1) return false; for (size_t i = 1; i < a.size(); i++) if (a[i]) return false; return true; }
Big big_mod(const Big& x, const Big& m) { Big q, r; big_divmod(x, m, &q, &r); return r; }       // m.size() words
Big big_mulmod(const Big& a, const Big& b, const Big& m) { return big_mod(big_mul(a, b), m); }
// base^e mod m, set-up time only (square and multiply over the bits of e; m.size() words)
Big big_powmod(const Big& base, const Big& e, const Big& m) {
  Big r(m.size(), 0); r[0] = 1;
  r = big_mod(r, m);
  const Big b = big_mod(base, m);
  for (int i = big_bits(e) - 1; i >= 0; i--) {
    r = big_mulmod(r, r, m);
    if ((e[i >> 5] >> (i & 31)) & 1) r = big_mulmod(r, b, m);
  }
  return r;
}
Big big_sub_small(Big a, uint32_t k) { Big b(a.size(), 0); b[0] = k; big_sub(a, b); return a; }
void big_add_inplace(Big& a, const Big& b) {   // same length; the carry out is dropped (callers keep a spare top word)
  uint64_t c = 0;
  for (size_t i = 0; i < a.size(); i++) { c += (uint64_t)a[i] + b[i]; a[i] = (uint32_t)c; c >>= 32; }
}
void big_shr1(Big& a) { for (size_t i = 0; i < a.size(); i++) a[i] = (a[i] >> 1) | ((i + 1 < a.size()) ? (a[i + 1] << 31) : 0u); }
// a^-1 mod m for odd m (binary extended Euclid, HAC 14.61 shape); empty when gcd(a, m) != 1.  Result has m.size() words.
Big big_modinv_odd(const Big& a_in, const Big& m_in) {
  const size_t W = m_in.size() + 1;                         // one spare word: x + m never overflows
  const Big m = big_fit(m_in, W);
  Big u = big_fit(big_mod(a_in, m_in), W), v = m, x1(W, 0), x2(W, 0);
  x1[0] = 1;
  if (big_is_zero(u)) return Big();
  auto halve = [&](Big& x) { if (x[0] & 1) big_add_inplace(x, m); big_shr1(x); };
  while (!big_is_one(u) && !big_is_one(v)) {
    while (!(u[0] & 1)) { big_shr1(u); halve(x1); }
    while (!(v[0] & 1)) { big_shr1(v); halve(x2); }
    if (big_cmp(u, v) >= 0) { big_sub(u, v); if (big_cmp(x1, x2) < 0) big_add_inplace(x1, m); big_sub(x1, x2); }
    else { big_sub(v, u); if (big_cmp(x2, x1) < 0) big_add_inplace(x2, m); big_sub(x2, x1); }
    if (big_is_zero(u) || big_is_zero(v)) return Big();     // a common factor was subtracted away
  }
  Big r = big_is_one(u) ? x1 : x2;
  while (big_cmp(r, m) >= 0) big_sub(r, m);
  r.resize(m_in.size());
  return r;
}

int reg_exp(sc_ctx* ctx, const Big& e, int* out) { Big t = big_trimmed(e); return sc_exp_create(ctx, t.data(), (int)t.size(), out); }
int reg_const(sc_ctx* ctx, int mod, const Big& v, int* out) {
  return sc_const_create_cached(ctx, mod, big_fit(big_trimmed(v), std::max<size_t>(big_trimmed(v).size(), (size_t)ctx->mods[mod].nwords)), out);
}

struct PaillierHalf {            // one prime of the key holder's CRT
  int m1 = -1, m2 = -1;          // moduli p and p^2
  int exp_small = -1;            // q mod (p - 1): first stage of rho^N mod p^2
  int exp_p = -1, exp_pm1 = -1;  // p (second stage) and p - 1 (decryption)
  int cst_h = -1;                // h_p = L_p((N+1)^(p-1) mod p^2)^-1 mod p  (constant of m1)
};
struct PaillierKey {
  Big n; int nw = 0, hw = 0;
  int mod_n = -1, mod_n2 = -1, cst_n = -1, exp_n = -1;
  bool secret = false, crt = true, pairs = true;
  int exp_lambda = -1, cst_mu = -1;
  PaillierHalf hp, hq;
  int r_k = -1, r_negk = -1, r_mq = -1;   // recombination modulo p^2, q^2 -> N^2 (randomizers)
  int d_k = -1, d_negk = -1, d_mq = -1;   // recombination modulo p, q -> N (decryption)
};
struct DgkHalf { int m = -1, m_v = -1, fbt = -1, c_g = -1; };   // c_g: g mod p (mod q)
struct DgkKey {
  Big n, g, h, u; int nw = 0, rbits = 0, window = 0, t = 0;
  int mod_n = -1, cst_g = -1, cst_ginv = -1, fbt_h = -1;
  bool secret = false, crt = true;
  int mod_p = -1, exp_vp = -1, exp_one = -1;
  DgkHalf hp, hq;
  int c_k = -1, c_negk = -1, c_mq = -1;   // recombination modulo p, q -> n
};

}  // namespace

// (the key tables live in the context; declared here, stored through the pointers below)
struct sc_scheme_keys { std::vector<PaillierKey> paillier; std::vector<DgkKey> dgk; };

namespace {

void free_scheme_keys(void* p) { delete (sc_scheme_keys*)p; }

sc_scheme_keys* keys_of(sc_ctx* ctx) {
  if (!ctx->scheme_keys) ctx->scheme_keys = new sc_scheme_keys();
  return (sc_scheme_keys*)ctx->scheme_keys;
}
PaillierKey* paillier_key(sc_ctx* ctx, int key) {
  if (!ctx || !ctx->scheme_keys) return nullptr;
  auto& v = ((sc_scheme_keys*)ctx->scheme_keys)->paillier;
  return (key >= 0 && key < (int)v.size()) ? &v[key] : nullptr;
}
DgkKey* dgk_key(sc_ctx* ctx, int key) {
  if (!ctx || !ctx->scheme_keys) return nullptr;
  auto& v = ((sc_scheme_keys*)ctx->scheme_keys)->dgk;
  return (key >= 0 && key < (int)v.size()) ? &v[key] : nullptr;
}

enum SchemeTmp { TMP_S_A = 40, TMP_S_B, TMP_S_C, TMP_S_D, TMP_S_E, TMP_S_F, TMP_S_G, TMP_S_H, TMP_S_I };
template <typename T>
int tmp_words(sc_ctx* ctx, int slot, uint64_t elems, T** out) { return tmp_buf(ctx, slot, (size_t)elems * sizeof(T), (void**)out); }


// rho^N mod N^2 for the key holder: ((rho mod p)^(q mod p-1) mod p)^p mod p^2 per prime, recombined (identical integers)
int paillier_crt_pow_n(sc_ctx* ctx, const PaillierKey& k, const uint32_t* rho, uint32_t* out, uint64_t count) {
  uint32_t *y, *part_p, *part_q;
  int rc = tmp_words(ctx, TMP_S_A, count * k.hw, &y); if (rc) return rc;
  rc = tmp_words(ctx, TMP_S_B, count * 2 * k.hw, &part_p); if (rc) return rc;
  rc = tmp_words(ctx, TMP_S_C, count * 2 * k.hw, &part_q); if (rc) return rc;
  AuxFork fork;
  const bool forked = small_enough_to_fork(ctx, ctx->mods[k.hp.m1], count);
  for (int pass = 0; pass < 2; pass++) {
    const int side = forked ? 1 - pass : pass;       // forked: the q-side is queued first, on the second stream, the p-side beside it
    const PaillierHalf& h = side ? k.hq : k.hp;
    uint32_t* part = side ? part_q : part_p;
    if (forked && pass == 0) {
      rc = fork.begin(ctx); if (rc) return rc;
      rc = tmp_words(ctx, TMP_S_A, count * k.hw, &y); if (rc) return rc;
    }
    if (forked && pass == 1) {
      rc = fork.suspend(); if (rc) return rc;
      rc = tmp_words(ctx, TMP_S_A, count * k.hw, &y); if (rc) return rc;
    }
    rc = sc_modexp_shared(ctx, h.m1, h.exp_small, rho, k.nw, nullptr, y, count); if (rc) return rc;       // wide operand reduced mod p
    if (k.pairs && sc_mod_supports_sq(ctx, h.m1) == 1) {
      rc = sc_modexp_shared_sq(ctx, h.m1, h.m2, h.exp_p, y, k.hw, nullptr, part, count);
    } else {
      rc = sc_modexp_shared(ctx, h.m2, h.exp_p, y, k.hw, nullptr, part, count);
    }
    if (rc) return rc;
  }
  rc = fork.join(); if (rc) return rc;
  return sc_crt_combine(ctx, k.hp.m2, k.mod_n2, k.r_k, k.r_negk, k.r_mq, part_p, 2 * k.hw, part_q, 2 * k.hw, out, count);
}

}  // namespace

extern "C" {

int sc_paillier_key_create(sc_ctx* ctx, const uint32_t* n_hptr, int nwords, const uint32_t* p_hptr, const uint32_t* q_hptr, int pwords,
                           int flags, int* out_key) {
  if (!ctx || !n_hptr || nwords <= 0 || !out_key || ((p_hptr == nullptr) != (q_hptr == nullptr)) || (p_hptr && pwords <= 0))
    return fail(ctx, SC_ERR_ARG, "sc_paillier_key_create: bad argument");
  PaillierKey k;
  k.n.assign(n_hptr, n_hptr + nwords);
  k.nw = nwords;
  k.crt = !(flags & SC_KEY_NO_CRT);
  k.pairs = !(flags & SC_KEY_NO_PAIRS);
  const Big n2 = big_fit(big_mul(k.n, k.n), 2 * (size_t)nwords);
  int rc = sc_mod_create(ctx, k.n.data(), nwords, &k.mod_n); if (rc) return rc;
  rc = sc_mod_create(ctx, n2.data(), 2 * nwords, &k.mod_n2); if (rc) return rc;
  rc = reg_const(ctx, k.mod_n2, k.n, &k.cst_n); if (rc) return rc;
  rc = reg_exp(ctx, k.n, &k.exp_n); if (rc) return rc;
  if (p_hptr) {
    const Big p = big_trimmed(Big(p_hptr, p_hptr + pwords)), q = big_trimmed(Big(q_hptr, q_hptr + pwords));
    {
      Big pq = big_fit(big_trimmed(big_mul(p, q)), (size_t)nwords);
      if (big_trimmed(big_mul(p, q)).size() > (size_t)nwords || big_cmp(pq, k.n) != 0) return fail(ctx, SC_ERR_ARG, "sc_paillier_key_create: p * q != n");
    }
    k.secret = true;
    k.hw = (std::max(big_bits(p), big_bits(q)) + 31) / 32;
    const Big pm1 = big_sub_small(p, 1), qm1 = big_sub_small(q, 1);
    // lambda = (p - 1)(q - 1), mu = lambda^-1 mod N  (literal decryption)
    const Big lambda = big_trimmed(big_mul(pm1, qm1));
    rc = reg_exp(ctx, lambda, &k.exp_lambda); if (rc) return rc;
    const Big mu = big_modinv_odd(lambda, k.n);
    if (mu.empty()) return fail(ctx, SC_ERR_ARG, "sc_paillier_key_create: lambda is not invertible modulo n");
    rc = reg_const(ctx, k.mod_n, mu, &k.cst_mu); if (rc) return rc;
    for (int side = 0; side < 2; side++) {
      const Big& pr = side ? q : p; const Big& other = side ? p : q; const Big& prm1 = side ? qm1 : pm1;
      PaillierHalf& h = side ? k.hq : k.hp;
      const Big pr_w = big_fit(pr, (size_t)k.hw), pr2 = big_fit(big_mul(pr, pr), 2 * (size_t)k.hw);
      rc = sc_mod_create(ctx, pr_w.data(), k.hw, &h.m1); if (rc) return rc;
      rc = sc_mod_create(ctx, pr2.data(), 2 * k.hw, &h.m2); if (rc) return rc;
      rc = reg_exp(ctx, big_mod(other, big_trimmed(prm1)), &h.exp_small); if (rc) return rc;
      rc = reg_exp(ctx, pr, &h.exp_p); if (rc) return rc;
      rc = reg_exp(ctx, prm1, &h.exp_pm1); if (rc) return rc;
      // (N+1)^(p-1) = 1 + (p-1) N (mod p^2), so L_p of it is (p-1) q = -q (mod p):  h_p = (-q)^-1 = p - (q^-1 mod p)
      const Big qinv = big_modinv_odd(other, pr_w);
      if (qinv.empty()) return fail(ctx, SC_ERR_ARG, "sc_paillier_key_create: p and q are not coprime");
      Big hval = pr_w; big_sub(hval, qinv);
      rc = reg_const(ctx, h.m1, hval, &h.cst_h); if (rc) return rc;
    }
    // recombination constants: x = a_q + m_q ((a_p - a_q) m_q^-1 mod m_p)
    {
      const Big p2 = big_fit(big_mul(p, p), 2 * (size_t)k.hw), q2 = big_trimmed(big_mul(q, q));
      const Big kk = big_modinv_odd(q2, p2);
      if (kk.empty()) return fail(ctx, SC_ERR_ARG, "sc_paillier_key_create: q^2 is not invertible modulo p^2");
      Big neg = p2; big_sub(neg, kk);
      rc = reg_const(ctx, k.hp.m2, kk, &k.r_k); if (rc) return rc;
      rc = reg_const(ctx, k.hp.m2, neg, &k.r_negk); if (rc) return rc;
      rc = reg_const(ctx, k.mod_n2, q2, &k.r_mq); if (rc) return rc;
      const Big pw = big_fit(p, (size_t)k.hw);
      const Big kd = big_modinv_odd(q, pw);
      Big negd = pw; big_sub(negd, kd);
      rc = reg_const(ctx, k.hp.m1, kd, &k.d_k); if (rc) return rc;
      rc = reg_const(ctx, k.hp.m1, negd, &k.d_negk); if (rc) return rc;
      rc = reg_const(ctx, k.mod_n, q, &k.d_mq); if (rc) return rc;
    }
  }
  keys_of(ctx)->paillier.push_back(k);
  *out_key = (int)keys_of(ctx)->paillier.size() - 1;
  // the key holder's CRT runs its half-size exponentiations under the one-lane policy: measure that policy's constants now (once
  // per device and process), not inside the first step
  if (k.secret && k.crt && ctx->onelane_mode == 1 && k.hw <= 32) (void)onelane_cal(ctx);
  return SC_OK;
}

int sc_paillier_key_mods(sc_ctx* ctx, int key, int* out_mod_n, int* out_mod_n2) {
  const PaillierKey* k = paillier_key(ctx, key);
  if (!k) return fail(ctx, SC_ERR_ARG, "sc_paillier_key_mods: bad key");
  if (out_mod_n) *out_mod_n = k->mod_n;
  if (out_mod_n2) *out_mod_n2 = k->mod_n2;
  return SC_OK;
}

int sc_paillier_encrypt(sc_ctx* ctx, int key, const uint32_t* m, int m_words, int negate, uint32_t* out, uint64_t count) {
  const PaillierKey* k = paillier_key(ctx, key);
  if (!k) return fail(ctx, SC_ERR_ARG, "sc_paillier_encrypt: bad key");
  return negate ? sc_paillier_encrypt_raw_neg(ctx, k->mod_n2, k->cst_n, m, m_words, out, count)
                : sc_paillier_encrypt_raw(ctx, k->mod_n2, k->cst_n, m, m_words, out, count);
}

int sc_paillier_randomize(sc_ctx* ctx, int key, const uint32_t* c, const uint32_t* rho, uint32_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  const PaillierKey* kp = paillier_key(ctx, key);
  if (!kp || !rho || !out) return fail(ctx, SC_ERR_ARG, "sc_paillier_randomize: bad argument");
  const PaillierKey k = *kp;    // (the key table may grow while programs are built: work on a copy)
  if (k.secret && k.crt) {
    if (!c) return paillier_crt_pow_n(ctx, k, rho, out, count);
    uint32_t* rn;
    int rc = tmp_words(ctx, TMP_S_D, count * 2 * k.nw, &rn); if (rc) return rc;
    rc = paillier_crt_pow_n(ctx, k, rho, rn, count); if (rc) return rc;
    return sc_modmul(ctx, k.mod_n2, c, 2 * k.nw, rn, 2 * k.nw, out, count);
  }
  if (k.pairs && sc_mod_supports_sq(ctx, k.mod_n) == 1)
    return sc_modexp_shared_sq(ctx, k.mod_n, k.mod_n2, k.exp_n, rho, k.nw, c, out, count);     // arithmetic modulo N only
  return sc_modexp_shared(ctx, k.mod_n2, k.exp_n, rho, k.nw, c, out, count);
}

int sc_paillier_decrypt(sc_ctx* ctx, int key, const uint32_t* c, uint32_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  const PaillierKey* kp = paillier_key(ctx, key);
  if (!kp || !c || !out) return fail(ctx, SC_ERR_ARG, "sc_paillier_decrypt: bad argument");
  if (!kp->secret) return fail(ctx, SC_ERR_ARG, "sc_paillier_decrypt: this key has no secret part");
  const PaillierKey k = *kp;
  int rc;
  if (!k.crt) {
    uint32_t* x;
    rc = tmp_words(ctx, TMP_S_A, count * 2 * k.nw, &x); if (rc) return rc;
    if (k.pairs && sc_mod_supports_sq(ctx, k.mod_n) == 1) rc = sc_modexp_shared_sq(ctx, k.mod_n, k.mod_n2, k.exp_lambda, c, 2 * k.nw, nullptr, x, count);
    else rc = sc_modexp_shared(ctx, k.mod_n2, k.exp_lambda, c, 2 * k.nw, nullptr, x, count);
    if (rc) return rc;
    return sc_paillier_l_mul(ctx, k.mod_n, k.cst_mu, x, 2 * k.nw, out, count);
  }
  // m_p = L_p(c^(p-1) mod p^2) h_p mod p, likewise q, recombined modulo N
  uint32_t *x, *m_p, *m_q;
  rc = tmp_words(ctx, TMP_S_A, count * 2 * k.hw, &x); if (rc) return rc;
  rc = tmp_words(ctx, TMP_S_B, count * k.hw, &m_p); if (rc) return rc;
  rc = tmp_words(ctx, TMP_S_C, count * k.hw, &m_q); if (rc) return rc;
  AuxFork fork;
  const bool forked = small_enough_to_fork(ctx, ctx->mods[k.hp.m1], count);
  for (int pass = 0; pass < 2; pass++) {
    const int side = forked ? 1 - pass : pass;       // forked: c^(q-1) mod q^2 first, on the second stream; c^(p-1) mod p^2 beside it
    const PaillierHalf& h = side ? k.hq : k.hp;
    if (forked && pass == 0) {
      rc = fork.begin(ctx); if (rc) return rc;
      rc = tmp_words(ctx, TMP_S_A, count * 2 * k.hw, &x); if (rc) return rc;
    }
    if (forked && pass == 1) {
      rc = fork.suspend(); if (rc) return rc;
      rc = tmp_words(ctx, TMP_S_A, count * 2 * k.hw, &x); if (rc) return rc;
    }
    if (k.pairs && sc_mod_supports_sq(ctx, h.m1) == 1) rc = sc_modexp_shared_sq(ctx, h.m1, h.m2, h.exp_pm1, c, 2 * k.nw, nullptr, x, count);
    else rc = sc_modexp_shared(ctx, h.m2, h.exp_pm1, c, 2 * k.nw, nullptr, x, count);
    if (rc) return rc;
    rc = sc_paillier_l_mul(ctx, h.m1, h.cst_h, x, 2 * k.hw, side ? m_q : m_p, count); if (rc) return rc;
  }
  rc = fork.join(); if (rc) return rc;
  return sc_crt_combine(ctx, k.hp.m1, k.mod_n, k.d_k, k.d_negk, k.d_mq, m_p, k.hw, m_q, k.hw, out, count);
}

int sc_dgk_key_create(sc_ctx* ctx, const uint32_t* n_hptr, const uint32_t* g_hptr, const uint32_t* h_hptr, int nwords, const uint32_t* u_hptr,
                      int uwords, int t_bits, const uint32_t* p_hptr, const uint32_t* q_hptr, int pwords, const uint32_t* vp_hptr,
                      const uint32_t* vq_hptr, int vwords, int randomizer_bits, int window, int flags, sc_ctx* table_src_ctx,
                      int table_src_key, int* out_key) {
  if (!ctx || !n_hptr || !g_hptr || !h_hptr || nwords <= 0 || !u_hptr || uwords <= 0 || randomizer_bits <= 0 || window < 1 || window > 24 || !out_key)
    return fail(ctx, SC_ERR_ARG, "sc_dgk_key_create: bad argument");
  const bool secret = p_hptr != nullptr;
  if (secret && (!q_hptr || !vp_hptr || !vq_hptr || pwords <= 0 || vwords <= 0)) return fail(ctx, SC_ERR_ARG, "sc_dgk_key_create: incomplete secret key");
  DgkKey k;
  k.n.assign(n_hptr, n_hptr + nwords); k.g.assign(g_hptr, g_hptr + nwords); k.h.assign(h_hptr, h_hptr + nwords);
  k.u.assign(u_hptr, u_hptr + uwords);
  k.nw = nwords; k.rbits = randomizer_bits; k.window = window; k.t = t_bits; k.secret = secret; k.crt = !(flags & SC_KEY_NO_CRT);
  const DgkKey* src = nullptr;
  if (table_src_ctx) {
    src = dgk_key(table_src_ctx, table_src_key);
    if (!src || src->n != k.n || src->h != k.h || src->window != window || src->rbits != randomizer_bits || src->secret != secret || src->crt != k.crt)
      return fail(ctx, SC_ERR_ARG, "sc_dgk_key_create: tables can only be shared between keys with the same modulus, h, window and randomizer width");
  }
  int rc = sc_mod_create(ctx, k.n.data(), nwords, &k.mod_n); if (rc) return rc;
  rc = reg_const(ctx, k.mod_n, k.g, &k.cst_g); if (rc) return rc;
  const Big ginv = big_modinv_odd(k.g, k.n);
  if (ginv.empty()) return fail(ctx, SC_ERR_ARG, "sc_dgk_key_create: g is not invertible modulo n");
  rc = reg_const(ctx, k.mod_n, ginv, &k.cst_ginv); if (rc) return rc;
  auto table = [&](int mod, const Big& base, int exp_bits, int w, int src_fbt, int* out_fbt) -> int {
    if (src) return sc_fbt_import(ctx, mod, table_src_ctx, src_fbt, out_fbt);
    const Big b = big_fit(base, (size_t)ctx->mods[mod].nwords);
    return sc_fbt_create(ctx, mod, b.data(), exp_bits, w, out_fbt);
  };
  if (!secret || !k.crt) {     // Alice -- and a key holder told not to use CRT -- randomize modulo n with the table for h
    rc = table(k.mod_n, k.h, randomizer_bits, window, src ? src->fbt_h : -1, &k.fbt_h); if (rc) return rc;
  }
  if (secret) {
    const Big p = big_trimmed(Big(p_hptr, p_hptr + pwords)), q = big_trimmed(Big(q_hptr, q_hptr + pwords));
    const Big vp = big_trimmed(Big(vp_hptr, vp_hptr + vwords)), vq = big_trimmed(Big(vq_hptr, vq_hptr + vwords));
    // a secret key that does not belong to the public one would silently give randomizers that are not h^r mod n (the CRT halves
    // reduce r modulo v_p, v_q) and zero tests that test nothing: check p q = n and that h has order dividing v_p (v_q) modulo p (q)
    {
      const Big pq = big_trimmed(big_mul(p, q));
      if (pq.size() > (size_t)nwords || big_cmp(big_fit(pq, (size_t)nwords), k.n) != 0) return fail(ctx, SC_ERR_ARG, "sc_dgk_key_create: p * q != n");
      if (!big_is_one(big_powmod(k.h, vp, p)) || !big_is_one(big_powmod(k.h, vq, q)))
        return fail(ctx, SC_ERR_ARG, "sc_dgk_key_create: h^v_p mod p and h^v_q mod q must be 1 (the secret key does not match h)");
    }
    rc = sc_mod_create(ctx, p.data(), (int)p.size(), &k.mod_p); if (rc) return rc;
    rc = reg_exp(ctx, vp, &k.exp_vp); if (rc) return rc;
    if (k.crt) {
      Big one(1, 1);
      rc = reg_exp(ctx, one, &k.exp_one); if (rc) return rc;
      for (int side = 0; side < 2; side++) {
        const Big& pr = side ? q : p; const Big& v = side ? vq : vp;
        DgkHalf& h = side ? k.hq : k.hp;
        rc = sc_mod_create(ctx, pr.data(), (int)pr.size(), &h.m); if (rc) return rc;
        if (!(v[0] & 1)) return fail(ctx, SC_ERR_ARG, "sc_dgk_key_create: v_p, v_q must be odd (they are primes)");
        rc = sc_mod_create(ctx, v.data(), (int)v.size(), &h.m_v); if (rc) return rc;
        // h has order v_p modulo p: h^r mod p = (h mod p)^(r mod v_p) -- a t-bit exponent and a half-size modulus
        rc = table(h.m, big_mod(k.h, pr), big_bits(v), std::min(window, 20), src ? (side ? src->hq.fbt : src->hp.fbt) : -1, &h.fbt); if (rc) return rc;
        rc = reg_const(ctx, h.m, big_mod(k.g, pr), &h.c_g); if (rc) return rc;
      }
      const Big kk = big_modinv_odd(q, p);
      if (kk.empty()) return fail(ctx, SC_ERR_ARG, "sc_dgk_key_create: p and q are not coprime");
      Big neg = big_fit(p, kk.size()); big_sub(neg, kk);
      rc = reg_const(ctx, k.hp.m, kk, &k.c_k); if (rc) return rc;
      rc = reg_const(ctx, k.hp.m, neg, &k.c_negk); if (rc) return rc;
      rc = reg_const(ctx, k.mod_n, q, &k.c_mq); if (rc) return rc;
    }
  }
  keys_of(ctx)->dgk.push_back(k);
  *out_key = (int)keys_of(ctx)->dgk.size() - 1;
  if (secret && ctx->onelane_mode == 1 && pwords <= 32) (void)onelane_cal(ctx);      // the zero tests run under the one-lane policy
  return SC_OK;
}

int sc_dgk_key_info(sc_ctx* ctx, int key, int* out_mod_n, int* out_mod_p, uint64_t* out_table_bytes) {
  const DgkKey* k = dgk_key(ctx, key);
  if (!k) return fail(ctx, SC_ERR_ARG, "sc_dgk_key_info: bad key");
  if (out_mod_n) *out_mod_n = k->mod_n;
  if (out_mod_p) *out_mod_p = k->mod_p;
  if (out_table_bytes) {
    uint64_t total = 0, b = 0;
    for (int f : {k->fbt_h, k->hp.fbt, k->hq.fbt})
      if (f >= 0 && sc_fbt_bytes(ctx, f, &b) == SC_OK) total += b;
    *out_table_bytes = total;
  }
  return SC_OK;
}

// h^r [* c] mod n; `bits` (nullable): additionally times g where bits[i] != 0 (the unrandomized encryption of a bit)
static int dgk_randomize_impl(sc_ctx* ctx, const DgkKey& k, const uint32_t* c, const uint8_t* bits, const uint32_t* r, int ewords, uint32_t* out, uint64_t count) {
  int rc;
  if (!k.secret || !k.crt) {
    rc = sc_fixedbase_pow(ctx, k.fbt_h, r, ewords, c, out, count); if (rc) return rc;
  } else if (!c) {
    // Key holder, CRT (SC/keyholder.py:106-108): h^r mod q, then h^r mod p with the first half of the recombination in the same
    // launch -- t = (a_p - a_q) q^-1 mod p leaves the fixed-base program instead of a_p --, then a_q + q t: five launches over the
    // l + 1 values of every comparison instead of seven (each pass over the batch costs a load, a canonical store and the launch's
    // ramp, whatever it multiplies).  The factor g^bit of the unrandomized encryption enters the two halves -- times g mod q, g mod p
    // where the bit is set, two half-size products -- instead of the recombined value (one full-size product).  Same residues.
    uint32_t *r_red, *tq, *a_q;
    const int pw = ctx->mods[k.hp.m].nwords, qw = ctx->mods[k.hq.m].nwords;
    const int vpw = ctx->mods[k.hp.m_v].nwords, vqw = ctx->mods[k.hq.m_v].nwords;
    rc = tmp_words(ctx, TMP_S_A, count * std::max(vpw, vqw), &r_red); if (rc) return rc;
    rc = tmp_words(ctx, TMP_S_B, count * pw, &tq); if (rc) return rc;
    rc = tmp_words(ctx, TMP_S_C, count * qw, &a_q); if (rc) return rc;
    rc = sc_modexp_shared(ctx, k.hq.m_v, k.exp_one, r, ewords, nullptr, r_red, count); if (rc) return rc;                 // r mod v_q
    if (!bits) {
      rc = sc_fixedbase_pow(ctx, k.hq.fbt, r_red, vqw, nullptr, a_q, count); if (rc) return rc;                             // a_q = h^r mod q
    } else {
      const Fbt f = ctx->fbts[k.hq.fbt];
      const Mod mq = ctx->mods[k.hq.m];
      std::string key = "fbcrt0:" + std::to_string(k.hq.fbt) + ":" + std::to_string(k.hq.c_g);
      auto it = ctx->progs.find(key);
      if (it == ctx->progs.end()) {
        Builder bd; const int cg = bd.use_const(k.hq.c_g);
        bd.loadt_fbt(0, 0, f.window, 0);
        for (int j = 1; j < f.nwin; j++) bd.mul_fbt(0, j * f.window, f.window, j);
        bd.emit(OP_MUL, AK_CONSTSEL, 0, 2, (uint32_t)1 | ((uint32_t)cg << 8)); bd.muls++;   // times g mod q where the bit is set
        bd.redc(); bd.storew(1); bd.end();                                                   // a_q = g^bit h^r mod q
        Prog pr; rc = finalize_prog(ctx, mq, bd, &pr); if (rc) return rc;
        it = ctx->progs.emplace(key, pr).first;
      }
      VmExt ex[3] = {mk_ext(r_red, vqw, vqw), mk_ext(a_q, qw, qw), mk_ext(bits, 0, 0)};
      rc = run_vm(ctx, k.hq.m, it->second, ex, 3, count, f.d_rows); if (rc) return rc;
    }
    rc = sc_modexp_shared(ctx, k.hp.m_v, k.exp_one, r, ewords, nullptr, r_red, count); if (rc) return rc;                 // r mod v_p
    {
      const Fbt f = ctx->fbts[k.hp.fbt];
      const Mod mp = ctx->mods[k.hp.m];
      std::string key = "fbcrt1:" + std::to_string(k.hp.fbt) + ":" + std::to_string(k.c_k) + ":" + std::to_string(k.c_negk) + ":" + std::to_string(qw) + (bits ? ":g" + std::to_string(k.hp.c_g) : "");
      auto it = ctx->progs.find(key);
      if (it == ctx->progs.end()) {
        Builder bd; const int ck = bd.use_const(k.c_k), cn = bd.use_const(k.c_negk);
        int kc = -1;
        if (qw > mp.nwords) { int cid; rc = get_const_kred(ctx, k.hp.m, &cid); if (rc) return rc; kc = bd.use_const(cid); }
        bd.loadt_fbt(0, 0, f.window, 0);
        for (int j = 1; j < f.nwin; j++) bd.mul_fbt(0, j * f.window, f.window, j);
        if (bits) { const int cg = bd.use_const(k.hp.c_g); bd.emit(OP_MUL, AK_CONSTSEL, 0, 3, (uint32_t)1 | ((uint32_t)cg << 8)); bd.muls++; }   // times g mod p where the bit is set
        bd.redc();                                                                         // a_p = [g^bit] h^r mod p
        bd.mul_const(ck); bd.stt(0);                                                       // a_p k,  k = q^-1 mod p
        if (qw > mp.nwords) emit_load_reduced(ctx, mp, bd, 1, qw, kc); else bd.loadw(1, 0, 0, qw);
        bd.mul_const(cn); bd.addt(0);                                                      // + a_q (p - k)
        bd.storew(2); bd.end();
        Prog pr; rc = finalize_prog(ctx, mp, bd, &pr); if (rc) return rc;
        it = ctx->progs.emplace(key, pr).first;
      }
      VmExt ex[4] = {mk_ext(r_red, vpw, vpw), mk_ext(a_q, qw, qw), mk_ext(tq, mp.nwords, mp.nwords), mk_ext(bits, 0, 0)};
      rc = run_vm(ctx, k.hp.m, it->second, ex, bits ? 4 : 3, count, f.d_rows); if (rc) return rc;
    }
    {
      const Mod mn = ctx->mods[k.mod_n];
      std::string key = "fbcrt2:" + std::to_string(k.mod_n) + ":" + std::to_string(k.c_mq) + ":" + std::to_string(pw) + ":" + std::to_string(qw);
      auto it = ctx->progs.find(key);
      if (it == ctx->progs.end()) {
        Builder bd; const int cm = bd.use_const(k.c_mq);
        bd.loadw(0, 0, 0, pw); bd.mul_const(cm);                                           // q t  (< p q: exact)
        bd.addw(1, 0, 0, qw);                                                              // + a_q = [g^bit] h^r mod n
        bd.storew(2); bd.end();
        Prog pr; rc = finalize_prog(ctx, mn, bd, &pr); if (rc) return rc;
        it = ctx->progs.emplace(key, pr).first;
      }
      VmExt ex[3] = {mk_ext(tq, pw, pw), mk_ext(a_q, qw, qw), mk_ext(out, mn.nwords, mn.nwords)};
      return run_vm(ctx, k.mod_n, it->second, ex, 3, count);
    }
  } else {
    uint32_t *r_red, *part_p, *part_q;
    const int pw = ctx->mods[k.hp.m].nwords, qw = ctx->mods[k.hq.m].nwords;
    const int vpw = ctx->mods[k.hp.m_v].nwords, vqw = ctx->mods[k.hq.m_v].nwords;
    rc = tmp_words(ctx, TMP_S_A, count * std::max(vpw, vqw), &r_red); if (rc) return rc;
    rc = tmp_words(ctx, TMP_S_B, count * pw, &part_p); if (rc) return rc;
    rc = tmp_words(ctx, TMP_S_C, count * qw, &part_q); if (rc) return rc;
    for (int side = 0; side < 2; side++) {
      const DgkHalf& h = side ? k.hq : k.hp;
      rc = sc_modexp_shared(ctx, h.m_v, k.exp_one, r, ewords, nullptr, r_red, count); if (rc) return rc;            // r mod v (wide operand reduced)
      rc = sc_fixedbase_pow(ctx, h.fbt, r_red, side ? vqw : vpw, nullptr, side ? part_q : part_p, count); if (rc) return rc;
    }
    uint32_t* hr = out;
    if (c) { rc = tmp_words(ctx, TMP_S_D, count * k.nw, &hr); if (rc) return rc; }
    rc = sc_crt_combine(ctx, k.hp.m, k.mod_n, k.c_k, k.c_negk, k.c_mq, part_p, pw, part_q, qw, hr, count); if (rc) return rc;
    if (c) { rc = sc_modmul(ctx, k.mod_n, c, k.nw, hr, k.nw, out, count); if (rc) return rc; }
  }
  if (bits) return sc_modmul_const_sel(ctx, k.mod_n, out, -1, k.cst_g, bits, out, count);
  return SC_OK;
}

int sc_dgk_randomize(sc_ctx* ctx, int key, const uint32_t* c, const uint32_t* r, int ewords, uint32_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  const DgkKey* kp = dgk_key(ctx, key);
  if (!kp || !r || ewords <= 0 || !out) return fail(ctx, SC_ERR_ARG, "sc_dgk_randomize: bad argument");
  const DgkKey k = *kp;
  return dgk_randomize_impl(ctx, k, c, nullptr, r, ewords, out, count);
}

int sc_dgk_encrypt_bits_randomized(sc_ctx* ctx, int key, const uint8_t* bits, const uint32_t* r, int ewords, uint32_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  const DgkKey* kp = dgk_key(ctx, key);
  if (!kp || !bits || !r || ewords <= 0 || !out) return fail(ctx, SC_ERR_ARG, "sc_dgk_encrypt_bits_randomized: bad argument");
  const DgkKey k = *kp;
  return dgk_randomize_impl(ctx, k, nullptr, bits, r, ewords, out, count);
}

int sc_dgk_is_zero(sc_ctx* ctx, int key, const uint32_t* c, uint8_t* flags, uint64_t count) {
  const DgkKey* k = dgk_key(ctx, key);
  if (!k || !k->secret) return fail(ctx, SC_ERR_ARG, "sc_dgk_is_zero: needs the secret key");
  return sc_modexp_shared_isone(ctx, k->mod_p, k->exp_vp, c, k->nw, flags, count);
}

int sc_dgk_any_zero(sc_ctx* ctx, int key, const uint32_t* c, int planes, uint64_t inner, uint64_t* any_flags) {
  const DgkKey* k = dgk_key(ctx, key);
  if (!k || !k->secret || planes <= 0) return fail(ctx, SC_ERR_ARG, "sc_dgk_any_zero: needs the secret key");
  return sc_modexp_shared_isone_any(ctx, k->mod_p, k->exp_vp, c, k->nw, inner, any_flags, (uint64_t)planes * inner);
}

// ---- protocol steps ----------------------------------------------------------------------------------------------------------
int sc_initiator_step1(sc_ctx* ctx, int paillier_key_id, int l, const uint32_t* x_enc, const uint32_t* y_enc, const uint32_t* r,
                       const uint32_t* rho_z, int flags, uint32_t* z_out, uint64_t* alpha, uint64_t* alpha_tilde, uint64_t* rsmall,
                       uint32_t* rshift, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  const PaillierKey* kp = paillier_key(ctx, paillier_key_id);
  if (!kp || !x_enc || !y_enc || !r || !z_out || !alpha || !alpha_tilde || !rsmall || !rshift || l <= 0 || l > 64)
    return fail(ctx, SC_ERR_ARG, "sc_initiator_step1: bad argument");
  const PaillierKey k = *kp;
  if (l + 3 >= big_bits(k.n) - 1) return fail(ctx, SC_ERR_ARG, "sc_initiator_step1: 2^(l+2) must be below N / 2 (SC/initiator.py:249)");
  uint32_t *m1, *xinv, *t;
  int rc = tmp_words(ctx, TMP_S_E, count * (k.nw + 1), &m1); if (rc) return rc;
  rc = tmp_words(ctx, TMP_S_F, count * 2 * k.nw, &xinv); if (rc) return rc;
  rc = sc_plain_alice(ctx, r, k.n.data(), k.nw, l, count, m1, alpha, alpha_tilde, rsmall, rshift); if (rc) return rc;
  int64_t bad = -1;
  rc = sc_modinv(ctx, k.mod_n2, x_enc, xinv, count, &bad);
  if (rc) return rc;
  // one launch for the rest of step 1: [[y]] [[x]]^-1 [[2^l + r]] (SC/initiator.py:254-256), times the finished randomizer when it
  // was computed ahead of time.  Same products as the separate modmul / encrypt launches, identical residues.
  const bool ready = rho_z && (flags & SC_STEP_RANDOMIZERS_READY);
  const bool fused_out = !rho_z || ready;                 // else the product goes to a temporary and .randomize() (:109) finishes
  uint32_t* dst = z_out;
  if (!fused_out) { rc = tmp_words(ctx, TMP_S_G, count * 2 * k.nw, &t); if (rc) return rc; dst = t; }
  const int w2 = ctx->mods[k.mod_n2].nwords;
  std::string key = "step1:" + std::to_string(k.mod_n2) + ":" + std::to_string(k.cst_n) + (ready ? ":r" : "");
  auto it = ctx->progs.find(key);
  if (it == ctx->progs.end()) {
    Builder bd; const int cn = bd.use_const(k.cst_n);
    bd.loadw(0); bd.mul_const(0); bd.mul_extw(1);        // [[y]] [[x]]^-1
    bd.mul_const(0); bd.stt(0);                          // ... in Montgomery form
    bd.loadw(2, 0, 0, k.nw + 1); bd.mul_const(cn); bd.emit(OP_ADD1);   // [[2^l + r]] = 1 + (2^l + r) N  (mod N^2)
    bd.mul_tbl(0);
    if (ready) { bd.mul_const(0); bd.mul_extw(4); }      // times rho_z^N
    bd.storew(3); bd.end();
    Prog p; rc = finalize_prog(ctx, ctx->mods[k.mod_n2], bd, &p); if (rc) return rc;
    it = ctx->progs.emplace(key, p).first;
  }
  VmExt ex[5] = {mk_ext(y_enc, w2, w2), mk_ext(xinv, w2, w2), mk_ext(m1, k.nw + 1, k.nw + 1),
                 mk_ext(dst, w2, w2), mk_ext(ready ? rho_z : nullptr, w2, w2)};
  rc = run_vm(ctx, k.mod_n2, it->second, ex, 5, count); if (rc) return rc;
  if (fused_out) return SC_OK;
  return sc_paillier_randomize(ctx, paillier_key_id, t, rho_z, z_out, count);                                         // .randomize() (:109)
}

int sc_initiator_step4i(sc_ctx* ctx, int dgk_key_id, int l, const uint32_t* c_in, const uint32_t* rhos, int rho_words, const int64_t* permutation,
                        const uint32_t* r_rand, int r_words, int flags, uint32_t* c_out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  const DgkKey* kp = dgk_key(ctx, dgk_key_id);
  if (!kp || l <= 0 || l > 64 || !c_in || !rhos || rho_words <= 0 || !c_out || (r_rand && r_words <= 0))
    return fail(ctx, SC_ERR_ARG, "sc_initiator_step4i: bad argument");
  const DgkKey k = *kp;
  const bool ready = r_rand && (flags & SC_STEP_RANDOMIZERS_READY);     // r_rand holds h^r itself ([l+1][count][nwords]), computed ahead
  if (r_rand && !ready && k.fbt_h < 0) return fail(ctx, SC_ERR_ARG, "sc_initiator_step4i: this key has no table for h modulo n (key holder with CRT)");
  const uint64_t planes = (uint64_t)l + 1, items = planes * count;
  const int ubits = big_bits(big_sub_small(k.u, 1));
  const int fbt = (r_rand && !ready) ? k.fbt_h : -1;
  const uint32_t* e2 = ready ? nullptr : r_rand;
  const uint32_t* premul = ready ? r_rand : nullptr;
  if (!permutation) return modexp_var_impl(ctx, k.mod_n, c_in, rhos, rho_words, ubits, fbt, e2, r_words, nullptr, c_out, items, premul);
  if (c_in == c_out) return fail(ctx, SC_ERR_ARG, "sc_initiator_step4i: a shuffled store cannot work in place");
  // the store of the blinding launch finds each item's output plane in the permutation itself (OP_STOREW, sc_vm.h): rows that
  // are not permutations act as the identity, so every output row is written; no destination array, no launch to build one
  // (round 3's 26-us k_perm_to_dest waited up to 18 ms for a wave slot beside the other shard's chip-filling launches)
  return modexp_var_impl(ctx, k.mod_n, c_in, rhos, rho_words, ubits, fbt, e2, r_words, nullptr, c_out, items, premul, permutation, (uint32_t)planes);
}

int sc_initiator_step4(sc_ctx* ctx, int dgk_key_id, int l, const uint32_t* d_enc, const uint32_t* beta_enc, const uint64_t* alpha,
                       const uint64_t* alpha_tilde, const uint64_t* rsmall, const uint64_t* delta_a, const uint32_t* rhos, int rho_words,
                       const int64_t* permutation, const uint32_t* r_rand, int r_words, int flags, uint32_t* c_unblinded_out, uint32_t* c_out,
                       uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  const DgkKey* kp = dgk_key(ctx, dgk_key_id);
  if (!kp || l <= 0 || l > 64 || !d_enc || !beta_enc || !alpha || !alpha_tilde || !rsmall || !delta_a || !c_out ||
      (rhos && rho_words <= 0) || (r_rand && (r_words <= 0 || !rhos)))
    return fail(ctx, SC_ERR_ARG, "sc_initiator_step4: bad argument");
  const DgkKey k = *kp;
  const uint64_t planes = (uint64_t)l + 1, items = planes * count;
  const size_t row = (size_t)k.nw;
  // one inversion pass over [d], [beta_0] .. [beta_{l-1}]: in place when they are the planes of one array, else joined first
  const uint32_t* joined = d_enc;
  uint32_t* inv;
  int rc = tmp_words(ctx, TMP_S_E, items * row, &inv); if (rc) return rc;
  if (beta_enc != d_enc + count * row) {
    uint32_t* j;
    rc = tmp_words(ctx, TMP_S_F, items * row, &j); if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(j, d_enc, count * row * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(j + count * row, beta_enc, (size_t)l * count * row * 4, hipMemcpyDeviceToDevice, ctx->stream));
    joined = j;
  }
  int64_t bad = -1;
  rc = sc_modinv(ctx, k.mod_n, joined, inv, items, &bad);
  if (rc) return rc;
  uint32_t* c_h = c_unblinded_out;
  if (!rhos) c_h = c_out;                                    // steps 4c-4h only
  else if (!c_h) { rc = tmp_words(ctx, TMP_S_G, items * row, &c_h); if (rc) return rc; }
  rc = sc_dgk_step4(ctx, k.mod_n, k.cst_g, k.cst_ginv, l, beta_enc, inv + count * row, d_enc, inv, alpha, alpha_tilde, rsmall, delta_a, c_h, count);
  if (rc || !rhos) return rc;
  return sc_initiator_step4i(ctx, dgk_key_id, l, c_h, rhos, rho_words, permutation, r_rand, r_words, flags, c_out, count);
}

int sc_keyholder_step2_4b(sc_ctx* ctx, int paillier_key_id, int dgk_key_id, int l, const uint32_t* z_enc, const uint32_t* r_rand, int r_words,
                          int flags, uint32_t* z_out, uint64_t* beta, uint64_t* dbit, uint32_t* zeta1, uint32_t* zeta2, uint32_t* d_beta_out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  const PaillierKey* pk = paillier_key(ctx, paillier_key_id);
  const DgkKey* dk = dgk_key(ctx, dgk_key_id);
  if (!pk || !dk || !pk->secret || l <= 0 || l > 64 || !z_enc || !z_out || !beta || !dbit || !zeta1 || !zeta2 || !d_beta_out || (r_rand && r_words <= 0))
    return fail(ctx, SC_ERR_ARG, "sc_keyholder_step2_4b: bad argument");
  if (big_bits(dk->u) <= l + 2) return fail(ctx, SC_ERR_ARG, "sc_keyholder_step2_4b: u must exceed 2^(l+2) (SC/keyholder.py:212)");
  const PaillierKey p = *pk; const DgkKey d = *dk;
  int rc = sc_paillier_decrypt(ctx, paillier_key_id, z_enc, z_out, count); if (rc) return rc;
  const uint64_t items = ((uint64_t)l + 1) * count;
  uint8_t* bits;
  rc = tmp_words(ctx, TMP_S_I, items, &bits); if (rc) return rc;
  rc = plain_bob_impl(ctx, z_out, p.n.data(), p.nw, l, count, beta, dbit, zeta1, zeta2, bits); if (rc) return rc;   // + the bits of steps 4a / 4b
  if (r_rand && (flags & SC_STEP_RANDOMIZERS_READY))        // r_rand holds h^r itself: g^bit * h^r is one selected-constant product
    return sc_modmul_const_sel(ctx, d.mod_n, r_rand, -1, d.cst_g, bits, d_beta_out, items);
  if (r_rand) return dgk_randomize_impl(ctx, d, nullptr, bits, r_rand, r_words, d_beta_out, items);
  // unrandomized: g^bit -- the residue 1 times (1 or g), chosen per item inside the launch
  std::string key = "bits1:" + std::to_string(d.mod_n) + ":" + std::to_string(d.cst_g);
  auto it = ctx->progs.find(key);
  if (it == ctx->progs.end()) {
    Builder bd; const int cg = bd.use_const(d.cst_g);
    bd.loadt_const(1);
    bd.emit(OP_MUL, AK_CONSTSEL, 0, 1, (uint32_t)1 | ((uint32_t)cg << 8)); bd.muls++;
    bd.redc(); bd.storew(0); bd.end();
    Prog pr; rc = finalize_prog(ctx, ctx->mods[d.mod_n], bd, &pr); if (rc) return rc;
    it = ctx->progs.emplace(key, pr).first;
  }
  VmExt ex[2] = {mk_ext(d_beta_out, d.nw, d.nw), mk_ext(bits, 0, 0)};
  return run_vm(ctx, d.mod_n, it->second, ex, 2, items);
}

int sc_keyholder_step4j_5(sc_ctx* ctx, int paillier_key_id, int dgk_key_id, int l, const uint32_t* c_enc, const uint32_t* zeta1,
                          const uint32_t* zeta2, const uint32_t* rho3, int flags, uint64_t* delta_b_out, uint32_t* out3, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  const PaillierKey* pk = paillier_key(ctx, paillier_key_id);
  const DgkKey* dk = dgk_key(ctx, dgk_key_id);
  if (!pk || !dk || !pk->secret || !dk->secret || l <= 0 || l > 64 || !c_enc || !zeta1 || !zeta2 || !delta_b_out || !out3)
    return fail(ctx, SC_ERR_ARG, "sc_keyholder_step4j_5: bad argument");
  const PaillierKey p = *pk;
  const DgkKey d = *dk;
  // step 4j: the l + 1 zero tests of a comparison OR their verdicts into a context-owned accumulator that is zero at rest -- the
  // launch that encrypts delta_B below reads it, hands it to delta_b_out and resets it (OP_TAKEFLAG): no clearing launch, which as a
  // runtime fill kernel waited up to 14 ms for a wave slot beside another context's chip-filling launch
  uint64_t* acc;
  int rc = zero_kept_flags(ctx, count, &acc); if (rc) return rc;
  struct Dirty {      // an error between the accumulation and the launch that resets the accumulator must not leave verdicts behind
    sc_ctx* c; bool armed = true;
    ~Dirty() { if (armed) drop_zero_kept_flags(c); }
  } dirty{ctx};
  rc = modexp_shared_impl(ctx, d.mod_p, d.exp_vp, c_enc, d.nw, nullptr, nullptr, nullptr, ((uint64_t)l + 1) * count, acc, count, true); if (rc) return rc;
  // step 5: three encryptions into the row blocks of one array (no copies of zeta_1 / zeta_2 into a joined plaintext array)
  uint32_t* enc = out3;
  if (rho3) { rc = tmp_words(ctx, TMP_S_F, 3 * count * 2 * p.nw, &enc); if (rc) return rc; }
  const size_t blk = (size_t)count * 2 * p.nw;
  rc = sc_paillier_encrypt_raw(ctx, p.mod_n2, p.cst_n, zeta1, p.nw, enc, count); if (rc) return rc;
  rc = sc_paillier_encrypt_raw(ctx, p.mod_n2, p.cst_n, zeta2, p.nw, enc + blk, count); if (rc) return rc;
  {
    std::string key = "encflag:" + std::to_string(p.mod_n2) + ":" + std::to_string(p.cst_n);
    auto it = ctx->progs.find(key);
    if (it == ctx->progs.end()) {
      Builder bd; const int c = bd.use_const(p.cst_n);
      bd.emit(OP_TAKEFLAG, 0, 0, 0, 1); bd.mul_const(c);       // delta_B N
      bd.emit(OP_ADD1); bd.storew(2); bd.end();
      Prog pr; rc = finalize_prog(ctx, ctx->mods[p.mod_n2], bd, &pr); if (rc) return rc;
      it = ctx->progs.emplace(key, pr).first;
    }
    const int w2 = 2 * p.nw;
    VmExt ex[3] = {mk_ext(acc, 2, 2), mk_ext(delta_b_out, 2, 2), mk_ext(enc + 2 * blk, w2, w2)};
    rc = run_vm(ctx, p.mod_n2, it->second, ex, 3, count); if (rc) return rc;
    dirty.armed = false;
  }
  if (!rho3) return SC_OK;                                                                                             // unrandomized
  if (flags & SC_STEP_RANDOMIZERS_READY) return sc_modmul(ctx, p.mod_n2, enc, 2 * p.nw, rho3, 2 * p.nw, out3, 3 * count);   // rho^N computed ahead
  return sc_paillier_randomize(ctx, paillier_key_id, enc, rho3, out3, 3 * count);                                    // the 3 .randomize() (:126-128)
}

int sc_initiator_step67(sc_ctx* ctx, int paillier_key_id, const uint64_t* delta_a, const uint32_t* delta_b_enc, const uint32_t* zeta1_enc,
                        const uint32_t* zeta2_enc, const uint64_t* rsmall, const uint32_t* rshift, int flags, uint32_t* out, uint64_t count) {
  if (ctx && count == 0) return SC_OK;
  const PaillierKey* kp = paillier_key(ctx, paillier_key_id);
  if (!kp || !delta_a || !delta_b_enc || !zeta1_enc || !zeta2_enc || !rsmall || !rshift || !out)
    return fail(ctx, SC_ERR_ARG, "sc_initiator_step67: bad argument");
  const PaillierKey k = *kp;
  const int w2 = 2 * k.nw;
  // [[x<=y]] = [[zeta]] * D * [[-(r div 2^l) - (1 - delta_A)]],  D = [[delta_B]]^-1 (delta_A = 1) or [[delta_B]] (delta_A = 0):
  // steps 6 and 7 (SC/initiator.py:529-531, 558-563) with ONE inversion pass -- [[a]] [[b]] = [[a + b]] holds exactly for
  // unrandomized g = N + 1 encryptions, so the residues equal the literal formula's
  uint32_t* inv;
  int rc = tmp_words(ctx, TMP_S_E, count * w2, &inv); if (rc) return rc;
  int64_t bad = -1;
  rc = sc_modinv(ctx, k.mod_n2, delta_b_enc, inv, count, &bad);
  if (rc) return rc;
  // one launch for the rest: the selections "zeta_1 if r < (N-1)/2 else zeta_2" and "D" are per-item operand choices of the loads,
  // and [[-(r div 2^l)]] [[-1]]^(1 - delta_A) = 1 - (r div 2^l + 1 - delta_A) N  (mod N^2) is one encryption
  std::string key = "step67:" + std::to_string(k.mod_n2) + ":" + std::to_string(k.cst_n);
  auto it = ctx->progs.find(key);
  if (it == ctx->progs.end()) {
    Builder bd; const int cn = bd.use_const(k.cst_n);
    bd.loadw(6, 0, 0, k.nw); bd.add_flag(5, 0, true);                      // r div 2^l + (1 - delta_A)
    bd.mul_const(cn); bd.emit(OP_NEG); bd.emit(OP_ADD1);                   // 1 - (...) N
    bd.mul_const(0); bd.stt(0);
    bd.loadw_sel(0, 1, 4, 0); bd.mul_const(0); bd.stt(1);                  // [[zeta]]  (SC/initiator.py:558-560)
    bd.loadw_sel(2, 3, 5, 0); bd.mul_const(0);                             // D: [[delta_B]]^-1 where delta_A = 1 (:529-531)
    bd.mul_tbl(1); bd.mul_tbl(0);
    bd.redc(); bd.storew(7); bd.end();
    Prog p; rc = finalize_prog(ctx, ctx->mods[k.mod_n2], bd, &p); if (rc) return rc;
    it = ctx->progs.emplace(key, p).first;
  }
  VmExt ex[8] = {mk_ext(zeta1_enc, w2, w2), mk_ext(zeta2_enc, w2, w2), mk_ext(inv, w2, w2), mk_ext(delta_b_enc, w2, w2),
                 mk_ext(rsmall, 2, 2), mk_ext(delta_a, 2, 2), mk_ext(rshift, k.nw, k.nw), mk_ext(out, w2, w2)};
  return run_vm(ctx, k.mod_n2, it->second, ex, 8, count);
}

int sc_clock_probe(sc_ctx* ctx, int paillier_key_id, const uint32_t* rho, uint64_t count, double* out_ghz, double* out_ms) {
  const PaillierKey* kp = paillier_key(ctx, paillier_key_id);
  if (!kp || !rho || count == 0 || !out_ghz) return fail(ctx, SC_ERR_ARG, "sc_clock_probe: bad argument");
  const PaillierKey k = *kp;
  if ((k.secret && k.crt) || !k.pairs || sc_mod_supports_sq(ctx, k.mod_n) != 1)
    return fail(ctx, SC_ERR_ARG, "sc_clock_probe: needs a public Paillier key with pair arithmetic (the dominant launch)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t max_waves = (size_t)ctx->num_cu * 16;
  uint64_t* d_st = nullptr; uint32_t* d_out = nullptr;
  HIPCHK(ctx, hipMalloc((void**)&d_st, max_waves * 4 * sizeof(uint64_t)));
  if (hipMalloc((void**)&d_out, (size_t)count * 2 * k.nw * 4) != hipSuccess) { (void)hipFree(d_st); return fail(ctx, SC_ERR_HIP, "sc_clock_probe: out of memory"); }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipMemsetAsync(d_st, 0, max_waves * 4 * sizeof(uint64_t), ctx->stream);
  ctx->stamps = d_st; ctx->stamp_grid = 0;
  const double saved_macs = ctx->mac_counter;
  (void)hipEventRecord(e0, ctx->stream);
  int rc = sc_paillier_randomize(ctx, paillier_key_id, nullptr, rho, d_out, count);
  (void)hipEventRecord(e1, ctx->stream);
  ctx->stamps = nullptr; ctx->mac_counter = saved_macs;
  const uint32_t grid = ctx->stamp_grid;
  std::vector<uint64_t> st((size_t)grid * 4);
  if (!rc && grid == 0) rc = fail(ctx, SC_ERR_UNSUPPORTED, "sc_clock_probe: this batch did not take the (4,18) modulus-multiple pair launch");
  if (!rc && hipMemcpyAsync(st.data(), d_st, st.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(ctx, SC_ERR_HIP, "sc_clock_probe: copy failed");
  if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = fail(ctx, SC_ERR_HIP, "sc_clock_probe: launch failed");
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d_st); (void)hipFree(d_out);
  if (rc) return rc;
  int rate_khz = 0;                                              // rate of s_memrealtime
  if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || rate_khz <= 0) rate_khz = 100000;
  double sum = 0; uint64_t used = 0;
  for (uint32_t w = 0; w < grid; w++) {
    const uint64_t dc = st[4 * w + 2] - st[4 * w], dr = st[4 * w + 3] - st[4 * w + 1];
    if (dr > 0 && st[4 * w + 3] != 0) { sum += (double)dc / (double)dr; used++; }
  }
  if (!used) return fail(ctx, SC_ERR_HIP, "sc_clock_probe: no wave recorded its clocks");
  *out_ghz = sum / (double)used * (double)rate_khz * 1e3 / 1e9;
  if (out_ms) *out_ms = ms;
  return SC_OK;
}

}  // extern "C"
