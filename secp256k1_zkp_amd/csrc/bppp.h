// bppp.h -- Bulletproofs++ norm-argument verification (secp256k1_bppp_rangeproof_norm_product_verify,
// src/modules/bppp/bppp_norm_product_impl.h:425-552; callbacks :375-420; point codec bppp_util.h:30-46;
// challenge bppp_transcript_impl.h:25-33).
//
// The reference computes two multi-scalar multiplications with ecmult_multi_var,
//      res1 = C + sum_i gamma_i X_i + (gamma_i^2 - 1) R_i            (2 n_rounds + 1 points, decompressed on the fly)
//      res2 = v G + sum_i s_g[i] G_i + sum_j s_h[j] H_j              (g_len + h_len points)
// and accepts iff res1 == res2.  Here both are folded into ONE sum  res2 - res1  that must be the point at infinity,
// and a batch of proofs becomes a flat list of (proof, term) lanes:
//   bp_prologue  1 lane / proof : transcript challenges, the scalar vectors s_g, s_h, v  (scalar.h) -> per-term scalars
//   bp_term      1 lane / term  : fetch / decompress the term's point, full double-and-add (ecmult.h) -- for the few points that
//                                 come from the proof; the generators G_i / H_j are the same for every proof of a deployment, so
//                                 the engine keeps a fixed-base table per generator set in HBM (16 windows x 65536 multiples x
//                                 72 B = 75.5 MB per generator, 5.4 GB for 64 + 8 generators, built once on the device) and a
//                                 generator term is 16 table additions with no doubling (bp_term_fixed)
//   gej_reduce   segmented tree sum over each proof's terms, then  result = ok && sum == infinity
#pragma once
#include "ecmult.h"
#include "sha256.h"

struct bp_shape { u32 g_len, h_len, n_rounds, log_g, n_terms, n_gens; };

static inline u32 bp_log2(size_t n) { u32 l = 0; while (((size_t)2 << l) <= n) l++; return l; }
// returns 0 when the shape itself is rejected by the reference (:446-461)
static inline int bp_make_shape(bp_shape& s, size_t g_len, size_t c_vec_len, size_t n_gens, size_t proof_len) {
    if (g_len == 0 || c_vec_len == 0) return 0;
    const u32 lg = bp_log2(g_len), lh = bp_log2(c_vec_len);
    const u32 nr = lg > lh ? lg : lh;
    if (n_gens != g_len + c_vec_len || proof_len != 65 * (size_t)nr + 64) return 0;
    if ((g_len & (g_len - 1)) || (c_vec_len & (c_vec_len - 1))) return 0;
    s.g_len = (u32)g_len; s.h_len = (u32)c_vec_len; s.n_rounds = nr; s.log_g = lg; s.n_gens = (u32)n_gens;
    s.n_terms = (u32)(g_len + c_vec_len + 1 + 2 * nr + 1);
    return 1;
}

// transcript: the reference's secp256k1_sha256 object {uint32 s[8]; uint8 buf[64]; uint64 bytes} (src/hash.h), 104 bytes
S2K_HD void bp_load_transcript(sha256_stream& h, const unsigned char* t104) {
    for (int i = 0; i < 8; i++) h.s[i] = (u32)t104[4 * i] | ((u32)t104[4 * i + 1] << 8) | ((u32)t104[4 * i + 2] << 16) | ((u32)t104[4 * i + 3] << 24);
    u64 bytes = 0;
    for (int i = 0; i < 8; i++) bytes |= (u64)t104[96 + i] << (8 * i);
    h.bytes = bytes;
    for (int i = 0; i < 16; i++) h.buf[i] = s2k_load_be32(t104 + 32 + 4 * i);
    // bytes beyond the fill level are don't-care in the reference; clear them so that partial-word writes start clean
    const u32 fill = (u32)(bytes & 63);
    for (u32 b = fill; b < 64; b++) { const u32 wi = b >> 2, sh = 24 - 8 * (b & 3); h.buf[wi] &= ~(0xFFu << sh); }
}

// One lane per proof.  term_sc: n_terms scalars (8 words each, little-endian limbs).  Returns the proof's ok flag.
// sg_factors (optional, 8 x 8 words): when given, the g_len - 1 scalars s_g[1..] are NOT computed here -- their recurrence
// s_g[i] = s_g[i - 2^b] * gamma_b * rho^-(2^b)  (b = top bit of i) is the product over the set bits of i of the factors
// f_b = gamma_b * rho^-(2^b), which are left in sg_factors for bp_sg_entry: one lane per (proof, i) instead of 2 (g_len - 1)
// dependent scalar products on the proof's single lane (40 % of this latency-bound stage at g_len 64).
#define BP_MAX_LOG_G 8
S2K_HD int bp_prologue(u32* term_sc, const bp_shape& sh, const unsigned char* proof, const unsigned char* transcript104,
                       const unsigned char* rho32, const unsigned char* c_vec32, u32* sg_factors = nullptr) {
    int ov;
    scalar n, l, rho;
    sc_set_b32(n, proof + 65 * sh.n_rounds, &ov); if (ov) return 0;
    sc_set_b32(l, proof + 65 * sh.n_rounds + 32, &ov); if (ov) return 0;
    sc_set_b32(rho, rho32, nullptr);
    if (sc_is_zero(rho)) return 0;
    scalar rho_inv, rho_f = rho;
    sc_inverse(rho_inv, rho);
    for (u32 i = 0; i < sh.log_g; i++) sc_sqr(rho_f, rho_f);
    // gammas (kept in the term array slots of the X_i terms until the end)
    const u32 base2 = sh.g_len + sh.h_len + 1;
    sha256_stream tr; bp_load_transcript(tr, transcript104);
    for (u32 i = 0; i < sh.n_rounds; i++) {
        sha256_stream_write(tr, proof + 65 * i, 65);
        sha256_stream c = tr;
        unsigned char le[8] = {0, 0, 0, 0, 0, 0, 0, 0}, d[32];
        sha256_stream_write(c, le, 8);
        sha256_stream_finalize(c, d);
        scalar g; sc_set_b32(g, d, nullptr);
        for (int k = 0; k < 8; k++) term_sc[8 * (base2 + 1 + 2 * i) + k] = g.d[k];
    }
    // s_g
    {
        scalar s0; sc_mul(s0, n, rho_f); sc_mul(s0, s0, rho_inv);
        for (int k = 0; k < 8; k++) term_sc[k] = s0.d[k];
        scalar pw = rho_inv;                       // rho_inv^(2^log_i)
        u32 log_i = 0;
        if (sg_factors) {
            for (u32 b = 0; b < sh.log_g; b++) {
                if (b) sc_sqr(pw, pw);
                scalar gm, f;
                for (int k = 0; k < 8; k++) gm.d[k] = term_sc[8 * (base2 + 1 + 2 * b) + k];
                sc_mul(f, gm, pw);
                for (int k = 0; k < 8; k++) sg_factors[8 * b + k] = f.d[k];
            }
        } else
        for (u32 i = 1; i < sh.g_len; i++) {
            if (i == (2u << log_i)) { log_i++; sc_sqr(pw, pw); }
            const u32 p2 = 1u << log_i;
            scalar a, gm;
            for (int k = 0; k < 8; k++) { a.d[k] = term_sc[8 * (i - p2) + k]; gm.d[k] = term_sc[8 * (base2 + 1 + 2 * log_i) + k]; }
            sc_mul(a, a, gm); sc_mul(a, a, pw);
            for (int k = 0; k < 8; k++) term_sc[8 * i + k] = a.d[k];
        }
    }
    // s_h and h_c = <c_vec, s_h>
    scalar h_c; sc_set_zero(h_c);
    {
        for (int k = 0; k < 8; k++) term_sc[8 * sh.g_len + k] = l.d[k];
        u32 log_i = 0;
        for (u32 i = 0; i < sh.h_len; i++) {
            scalar a;
            if (i > 0) {
                if (i == (2u << log_i)) log_i++;
                const u32 p2 = 1u << log_i;
                scalar gm;
                for (int k = 0; k < 8; k++) { a.d[k] = term_sc[8 * (sh.g_len + i - p2) + k]; gm.d[k] = term_sc[8 * (base2 + 1 + 2 * log_i) + k]; }
                sc_mul(a, a, gm);
                for (int k = 0; k < 8; k++) term_sc[8 * (sh.g_len + i) + k] = a.d[k];
            } else a = l;
            scalar c, t; sc_set_b32(c, c_vec32 + 32 * i, nullptr);
            sc_mul(t, c, a); sc_add(h_c, h_c, t);
        }
    }
    // v = n^2 mu_f + h_c, mu_f = rho_f^2
    {
        scalar mu_f, v; sc_sqr(mu_f, rho_f); sc_mul(v, n, n); sc_mul(v, v, mu_f); sc_add(v, v, h_c);
        for (int k = 0; k < 8; k++) term_sc[8 * (sh.g_len + sh.h_len) + k] = v.d[k];
    }
    // res1 side, negated: -1 * C, -gamma_i * X_i, -(gamma_i^2 - 1) * R_i
    {
        scalar one, m1; sc_set_int(one, 1); sc_negate(m1, one);
        for (int k = 0; k < 8; k++) term_sc[8 * base2 + k] = m1.d[k];
        for (u32 i = 0; i < sh.n_rounds; i++) {
            scalar g, g2, t;
            for (int k = 0; k < 8; k++) g.d[k] = term_sc[8 * (base2 + 1 + 2 * i) + k];
            sc_sqr(g2, g); sc_add(g2, g2, m1);          // gamma^2 - 1
            sc_negate(t, g);
            for (int k = 0; k < 8; k++) term_sc[8 * (base2 + 1 + 2 * i) + k] = t.d[k];
            sc_negate(t, g2);
            for (int k = 0; k < 8; k++) term_sc[8 * (base2 + 2 + 2 * i) + k] = t.d[k];
        }
    }
    return 1;
}

// s_g[i], 1 <= i < g_len, from s_g[0] and the factors bp_prologue left (see there)
S2K_HD void bp_sg_entry(u32* term_sc, const u32* sg_factors, const bp_shape& sh, u32 i) {
    scalar acc;
    for (int k = 0; k < 8; k++) acc.d[k] = term_sc[k];
    for (u32 b = 0; b < sh.log_g; b++) {
        if ((i >> b) & 1u) { scalar f; for (int k = 0; k < 8; k++) f.d[k] = sg_factors[8 * b + k]; sc_mul(acc, acc, f); }
    }
    for (int k = 0; k < 8; k++) term_sc[8 * i + k] = acc.d[k];
}
// compressed point (0x02/0x03 || x) -> affine; cf. secp256k1_eckey_pubkey_parse (src/eckey_impl.h:18-22)
S2K_HD int bp_parse33(ge& p, const unsigned char* in33) {
    if (in33[0] != 2 && in33[0] != 3) return 0;
    fe x; if (!fe_set_b32_limit(x, in33 + 1)) return 0;
    return ge_set_xo(p, x, in33[0] == 3);
}
// cf. secp256k1_bppp_parse_one_of_points (bppp_util.h:30-46); *inf set when the encoded point is infinity
S2K_HD int bp_parse_one_of_points(ge& p, int& inf, const unsigned char* in65, int idx) {
    inf = 0;
    if (in65[0] > 3) return 0;
    const unsigned char* x = in65 + 1 + 32 * idx;
    u32 nz = 0; for (int i = 0; i < 32; i++) nz |= x[i];
    const u32 mask = 2u - (u32)idx;
    if (!nz) { if (in65[0] & mask) return 0; inf = 1; return 1; }
    unsigned char tmp[33];
    tmp[0] = (unsigned char)(2u | ((in65[0] & mask) >> (1 - idx)));
    for (int i = 0; i < 32; i++) tmp[1 + i] = x[i];
    return bp_parse33(p, tmp);
}

// One lane per (proof, term): returns the term's contribution (Jacobian) and whether its point parsed.
S2K_HD int bp_term(gej& out, const bp_shape& sh, u32 t, const u32* term_sc, const u32* gens18, const unsigned char* proof,
                   const unsigned char* commit33, int live, const u32* gtab, const lane_mem& lm) {
    scalar k, g; sc_set_zero(g);
    for (int i = 0; i < 8; i++) k.d[i] = term_sc[8 * t + i];
    gej A; gej_set_infinity(A);
    int ok = 1, has_g = 0;
    const u32 ngen = sh.g_len + sh.h_len;
    if (t < ngen) {
        ge p; for (int i = 0; i < 9; i++) { p.x.n[i] = gens18[18 * t + i]; p.y.n[i] = gens18[18 * t + 9 + i]; }
        gej_set_ge(A, p);
    } else if (t == ngen) {
        g = k; sc_set_zero(k); has_g = 1;
    } else if (t == ngen + 1) {
        u32 nz = 0; for (int i = 0; i < 33; i++) nz |= commit33[i];
        if (nz) { ge p; ok = bp_parse33(p, commit33); gej_set_ge(A, p); A.inf = !ok; }
    } else {
        const u32 r = t - (ngen + 2), i = r >> 1, idx = r & 1;
        ge p; int inf; ok = bp_parse_one_of_points(p, inf, proof + 65 * i, (int)idx);
        gej_set_ge(A, p); A.inf = inf | !ok;
    }
    if (!live) { sc_set_zero(k); sc_set_zero(g); gej_set_infinity(A); }
    ecmult_lane(out, A, k, g, has_g, gtab, lm);
    return ok;
}

// ---- fixed-base tables for a generator set -------------------------------------------------------------------------------
// One table per generator in the format of the table of G (gtable.h: signed D-bit digits, 64-byte records of canonical words, the header
// in slot 0), back to back, built by the same seeded construction (window bases -> seeds -> one affine addition per entry with a shared
// inversion per run of rows).  Rounds 2-5 had a format of their own here -- 16 unsigned 16-bit windows of 72-byte limb records, every entry
// its own 16-step double-and-add with its own inversion: 119 ms for the 72 generators of config 4 the first time a set is seen.  The width
// follows the size of the set (bp_tab_bits_for): fewer, wider windows for the usual sets (13 or 15 additions per generator term instead of
// 16), narrower ones when a large set would not fit.
S2K_HD u32 bp_tab_bits_for(size_t n_gens) {
    // table bytes: n_gens x W(D) x 2^(D-1) x 64 -- 20 bits: 0.44 GB per generator (13 additions), 19: 0.23 GB (14), 18: 0.13 GB (15),
    // 17: 0.07 GB (16).  Budget ~16 GB.
    return n_gens <= 36 ? 20u : (n_gens <= 68 ? 19u : (n_gens <= 127 ? 18u : 17u));
}
S2K_HD size_t bp_tab_stride(u32 D) { return gtab_words_for(D); }                       // words between the tables of consecutive generators
S2K_HD size_t bp_tab_words(size_t n_gens, u32 D) { return n_gens * bp_tab_stride(D); }
// k * P_gen through the generator's table (k: 8 little-endian words): one addition per window, no doubling; the record of window w + 1 is
// requested before the addition of window w.  `tab`: the generator's own table (its header gives the geometry).
S2K_HD void bp_term_fixed(gej& out, const u32* tab, const u32* k8) {
    gej acc; gej_set_infinity(acc);
    const gtab_geom GG = gtab_geometry(tab);
    u32 kr[S2K_GTAB_SWORDS]; gtab_recode(kr, k8, tab);
    u32 raw[16]; int have = 0, neg = 0;
    {   const u32* rec = tab;
        have = gtab_locate(rec, neg, tab, GG, 0, kr[0], kr[1]);
        if (have) { for (int i = 0; i < 16; i++) raw[i] = rec[i]; } }
    for (int w = 0; w < (int)GG.W; w++) {
        u32 nraw[16]; int nhave = 0, nneg = 0;
        if (w + 1 < (int)GG.W) {
            const int word = (int)(((u32)(w + 1) * GG.D) >> 5);
            const u32* rec = tab;
            nhave = gtab_locate(rec, nneg, tab, GG, w + 1, kr[word], word + 1 < S2K_GTAB_SWORDS ? kr[word + 1] : 0u);
            if (nhave) { for (int i = 0; i < 16; i++) nraw[i] = rec[i]; }
        }
        if (have) {
            ge p; fe_from_words(p.x, raw); fe_from_words(p.y, raw + 8);
            if (neg) { fe_neg(p.y, p.y, 1); fe_norm_weak(p.y); }
            gej t; const int f = gej_add_ge(t, acc, p); acc = t;
            if (f == GEJ_ADD_NEEDS_DOUBLE) { gej_double(t, acc); acc = t; }
        }
        have = nhave; neg = nneg;
        if (nhave) { for (int i = 0; i < 16; i++) raw[i] = nraw[i]; }
    }
    out = acc;
}

// ---- secp256k1_bppp_commit (bppp_norm_product_impl.h:105-151), batched ---------------------------------------
//   commit = v G + sum_i n_i G_i + sum_j l_j H_j ,   v = sum_i n_i^2 mu^(i+1) + <l, c>     (:33-69, :131-134)
// Every base is fixed: the generators use the set's fixed-base tables above, G the engine's generator table -- the same format, the same
// routine (bp_term_fixed: one addition per window, no doubling).  One lane per (commitment, base); the per-commitment sums reuse the segmented tree reduction.
S2K_HD void bpc_v_scalar(u32 v8[8], const unsigned char* n_vec32, u32 g_len, const unsigned char* l_vec32, const unsigned char* c_vec32, u32 h_len,
                         const unsigned char* mu32) {
    scalar mu, mu_pow, v; sc_set_b32(mu, mu32, nullptr); mu_pow = mu; sc_set_zero(v);
    for (u32 i = 0; i < g_len; i++) {
        scalar a, t; sc_set_b32(a, n_vec32 + 32 * i, nullptr);
        sc_mul(t, a, a); sc_mul(t, t, mu_pow); sc_mul(mu_pow, mu_pow, mu);
        sc_add(v, v, t);
    }
    for (u32 j = 0; j < h_len; j++) {
        scalar a, b, t; sc_set_b32(a, l_vec32 + 32 * j, nullptr); sc_set_b32(b, c_vec32 + 32 * j, nullptr);
        sc_mul(t, a, b); sc_add(v, v, t);
    }
    for (int k = 0; k < 8; k++) v8[k] = v.d[k];
}
