// rangeproof_rewind.h -- the wallet side of a Borromean rangeproof: given the nonce, recover value, blinding factor and the
// embedded message from a proof that verified (secp256k1_rangeproof_rewind, src/modules/rangeproof/main_impl.h:31-52;
// rangeproof_impl.h: genrand :61-108, recover_x / recover_k :339-356, rewind_inner :364-485, the commitment check :652-680).
// SURVEY.md section 8f, rank 3.
//
// Verification runs unchanged (rangeproof.h) with one addition: the rings kernel also stores the challenge e of every ring
// position (the reference's `evalues`, borromean_impl.h:80-83).  Rewinding is then one lane per proof:
//   1. replay the prover's RFC 6979 HMAC-SHA256 stream seeded with nonce | ser(commit) | ser(gen) | proof header (~190 draws,
//      8 SHA-256 compressions each): the ring nonces k_i ("sec") and the pad of every ring position ("prep"),
//   2. the last ring hides the value: s XOR pad at one of its last two positions carries  1|..|v|v|v,
//   3. the non-forged position of the last ring gives the blinding factor  x = (k - s)/e - sum_i k_i,
//   4. every other position gives 32 message bytes: (s or the recovered k) XOR pad,
//   5. accept only if  x*G + (v*scale + min_value)*H  equals the commitment (one ecmult_lane).
#pragma once
#include "rangeproof.h"

// ---- HMAC-SHA256 DRBG (RFC 6979 3.2) exactly as src/hash_impl.h:211-314 drives it --------------------------------------
struct rp_drbg { u32 k[8], v[8], ipad[8], opad[8]; int retry; };

S2K_HD void drbg_setkey(rp_drbg& d) {             // midstates after the K^ipad and K^opad blocks (hmac_sha256_initialize :225-235)
    u32 w[16];
    for (int i = 0; i < 8; i++) { w[i] = d.k[i] ^ 0x36363636u; w[8 + i] = 0x36363636u; }
    sha256_init(d.ipad); sha256_compress(d.ipad, w);
    for (int i = 0; i < 8; i++) { w[i] = d.k[i] ^ 0x5c5c5c5cu; w[8 + i] = 0x5c5c5c5cu; }
    sha256_init(d.opad); sha256_compress(d.opad, w);
}
// out = HMAC_K(V) or HMAC_K(V || 0x00): both inner messages fit one padded block after the pad block
S2K_HD void drbg_hmac_v(u32 out[8], const rp_drbg& d, int with_zero) {
    u32 st[8], w[16];
    for (int i = 0; i < 8; i++) { st[i] = d.ipad[i]; w[i] = d.v[i]; w[8 + i] = 0; }
    w[8] = with_zero ? 0x00800000u : 0x80000000u;
    w[15] = (64 + 32 + (with_zero ? 1 : 0)) * 8;
    sha256_compress(st, w);
    u32 so[8];
    for (int i = 0; i < 8; i++) { so[i] = d.opad[i]; w[i] = st[i]; w[8 + i] = 0; }
    w[8] = 0x80000000u; w[15] = (64 + 32) * 8;
    sha256_compress(so, w);
    for (int i = 0; i < 8; i++) out[i] = so[i];
}
// K = HMAC_K(V || sep || seed), the two long messages of initialisation (:264-268, :274-278)
S2K_HD void drbg_hmac_seed(u32 out[8], const rp_drbg& d, unsigned char sep, const unsigned char* seed, u32 seedlen) {
    sha256_stream h;
    for (int i = 0; i < 8; i++) h.s[i] = d.ipad[i];
    for (int i = 0; i < 16; i++) h.buf[i] = 0;
    h.bytes = 64;
    unsigned char b[32];
    for (int i = 0; i < 8; i++) s2k_store_be32(b + 4 * i, d.v[i]);
    sha256_stream_write(h, b, 32); sha256_stream_put(h, sep); sha256_stream_write(h, seed, seedlen);
    sha256_stream_finalize(h, b);
    u32 so[8], w[16];
    for (int i = 0; i < 8; i++) { so[i] = d.opad[i]; w[i] = s2k_load_be32(b + 4 * i); w[8 + i] = 0; }
    w[8] = 0x80000000u; w[15] = (64 + 32) * 8;
    sha256_compress(so, w);
    for (int i = 0; i < 8; i++) out[i] = so[i];
}
S2K_HD void drbg_init(rp_drbg& d, const unsigned char* seed, u32 seedlen) {
    u32 t[8];
    for (int i = 0; i < 8; i++) { d.v[i] = 0x01010101u; d.k[i] = 0; }
    drbg_setkey(d);
    drbg_hmac_seed(t, d, 0x00, seed, seedlen); for (int i = 0; i < 8; i++) d.k[i] = t[i];
    drbg_setkey(d);
    drbg_hmac_v(t, d, 0); for (int i = 0; i < 8; i++) d.v[i] = t[i];
    drbg_hmac_seed(t, d, 0x01, seed, seedlen); for (int i = 0; i < 8; i++) d.k[i] = t[i];
    drbg_setkey(d);
    drbg_hmac_v(t, d, 0); for (int i = 0; i < 8; i++) d.v[i] = t[i];
    d.retry = 0;
}
// 32 bytes (as 8 big-endian words)  (rfc6979_hmac_sha256_generate :285-314 with outlen = 32)
S2K_HD void drbg_generate(u32 out[8], rp_drbg& d) {
    u32 t[8];
    if (d.retry) {
        drbg_hmac_v(t, d, 1); for (int i = 0; i < 8; i++) d.k[i] = t[i];
        drbg_setkey(d);
        drbg_hmac_v(t, d, 0); for (int i = 0; i < 8; i++) d.v[i] = t[i];
    }
    drbg_hmac_v(t, d, 0);
    for (int i = 0; i < 8; i++) { d.v[i] = t[i]; out[i] = t[i]; }
    d.retry = 1;
}

// ---- recovery ----------------------------------------------------------------------------------------------------------
S2K_HD void rp_recover_x(scalar& x, const scalar& k, const scalar& e, const scalar& s) {     // (k - s) / e   (:339-346)
    scalar t, ei;
    sc_negate(t, s); sc_add(t, t, k);
    sc_inverse(ei, e);
    sc_mul(x, t, ei);
}
S2K_HD void rp_load_scalar_words(scalar& s, const u32* w8) { int ov; rp_words_to_scalar(s, ov, w8); }

// Rewinding in two parts, one lane per proof each.
// rp_rewind_draws: the replay of the prover's RFC 6979 stream -- the ring nonces ("secs", [32][8] words) and the pad of every ring
// position ("prep", [128][8] words).  It needs the nonce, the commitment, the generator and the proof's header only, NOT the outcome
// of the verification, and it is a serial chain of ~1 500 SHA-256 compressions: the engine runs it on a side stream underneath the
// rings kernel (engine_rangeproof.hip, rp_launch).
S2K_HD void rp_rewind_draws(const rp_rec& rec, const unsigned char* proof, const unsigned char* nonce32, const unsigned char* gen64, u32* prep, u32* secs) {
    const u32 rings = rec.rings, last = rec.last_rsize;
    rp_drbg rng;
    {   // seed = nonce | ser(commit) | ser(gen) | proof[0 .. header)    (genrand :74-78; ser = [!is_square(y)] | x, :53-59)
        unsigned char seed[32 + 33 + 33 + 10];
        for (int i = 0; i < 32; i++) seed[i] = nonce32[i];
        fe cx, cy, r;
        for (int i = 0; i < 9; i++) { cx.n[i] = rec.commit[i]; cy.n[i] = rec.commit[9 + i]; }
        seed[32] = (unsigned char)!fe_sqrt(r, cy);
        fe_get_b32(seed + 33, cx);
        fe gx, gy;
        fe_set_b32_mod(gx, gen64); fe_set_b32_mod(gy, gen64 + 32); fe_normalize(gx); fe_normalize(gy);
        seed[65] = (unsigned char)!fe_sqrt(r, gy);
        fe_get_b32(seed + 66, gx);
        const u32 hdr = rec.off_signs;                       // offset_post_header, <= 10
        for (u32 i = 0; i < hdr; i++) seed[98 + i] = proof[i];
        drbg_init(rng, seed, 98 + hdr);
    }
    scalar acc; sc_set_zero(acc);
    u32 npub = 0;
    for (u32 i = 0; i < rings; i++) {
        scalar sec; u32 t[8]; int ov;
        if (i + 1 < rings) {
            drbg_generate(t, rng);                            // one draw is thrown away (:84)
            do { drbg_generate(t, rng); rp_words_to_scalar(sec, ov, t); } while (ov || sc_is_zero(sec));
            sc_add(acc, acc, sec);
        } else {
            sc_negate(sec, acc);
        }
        for (int k = 0; k < 8; k++) secs[8 * i + k] = sec.d[k];
        const u32 rsize = (i + 1 == rings) ? last : 4u;
        for (u32 j = 0; j < rsize; j++) {
            drbg_generate(t, rng);
            for (int k = 0; k < 8; k++) prep[8 * npub + k] = t[k];
            npub++;
        }
    }
}
// rp_rewind_recover: for a proof that passed verification.  ev: [128][8] challenge words written by the rings kernel; prep / secs:
// what rp_rewind_draws left.  On success returns 1 with the blinding factor and the *raw* mantissa value (the caller scales it and
// checks the commitment); msg_out receives min(*mlen, recovered) bytes and *mlen the count (:364-485).
S2K_HD int rp_rewind_recover(scalar& blind, u64& value, unsigned char* msg_out, u64* mlen, const rp_rec& rec, const unsigned char* proof,
                             const u32* ev, const u32* prep, const u32* secs) {
    const u32 rings = rec.rings, last = rec.last_rsize;
    scalar s_orig_last[4];
    for (int i = 0; i < 4; i++) sc_set_zero(s_orig_last[i]);
    for (u32 j = 0; j < last && j < 4; j++) { int ov; rp_words_to_scalar(s_orig_last[j], ov, prep + 8 * (((rings - 1) << 2) + j)); }
    u32 npub = 0;
    value = 0xFFFFFFFFFFFFFFFFull;
    sc_set_zero(blind);
    const unsigned char* sbytes = proof + rec.off_s;
    if (rings == 1 && last == 1) {                            // a single exact-value proof: only the blinding factor (:386-396)
        scalar e, s;
        rp_load_scalar_words(e, ev); sc_set_b32(s, sbytes, nullptr);
        rp_recover_x(blind, s_orig_last[0], e, s);
        value = 0;
        if (mlen) *mlen = 0;
        return 1;
    }
    const u32 npub0 = (rings - 1) << 2;
    u32 j;
    for (j = 0; j < 2; j++) {                                 // look for the value encoding in the last ring (:398-417)
        const u32 idx = npub0 + last - 1 - j;
        u32 w[8];
        for (int k = 0; k < 8; k++) w[k] = s2k_load_be32(sbytes + 32 * idx + 4 * k) ^ prep[8 * idx + k];
        if ((w[0] & 0x80000000u) && w[4] == w[6] && w[5] == w[7] && w[2] == w[4] && w[3] == w[5]) {
            value = ((u64)w[6] << 32) | w[7];
            break;
        }
    }
    if (j > 1) { if (mlen) *mlen = 0; return 0; }
    u32 skip1 = last - 1 - j;
    u32 skip2 = (u32)((value >> ((rings - 1) << 1)) & 3);
    if (skip1 == skip2) { if (mlen) *mlen = 0; return 0; }
    if (skip2 >= last) { if (mlen) *mlen = 0; return 0; }      // the reference would read past the ring here; never a valid rewind
    {
        scalar e, s, x, seclast;
        rp_load_scalar_words(e, ev + 8 * (npub0 + skip2)); sc_set_b32(s, sbytes + 32 * (npub0 + skip2), nullptr);
        rp_recover_x(x, s_orig_last[skip2], e, s);
        for (int k = 0; k < 8; k++) seclast.d[k] = secs[8 * (rings - 1) + k];
        sc_negate(seclast, seclast);
        sc_add(blind, x, seclast);
    }
    skip1 += npub0; skip2 += npub0;
    if (!msg_out || !mlen || *mlen == 0) { if (mlen) *mlen = 0; return 1; }
    u64 offset = 0; const u64 cap = *mlen;
    npub = 0;
    for (u32 i = 0; i < rings; i++) {
        const u32 idx = (u32)((value >> (i << 1)) & 3);
        const u32 rsize = (i + 1 == rings) ? last : 4u;
        for (u32 jj = 0; jj < rsize; jj++) {
            if (npub == skip1 || npub == skip2) { npub++; continue; }
            u32 w[8];
            if (idx == jj) {                                  // k = s + x*e for the non-forged position (:462, recover_k :349-356)
                scalar x, e, s, k;
                for (int q = 0; q < 8; q++) x.d[q] = secs[8 * i + q];
                rp_load_scalar_words(e, ev + 8 * npub); sc_set_b32(s, sbytes + 32 * npub, nullptr);
                sc_mul(k, x, e); sc_add(k, k, s);
                unsigned char kb[32]; sc_get_b32(kb, k);
                for (int q = 0; q < 8; q++) w[q] = s2k_load_be32(kb + 4 * q);
            } else {
                for (int q = 0; q < 8; q++) w[q] = s2k_load_be32(sbytes + 32 * npub + 4 * q);
            }
            for (int q = 0; q < 8; q++) w[q] ^= prep[8 * npub + q];
            for (int b = 0; b < 32 && offset < cap; b++) { msg_out[offset] = (unsigned char)(w[b >> 2] >> (24 - 8 * (b & 3))); offset++; }
            npub++;
        }
    }
    *mlen = offset;
    return 1;
}
// both parts back to back (host emulation)
S2K_HD int rp_rewind(scalar& blind, u64& value, unsigned char* msg_out, u64* mlen, const rp_rec& rec, const unsigned char* proof,
                     const unsigned char* nonce32, const unsigned char* gen64, const u32* ev, u32* prep, u32* secs) {
    rp_rewind_draws(rec, proof, nonce32, gen64, prep, secs);
    return rp_rewind_recover(blind, value, msg_out, mlen, rec, proof, ev, prep, secs);
}
