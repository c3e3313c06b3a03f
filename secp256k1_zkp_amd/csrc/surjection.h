// surjection.h -- surjection-proof verification (secp256k1_surjectionproof_verify,
// src/modules/surjection/main_impl.h:360-402; public keys surjection_impl.h:66-95, message :19-37; wire format
// main_impl.h:45-82).  A surjection proof is ONE Borromean ring over the used inputs: the chain
//      e_{j+1} = H( ser33( s_j G + e_j (T_out - T_in[j]) ) || m || 0 || j+1 )
// is serial inside a proof, so the mapping is one lane per proof (Elements uses at most 3 used inputs) on top of the
// same ecmult_lane / SHA-256 pieces as the rangeproof rings kernel.
#pragma once
#include "rangeproof.h"

#define SJ_MAX_INPUTS 256      // SECP256K1_SURJECTIONPROOF_MAX_N_INPUTS (include/secp256k1_surjectionproof.h)

// proof: serialised form (2-byte LE n_inputs | bitmap | e0 | s_0..).  tags: n_tags 64-byte generators (x||y).
S2K_HD int sj_verify_lane(const unsigned char* proof, u64 plen, const unsigned char* in_tags64, u64 n_tags, const unsigned char* out_tag64,
                          int live, const u32* gtab, const lane_mem& lm) {
    int ok = live;
    // ---- parse (secp256k1_surjectionproof_parse :45-82)
    u32 n_inputs = 0, bm_len = 0, n_used = 0;
    if (ok) {
        if (plen < 2) ok = 0;
        else {
            n_inputs = ((u32)proof[1] << 8) + proof[0];
            bm_len = (n_inputs + 7) / 8;
            if (n_inputs > SJ_MAX_INPUTS || plen < 2 + (u64)bm_len) ok = 0;
        }
    }
    if (ok) {
        if (n_inputs % 8 != 0) {
            const u32 mask = (0xFFu << (n_inputs % 8)) & 0xFFu;
            if (proof[2 + bm_len - 1] & mask) ok = 0;
        }
        for (u32 i = 0; i < bm_len; i++) { u32 b = proof[2 + i]; while (b) { n_used += b & 1u; b >>= 1; } }
        if (plen != 2 + (u64)bm_len + 32 * (u64)(1 + n_used)) ok = 0;
    }
    // ---- verify preconditions (:371-380)
    if (ok && (n_used == 0 || n_used > n_inputs || n_inputs != n_tags)) ok = 0;
    const unsigned char* data = proof + 2 + bm_len;          // e0 || s_0 || s_1 ...
    u32 m[8] = {0, 0, 0, 0, 0, 0, 0, 0}, e[8] = {0, 0, 0, 0, 0, 0, 0, 0}, e0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    ge out; fe_set_zero(out.x); fe_set_zero(out.y);
    if (ok) {
        // every s must be a canonical scalar (:389-395)
        for (u32 i = 0; i < n_used; i++) { scalar s; int ov; sc_set_b32(s, data + 32 + 32 * i, &ov); if (ov) ok = 0; }
        // message (secp256k1_surjection_genmessage, surjection_impl.h:19-37)
        sha256_stream h; sha256_stream_init(h);
        for (u64 i = 0; i <= n_tags; i++) {
            const unsigned char* t = (i < n_tags) ? in_tags64 + 64 * i : out_tag64;
            sha256_stream_put(h, (unsigned char)(2 + (t[63] & 1)));
            sha256_stream_write(h, t, 32);
        }
        unsigned char mb[32]; sha256_stream_finalize(h, mb);
        for (int i = 0; i < 8; i++) { m[i] = s2k_load_be32(mb + 4 * i); e0[i] = s2k_load_be32(data + 4 * i); }
        rp_hash_e0(e, e0, m, 0);
        fe_set_b32_mod(out.x, out_tag64); fe_set_b32_mod(out.y, out_tag64 + 32);      // generator_load (generator/main_impl.h:40-49)
    }
    // ---- the ring (borromean_verify with nrings = 1, borromean_impl.h:70-98)
    u32 pos = 0;                 // next input index to look at
    u32 last_prefix = 0, last_x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const u32 steps = ok ? n_used : 0;
    u32 max_steps = steps;
#if defined(__HIP_DEVICE_COMPILE__)
    for (int off = 32; off > 0; off >>= 1) { const u32 o = __shfl_xor(max_steps, off); max_steps = o > max_steps ? o : max_steps; }
#endif
#pragma unroll 1
    for (u32 j = 0; j < max_steps; j++) {
        const int step_live = ok & (j < steps);
        scalar ens, s; int ov_e, ov_s = 0;
        rp_words_to_scalar(ens, ov_e, e);
        sc_set_zero(s);
        gej pub; gej_set_infinity(pub);
        if (step_live) {
            sc_set_b32(s, data + 32 + 32 * j, &ov_s);
            while (!((proof[2 + (pos >> 3)] >> (pos & 7)) & 1)) pos++;             // j-th used input
            ge tin; fe_set_b32_mod(tin.x, in_tags64 + 64 * pos); fe_set_b32_mod(tin.y, in_tags64 + 64 * pos + 32);
            pos++;
            fe_neg(tin.y, tin.y, 1); fe_norm_weak(tin.y);
            gej a; gej_set_ge(a, tin);
            gej o; gej_set_ge(o, out);
            gej_add_var(pub, a, o);                                                  // T_out - T_in  (surjection_impl.h:66-95)
        }
        int good = step_live & !ov_e & !ov_s & !sc_is_zero(s) & !sc_is_zero(ens) & !pub.inf;
        if (!good) { sc_set_zero(ens); sc_set_zero(s); }
        gej R;
        ecmult_lane(R, pub, ens, s, 1, gtab, lm);
        good &= !R.inf;
        ge a; ge_set_gej(a, R);
        u32 xw[8]; fe_to_words(xw, a.x);
        u32 xb[8];
#pragma unroll
        for (int i = 0; i < 8; i++) xb[i] = xw[7 - i];
        const u32 prefix = 2u | (u32)fe_is_odd(a.y);
        if (step_live) {
            ok &= good;
            if (j + 1 < steps) rp_hash_step(e, prefix, xb, m, 0, j + 1);
            else { last_prefix = prefix; for (int i = 0; i < 8; i++) last_x[i] = xb[i]; }
        }
    }
    if (!ok) return 0;
    // e0 == SHA256( r_last || m )   (:100-103)
    sha256_stream h; sha256_stream_init(h);
    sha256_stream_put(h, (unsigned char)last_prefix);
    for (int i = 0; i < 8; i++) { unsigned char b[4]; s2k_store_be32(b, last_x[i]); sha256_stream_write(h, b, 4); }
    for (int i = 0; i < 8; i++) { unsigned char b[4]; s2k_store_be32(b, m[i]); sha256_stream_write(h, b, 4); }
    unsigned char d[32]; sha256_stream_finalize(h, d);
    int diff = 0;
    for (int i = 0; i < 32; i++) diff |= d[i] ^ data[i];
    return diff == 0;
}
