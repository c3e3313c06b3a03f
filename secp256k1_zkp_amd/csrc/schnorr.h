// schnorr.h -- BIP-340 verification of one signature per lane (secp256k1_schnorrsig_verify,
// src/modules/schnorrsig/main_impl.h:215-261; challenge hash :106-120).
//   accept  <=>  R = s*G + (-e)*P is finite, has even y and x(R) == r,   e = H_tag(r || x(P) || msg) mod n
// The double multiplication is ecmult_lane (ecmult.h); everything else is a few hundred instructions.
#pragma once
#include "ecmult.h"
#include "sha256.h"

struct schnorr_midstate { u32 s[8]; };   // SHA256 state after the 64-byte tag prefix SHA256("BIP0340/challenge") x 2

S2K_HD void schnorr_tag_midstate(schnorr_midstate& m) {
    const char tag[] = "BIP0340/challenge";
    sha256_stream h; sha256_stream_init(h);
    sha256_stream_write(h, (const unsigned char*)tag, sizeof(tag) - 1);
    unsigned char th[32]; sha256_stream_finalize(h, th);
    sha256_stream g; sha256_stream_init(g);
    sha256_stream_write(g, th, 32); sha256_stream_write(g, th, 32);     // exactly one block -> compressed
    for (int i = 0; i < 8; i++) m.s[i] = g.s[i];
}
// 32 little-endian bytes (the in-memory secp256k1_ge_storage on little-endian hosts, cf. secp256k1_ge_from_bytes) -> fe
S2K_HD void fe_set_le32(fe& r, const unsigned char* b) {
    u32 w[8];
    for (int j = 0; j < 8; j++) w[j] = (u32)b[4 * j] | ((u32)b[4 * j + 1] << 8) | ((u32)b[4 * j + 2] << 16) | ((u32)b[4 * j + 3] << 24);
    fe_from_words(r, w);
}
// pk_format 0: 32-byte x-only serialisation (lift with even y, as secp256k1_xonly_pubkey_parse); 1: 64-byte opaque object
S2K_HD int schnorr_verify_lane(const schnorr_midstate& mid, const unsigned char* sig64, const unsigned char* msg, size_t msglen,
                               const unsigned char* pk, int pk_format, int live, const u32* gtab, const lane_mem& lm) {
    int ok = live;
    fe rx; scalar s, e; ge P; int ov;
    ok &= fe_set_b32_limit(rx, sig64);
    sc_set_b32(s, sig64 + 32, &ov); ok &= !ov;
    if (pk_format == 0) {
        fe x; ok &= fe_set_b32_limit(x, pk);
        ok &= ge_set_xo(P, x, 0);
    } else {
        fe_set_le32(P.x, pk); fe_set_le32(P.y, pk + 32);
        fe_normalize(P.x); fe_normalize(P.y);
        ok &= !fe_is_zero_normalized(P.x);
    }
    {
        sha256_stream h;
        for (int i = 0; i < 8; i++) h.s[i] = mid.s[i];
        for (int i = 0; i < 16; i++) h.buf[i] = 0;
        h.bytes = 64;
        unsigned char buf[32];
        sha256_stream_write(h, sig64, 32);
        fe px = P.x; fe_normalize(px); fe_get_b32(buf, px);
        sha256_stream_write(h, buf, 32);
        sha256_stream_write(h, msg, msglen);
        sha256_stream_finalize(h, buf);
        sc_set_b32(e, buf, nullptr);
        sc_negate(e, e);
    }
    if (!ok) { sc_set_zero(e); sc_set_zero(s); }
    gej Pj, R; gej_set_ge(Pj, P);
    ecmult_lane(R, Pj, e, s, 1, gtab, lm);
    ge a; ge_set_gej(a, R);
    ok &= !R.inf;
    ok &= !fe_is_odd(a.y);
    fe d; fe_neg(d, a.x, 1); fe_add(d, rx);
    ok &= fe_normalizes_to_zero(d);
    return ok;
}
