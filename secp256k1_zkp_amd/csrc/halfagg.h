// halfagg.h -- half-aggregated BIP-340 signatures: verification as ONE multi-scalar multiplication
// (secp256k1_schnorrsig_aggverify, src/modules/schnorrsig_halfagg/main_impl.h:108-198).
//
// The reference checks  s*G == sum_i z_i*(R_i + e_i*P_i)  with two single multiplications per signature.  The same
// statement is  (-s)*G + sum_i z_i*R_i + sum_i (z_i e_i)*P_i == infinity : a (2n+1)-term MSM, which is what runs here
// (msm.h), fed by four data-parallel passes and one serial one:
//
//   ha_points    1 lane / signature : R_i = lift_x(r_i) (even y), P_i from the key; writes the MSM's point array and x(P_i)
//   ha_schedule  1 lane / block     : message schedule W_t + K_t of every full 64-byte block of the randomizer hash's input
//                                     r_0|pk_0|m_0|r_1|pk_1|m_1|...  (the expansion does not depend on the chain)
//   ha_chain     1 wave, serial     : the 64 rounds of every block in order -- z_i hashes the whole prefix up to i
//                                     (main_impl.h:153-163), so this Merkle-Damgard chain is the serial floor of the scheme;
//                                     all values are wave-uniform, so it runs on the scalar unit
//   ha_scalars   1 lane / signature : z_i = finalize(copy of the state after item i), e_i = BIP-340 challenge,
//                                     MSM scalars z_i and z_i*e_i (z_0 = 1, :185), and -s
//   (msm)                           : msm.h on the 2n points, g_sc = -s;  accept iff the sum is the point at infinity
#pragma once
#include "schnorr.h"
#include "msm.h"

// SHA256 state after the 64-byte prefix SHA256("HalfAgg/randomizer") x 2 (main_impl.h:12-18)
S2K_HD void ha_tag_midstate(u32 s[8]) {
    const u32 m[8] = {0xd11f5532u, 0xfa57f70fu, 0x5db0d728u, 0xf806ffe1u, 0x1d4db069u, 0xb4d587e1u, 0x50451c2au, 0x10fb63e9u};
    for (int i = 0; i < 8; i++) s[i] = m[i];
}
// 32-byte unit u of the hashed stream: unit 3i = r_i, 3i+1 = x(P_i), 3i+2 = m_i
S2K_HD const unsigned char* ha_unit(const unsigned char* aggsig, const unsigned char* pkx32, const unsigned char* msgs32, size_t u) {
    const size_t i = u / 3; const u32 part = (u32)(u % 3);
    return (part == 0 ? aggsig : part == 1 ? pkx32 : msgs32) + 32 * i;
}
S2K_HD u32 ha_be32(const unsigned char* p) { return ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3]; }

// pt128: R_i (64 bytes x|y) then P_i (64 bytes); pkx32: x(P_i).  Returns 0 if r_i >= p, r_i is not an x coordinate, or the
// key is invalid (:143-145, :164-169).  pk_format as in schnorr.h.
S2K_HD int ha_points(unsigned char* pt128, unsigned char* pkx32, const unsigned char* r32, const unsigned char* pk, int pk_format) {
    int ok = 1;
    fe rx; ge R, P;
    ok &= fe_set_b32_limit(rx, r32);
    ok &= ge_set_xo(R, rx, 0);
    if (pk_format == 0) {
        fe x; ok &= fe_set_b32_limit(x, pk);
        ok &= ge_set_xo(P, x, 0);
    } else {
        fe_set_le32(P.x, pk); fe_set_le32(P.y, pk + 32);
        ok &= !fe_normalizes_to_zero(P.x);
    }
    fe_normalize(R.x); fe_normalize(R.y); fe_normalize(P.x); fe_normalize(P.y);
    fe_get_b32(pt128, R.x); fe_get_b32(pt128 + 32, R.y);
    fe_get_b32(pt128 + 64, P.x); fe_get_b32(pt128 + 96, P.y);
    fe_get_b32(pkx32, P.x);
    return ok;
}
// W_t + K_t (t = 0..63) of full block j (units 2j, 2j+1)
S2K_HD void ha_schedule(u32* wk, const unsigned char* aggsig, const unsigned char* pkx32, const unsigned char* msgs32, size_t j) {
    u32 w[64];
    const unsigned char* u0 = ha_unit(aggsig, pkx32, msgs32, 2 * j);
    const unsigned char* u1 = ha_unit(aggsig, pkx32, msgs32, 2 * j + 1);
    for (int t = 0; t < 8; t++) { w[t] = ha_be32(u0 + 4 * t); w[8 + t] = ha_be32(u1 + 4 * t); }
    for (int t = 16; t < 64; t++) {
        const u32 w15 = w[t - 15], w2 = w[t - 2];
        const u32 s0 = sha_rotr(w15, 7) ^ sha_rotr(w15, 18) ^ (w15 >> 3);
        const u32 s1 = sha_rotr(w2, 17) ^ sha_rotr(w2, 19) ^ (w2 >> 10);
        w[t] = w[t - 16] + s0 + w[t - 7] + s1;
    }
    for (int t = 0; t < 64; t++) wk[t] = w[t] + S2K_SHA_K(t);
}
// the 64 rounds of one block on a prepared schedule; s <- s + rounds(s)
S2K_HD void ha_rounds(u32 s[8], const u32* wk) {
    u32 a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
#pragma unroll
    for (int t = 0; t < 64; t++) {
        const u32 S1 = sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25);
        const u32 ch = (e & f) ^ (~e & g);
        const u32 t1 = h + S1 + ch + wk[t];
        const u32 S0 = sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22);
        const u32 mj = (a & b) ^ (a & c) ^ (b & c);
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + S0 + mj;
    }
    s[0] += a; s[1] += b; s[2] += c; s[3] += d; s[4] += e; s[5] += f; s[6] += g; s[7] += h;
}
// MSM scalars of signature i: sc64 = z_i (32 bytes big-endian) then z_i*e_i.  states: chain state after every full block.
S2K_HD void ha_scalars(unsigned char* sc64, const u32* states, const schnorr_midstate& bip340, const unsigned char* aggsig,
                       const unsigned char* pkx32, const unsigned char* msgs32, size_t i) {
    scalar z, e;
    {   // z_i: the stream holds 96*(i+1) bytes after item i: (3(i+1))/2 full blocks, plus m_i when i+1 is odd
        const size_t full = (3 * (i + 1)) >> 1;
        u32 st[8], w[16];
        for (int k = 0; k < 8; k++) st[k] = states[(full - 1) * 8 + k];
        for (int k = 0; k < 16; k++) w[k] = 0;
        int pos = 0;
        if ((i + 1) & 1) { for (int k = 0; k < 8; k++) w[k] = ha_be32(msgs32 + 32 * i + 4 * k); pos = 8; }
        w[pos] = 0x80000000u;
        const u64 bits = (u64)(64 + 96 * (i + 1)) * 8;
        w[14] = (u32)(bits >> 32); w[15] = (u32)bits;
        sha256_compress(st, w);
        unsigned char out[32];
        for (int k = 0; k < 8; k++) { out[4 * k] = (unsigned char)(st[k] >> 24); out[4 * k + 1] = (unsigned char)(st[k] >> 16); out[4 * k + 2] = (unsigned char)(st[k] >> 8); out[4 * k + 3] = (unsigned char)st[k]; }
        sc_set_b32(z, out, nullptr);
        if (i == 0) sc_set_int(z, 1);
    }
    {   // e_i = H_BIP0340/challenge(r_i | x(P_i) | m_i) mod n  (schnorrsig/main_impl.h:106-120)
        sha256_stream h;
        for (int k = 0; k < 8; k++) h.s[k] = bip340.s[k];
        for (int k = 0; k < 16; k++) h.buf[k] = 0;
        h.bytes = 64;
        unsigned char buf[32];
        sha256_stream_write(h, aggsig + 32 * i, 32);
        sha256_stream_write(h, pkx32 + 32 * i, 32);
        sha256_stream_write(h, msgs32 + 32 * i, 32);
        sha256_stream_finalize(h, buf);
        sc_set_b32(e, buf, nullptr);
    }
    scalar ze; sc_mul(ze, z, e);
    sc_get_b32(sc64, z); sc_get_b32(sc64 + 32, ze);
}
// g_sc = -s; returns 0 if s >= n (:187-190)
S2K_HD int ha_gscalar(unsigned char* g32, const unsigned char* s32) {
    scalar s; int ov; sc_set_b32(s, s32, &ov);
    sc_negate(s, s);
    sc_get_b32(g32, s);
    return !ov;
}
