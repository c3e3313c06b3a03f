// modinv.h -- modular inverse by division steps ("safegcd", Bernstein-Yang 2019; the reference's secp256k1_modinv32,
// src/modinv32_impl.h, behind secp256k1_fe_inv(_var) field_5x52_impl.h:481-522 and secp256k1_scalar_inverse(_var)).
//
// Fixed 20 x 30 = 600 division steps (>= the 590 that suffice for a 256-bit modulus), branch-free, so every lane of a wave
// runs the same instructions.  Operands are 9 signed limbs of 30 bits; each batch of 30 steps works on the low words only and
// yields a 2x2 transition matrix that is then applied to (f, g) exactly and to (d, e) modulo the modulus.  About 17 000
// 32-bit instructions against ~40 000 issue slots for a Fermat exponentiation.  Used for the field (fe.h) and for the group
// order (scalar.h); the modulus is a compile-time constant, so its zero limbs cost nothing.
#pragma once
#include "s2k_common.h"

struct s30 { int32_t v[9]; };
struct s30_modulus { int32_t m[9]; u32 inv30; };          // signed 30-bit limbs of the modulus, modulus^-1 mod 2^30
#define S30_M ((int32_t)0x3FFFFFFF)
// p = 65536*2^240 - 4*2^30 - 977
#define S30_MOD_P s30_modulus{{-977, -4, 0, 0, 0, 0, 0, 0, 65536}, 0x2DDACACFu}
// n = group order
#define S30_MOD_N s30_modulus{{0x10364141, 0x3F497A33, 0x348A03BB, 0x2BB739AB, 0x3FFFFEBA, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0xFFFF}, 0x2A774EC1u}

S2K_HD int32_t s30_divsteps_30(int32_t zeta, u32 f0, u32 g0, int32_t t[4]) {
    u32 u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 5
    for (int i = 0; i < 30; i++) {
        u32 c1 = (u32)(zeta >> 31);                      // all ones when zeta < 0
        const u32 c2 = 0u - (g & 1u);                    // all ones when g is odd
        const u32 x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;     // -f, -u, -v when zeta < 0
        g += x & c2; q += y & c2; r += z & c2;
        c1 &= c2;                                        // swap (f, g) <- (g, g - f) only when zeta < 0 and g odd
        zeta = (zeta ^ (int32_t)c1) - 1;
        f += g & c1; u += q & c1; v += r & c1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    t[0] = (int32_t)u; t[1] = (int32_t)v; t[2] = (int32_t)q; t[3] = (int32_t)r;
    return zeta;
}
// (f, g) <- t * (f, g) / 2^30   (exact)
S2K_HD void s30_update_fg(s30& f, s30& g, const int32_t t[4]) {
    const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
    int64_t cf = u * f.v[0] + v * g.v[0], cg = q * f.v[0] + r * g.v[0];
    cf >>= 30; cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cf += u * f.v[i] + v * g.v[i];
        cg += q * f.v[i] + r * g.v[i];
        f.v[i - 1] = (int32_t)cf & S30_M; cf >>= 30;
        g.v[i - 1] = (int32_t)cg & S30_M; cg >>= 30;
    }
    f.v[8] = (int32_t)cf; g.v[8] = (int32_t)cg;
}
// (d, e) <- t * (d, e) / 2^30 mod m, both kept in (-2m, m)
S2K_HD void s30_update_de(s30& d, s30& e, const int32_t t[4], const s30_modulus md) {
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t xd = (u & sd) + (v & se), xe = (q & sd) + (r & se);              // add m to a negative d / e before multiplying
    int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0], ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
    xd -= (int32_t)((md.inv30 * (u32)cd + (u32)xd) & (u32)S30_M);            // multiples of m that clear the low 30 bits
    xe -= (int32_t)((md.inv30 * (u32)ce + (u32)xe) & (u32)S30_M);
    cd += (int64_t)md.m[0] * xd; ce += (int64_t)md.m[0] * xe;
    cd >>= 30; ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cd += (int64_t)u * d.v[i] + (int64_t)v * e.v[i];
        ce += (int64_t)q * d.v[i] + (int64_t)r * e.v[i];
        if (md.m[i] != 0) { cd += (int64_t)md.m[i] * xd; ce += (int64_t)md.m[i] * xe; }
        d.v[i - 1] = (int32_t)cd & S30_M; cd >>= 30;
        e.v[i - 1] = (int32_t)ce & S30_M; ce >>= 30;
    }
    d.v[8] = (int32_t)cd; e.v[8] = (int32_t)ce;
}
// d in (-2m, m), negated when sign < 0, brought to [0, m)
S2K_HD void s30_normalize(s30& r, int32_t sign, const s30_modulus md) {
    int32_t add = r.v[8] >> 31;
    const int32_t neg = sign >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) { r.v[i] += md.m[i] & add; r.v[i] = (r.v[i] ^ neg) - neg; }
#pragma unroll
    for (int i = 0; i < 8; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= S30_M; }
    add = r.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] += md.m[i] & add;
#pragma unroll
    for (int i = 0; i < 8; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= S30_M; }
}
// w (8 little-endian 32-bit words, value < m) -> w^-1 mod m (0 for 0)
S2K_HD void s30_inverse_words(u32 o[8], const u32 w[8], const s30_modulus md) {
    s30 f, g, d, e;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 30 * i, idx = bit >> 5, sh = bit & 31;
        u64 v = w[idx];
        if (idx + 1 < 8) v |= (u64)w[idx + 1] << 32;
        g.v[i] = (int32_t)((u32)(v >> sh) & (u32)S30_M);
        d.v[i] = 0; e.v[i] = 0; f.v[i] = md.m[i];
    }
    e.v[0] = 1;
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 20; it++) {
        int32_t t[4];
        zeta = s30_divsteps_30(zeta, (u32)f.v[0], (u32)g.v[0], t);
        s30_update_de(d, e, t, md);
        s30_update_fg(f, g, t);
    }
    s30_normalize(d, f.v[8], md);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int bit = 32 * j, i = bit / 30, sh = bit % 30;
        u64 v = (u64)(u32)d.v[i] >> sh;
        if (i + 1 < 9) v |= (u64)(u32)d.v[i + 1] << (30 - sh);
        if (i + 2 < 9) v |= (u64)(u32)d.v[i + 2] << (60 - sh);
        o[j] = (u32)v;
    }
}
