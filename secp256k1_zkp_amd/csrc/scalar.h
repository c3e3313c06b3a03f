// scalar.h -- integers mod the group order n of secp256k1, 8 x 32-bit saturated limbs, always fully reduced.
//
// Role of the reference's src/scalar_4x64_impl.h (set_b32 :155-167, negate :184-203, is_high :252-264,
// mul_512 :849-880, reduce_512 :636-711, mul_shift_var :1071-1091) and src/scalar_impl.h:142-180
// (GLV split_lambda).  Scalars are touched a handful of times per point multiplication (split, recode), so
// this file favours compact straight-line code over instruction-level tuning; the field (fe.h) is where the
// cycles go.  Encodings: 32-byte big-endian, identical to secp256k1_scalar_get_b32 / set_b32.
#pragma once
#include "s2k_common.h"
#include "modinv.h"

struct scalar { u32 d[8]; };   // d[0] least significant

// n and 2^256 - n
#define SC_N0 0xD0364141u
#define SC_N1 0xBFD25E8Cu
#define SC_N2 0xAF48A03Bu
#define SC_N3 0xBAAEDCE6u
#define SC_N4 0xFFFFFFFEu
#define SC_N5 0xFFFFFFFFu
#define SC_N6 0xFFFFFFFFu
#define SC_N7 0xFFFFFFFFu

S2K_HD u32 sc_n_limb(int i) {
    return i == 0 ? SC_N0 : i == 1 ? SC_N1 : i == 2 ? SC_N2 : i == 3 ? SC_N3 : i == 4 ? SC_N4 : 0xFFFFFFFFu;
}
// c = 2^256 - n (129 bits): limbs {~N0+1, ~N1, ~N2, ~N3, 1}
S2K_HD u32 sc_nc_limb(int i) {
    return i == 0 ? (~SC_N0 + 1u) : i == 1 ? ~SC_N1 : i == 2 ? ~SC_N2 : i == 3 ? ~SC_N3 : i == 4 ? 1u : 0u;
}

S2K_HD void sc_set_zero(scalar& r) {
#pragma unroll
    for (int i = 0; i < 8; i++) r.d[i] = 0;
}
S2K_HD void sc_set_int(scalar& r, u32 v) { sc_set_zero(r); r.d[0] = v; }
S2K_HD void sc_set_u64(scalar& r, u64 v) { sc_set_zero(r); r.d[0] = (u32)v; r.d[1] = (u32)(v >> 32); }
S2K_HD int sc_is_zero(const scalar& a) {
    u32 z = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) z |= a.d[i];
    return z == 0;
}
S2K_HD int sc_eq(const scalar& a, const scalar& b) {
    u32 z = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) z |= a.d[i] ^ b.d[i];
    return z == 0;
}
// a >= n ?
S2K_HD int sc_check_overflow(const u32 d[8]) {
    int yes = 0, no = 0;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        const u32 ni = sc_n_limb(i);
        no |= (d[i] < ni) & ~yes;
        yes |= (d[i] > ni) & ~no;
    }
    return yes | (!no);   // equal counts as overflow
}
// r = a - overflow*n  (a < 2n)
S2K_HD void sc_reduce_once(u32 d[8], int overflow) {
    u64 t = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        t += (u64)d[i] + (overflow ? sc_nc_limb(i) : 0u);
        d[i] = (u32)t; t >>= 32;
    }
}
// cf. secp256k1_scalar_set_b32 (scalar_4x64_impl.h:155-167): value mod n, *overflow = (value >= n)
S2K_HD void sc_set_b32(scalar& r, const unsigned char* b, int* overflow) {
#pragma unroll
    for (int i = 0; i < 8; i++) r.d[i] = s2k_load_be32(b + 4 * (7 - i));
    const int o = sc_check_overflow(r.d);
    sc_reduce_once(r.d, o);
    if (overflow) *overflow = o;
}
S2K_HD void sc_get_b32(unsigned char* b, const scalar& a) {
#pragma unroll
    for (int i = 0; i < 8; i++) s2k_store_be32(b + 4 * (7 - i), a.d[i]);
}
// cf. secp256k1_scalar_negate :184-203
S2K_HD void sc_negate(scalar& r, const scalar& a) {
    const u32 nz = sc_is_zero(a) ? 0u : 0xFFFFFFFFu;
    u64 t = 1;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        t += (u64)(~a.d[i]) + sc_n_limb(i);
        r.d[i] = (u32)t & nz; t >>= 32;
    }
}
// cf. secp256k1_scalar_add :84-105
S2K_HD int sc_add(scalar& r, const scalar& a, const scalar& b) {
    u64 t = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { t += (u64)a.d[i] + b.d[i]; r.d[i] = (u32)t; t >>= 32; }
    const int o = (int)t | sc_check_overflow(r.d);
    sc_reduce_once(r.d, o);
    return o;
}
// a > n/2 ?  (cf. secp256k1_scalar_is_high :252-264); n/2 = 7FFFFFFF FFFFFFFF FFFFFFFF FFFFFFFF 5D576E73 57A4501D DFE92F46 681B20A0
S2K_HD int sc_is_high(const scalar& a) {
    const u32 h[8] = {0x681B20A0u, 0xDFE92F46u, 0x57A4501Du, 0x5D576E73u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu};
    int yes = 0, no = 0;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        no |= (a.d[i] < h[i]) & ~yes;
        yes |= (a.d[i] > h[i]) & ~no;
    }
    return yes;
}

// 8x8 -> 16 limb schoolbook product
S2K_HD void sc_mul_wide(u32 l[16], const u32 a_in[8], const u32 b_in[8]) {
    u64 acc = 0; u32 ex = 0;
    u32 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = a_in[i]; b[i] = b_in[i]; S2K_OPAQUE(a[i]); S2K_OPAQUE(b[i]); }
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = k - i;
            if (j < 0 || j > 7) continue;
            const u64 p = (u64)a[i] * b[j];
            acc += p;
            ex += (acc < p);
        }
        l[k] = (u32)acc;
        acc = (acc >> 32) | ((u64)ex << 32);
        ex = 0;
    }
    l[15] = (u32)acc;
}
// 512 -> 256 mod n by folding with 2^256 == c (c = 2^256 - n, 129 bits)  (role of secp256k1_scalar_reduce_512)
S2K_HD void sc_reduce_512(scalar& r, const u32 l[16]) {
    // pass 1: m = lo + hi*c            (hi 256 bits, c 129 bits -> m < 2^386)
    u32 m[13];
    {
        u64 acc = 0; u32 ex = 0;
#pragma unroll
        for (int k = 0; k < 13; k++) {
            if (k < 8) { acc += l[k]; }
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int j = k - i;
                if (j < 0 || j > 7) continue;
                u32 lv = l[8 + j]; S2K_OPAQUE(lv);
                const u64 p = (u64)sc_nc_limb(i) * lv;
                acc += p; ex += (acc < p);
            }
            m[k] = (u32)acc; acc = (acc >> 32) | ((u64)ex << 32); ex = 0;
        }
    }
    // pass 2: q = m_lo + m_hi*c        (m_hi = m[8..12] 130 bits -> q < 2^260)
    u32 q[9];
    {
        u64 acc = 0; u32 ex = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            if (k < 8) { acc += m[k]; }
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int j = k - i;
                if (j < 0 || j > 4) continue;
                u32 mv = m[8 + j]; S2K_OPAQUE(mv);
                const u64 p = (u64)sc_nc_limb(i) * mv;
                acc += p; ex += (acc < p);
            }
            q[k] = (u32)acc; acc = (acc >> 32) | ((u64)ex << 32); ex = 0;
        }
    }
    // pass 3: r = q_lo + q[8]*c        (q[8] < 2^4) -> r < 2^256 + 2^134
    u64 t = 0; u32 top;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 q8 = q[8]; S2K_OPAQUE(q8);
        t += (u64)q[i] + (u64)sc_nc_limb(i) * q8;
        r.d[i] = (u32)t; t >>= 32;
    }
    top = (u32)t;    // 0 or 1
    const int o = (int)top | sc_check_overflow(r.d);
    sc_reduce_once(r.d, o);
}
S2K_HD void sc_mul(scalar& r, const scalar& a, const scalar& b) {
    u32 l[16];
    sc_mul_wide(l, a.d, b.d);
    sc_reduce_512(r, l);
}
S2K_HD void sc_sqr(scalar& r, const scalar& a) { sc_mul(r, a, a); }

// r = round(a*b / 2^384)  (cf. secp256k1_scalar_mul_shift_var with shift = 384)
S2K_HD void sc_mul_shift384(scalar& r, const scalar& a, const scalar& b) {
    u32 l[16];
    sc_mul_wide(l, a.d, b.d);
    const u32 rnd = (l[11] >> 31) & 1u;
    u64 t = rnd;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        t += (i < 4) ? l[12 + i] : 0u;
        r.d[i] = (u32)t; t >>= 32;
    }
}

// lambda: the cube root of unity mod n with lambda*(x, y) = (beta*x, y)
S2K_HD void sc_set_lambda(scalar& r) {
    const u32 l[8] = {0x1B23BD72u, 0xDF02967Cu, 0x20816678u, 0x122E22EAu, 0x8812645Au, 0xA5261C02u, 0xC05C30E0u, 0x5363AD4Cu};
    for (int i = 0; i < 8; i++) r.d[i] = l[i];
}
// GLV decomposition k = r1 + lambda*r2 (mod n), |r1|,|r2| < 2^128 as signed residues.
// Same lattice constants and rounding as secp256k1_scalar_split_lambda (scalar_impl.h:142-180).
S2K_HD void sc_split_lambda(scalar& r1, scalar& r2, const scalar& k) {
    const scalar minus_b1 = {{0x0ABFE4C3u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u, 0, 0, 0, 0}};
    const scalar minus_b2 = {{0x3DB1562Cu, 0xD765CDA8u, 0x0774346Du, 0x8A280AC5u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}};
    const scalar g1 = {{0x45DBB031u, 0xE893209Au, 0x71E8CA7Fu, 0x3DAA8A14u, 0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u}};
    const scalar g2 = {{0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu, 0x0ABFE4C4u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u}};
    scalar lambda; sc_set_lambda(lambda);
    scalar c1, c2;
    sc_mul_shift384(c1, k, g1);
    sc_mul_shift384(c2, k, g2);
    sc_mul(c1, c1, minus_b1);
    sc_mul(c2, c2, minus_b2);
    sc_add(r2, c1, c2);
    sc_mul(r1, r2, lambda);
    sc_negate(r1, r1);
    sc_add(r1, r1, k);
}

// Signed 128-bit half-scalar produced by the split: value = (neg ? -1 : 1) * mag, mag < 2^128 (4 words + spare bit)
struct half_scalar { u32 w[5]; int neg; };
S2K_HD void sc_to_half(half_scalar& h, const scalar& s) {
    scalar t = s;
    h.neg = sc_is_high(s);
    if (h.neg) sc_negate(t, s);
#pragma unroll
    for (int i = 0; i < 5; i++) h.w[i] = t.d[i];
}

// GLV split with BOTH halves odd.  The split is only defined up to the lattice {(a, b) : a + lambda b == 0 mod n}; its basis
// (a1, b1), (a2, b2) (the constants of secp256k1_scalar_split_lambda, scalar_impl.h:106-141) has parities (odd, odd), (even, odd),
// so (b1, -a2) = v1 - v2 flips the parity of k1 alone, v2 that of k2 alone and v1 both.  Adding the right one (with the sign that
// shrinks the component it moves most) makes both halves odd at the price of ~1 bit: |k1|, |k2| < 2^129.  The signed-odd-digit
// recoding of ecmult.h then needs no "+1, subtract P afterwards" correction -- two point additions per multiplication saved.
S2K_HD void sc_split_lambda_odd(half_scalar& h0, half_scalar& h1, const scalar& k) {
    const scalar a1 = {{0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u, 0, 0, 0, 0}};
    const scalar mb1 = {{0x0ABFE4C3u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u, 0, 0, 0, 0}};                   // -b1 (b1 < 0)
    const scalar a2 = {{0x9D44CFD8u, 0x57C1108Du, 0xA8E2F3F6u, 0x14CA50F7u, 1, 0, 0, 0}};
    scalar k1, k2; sc_split_lambda(k1, k2, k);
    sc_to_half(h0, k1); sc_to_half(h1, k2);
    const int e1 = !(h0.w[0] & 1u), e2 = !(h1.w[0] & 1u);
    // (d1, d2): the lattice vector to add, as (magnitude, negative?) pairs
    scalar m1, m2; int n1, n2;
    sc_set_zero(m1); sc_set_zero(m2); n1 = 0; n2 = 0;
    if (e1 & e2) {                    // +-v1 = +-(a1, b1): k2 moves by |b1| ~ 2^127.8, towards zero
        const int s_neg = h1.neg;     // k2 >= 0: add v1 (b1 < 0 pulls k2 down); k2 < 0: subtract it
        m1 = a1; n1 = s_neg; m2 = mb1; n2 = !s_neg;
    } else if (e1) {                  // +-(b1, -a2): k2 moves by a2 ~ 2^128.1, towards zero
        const int s_neg = h1.neg;     // k2 >= 0: (b1, -a2); k2 < 0: (-b1, a2)
        m1 = mb1; n1 = !s_neg; m2 = a2; n2 = !s_neg;
    } else if (e2) {                  // +-v2 = +-(a2, a1): k1 moves by a2, towards zero
        const int s_neg = !h0.neg;    // k1 >= 0: subtract v2
        m1 = a2; n1 = s_neg; m2 = a1; n2 = s_neg;
    }
    scalar t;
    if (n1) sc_negate(t, m1); else t = m1;
    sc_add(k1, k1, t);
    if (n2) sc_negate(t, m2); else t = m2;
    sc_add(k2, k2, t);
    sc_to_half(h0, k1); sc_to_half(h1, k2);
}

// a^-1 mod n (0 for 0) by division steps (modinv.h); cf. secp256k1_scalar_inverse_var
S2K_HD void sc_inverse(scalar& r, const scalar& a) {
    u32 o[8];
    ds_inverse_words(o, a.d, DS_MOD_N);
#pragma unroll
    for (int i = 0; i < 8; i++) r.d[i] = o[i];
}
