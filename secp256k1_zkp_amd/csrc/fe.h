// fe.h -- secp256k1 base field Fp, p = 2^256 - 2^32 - 977, for the gfx950 vector ALU.
//
// Role of the reference's src/field_5x52_impl.h + field_5x52_int128_impl.h (fe_mul_inner :18-152,
// fe_sqr_inner :154-272, normalize family field_5x52_impl.h:43-199, codecs :228-304) and of the
// addition chains in src/field_impl.h:37-146 -- re-designed, not translated:
//
//  * Layout: 9 limbs x 29 bits in 9 VGPRs (value = sum n[i] * 2^(29 i)), lazily reduced.  Measured on
//    MI355X (tools/ubench): v_mad_u64_u32 issues at half rate (4.4 cyc/wave64/SIMD) and the 2-cycle
//    VALU->carry-in hazard makes saturated carry chains expensive, so the fastest exact product is the
//    one with the fewest 32x32->64 multiply-accumulates and *no* carry flags: 81 MACs straight into a
//    64-bit column accumulator (9x29), versus 100 for 10x26 and 124 + 128-bit carry chains for the
//    reference's 5x52 under the compiler.  fe_mul 9x29 = 1.97e11/s chip-wide vs 1.69e11 (10x26) vs
//    1.13e11 (5x52/__int128).
//  * Reduction: 2^261 = 2^5 * 2^256 == 2^37 + 31264 (mod p), i.e. a limb at index 9+k folds to
//    31264*h at limb k plus h<<8 at limb k+1.  Bits >= 2^256 of the top limb fold with 2^256 == 2^32+977.
//
// Magnitude contract (checked in the VERIFY host build, tests/host_emul):
//    magnitude m  <=>  n[i] <= m*(2^29 + 2^20) for i<8  and  n[8] <= m*(2^24 + 2^10).
//    fe_mul / fe_sqr inputs need  m_a * m_b <= 7  (9 products of 58+ bits must fit a u64 column);
//    outputs have magnitude 1.  fe_neg(a, m) has magnitude m+1, fe_add adds magnitudes.
// Bit-exactness is defined on the canonical 32-byte encodings (fe_get_b32 after fe_normalize), never on limbs.
#pragma once
#include "s2k_common.h"
#include "modinv.h"

#define FE_LIMBS 9
#define FE_BITS 29
#define FE_M 0x1FFFFFFFu
#define FE_TOPM 0x00FFFFFFu

struct fe { u32 n[FE_LIMBS]; };

// S2K_VERIFY (host test build only): abort when a lazy-limb operation would wrap -- the dynamic counterpart of
// the reference's VERIFY magnitude tracking (field.h:32-38).
#ifdef S2K_VERIFY
#include <stdio.h>
#include <stdlib.h>
#define S2K_CHECK(c) do { if (!(c)) { fprintf(stderr, "S2K_VERIFY failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); abort(); } } while (0)
#else
#define S2K_CHECK(c) do { } while (0)
#endif

// p in 9x29
#define FE_P0 0x1FFFFC2Fu
#define FE_P1 0x1FFFFFF7u

S2K_HD u32 fe_p_limb(int i) { return i == 0 ? FE_P0 : (i == 1 ? FE_P1 : (i == 8 ? FE_TOPM : FE_M)); }

S2K_HD void fe_set_zero(fe& r) {
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) r.n[i] = 0;
}
S2K_HD void fe_set_int(fe& r, u32 v) { fe_set_zero(r); r.n[0] = v; }   // v < 2^29

// ---- codecs -------------------------------------------------------------------------------
// w[0] = least significant 32-bit word of the 256-bit integer.
S2K_HD void fe_from_words(fe& r, const u32 w[8]) {
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) {
        const int bit = FE_BITS * i, idx = bit >> 5, sh = bit & 31;
        u32 v = w[idx] >> sh;
        if (sh > 32 - FE_BITS && idx + 1 < 8) v |= w[idx + 1] << (32 - sh);
        r.n[i] = v & FE_M;
    }
}
// requires a fully normalised input
S2K_HD void fe_to_words(u32 w[8], const fe& a) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int bit = 32 * j, i = bit / FE_BITS, sh = bit % FE_BITS;   // word j starts inside limb i at bit sh
        u32 v = a.n[i] >> sh;
        if (i + 1 < FE_LIMBS) v |= a.n[i + 1] << (FE_BITS - sh);
        if (FE_BITS - sh + FE_BITS < 32 && i + 2 < FE_LIMBS) v |= a.n[i + 2] << (2 * FE_BITS - sh);
        w[j] = v;
    }
}
// big-endian 32 bytes -> field element, no reduction (value < 2^256, magnitude 1, maybe >= p).
// cf. secp256k1_fe_set_b32_mod (field_5x52_impl.h:228-245)
S2K_HD void fe_set_b32_mod(fe& r, const unsigned char* b) {
    u32 w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = s2k_load_be32(b + 4 * (7 - j));
    fe_from_words(r, w);
}
// returns 0 when the encoded integer is >= p (cf. secp256k1_fe_set_b32_limit :247-250)
S2K_HD int fe_set_b32_limit(fe& r, const unsigned char* b) {
    u32 w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = s2k_load_be32(b + 4 * (7 - j));
    fe_from_words(r, w);
    const u32 hi = w[2] & w[3] & w[4] & w[5] & w[6] & w[7];
    const int ge_p = (hi == 0xFFFFFFFFu) && (w[1] == 0xFFFFFFFFu || (w[1] == 0xFFFFFFFEu && w[0] >= 0xFFFFFC2Fu));
    return !ge_p;
}
// requires normalised input (cf. secp256k1_fe_get_b32 :253-287)
S2K_HD void fe_get_b32(unsigned char* b, const fe& a) {
    u32 w[8];
    fe_to_words(w, a);
#pragma unroll
    for (int j = 0; j < 8; j++) s2k_store_be32(b + 4 * (7 - j), w[j]);
}

// ---- linear ops (lazy) ----------------------------------------------------------------------
S2K_HD void fe_add(fe& r, const fe& a) {
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) { S2K_CHECK((u64)r.n[i] + a.n[i] < (1ull << 32)); r.n[i] += a.n[i]; }
}
S2K_HD void fe_add2(fe& r, const fe& a, const fe& b) {
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) { S2K_CHECK((u64)a.n[i] + b.n[i] < (1ull << 32)); r.n[i] = a.n[i] + b.n[i]; }
}
// r = (m+1)*p - a : magnitude m -> m+1  (cf. secp256k1_fe_negate_unchecked :306-322)
S2K_HD void fe_neg(fe& r, const fe& a, u32 m) {
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) {
        S2K_CHECK((u64)(m + 1) * fe_p_limb(i) < (1ull << 32) && (m + 1) * fe_p_limb(i) >= a.n[i]);
        r.n[i] = (m + 1) * fe_p_limb(i) - a.n[i];
    }
}
// r *= k (k small; magnitude multiplies)
S2K_HD void fe_mul_int(fe& r, u32 k) {
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) { S2K_CHECK((u64)r.n[i] * k < (1ull << 32)); r.n[i] *= k; }
}
S2K_HD void fe_select(fe& r, const fe& a, const fe& b, int take_a) {
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) r.n[i] = take_a ? a.n[i] : b.n[i];
}
S2K_HD void fe_cmov(fe& r, const fe& a, int flag) {
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) r.n[i] = flag ? a.n[i] : r.n[i];
}

// ---- normalisation ----------------------------------------------------------------------------
// Parallel (dependency-free) weak normalisation: any magnitude <= 7 -> magnitude 1.  Value unchanged mod p.
// (role of secp256k1_fe_normalize_weak, field_5x52_impl.h:79-104)
S2K_HD void fe_norm_weak(fe& r) {
    u32 c[FE_LIMBS];
    const u32 t = r.n[8] >> 24;
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = r.n[i] >> FE_BITS;
    r.n[8] = (r.n[8] & FE_TOPM) + c[7];
#pragma unroll
    for (int i = 7; i >= 1; i--) r.n[i] = (r.n[i] & FE_M) + c[i - 1];
    r.n[0] = (r.n[0] & FE_M) + t * 977u;
    r.n[1] += t << 3;
}
// Sequential carry pass: limbs 0..7 become < 2^29 exactly; top limb may keep bit 24.
S2K_HD void fe_norm_seq(fe& r) {
    const u32 t = r.n[8] >> 24;
    r.n[8] &= FE_TOPM;
    r.n[0] += t * 977u;
    r.n[1] += t << 3;
#pragma unroll
    for (int i = 0; i < 8; i++) { r.n[i + 1] += r.n[i] >> FE_BITS; r.n[i] &= FE_M; }
}
// Full normalisation to the canonical representative in [0, p) (cf. secp256k1_fe_normalize :43-77)
S2K_HD void fe_normalize(fe& r) {
    fe_norm_seq(r);                 // value < 2^256 + 2^237 < 2p, limbs 0..7 clean
    fe u = r;                       // u = r + (2^256 - p)
    u.n[0] += 977u; u.n[1] += 8u;
#pragma unroll
    for (int i = 0; i < 8; i++) { u.n[i + 1] += u.n[i] >> FE_BITS; u.n[i] &= FE_M; }
    const int ge = (u.n[8] >> 24) != 0;     // r >= p  <=>  r + 2^256 - p >= 2^256
    u.n[8] &= FE_TOPM;
    fe_cmov(r, u, ge);
}
// after fe_norm_seq: is the value 0 mod p?  (cf. secp256k1_fe_normalizes_to_zero :138-167)
S2K_HD int fe_seq_is_zero(const fe& a) {
    u32 z0 = 0, z1 = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) { z0 |= a.n[i]; z1 &= (a.n[i] ^ ~fe_p_limb(i)); }
    // z1 == all-ones  <=>  every limb equals the corresponding limb of p
    return (z0 == 0) | (z1 == 0xFFFFFFFFu);
}
S2K_HD int fe_normalizes_to_zero(const fe& a) { fe t = a; fe_norm_seq(t); return fe_seq_is_zero(t); }
// requires normalised input
S2K_HD int fe_is_odd(const fe& a) { return a.n[0] & 1; }
S2K_HD int fe_is_zero_normalized(const fe& a) {
    u32 z = 0;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) z |= a.n[i];
    return z == 0;
}
// equality mod p; magnitudes of a, b <= 1 (cf. secp256k1_fe_equal field_impl.h:25-35)
S2K_HD int fe_equal(const fe& a, const fe& b) {
    fe t; fe_neg(t, a, 1); fe_add(t, b);
    return fe_normalizes_to_zero(t);
}
// r = a/2 (cf. secp256k1_fe_half field_5x52_impl.h:358-398).  Magnitude m -> floor(m/2)+1.
S2K_HD void fe_half(fe& r) {
    const u32 odd = 0u - (r.n[0] & 1u);
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) r.n[i] += fe_p_limb(i) & odd;   // now even
#pragma unroll
    for (int i = 0; i < 8; i++) r.n[i] = (r.n[i] >> 1) + ((r.n[i + 1] & 1u) << (FE_BITS - 1));
    r.n[8] >>= 1;
}

// ---- multiplication ----------------------------------------------------------------------------------------
// Product scanning with the reduction folded in, two interleaved carry chains (the structure of the reference's
// fe_mul_inner, field_5x52_int128_impl.h:18-152, re-derived for 9x29):
//     T = sum_{k<17} col_k 2^(29k),  col_k = sum_{i+j=k} a_i b_j
//     high chain d walks columns 9..16 and emits clean limbs u_0..u_8 of  H = T >> 261;
//     low chain c walks columns 0..8 and, since 2^261 == 31264 + 2^8 * 2^29 (mod p), absorbs 31264*u_k + 256*u_{k-1}
//     in the same step -- two extra multiply-accumulates per limb instead of separate shifts and 64-bit adds.
// Every partial sum fits a u64 for mag(a)*mag(b) <= 7:  9 products <= 63.3*2^58, fold terms < 2^48, carry < 2^35.
// Instruction budget per product: 81 + 18 v_mad_u64_u32, 18 v_and, 18 v_lshrrev_b64 and a ~10-instruction tail.
S2K_HD void fe_mul_tail(fe& r, u64 c, u32 u8) {
    u32 k256 = 256u; S2K_OPAQUE(k256);
    c += (u64)u8 * k256;                                     // everything of weight 2^261
    const u64 hi = (u64)(r.n[8] >> 24) + (c << 5);           // everything >= 2^256, in units of 2^256 (< 2^46)
    r.n[8] &= FE_TOPM;
    u32 hi_lo = (u32)hi, hi_hi = (u32)(hi >> 32);
    S2K_OPAQUE(hi_lo); S2K_OPAQUE(hi_hi);
    u64 e = (u64)r.n[0] + (u64)hi_lo * 977u + ((u64)(hi_hi * 977u) << 32);     // 2^256 == 2^32 + 977
    r.n[0] = (u32)e & FE_M; e >>= FE_BITS;
    e += (u64)r.n[1] + (hi << 3);
    r.n[1] = (u32)e & FE_M; e >>= FE_BITS;
    r.n[2] += (u32)e;
}
// The "+ 256 * u" term of the fold: as a multiply-accumulate (default: one v_mad_u64_u32) or, with -DS2K_FOLD_SHIFT, as a 64-bit shift-add.
// Measured on MI355X (round 3, same box): the shift form has 8 % fewer multiply-accumulates and 8 % more instructions in the rings' main
// loop (the zero-extensions) and is 3 % SLOWER -- a non-MAC instruction costs ~0.6 of a MAC there: the kernel is bound by issue slots,
// not by the multiplier's energy.
#ifdef S2K_FOLD_SHIFT
#define S2K_FOLD256(c, u, k256) ((c) += ((u64)(u) << 8))
#else
#define S2K_FOLD256(c, u, k256) ((c) += (u64)(u) * (k256))
#endif
// r = a*b; needs mag(a)*mag(b) <= 7.
S2K_HD void fe_mul(fe& r, const fe& a_in, const fe& b_in) {
    u32 a[FE_LIMBS], b[FE_LIMBS];
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) { a[i] = a_in.n[i]; b[i] = b_in.n[i]; }
    S2K_OPAQUE(a[8]); S2K_OPAQUE(b[8]);          // the only limbs whose range (< 2^24) the optimiser can prove: see S2K_OPAQUE
    u32 k256 = 256u; S2K_OPAQUE(k256);           // keeps "x * 256 + acc" one v_mad_u64_u32 instead of a 64-bit shift and add
    u64 c = 0, d = 0; u32 u = 0, uprev = 0;
#pragma unroll
    for (int k = 0; k < FE_LIMBS; k++) {
        // column 9+k (high chain d, 8-k products) and column k (low chain c, k+1 products), issued alternately: the two
        // accumulators are independent, so each v_mad_u64_u32 has the other chain's between itself and its successor
#pragma unroll
        for (int t = 0; t < FE_LIMBS; t++) {
            if (k < 8 && k + 1 + t < FE_LIMBS) {
                const int i = k + 1 + t, j = 9 + k - i;
                S2K_CHECK(d + (u64)a[i] * b[j] >= d);
                d += (u64)a[i] * b[j]; S2K_CHAIN(d);
            }
            if (t <= k) {
                const int i = t, j = k - t;
                S2K_CHECK(c + (u64)a[i] * b[j] >= c);
                c += (u64)a[i] * b[j]; S2K_CHAIN(c);
            }
        }
        if (k < 8) {
            u = (u32)d & FE_M; d >>= FE_BITS;
        } else {
            S2K_CHECK((d >> 32) == 0);
            u = (u32)d;                           // what is left of the high chain (< 7*2^29)
        }
        S2K_CHECK(c + (u64)u * 31264u >= c);
        c += (u64)u * 31264u; S2K_CHAIN(c);
        if (k > 0) { S2K_FOLD256(c, uprev, k256); S2K_CHAIN(c); }
        uprev = u;
        r.n[k] = (u32)c & FE_M; c >>= FE_BITS;
    }
    fe_mul_tail(r, c, u);                         // c: carry of weight 2^261; 256*u_8 has that weight too
}
// r = a^2; needs mag(a) <= 2.
S2K_HD void fe_sqr(fe& r, const fe& a_in) {
    u32 a[FE_LIMBS], a2[FE_LIMBS];
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) a[i] = a_in.n[i];
    S2K_OPAQUE(a[8]);
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) { S2K_CHECK(a[i] < (1u << 31)); a2[i] = a[i] << 1; }
    S2K_OPAQUE(a2[8]);
    u32 k256 = 256u; S2K_OPAQUE(k256);
    u64 c = 0, d = 0; u32 u = 0, uprev = 0;
#pragma unroll
    for (int k = 0; k < FE_LIMBS; k++) {
        // as in fe_mul: the products of column 9+k (chain d) and of column k (chain c) are issued alternately
#pragma unroll
        for (int t = 0; t < FE_LIMBS; t++) {
            if (k < 8) {
                const int i = k + 1 + t, j = 9 + k - i;              // i + j = 9 + k, i <= j
                if (i < FE_LIMBS && i <= j) {
                    const u64 pr = (i == j) ? (u64)a[i] * a[i] : (u64)a2[i] * a[j];
                    S2K_CHECK(d + pr >= d);
                    d += pr; S2K_CHAIN(d);
                }
            }
            {
                const int i = t, j = k - t;                          // i + j = k, i <= j
                if (j >= 0 && i <= j) {
                    const u64 pr = (i == j) ? (u64)a[i] * a[i] : (u64)a2[i] * a[j];
                    S2K_CHECK(c + pr >= c);
                    c += pr; S2K_CHAIN(c);
                }
            }
        }
        if (k < 8) {
            u = (u32)d & FE_M; d >>= FE_BITS;
        } else {
            S2K_CHECK((d >> 32) == 0);
            u = (u32)d;
        }
        c += (u64)u * 31264u; S2K_CHAIN(c);
        if (k > 0) { S2K_FOLD256(c, uprev, k256); S2K_CHAIN(c); }
        uprev = u;
        r.n[k] = (u32)c & FE_M; c >>= FE_BITS;
    }
    fe_mul_tail(r, c, u);
}

// ---- two independent products in lockstep ------------------------------------------------------------------
// r1 = a1*b1 (or a1^2 when SQ1) and r2 = a2*b2 (or a2^2 when SQ2), both instruction streams interleaved: four accumulator
// chains instead of two, so that almost no v_mad_u64_u32 follows the one it depends on (the one-wait-state hazard that
// costs ~45 s_nop per single product).  8 % faster per product on MI355X at 2 waves/SIMD.  Same magnitude contract as
// fe_mul / fe_sqr for each operand pair.  The outputs may alias the inputs.
template <bool SQ1, bool SQ2>
S2K_HD void fe_dual(fe& r1, const fe& a1_in, const fe& b1_in, fe& r2, const fe& a2_in, const fe& b2_in) {
    u32 a1[FE_LIMBS], b1[FE_LIMBS], a2[FE_LIMBS], b2[FE_LIMBS], x1[FE_LIMBS], x2[FE_LIMBS];      // x = 2a for squarings
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) {
        a1[i] = a1_in.n[i]; b1[i] = SQ1 ? a1_in.n[i] : b1_in.n[i]; x1[i] = a1[i] << 1;
        a2[i] = a2_in.n[i]; b2[i] = SQ2 ? a2_in.n[i] : b2_in.n[i]; x2[i] = a2[i] << 1;
        if (SQ1) S2K_CHECK(a1[i] < (1u << 31));
        if (SQ2) S2K_CHECK(a2[i] < (1u << 31));
    }
    S2K_OPAQUE(a1[8]); S2K_OPAQUE(b1[8]); S2K_OPAQUE(a2[8]); S2K_OPAQUE(b2[8]); S2K_OPAQUE(x1[8]); S2K_OPAQUE(x2[8]);
    u32 k256 = 256u; S2K_OPAQUE(k256);
    u64 c1 = 0, d1 = 0, c2 = 0, d2 = 0; u32 u1 = 0, up1 = 0, u2 = 0, up2 = 0;
#pragma unroll
    for (int k = 0; k < FE_LIMBS; k++) {
#pragma unroll
        for (int t = 0; t < FE_LIMBS; t++) {
            if (k < 8) {
                const int i = k + 1 + t, j = 9 + k - i;                  // high column 9 + k
                if (i < FE_LIMBS && (!SQ1 || i <= j)) {
                    const u64 pr = !SQ1 ? (u64)a1[i] * b1[j] : (i == j) ? (u64)a1[i] * a1[i] : (u64)x1[i] * a1[j];
                    S2K_CHECK(d1 + pr >= d1); d1 += pr; S2K_CHAIN(d1);
                }
                if (i < FE_LIMBS && (!SQ2 || i <= j)) {
                    const u64 pr = !SQ2 ? (u64)a2[i] * b2[j] : (i == j) ? (u64)a2[i] * a2[i] : (u64)x2[i] * a2[j];
                    S2K_CHECK(d2 + pr >= d2); d2 += pr; S2K_CHAIN(d2);
                }
            }
            {
                const int i = t, j = k - t;                              // low column k
                if (j >= 0 && (!SQ1 || i <= j)) {
                    const u64 pr = !SQ1 ? (u64)a1[i] * b1[j] : (i == j) ? (u64)a1[i] * a1[i] : (u64)x1[i] * a1[j];
                    S2K_CHECK(c1 + pr >= c1); c1 += pr; S2K_CHAIN(c1);
                }
                if (j >= 0 && (!SQ2 || i <= j)) {
                    const u64 pr = !SQ2 ? (u64)a2[i] * b2[j] : (i == j) ? (u64)a2[i] * a2[i] : (u64)x2[i] * a2[j];
                    S2K_CHECK(c2 + pr >= c2); c2 += pr; S2K_CHAIN(c2);
                }
            }
        }
        if (k < 8) { u1 = (u32)d1 & FE_M; d1 >>= FE_BITS; u2 = (u32)d2 & FE_M; d2 >>= FE_BITS; }
        else { S2K_CHECK((d1 >> 32) == 0); S2K_CHECK((d2 >> 32) == 0); u1 = (u32)d1; u2 = (u32)d2; }
        c1 += (u64)u1 * 31264u; S2K_CHAIN(c1);
        c2 += (u64)u2 * 31264u; S2K_CHAIN(c2);
        if (k > 0) { S2K_FOLD256(c1, up1, k256); S2K_CHAIN(c1); S2K_FOLD256(c2, up2, k256); S2K_CHAIN(c2); }
        up1 = u1; up2 = u2;
        r1.n[k] = (u32)c1 & FE_M; c1 >>= FE_BITS;
        r2.n[k] = (u32)c2 & FE_M; c2 >>= FE_BITS;
    }
    fe_mul_tail(r1, c1, u1);
    fe_mul_tail(r2, c2, u2);
}
S2K_HD void fe_mul2(fe& r1, const fe& a1, const fe& b1, fe& r2, const fe& a2, const fe& b2) { fe_dual<false, false>(r1, a1, b1, r2, a2, b2); }
S2K_HD void fe_mul_sqr(fe& r1, const fe& a1, const fe& b1, fe& r2, const fe& a2) { fe_dual<false, true>(r1, a1, b1, r2, a2, a2); }
S2K_HD void fe_sqr2(fe& r1, const fe& a1, fe& r2, const fe& a2) { fe_dual<true, true>(r1, a1, a1, r2, a2, a2); }

// ---- sum of two products with ONE reduction ---------------------------------------------------------------------
// r = a1*b1 + a2*b2  (a1^2 when SQ1, a2^2 when SQ2): both products accumulate into the same pair of column chains, so the
// whole reduction (18 fold multiply-accumulates, 17 column carries, the tail) is paid once instead of twice -- the "lazy
// reduction" that the point formulas' Y3 = A*B + C*D lines ask for.  Needs mag(a1)*mag(b1) + mag(a2)*mag(b2) <= 7.
template <bool SQ1, bool SQ2>
S2K_HD void fe_muladd(fe& r, const fe& a1_in, const fe& b1_in, const fe& a2_in, const fe& b2_in) {
    u32 a1[FE_LIMBS], b1[FE_LIMBS], a2[FE_LIMBS], b2[FE_LIMBS], x1[FE_LIMBS], x2[FE_LIMBS];      // x = 2a for squarings
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) {
        a1[i] = a1_in.n[i]; b1[i] = SQ1 ? a1_in.n[i] : b1_in.n[i]; x1[i] = a1[i] << 1;
        a2[i] = a2_in.n[i]; b2[i] = SQ2 ? a2_in.n[i] : b2_in.n[i]; x2[i] = a2[i] << 1;
        if (SQ1) S2K_CHECK(a1[i] < (1u << 31));
        if (SQ2) S2K_CHECK(a2[i] < (1u << 31));
    }
    S2K_OPAQUE(a1[8]); S2K_OPAQUE(b1[8]); S2K_OPAQUE(a2[8]); S2K_OPAQUE(b2[8]); S2K_OPAQUE(x1[8]); S2K_OPAQUE(x2[8]);
    u32 k256 = 256u; S2K_OPAQUE(k256);
    u64 c = 0, d = 0; u32 u = 0, uprev = 0;
#pragma unroll
    for (int k = 0; k < FE_LIMBS; k++) {
#pragma unroll
        for (int t = 0; t < FE_LIMBS; t++) {
            // issue order d, c, d, c: the two chains alternate, so a multiply-accumulate rarely follows the one it depends on
            const int ih = k + 1 + t, jh = 8 - t;                            // high column 9 + k (ih + jh = 9 + k)
            const int il = t, jl = k - t;                                    // low column k
            if (k < 8 && ih < FE_LIMBS && (!SQ1 || ih <= jh)) {
                const u64 pr = !SQ1 ? (u64)a1[ih] * b1[jh] : (ih == jh) ? (u64)a1[ih] * a1[ih] : (u64)x1[ih] * a1[jh];
                S2K_CHECK(d + pr >= d); d += pr; S2K_CHAIN(d);
            }
            if (jl >= 0 && (!SQ1 || il <= jl)) {
                const u64 pr = !SQ1 ? (u64)a1[il] * b1[jl] : (il == jl) ? (u64)a1[il] * a1[il] : (u64)x1[il] * a1[jl];
                S2K_CHECK(c + pr >= c); c += pr; S2K_CHAIN(c);
            }
            if (k < 8 && ih < FE_LIMBS && (!SQ2 || ih <= jh)) {
                const u64 pr = !SQ2 ? (u64)a2[ih] * b2[jh] : (ih == jh) ? (u64)a2[ih] * a2[ih] : (u64)x2[ih] * a2[jh];
                S2K_CHECK(d + pr >= d); d += pr; S2K_CHAIN(d);
            }
            if (jl >= 0 && (!SQ2 || il <= jl)) {
                const u64 pr = !SQ2 ? (u64)a2[il] * b2[jl] : (il == jl) ? (u64)a2[il] * a2[il] : (u64)x2[il] * a2[jl];
                S2K_CHECK(c + pr >= c); c += pr; S2K_CHAIN(c);
            }
        }
        if (k < 8) { u = (u32)d & FE_M; d >>= FE_BITS; }
        else { S2K_CHECK((d >> 32) == 0); u = (u32)d; }
        S2K_CHECK(c + (u64)u * 31264u >= c);
        c += (u64)u * 31264u; S2K_CHAIN(c);
        if (k > 0) { S2K_FOLD256(c, uprev, k256); S2K_CHAIN(c); }
        uprev = u;
        r.n[k] = (u32)c & FE_M; c >>= FE_BITS;
    }
    fe_mul_tail(r, c, u);
}

// ---- exponentiation chains ------------------------------------------------------------------------
// r = x^(2^n) * y.  Inlined, with the run of squarings as a rolled loop: one fe_sqr body per call site (~1 KB of code each)
// instead of a call whose register spills around it cost more than the squarings' own bookkeeping (the out-of-line form
// left k_rp_lift with 448 bytes of scratch per lane).
S2K_HD void fe_sqrn_mul(fe& r, const fe& x, int n, const fe& y) {
    fe t = x;
#pragma unroll 1
    for (int i = 0; i < n; i++) fe_sqr(t, t);
    fe_mul(r, t, y);
}
S2K_HD void fe_sqrn(fe& r, const fe& x, int n) {
    fe t = x;
#pragma unroll 1
    for (int i = 0; i < n; i++) fe_sqr(t, t);
    r = t;
}
// x^(2^223 - 1) and helpers x2 = x^3, x22 = x^(2^22 - 1): the common prefix of a^(p-2) and a^((p+1)/4)
S2K_HD void fe_pow_x223(fe& x223, fe& x22, fe& x2, const fe& a) {
    fe x3, x6, x9, x11, x44, x88, x176, x220;
    fe_sqrn_mul(x2, a, 1, a);
    fe_sqrn_mul(x3, x2, 1, a);
    fe_sqrn_mul(x6, x3, 3, x3);
    fe_sqrn_mul(x9, x6, 3, x3);
    fe_sqrn_mul(x11, x9, 2, x2);
    fe_sqrn_mul(x22, x11, 11, x11);
    fe_sqrn_mul(x44, x22, 22, x22);
    fe_sqrn_mul(x88, x44, 44, x44);
    fe_sqrn_mul(x176, x88, 88, x88);
    fe_sqrn_mul(x220, x176, 44, x44);
    fe_sqrn_mul(x223, x220, 3, x3);
}
// r = a^(p-2) (Fermat): kept as the independent cross-check of fe_inv (tests) -- 255 squarings + 15 multiplications.
// Input magnitude <= 2.
S2K_HD void fe_inv_fermat(fe& r, const fe& a) {
    fe x223, x22, x2, t;
    fe_pow_x223(x223, x22, x2, a);
    fe_sqrn_mul(t, x223, 23, x22);
    fe_sqrn_mul(t, t, 5, a);
    fe_sqrn_mul(t, t, 3, x2);
    fe_sqrn_mul(r, t, 2, a);
}

// r = a^-1 mod p (0 for a = 0) by division steps (modinv.h).  Same value as secp256k1_fe_inv_var.  Input magnitude <= 2.
S2K_HD void fe_inv(fe& r, const fe& a) {
    fe an = a; fe_normalize(an);
    u32 w[8], o[8]; fe_to_words(w, an);
    ds_inverse_words(o, w, DS_MOD_P);
    fe_from_words(r, o);
}
// r = a^((p+1)/4); returns 1 iff r^2 == a, i.e. a is a square (cf. secp256k1_fe_sqrt, field_impl.h:37-146).
// Input magnitude <= 1.
S2K_HD int fe_sqrt(fe& r, const fe& a) {
    fe x223, x22, x2, t, chk;
    fe_pow_x223(x223, x22, x2, a);
    fe_sqrn_mul(t, x223, 23, x22);
    fe_sqrn_mul(t, t, 6, x2);
    fe_sqrn(t, t, 2);
    fe_sqrn(chk, t, 1);
    r = t;
    return fe_equal(chk, a);
}
