/* oracle/zkp_oracle.c -- TEST INFRASTRUCTURE ONLY.  Plain-C CPU restatement of the reference's algorithms for the hot
 * path (BlockstreamResearch/secp256k1-zkp; all file:line citations are relative to the reference tree).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (secp256k1_zkp_amd/libsecp256k1_zkp_amd.so) never links, calls or falls back to it.
 *
 * PARITY PINNING: this restatement is pinned two ways by tests/test_cpu_restatement.py --
 *   (1) against every golden vector the reference's own tests carry for this path (tests/golden/ JSON files: 6 fixed
 *       rangeproofs, 19 BIP-340 vectors, 13 BP++ norm-argument vectors), and
 *   (2) against the unmodified reference itself (oracle/_ref, built by oracle/Makefile from the reference's sources)
 *       on random and edge inputs, primitive by primitive.
 *
 * What is restated (the *algorithms*, at the level where the reference's results are defined -- serialised bytes):
 *   field      Fp arithmetic; the reference's 5x52 lazy limbs (src/field_5x52_int128_impl.h:18-272,
 *              src/field_5x52_impl.h:43-304) are an implementation detail that never shows in a result, so this file
 *              keeps elements canonical in 4x64 limbs and reduces with 2^256 = 2^32 + 977 (mod p) after every product
 *   sqrt/inv   a^((p+1)/4) (src/field_impl.h:37-146) and a^(p-2) (value of fe_inv_var, src/field_5x52_impl.h:481-499)
 *   scalar     Zn arithmetic, set_b32 with overflow (src/scalar_4x64_impl.h:155-167), GLV split (src/scalar_impl.h:142-180)
 *   group      Jacobian doubling / addition / mixed addition (src/group_impl.h:468-659), lift_x (:347-373)
 *   ecmult     Strauss interleaved wNAF with GLV (src/ecmult_impl.h:252-375; wnaf :162-236; odd-multiples :73-115)
 *   multi      dispatcher + Pippenger bucket method with fixed wNAF and endomorphism (src/ecmult_impl.h:437-867)
 *   hash       SHA-256 (src/hash_impl.h:51-194)
 *   rangeproof header / verify / pub_expand / borromean (src/modules/rangeproof/rangeproof_impl.h:20-51,487-683,
 *              borromean_impl.h:23-104, generator/main_impl.h:40-49,266-273)
 *   schnorr    BIP-340 verify (src/modules/schnorrsig/main_impl.h:106-120,215-261)
 *   bppp       norm-argument verify (src/modules/bppp/bppp_norm_product_impl.h:375-552, bppp_util.h:30-46)
 */
#include "zkp_oracle.h"
#include <string.h>
#include <stdlib.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef uint32_t u32;

/* ======================================================= 256-bit helpers ================================================ */
typedef struct { u64 w[4]; } n256;          /* w[0] least significant */

static void n_from_be(n256 *r, const unsigned char *b) {
    int i, j;
    for (i = 0; i < 4; i++) { u64 v = 0; for (j = 0; j < 8; j++) v = (v << 8) | b[8 * (3 - i) + j]; r->w[i] = v; }
}
static void n_to_be(unsigned char *b, const n256 *a) {
    int i, j;
    for (i = 0; i < 4; i++) for (j = 0; j < 8; j++) b[8 * (3 - i) + j] = (unsigned char)(a->w[i] >> (8 * (7 - j)));
}
static int n_cmp(const n256 *a, const n256 *b) {
    int i;
    for (i = 3; i >= 0; i--) { if (a->w[i] < b->w[i]) return -1; if (a->w[i] > b->w[i]) return 1; }
    return 0;
}
static int n_is_zero(const n256 *a) { return (a->w[0] | a->w[1] | a->w[2] | a->w[3]) == 0; }
static u64 n_add(n256 *r, const n256 *a, const n256 *b) {
    u128 c = 0; int i;
    for (i = 0; i < 4; i++) { c += (u128)a->w[i] + b->w[i]; r->w[i] = (u64)c; c >>= 64; }
    return (u64)c;
}
static u64 n_sub(n256 *r, const n256 *a, const n256 *b) {
    u64 borrow = 0; int i;
    for (i = 0; i < 4; i++) { u128 d = (u128)a->w[i] - b->w[i] - borrow; r->w[i] = (u64)d; borrow = (u64)(d >> 64) & 1; }
    return borrow;
}
static void n_mul_wide(u64 l[8], const n256 *a, const n256 *b) {
    int i, j;
    memset(l, 0, 8 * sizeof(u64));
    for (i = 0; i < 4; i++) {
        u64 carry = 0;
        for (j = 0; j < 4; j++) { u128 t = (u128)a->w[i] * b->w[j] + l[i + j] + carry; l[i + j] = (u64)t; carry = (u64)(t >> 64); }
        l[i + 4] = carry;
    }
}
/* reduce the 512-bit l modulo m = 2^256 - c (c given as a short little-endian limb array, clen limbs) by repeated folding */
static void n_reduce_wide(n256 *r, const u64 l_in[8], const n256 *m, const u64 *c, int clen) {
    u64 l[8]; int i, j, round;
    memcpy(l, l_in, sizeof(l));
    for (round = 0; round < 4; round++) {
        /* l = lo + hi * c */
        u64 t[8]; u64 hi[4];
        memcpy(hi, l + 4, sizeof(hi));
        memcpy(t, l, 4 * sizeof(u64)); memset(t + 4, 0, 4 * sizeof(u64));
        for (i = 0; i < 4; i++) {
            u64 carry = 0;
            if (!hi[i]) continue;
            for (j = 0; j < clen; j++) { u128 v = (u128)hi[i] * c[j] + t[i + j] + carry; t[i + j] = (u64)v; carry = (u64)(v >> 64); }
            for (j = i + clen; carry && j < 8; j++) { u128 v = (u128)t[j] + carry; t[j] = (u64)v; carry = (u64)(v >> 64); }
        }
        memcpy(l, t, sizeof(l));
        if (!(l[4] | l[5] | l[6] | l[7])) break;
    }
    memcpy(r->w, l, 4 * sizeof(u64));
    while (n_cmp(r, m) >= 0) n_sub(r, r, m);
}

/* ============================================================ field ===================================================== */
static const n256 FP = {{0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL}};
static const u64 FP_C[1] = {0x1000003D1ULL};
typedef n256 fe;     /* always in [0, p) */

static void fe_set_b32_mod(fe *r, const unsigned char *b) { n_from_be(r, b); if (n_cmp(r, &FP) >= 0) n_sub(r, r, &FP); }   /* field_5x52_impl.h:228-245 */
static int fe_set_b32_limit(fe *r, const unsigned char *b) { n_from_be(r, b); return n_cmp(r, &FP) < 0; }                  /* :247-250 */
static void fe_get_b32(unsigned char *b, const fe *a) { n_to_be(b, a); }
static void fe_add(fe *r, const fe *a, const fe *b) { u64 c = n_add(r, a, b); if (c || n_cmp(r, &FP) >= 0) n_sub(r, r, &FP); }
static void fe_neg(fe *r, const fe *a) { if (n_is_zero(a)) *r = *a; else n_sub(r, &FP, a); }
static void fe_sub(fe *r, const fe *a, const fe *b) { fe t; fe_neg(&t, b); fe_add(r, a, &t); }
static void fe_mul(fe *r, const fe *a, const fe *b) { u64 l[8]; n_mul_wide(l, a, b); n_reduce_wide(r, l, &FP, FP_C, 1); }
static void fe_sqr(fe *r, const fe *a) { fe_mul(r, a, a); }
static void fe_set_int(fe *r, u64 v) { r->w[0] = v; r->w[1] = r->w[2] = r->w[3] = 0; }
static int fe_is_zero(const fe *a) { return n_is_zero(a); }
static int fe_is_odd(const fe *a) { return (int)(a->w[0] & 1); }
static int fe_equal(const fe *a, const fe *b) { return n_cmp(a, b) == 0; }
static void fe_pow(fe *r, const fe *a, const n256 *e) {
    fe acc; int i; fe_set_int(&acc, 1);
    for (i = 255; i >= 0; i--) { fe_sqr(&acc, &acc); if ((e->w[i >> 6] >> (i & 63)) & 1) fe_mul(&acc, &acc, a); }
    *r = acc;
}
static void fe_inv(fe *r, const fe *a) { n256 e = FP; e.w[0] -= 2; fe_pow(r, a, &e); }
/* r = a^((p+1)/4); returns 1 iff r^2 == a  (field_impl.h:37-146) */
static int fe_sqrt(fe *r, const fe *a) {
    n256 e = {{0xFFFFFFFFBFFFFF0CULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x3FFFFFFFFFFFFFFFULL}};
    fe t, c; fe_pow(&t, a, &e); fe_sqr(&c, &t); *r = t; return fe_equal(&c, a);
}
static int fe_is_square(const fe *a) { fe r; return fe_sqrt(&r, a); }

/* ============================================================ scalar ==================================================== */
static const n256 SN = {{0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL}};
static const u64 SN_C[3] = {0x402DA1732FC9BEBFULL, 0x4551231950B75FC4ULL, 1ULL};
typedef n256 sc;     /* always in [0, n) */

static void sc_set_b32(sc *r, const unsigned char *b, int *overflow) {        /* scalar_4x64_impl.h:155-167 */
    int o; n_from_be(r, b); o = n_cmp(r, &SN) >= 0; if (o) n_sub(r, r, &SN); if (overflow) *overflow = o;
}
static void sc_get_b32(unsigned char *b, const sc *a) { n_to_be(b, a); }
static void sc_add(sc *r, const sc *a, const sc *b) { u64 c = n_add(r, a, b); if (c || n_cmp(r, &SN) >= 0) n_sub(r, r, &SN); }
static void sc_neg(sc *r, const sc *a) { if (n_is_zero(a)) *r = *a; else n_sub(r, &SN, a); }
static void sc_mul(sc *r, const sc *a, const sc *b) { u64 l[8]; n_mul_wide(l, a, b); n_reduce_wide(r, l, &SN, SN_C, 3); }
static void sc_set_int(sc *r, u64 v) { r->w[0] = v; r->w[1] = r->w[2] = r->w[3] = 0; }
static int sc_is_zero(const sc *a) { return n_is_zero(a); }
static int sc_is_high(const sc *a) {                                          /* :252-264 */
    static const n256 half = {{0xDFE92F46681B20A0ULL, 0x5D576E7357A4501DULL, 0xFFFFFFFFFFFFFFFFULL, 0x7FFFFFFFFFFFFFFFULL}};
    return n_cmp(a, &half) > 0;
}
static void sc_inverse(sc *r, const sc *a) {
    n256 e = SN; sc acc; int i; e.w[0] -= 2; sc_set_int(&acc, 1);
    for (i = 255; i >= 0; i--) { sc_mul(&acc, &acc, &acc); if ((e.w[i >> 6] >> (i & 63)) & 1) sc_mul(&acc, &acc, a); }
    *r = acc;
}
/* round(a*b / 2^384)  (scalar_4x64_impl.h:1071-1091 with shift = 384) */
static void sc_mul_shift384(sc *r, const sc *a, const sc *b) {
    u64 l[8]; n256 one = {{1, 0, 0, 0}};
    n_mul_wide(l, a, b);
    r->w[0] = l[6]; r->w[1] = l[7]; r->w[2] = r->w[3] = 0;
    if ((l[5] >> 63) & 1) n_add(r, r, &one);
}
/* k = r1 + lambda*r2  (scalar_impl.h:142-180) */
static void sc_split_lambda(sc *r1, sc *r2, const sc *k) {
    static const sc minus_b1 = {{0x6F547FA90ABFE4C3ULL, 0xE4437ED6010E8828ULL, 0, 0}};
    static const sc minus_b2 = {{0xD765CDA83DB1562CULL, 0x8A280AC50774346DULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL}};
    static const sc g1 = {{0xE893209A45DBB031ULL, 0x3DAA8A1471E8CA7FULL, 0xE86C90E49284EB15ULL, 0x3086D221A7D46BCDULL}};
    static const sc g2 = {{0x1571B4AE8AC47F71ULL, 0x221208AC9DF506C6ULL, 0x6F547FA90ABFE4C4ULL, 0xE4437ED6010E8828ULL}};
    static const sc lambda = {{0xDF02967C1B23BD72ULL, 0x122E22EA20816678ULL, 0xA5261C028812645AULL, 0x5363AD4CC05C30E0ULL}};
    sc c1, c2;
    sc_mul_shift384(&c1, k, &g1); sc_mul_shift384(&c2, k, &g2);
    sc_mul(&c1, &c1, &minus_b1); sc_mul(&c2, &c2, &minus_b2);
    sc_add(r2, &c1, &c2);
    sc_mul(r1, r2, &lambda); sc_neg(r1, r1); sc_add(r1, r1, k);
}

/* ============================================================ group ===================================================== */
typedef struct { fe x, y; int inf; } ge;
typedef struct { fe x, y, z; int inf; } gej;
static const fe BETA = {{0xC1396C28719501EEULL, 0x9CF0497512F58995ULL, 0x6E64479EAC3434E9ULL, 0x7AE96A2B657C0710ULL}};
static const ge GEN = {{{0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL, 0x79BE667EF9DCBBACULL}},
                       {{0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL, 0x483ADA7726A3C465ULL}}, 0};

static void gej_set_inf(gej *r) { memset(r, 0, sizeof(*r)); r->inf = 1; }
static void gej_set_ge(gej *r, const ge *a) { r->x = a->x; r->y = a->y; fe_set_int(&r->z, 1); r->inf = a->inf; }
static void ge_neg(ge *r, const ge *a) { *r = *a; fe_neg(&r->y, &a->y); }
static void gej_neg(gej *r, const gej *a) { *r = *a; fe_neg(&r->y, &a->y); }
/* group_impl.h:468-501 (L = 3/2 X^2 formulation restated as the textbook a=0 doubling; same group element) */
static void gej_double(gej *r, const gej *a) {
    fe s, m, t, x3, y3, z3, y2, y4;
    if (a->inf) { gej_set_inf(r); return; }
    fe_sqr(&y2, &a->y);
    fe_mul(&s, &a->x, &y2); fe_add(&s, &s, &s); fe_add(&s, &s, &s);          /* S = 4 X Y^2 */
    fe_sqr(&m, &a->x); fe_add(&t, &m, &m); fe_add(&m, &m, &t);                /* M = 3 X^2 */
    fe_sqr(&x3, &m); fe_sub(&x3, &x3, &s); fe_sub(&x3, &x3, &s);             /* X3 = M^2 - 2S */
    fe_sqr(&y4, &y2); fe_add(&y4, &y4, &y4); fe_add(&y4, &y4, &y4); fe_add(&y4, &y4, &y4);   /* 8 Y^4 */
    fe_sub(&t, &s, &x3); fe_mul(&y3, &m, &t); fe_sub(&y3, &y3, &y4);
    fe_mul(&z3, &a->y, &a->z); fe_add(&z3, &z3, &z3);
    r->x = x3; r->y = y3; r->z = z3; r->inf = 0;
}
/* group_impl.h:534-596 */
static void gej_add(gej *r, const gej *a, const gej *b) {
    fe z12, z22, u1, u2, s1, s2, h, i, h2, h3, t, x3, y3, z3;
    if (a->inf) { *r = *b; return; }
    if (b->inf) { *r = *a; return; }
    fe_sqr(&z22, &b->z); fe_sqr(&z12, &a->z);
    fe_mul(&u1, &a->x, &z22); fe_mul(&u2, &b->x, &z12);
    fe_mul(&s1, &a->y, &z22); fe_mul(&s1, &s1, &b->z);
    fe_mul(&s2, &b->y, &z12); fe_mul(&s2, &s2, &a->z);
    fe_sub(&h, &u2, &u1); fe_sub(&i, &s2, &s1);
    if (fe_is_zero(&h)) { if (fe_is_zero(&i)) gej_double(r, a); else gej_set_inf(r); return; }
    fe_sqr(&h2, &h); fe_mul(&h3, &h2, &h); fe_mul(&t, &u1, &h2);
    fe_sqr(&x3, &i); fe_sub(&x3, &x3, &h3); fe_sub(&x3, &x3, &t); fe_sub(&x3, &x3, &t);
    fe_sub(&y3, &t, &x3); fe_mul(&y3, &y3, &i); fe_mul(&h3, &h3, &s1); fe_sub(&y3, &y3, &h3);
    fe_mul(&z3, &a->z, &b->z); fe_mul(&z3, &z3, &h);
    r->x = x3; r->y = y3; r->z = z3; r->inf = 0;
}
/* group_impl.h:598-659 */
static void gej_add_ge(gej *r, const gej *a, const ge *b) { gej bj; gej_set_ge(&bj, b); gej_add(r, a, &bj); }
/* group_impl.h:177-196 */
static void ge_set_gej(ge *r, const gej *a) {
    fe zi, zi2, zi3;
    if (a->inf) { memset(r, 0, sizeof(*r)); r->inf = 1; return; }
    fe_inv(&zi, &a->z); fe_sqr(&zi2, &zi); fe_mul(&zi3, &zi2, &zi);
    fe_mul(&r->x, &a->x, &zi2); fe_mul(&r->y, &a->y, &zi3); r->inf = 0;
}
/* group_impl.h:347-355: y = sqrt(x^3 + 7), the root that is itself a square; returns validity, y written either way */
static int ge_set_xquad(ge *r, const fe *x) {
    fe c, seven; fe_sqr(&c, x); fe_mul(&c, &c, x); fe_set_int(&seven, 7); fe_add(&c, &c, &seven);
    r->x = *x; r->inf = 0;
    return fe_sqrt(&r->y, &c);
}
/* group_impl.h:357-373 */
static int ge_set_xo(ge *r, const fe *x, int odd) {
    if (!ge_set_xquad(r, x)) return 0;
    if (fe_is_odd(&r->y) != odd) fe_neg(&r->y, &r->y);
    return 1;
}
static void ge_from_b64(ge *r, const unsigned char *b, int inf) { fe_set_b32_mod(&r->x, b); fe_set_b32_mod(&r->y, b + 32); r->inf = inf; }
static int gej_to_b64(unsigned char *b, const gej *j) {
    ge a; ge_set_gej(&a, j);
    if (a.inf) { memset(b, 0, 64); return 1; }
    fe_get_b32(b, &a.x); fe_get_b32(b + 32, &a.y); return 0;
}

/* ============================================================ ecmult ==================================================== */
/* width-w NAF of a (<= 129-bit) non-negative number, ecmult_impl.h:162-236.  Returns number of digits. */
static int wnaf(int *out, int len, const n256 *k, int w) {
    n256 a = *k; int i, bits = 0;
    memset(out, 0, len * sizeof(int));
    for (i = 0; i < len && !n_is_zero(&a); i++) {
        if (a.w[0] & 1) {
            int d = (int)(a.w[0] & ((1u << w) - 1));
            n256 t;
            if (d >= (1 << (w - 1))) d -= (1 << w);
            out[i] = d; bits = i + 1;
            if (d > 0) { t.w[0] = (u64)d; t.w[1] = t.w[2] = t.w[3] = 0; n_sub(&a, &a, &t); }
            else { t.w[0] = (u64)(-d); t.w[1] = t.w[2] = t.w[3] = 0; n_add(&a, &a, &t); }
        }
        /* a >>= 1 */
        a.w[0] = (a.w[0] >> 1) | (a.w[1] << 63); a.w[1] = (a.w[1] >> 1) | (a.w[2] << 63); a.w[2] = (a.w[2] >> 1) | (a.w[3] << 63); a.w[3] >>= 1;
    }
    return bits;
}
#define OW 5
#define OTAB (1 << (OW - 2))
/* odd multiples 1,3,..,(2*OTAB-1) of a (ecmult_impl.h:73-115), kept Jacobian here */
static void odd_multiples(gej *tab, const gej *a) {
    gej d; int i; gej_double(&d, a); tab[0] = *a;
    for (i = 1; i < OTAB; i++) gej_add(&tab[i], &tab[i - 1], &d);
}
static void table_get(gej *r, const gej *tab, int d) { if (d > 0) *r = tab[(d - 1) / 2]; else gej_neg(r, &tab[(-d - 1) / 2]); }
static void gej_mul_lambda(gej *r, const gej *a) { *r = *a; fe_mul(&r->x, &a->x, &BETA); }       /* group_impl.h:925-932 */

/* R = na*A + ng*G : interleaved wNAF over the GLV halves of both scalars (ecmult_impl.h:252-375) */
static void ecmult(gej *r, const gej *a, const sc *na, const sc *ng) {
    int digits[4][132]; int bits[4] = {0, 0, 0, 0}, neg[4] = {0, 0, 0, 0}, i, t, top = 0;
    gej tab[4][OTAB]; int used[4] = {0, 0, 0, 0};
    gej g; sc half[4];
    gej_set_ge(&g, &GEN);
    if (!a->inf && !sc_is_zero(na)) { sc_split_lambda(&half[0], &half[1], na); used[0] = used[1] = 1; }
    if (ng && !sc_is_zero(ng)) { sc_split_lambda(&half[2], &half[3], ng); used[2] = used[3] = 1; }
    for (t = 0; t < 4; t++) {
        if (!used[t]) continue;
        if (sc_is_high(&half[t])) { sc_neg(&half[t], &half[t]); neg[t] = 1; }
        bits[t] = wnaf(digits[t], 132, &half[t], OW);
        if (bits[t] > top) top = bits[t];
    }
    if (used[0]) { gej l; odd_multiples(tab[0], a); gej_mul_lambda(&l, a); odd_multiples(tab[1], &l); }
    if (used[2]) { gej l; odd_multiples(tab[2], &g); gej_mul_lambda(&l, &g); odd_multiples(tab[3], &l); }
    gej_set_inf(r);
    for (i = top - 1; i >= 0; i--) {
        gej_double(r, r);
        for (t = 0; t < 4; t++) {
            int d;
            if (!used[t] || i >= bits[t] || !(d = digits[t][i])) continue;
            { gej p; table_get(&p, tab[t], neg[t] ? -d : d); gej_add(r, r, &p); }
        }
    }
}

/* fixed-length signed odd-digit recoding + skew (role of secp256k1_wnaf_fixed, ecmult_impl.h:437-497):
 * s + skew = sum_i out[i] * 2^(w i), every out[i] odd, |out[i]| < 2^w; skew = 1 iff s is even (s = 0 -> all zero). */
static int wnaf_fixed(int *out, int n_wnaf, const n256 *s, int w) {
    n256 work = *s; int skew = 0, i;
    if (n_is_zero(s)) { for (i = 0; i < n_wnaf; i++) out[i] = 0; return 0; }
    if (!(work.w[0] & 1)) { n256 one = {{1, 0, 0, 0}}; skew = 1; n_add(&work, &work, &one); }
    for (i = 0; i < n_wnaf; i++) {
        int d, k;
        if (i == n_wnaf - 1) { d = (int)work.w[0]; }
        else {
            n256 t; d = (int)(work.w[0] & ((2u << w) - 1)) - (1 << w);
            if (d > 0) { t.w[0] = (u64)d; t.w[1] = t.w[2] = t.w[3] = 0; n_sub(&work, &work, &t); }
            else { t.w[0] = (u64)(-d); t.w[1] = t.w[2] = t.w[3] = 0; n_add(&work, &work, &t); }
            for (k = 0; k < w; k++) { work.w[0] = (work.w[0] >> 1) | (work.w[1] << 63); work.w[1] = (work.w[1] >> 1) | (work.w[2] << 63); work.w[2] = (work.w[2] >> 1) | (work.w[3] << 63); work.w[3] >>= 1; }
        }
        out[i] = d;
    }
    return skew;
}
static int pippenger_bucket_window(size_t n) {          /* ecmult_impl.h:597-621 */
    static const size_t lim[11] = {1, 4, 20, 57, 136, 235, 1260, 4420, 7880, 16050, 0};
    static const int win[11] = {1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12};
    int i;
    for (i = 0; i < 10; i++) if (n <= lim[i]) return win[i];
    return win[10];
}
/* Pippenger bucket method over GLV-split entries (ecmult_impl.h:516-591, 645-731) */
static void pippenger(gej *r, const sc *scalars, const ge *points, size_t n) {
    int c = pippenger_bucket_window(n), w = c + 1, n_wnaf = (130 + w - 1) / w, i; size_t e, ne = 0, nb = (size_t)1 << c;
    ge *pts = (ge *)malloc(2 * n * sizeof(ge) + sizeof(ge)); int *dig = (int *)malloc((2 * n + 1) * n_wnaf * sizeof(int)); int *skew = (int *)malloc((2 * n + 1) * sizeof(int));
    gej *buckets = (gej *)malloc(nb * sizeof(gej));
    for (e = 0; e < n; e++) {
        sc k1, k2; ge p1, p2;
        if (sc_is_zero(&scalars[e]) || points[e].inf) continue;          /* :523 */
        sc_split_lambda(&k1, &k2, &scalars[e]);
        p1 = points[e]; p2 = points[e]; fe_mul(&p2.x, &p2.x, &BETA);
        if (sc_is_high(&k1)) { sc_neg(&k1, &k1); ge_neg(&p1, &p1); }     /* endo_split :645-658 */
        if (sc_is_high(&k2)) { sc_neg(&k2, &k2); ge_neg(&p2, &p2); }
        pts[ne] = p1; skew[ne] = wnaf_fixed(dig + ne * n_wnaf, n_wnaf, &k1, w); ne++;
        pts[ne] = p2; skew[ne] = wnaf_fixed(dig + ne * n_wnaf, n_wnaf, &k2, w); ne++;
    }
    gej_set_inf(r);
    for (i = n_wnaf - 1; i >= 0; i--) {
        size_t b; gej running, sum;
        for (b = 0; b < nb; b++) gej_set_inf(&buckets[b]);
        for (e = 0; e < ne; e++) {
            int d = dig[e * n_wnaf + i]; ge t;
            if (i == 0 && skew[e]) { ge_neg(&t, &pts[e]); gej_add_ge(&buckets[0], &buckets[0], &t); }      /* skew correction :550-557 */
            if (d > 0) gej_add_ge(&buckets[(d - 1) / 2], &buckets[(d - 1) / 2], &pts[e]);
            else if (d < 0) { ge_neg(&t, &pts[e]); gej_add_ge(&buckets[(-d - 1) / 2], &buckets[(-d - 1) / 2], &t); }
        }
        for (b = 0; b < (size_t)w; b++) gej_double(r, r);
        /* sum_b (2b+1) bucket[b] by the running-sum trick (:572-588): with running_b = sum_{j>=b} bucket[j],
         * sum_b running_b = sum_j (j+1) bucket[j], so the odd-weight total is 2*that - running_0 */
        gej_set_inf(&running); gej_set_inf(&sum);
        for (b = nb; b-- > 0;) { gej_add(&running, &running, &buckets[b]); gej_add(&sum, &sum, &running); }
        gej_double(&sum, &sum); gej_neg(&running, &running); gej_add(&sum, &sum, &running);
        gej_add(r, r, &sum);
    }
    free(pts); free(dig); free(skew); free(buckets);
}
/* dispatcher (ecmult_impl.h:823-867): Pippenger from 88 points on, Strauss-style sum of single multiplications below */
static void ecmult_multi(gej *r, const sc *g_sc, const sc *scalars, const ge *points, size_t n) {
    gej acc; size_t i;
    gej_set_inf(&acc);
    if (n >= 88) pippenger(&acc, scalars, points, n);
    else for (i = 0; i < n; i++) { gej pj, t; gej_set_ge(&pj, &points[i]); ecmult(&t, &pj, &scalars[i], NULL); gej_add(&acc, &acc, &t); }
    if (g_sc) { gej inf_pt, t; gej_set_inf(&inf_pt); ecmult(&t, &inf_pt, g_sc, g_sc); gej_add(&acc, &acc, &t); }
    *r = acc;
}

/* ============================================================ sha256 ==================================================== */
typedef struct { u32 s[8]; unsigned char buf[64]; u64 bytes; } sha256;
static const u32 K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_init(sha256 *h) {
    static const u32 iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(h->s, iv, sizeof(iv)); h->bytes = 0;
}
static void sha256_block(u32 *s, const unsigned char *p) {             /* hash_impl.h:51-138 */
    u32 w[64], a, b, c, d, e, f, g, h, t1, t2; int i;
    for (i = 0; i < 16; i++) w[i] = ((u32)p[4 * i] << 24) | ((u32)p[4 * i + 1] << 16) | ((u32)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (i = 16; i < 64; i++) w[i] = w[i - 16] + (ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10));
    a = s[0]; b = s[1]; c = s[2]; d = s[3]; e = s[4]; f = s[5]; g = s[6]; h = s[7];
    for (i = 0; i < 64; i++) {
        t1 = h + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    s[0] += a; s[1] += b; s[2] += c; s[3] += d; s[4] += e; s[5] += f; s[6] += g; s[7] += h;
}
static void sha256_write(sha256 *h, const unsigned char *p, size_t n) {  /* :145-165 */
    while (n--) { h->buf[h->bytes++ & 63] = *p++; if (!(h->bytes & 63)) sha256_block(h->s, h->buf); }
}
static void sha256_final(sha256 *h, unsigned char *out) {               /* :167-194 */
    u64 bits = h->bytes << 3; unsigned char pad = 0x80, z = 0, len[8]; int i;
    sha256_write(h, &pad, 1);
    while ((h->bytes & 63) != 56) sha256_write(h, &z, 1);
    for (i = 0; i < 8; i++) len[i] = (unsigned char)(bits >> (8 * (7 - i)));
    sha256_write(h, len, 8);
    for (i = 0; i < 8; i++) { out[4 * i] = h->s[i] >> 24; out[4 * i + 1] = h->s[i] >> 16; out[4 * i + 2] = h->s[i] >> 8; out[4 * i + 3] = h->s[i]; }
}

/* ========================================================== rangeproof ================================================== */
static void borromean_hash(unsigned char *out, const unsigned char *m, const unsigned char *e, size_t elen, u32 ridx, u32 eidx) {   /* borromean_impl.h:23-37 */
    sha256 h; unsigned char t[8];
    t[0] = ridx >> 24; t[1] = ridx >> 16; t[2] = ridx >> 8; t[3] = ridx; t[4] = eidx >> 24; t[5] = eidx >> 16; t[6] = eidx >> 8; t[7] = eidx;
    sha256_init(&h); sha256_write(&h, e, elen); sha256_write(&h, m, 32); sha256_write(&h, t, 8); sha256_final(&h, out);
}
static void ser33(unsigned char *out, const ge *p) { out[0] = 2 | fe_is_odd(&p->y); fe_get_b32(out + 1, &p->x); }    /* eckey_impl.h:38-45 */
/* borromean_impl.h:53-104 */
static int borromean_verify(sc *evalues, const unsigned char *e0, const sc *s, const gej *pubs, const size_t *rsizes, size_t nrings, const unsigned char *m) {
    sha256 he0; unsigned char tmp[33]; size_t i, j, count = 0; int overflow;
    sha256_init(&he0);
    for (i = 0; i < nrings; i++) {
        sc ens;
        borromean_hash(tmp, m, e0, 32, (u32)i, 0);
        sc_set_b32(&ens, tmp, &overflow);
        for (j = 0; j < rsizes[i]; j++) {
            gej rj; ge ra;
            if (overflow || sc_is_zero(&s[count]) || sc_is_zero(&ens) || pubs[count].inf) return 0;
            if (evalues) evalues[count] = ens;                                 /* :80-83 */
            ecmult(&rj, &pubs[count], &ens, &s[count]);
            if (rj.inf) return 0;
            ge_set_gej(&ra, &rj); ser33(tmp, &ra);
            if (j != rsizes[i] - 1) { borromean_hash(tmp, m, tmp, 33, (u32)i, (u32)(j + 1)); sc_set_b32(&ens, tmp, &overflow); }
            else sha256_write(&he0, tmp, 33);
            count++;
        }
    }
    sha256_write(&he0, m, 32); sha256_final(&he0, tmp);
    return memcmp(e0, tmp, 32) == 0;
}
/* rangeproof_impl.h:487-538 */
static int rp_getheader(size_t *offset, int *exp, int *mantissa, u64 *scale, u64 *min_value, u64 *max_value, const unsigned char *proof, size_t plen) {
    int i, has_nz_range, has_min;
    if (plen < 65 || (proof[*offset] & 128)) return 0;
    has_nz_range = proof[*offset] & 64; has_min = proof[*offset] & 32; *exp = -1; *mantissa = 0;
    if (has_nz_range) {
        *exp = proof[*offset] & 31; *offset += 1;
        if (*exp > 18) return 0;
        *mantissa = proof[*offset] + 1;
        if (*mantissa > 64) return 0;
        *max_value = UINT64_MAX >> (64 - *mantissa);
    } else *max_value = 0;
    *offset += 1; *scale = 1;
    for (i = 0; i < *exp; i++) { if (*max_value > UINT64_MAX / 10) return 0; *max_value *= 10; *scale *= 10; }
    *min_value = 0;
    if (has_min) {
        if (plen - *offset < 8) return 0;
        for (i = 0; i < 8; i++) *min_value = (*min_value << 8) | proof[*offset + i];
        *offset += 8;
    }
    if (*max_value > UINT64_MAX - *min_value) return 0;
    *max_value += *min_value;
    return 1;
}
static void ser_point_rp(unsigned char *out, const ge *p) { out[0] = !fe_is_square(&p->y); fe_get_b32(out + 1, &p->x); }   /* rangeproof_impl.h:53-59 */
static void mul_u64(gej *r, u64 k, const ge *p) {                      /* value of pedersen_ecmult_small (generator/pedersen_impl.h:33-38) */
    int i; gej_set_inf(r);
    for (i = 63; i >= 0; i--) { gej_double(r, r); if ((k >> i) & 1) gej_add_ge(r, r, p); }
}
/* ---- HMAC-SHA256 and the RFC 6979 generator (hash_impl.h:211-314) ---- */
typedef struct { sha256 inner, outer; } hmac256;
static void hmac_init(hmac256 *h, const unsigned char *key, size_t keylen) {
    unsigned char rkey[64]; int n;
    memset(rkey, 0, 64);
    if (keylen <= 64) memcpy(rkey, key, keylen); else { sha256 t; sha256_init(&t); sha256_write(&t, key, keylen); sha256_final(&t, rkey); }
    sha256_init(&h->outer); for (n = 0; n < 64; n++) rkey[n] ^= 0x5c; sha256_write(&h->outer, rkey, 64);
    sha256_init(&h->inner); for (n = 0; n < 64; n++) rkey[n] ^= 0x5c ^ 0x36; sha256_write(&h->inner, rkey, 64);
}
static void hmac_final(hmac256 *h, unsigned char *out32) { unsigned char t[32]; sha256_final(&h->inner, t); sha256_write(&h->outer, t, 32); sha256_final(&h->outer, out32); }
typedef struct { unsigned char v[32], k[32]; int retry; } rfc6979;
static void rfc6979_init(rfc6979 *r, const unsigned char *key, size_t keylen) {
    hmac256 h; static const unsigned char zero = 0, one = 1;
    memset(r->v, 1, 32); memset(r->k, 0, 32);
    hmac_init(&h, r->k, 32); sha256_write(&h.inner, r->v, 32); sha256_write(&h.inner, &zero, 1); sha256_write(&h.inner, key, keylen); hmac_final(&h, r->k);
    hmac_init(&h, r->k, 32); sha256_write(&h.inner, r->v, 32); hmac_final(&h, r->v);
    hmac_init(&h, r->k, 32); sha256_write(&h.inner, r->v, 32); sha256_write(&h.inner, &one, 1); sha256_write(&h.inner, key, keylen); hmac_final(&h, r->k);
    hmac_init(&h, r->k, 32); sha256_write(&h.inner, r->v, 32); hmac_final(&h, r->v);
    r->retry = 0;
}
static void rfc6979_gen32(rfc6979 *r, unsigned char *out) {
    hmac256 h; static const unsigned char zero = 0;
    if (r->retry) {
        hmac_init(&h, r->k, 32); sha256_write(&h.inner, r->v, 32); sha256_write(&h.inner, &zero, 1); hmac_final(&h, r->k);
        hmac_init(&h, r->k, 32); sha256_write(&h.inner, r->v, 32); hmac_final(&h, r->v);
    }
    hmac_init(&h, r->k, 32); sha256_write(&h.inner, r->v, 32); hmac_final(&h, r->v);
    memcpy(out, r->v, 32);
    r->retry = 1;
}
/* rangeproof_impl.h:61-108 with message = the zeroed `prep` buffer (as rewind_inner calls it, :385) */
static void rp_genrand(sc *sec, sc *s, unsigned char *prep, const size_t *rsizes, size_t rings, const unsigned char *nonce, const ge *commit,
                       const unsigned char *proof, size_t len, const ge *genp) {
    unsigned char tmp[32], seed[32 + 33 + 33 + 10]; rfc6979 rng; sc acc; int overflow; size_t i, j, npub = 0; int b;
    memcpy(seed, nonce, 32); ser_point_rp(seed + 32, commit); ser_point_rp(seed + 65, genp); memcpy(seed + 98, proof, len);
    rfc6979_init(&rng, seed, 98 + len);
    sc_set_int(&acc, 0);
    for (i = 0; i < rings; i++) {
        if (i < rings - 1) {
            rfc6979_gen32(&rng, tmp);
            do { rfc6979_gen32(&rng, tmp); sc_set_b32(&sec[i], tmp, &overflow); } while (overflow || sc_is_zero(&sec[i]));
            sc_add(&acc, &acc, &sec[i]);
        } else { sc_neg(&acc, &acc); sec[i] = acc; }
        for (j = 0; j < rsizes[i]; j++) {
            rfc6979_gen32(&rng, tmp);
            for (b = 0; b < 32; b++) { tmp[b] ^= prep[(i * 4 + j) * 32 + b]; prep[(i * 4 + j) * 32 + b] = tmp[b]; }
            sc_set_b32(&s[npub], tmp, &overflow);
            npub++;
        }
    }
}
static void rp_recover_x(sc *x, const sc *k, const sc *e, const sc *s) { sc t; sc_neg(x, s); sc_add(x, x, k); sc_inverse(&t, e); sc_mul(x, x, &t); }   /* :339-346 */
/* rangeproof_impl.h:364-485 */
static int rp_rewind_inner(sc *blind, u64 *v, unsigned char *m, size_t *mlen, const sc *ev, const sc *s, const size_t *rsizes, size_t rings,
                           const unsigned char *nonce, const ge *commit, const unsigned char *proof, size_t len, const ge *genp) {
    sc s_orig[128], sec[32], stmp; unsigned char prep[4096], tmp[32]; u64 value = 0; size_t offset, i, j, skip1, skip2, npub; int b;
    memset(prep, 0, 4096); memset(s_orig, 0, sizeof(s_orig));
    rp_genrand(sec, s_orig, prep, rsizes, rings, nonce, commit, proof, len, genp);
    *v = UINT64_MAX; sc_set_int(blind, 0);
    if (rings == 1 && rsizes[0] == 1) { rp_recover_x(blind, &s_orig[0], &ev[0], &s[0]); *v = 0; if (mlen) *mlen = 0; return 1; }
    npub = (rings - 1) << 2;
    for (j = 0; j < 2; j++) {
        size_t idx = npub + rsizes[rings - 1] - 1 - j;
        sc_get_b32(tmp, &s[idx]);
        for (b = 0; b < 32; b++) tmp[b] ^= prep[idx * 32 + b];
        if ((tmp[0] & 128) && memcmp(&tmp[16], &tmp[24], 8) == 0 && memcmp(&tmp[8], &tmp[16], 8) == 0) {
            value = 0; for (i = 0; i < 8; i++) value = (value << 8) + tmp[24 + i];
            *v = value; memcpy(&prep[idx * 32], tmp, 32);
            break;
        }
    }
    if (j > 1) { if (mlen) *mlen = 0; return 0; }
    skip1 = rsizes[rings - 1] - 1 - j;
    skip2 = (value >> ((rings - 1) << 1)) & 3;
    if (skip1 == skip2) { if (mlen) *mlen = 0; return 0; }
    if (skip2 >= rsizes[rings - 1]) { if (mlen) *mlen = 0; return 0; }       /* the reference indexes past the ring here (unspecified stack contents) */
    skip1 += (rings - 1) << 2; skip2 += (rings - 1) << 2;
    rp_recover_x(&stmp, &s_orig[skip2], &ev[skip2], &s[skip2]);
    sc_neg(&sec[rings - 1], &sec[rings - 1]);
    sc_add(blind, &stmp, &sec[rings - 1]);
    if (!m || !mlen || *mlen == 0) { if (mlen) *mlen = 0; return 1; }
    offset = 0; npub = 0;
    for (i = 0; i < rings; i++) {
        size_t idx = (value >> (i << 1)) & 3;
        for (j = 0; j < rsizes[i]; j++) {
            if (npub == skip1 || npub == skip2) { npub++; continue; }
            if (idx == j) { sc t; sc_mul(&t, &sec[i], &ev[npub]); sc_add(&stmp, &s[npub], &t); }       /* recover_k :349-356 */
            else stmp = s[npub];
            sc_get_b32(tmp, &stmp);
            for (b = 0; b < 32; b++) tmp[b] ^= prep[npub * 32 + b];
            for (b = 0; b < 32 && offset < *mlen; b++) { m[offset] = tmp[b]; offset++; }
            npub++;
        }
    }
    *mlen = offset;
    return 1;
}
/* rangeproof_impl.h:541-683 with main_impl.h:31-71 and generator/main_impl.h:40-49,266-273; nonce == NULL: verification only */
static int rp_verify_impl(unsigned char *blindout, uint64_t *value_out, unsigned char *message_out, size_t *outlen, const unsigned char *nonce,
                          uint64_t *min_value, uint64_t *max_value, const unsigned char *commit33, const unsigned char *proof, size_t plen,
                          const unsigned char *extra, size_t extra_len, const unsigned char *gen64) {
    gej accj, pubs[128], base; ge c, commit, genp; sc s[128], evalues[128]; sha256 hm; size_t rsizes[32], offset = 0, offset_post_header, rings, npub, i, j; int exp, mantissa, overflow, ret; u64 scale;
    unsigned char signs[31], m[33]; const unsigned char *e0;
    { fe x; fe_set_b32_mod(&x, commit33 + 1); ge_set_xquad(&commit, &x); if (commit33[0] & 1) ge_neg(&commit, &commit); }
    ge_from_b64(&genp, gen64, 0);
    if (!rp_getheader(&offset, &exp, &mantissa, &scale, min_value, max_value, proof, plen)) return 0;
    offset_post_header = offset;
    rings = 1; rsizes[0] = 1; npub = 1;
    if (mantissa != 0) {
        rings = mantissa >> 1;
        for (i = 0; i < rings; i++) rsizes[i] = 4;
        npub = (size_t)(mantissa >> 1) << 2;
        if (mantissa & 1) { rsizes[rings] = 2; npub += 2; rings++; }
    }
    if (plen - offset < 32 * (npub + rings - 1) + 32 + ((rings + 6) >> 3)) return 0;
    sha256_init(&hm);
    ser_point_rp(m, &commit); sha256_write(&hm, m, 33);
    ser_point_rp(m, &genp); sha256_write(&hm, m, 33);
    sha256_write(&hm, proof, offset);
    for (i = 0; i < rings - 1; i++) signs[i] = (proof[offset + (i >> 3)] & (1 << (i & 7))) != 0;
    offset += (rings + 6) >> 3;
    if ((rings - 1) & 7) { if ((proof[offset - 1] >> ((rings - 1) & 7)) != 0) return 0; }
    npub = 0; gej_set_inf(&accj);
    if (*min_value) mul_u64(&accj, *min_value, &genp);
    for (i = 0; i < rings - 1; i++) {
        fe x;
        if (!fe_set_b32_limit(&x, proof + offset) || !ge_set_xquad(&c, &x)) return 0;
        if (signs[i]) ge_neg(&c, &c);
        sha256_write(&hm, &signs[i], 1); sha256_write(&hm, proof + offset, 32);
        gej_set_ge(&pubs[npub], &c); gej_add_ge(&accj, &accj, &c);
        offset += 32; npub += rsizes[i];
    }
    gej_neg(&accj, &accj); gej_add_ge(&pubs[npub], &accj, &commit);
    if (pubs[npub].inf) return 0;
    /* pub_expand (:20-51) */
    { ge ng; int e; size_t k = 0; ge_neg(&ng, &genp); gej_set_ge(&base, &ng);
      for (e = 0; e < (exp < 0 ? 0 : exp); e++) { gej t2, t8; gej_double(&t2, &base); gej_double(&t8, &t2); gej_double(&t8, &t8); gej_add(&base, &t8, &t2); }
      for (i = 0; i < rings; i++) {
          for (j = 1; j < rsizes[i]; j++) gej_add(&pubs[k + j], &pubs[k + j - 1], &base);
          if (i < rings - 1) { gej_double(&base, &base); gej_double(&base, &base); }
          k += rsizes[i];
      } }
    npub += rsizes[rings - 1];
    e0 = proof + offset; offset += 32;
    for (i = 0; i < npub; i++) { sc_set_b32(&s[i], proof + offset, &overflow); if (overflow) return 0; offset += 32; }
    if (offset != plen) return 0;
    if (extra) sha256_write(&hm, extra, extra_len);
    sha256_final(&hm, m);
    ret = borromean_verify(nonce ? evalues : NULL, e0, s, pubs, rsizes, rings, m);
    if (ret && nonce) {                                                 /* :652-680 */
        sc blind; u64 vv; gej t1, t2; sc svv;
        if (!rp_rewind_inner(&blind, &vv, message_out, outlen, evalues, s, rsizes, rings, nonce, &commit, proof, offset_post_header, &genp)) return 0;
        vv = (vv * scale) + *min_value;
        { gej gj; gej_set_ge(&gj, &genp); memset(&svv, 0, sizeof(svv)); sc_set_int(&svv, vv); ecmult(&t1, &gj, &svv, &blind); }   /* blind*G + vv*gen (pedersen_ecmult) */
        if (t1.inf) return 0;
        gej_neg(&t2, &t1); gej_add_ge(&t1, &t2, &commit);
        if (!t1.inf) return 0;
        if (blindout) sc_get_b32(blindout, &blind);
        if (value_out) *value_out = vv;
    }
    return ret;
}
int zo_rangeproof_verify(uint64_t *min_value, uint64_t *max_value, const unsigned char *commit33, const unsigned char *proof, size_t plen,
                         const unsigned char *extra, size_t extra_len, const unsigned char *gen64) {
    return rp_verify_impl(NULL, NULL, NULL, NULL, NULL, min_value, max_value, commit33, proof, plen, extra, extra_len, gen64);
}
int zo_rangeproof_rewind(unsigned char *blind_out, uint64_t *value_out, unsigned char *message_out, size_t *outlen, const unsigned char *nonce32,
                         uint64_t *min_value, uint64_t *max_value, const unsigned char *commit33, const unsigned char *proof, size_t plen,
                         const unsigned char *extra, size_t extra_len, const unsigned char *gen64) {
    return rp_verify_impl(blind_out, value_out, message_out, outlen, nonce32, min_value, max_value, commit33, proof, plen, extra, extra_len, gen64);
}

/* ============================================================ schnorr =================================================== */
/* main_impl.h:215-261 with the tagged challenge hash :106-120; pk32 = x-only serialisation */
int zo_schnorrsig_verify(const unsigned char *sig64, const unsigned char *msg, size_t msglen, const unsigned char *pk32) {
    static const char tag[] = "BIP0340/challenge";
    fe rx, px; sc s, e; ge pk, ra; gej pkj, rj; sha256 h; unsigned char th[32], buf[32]; int overflow;
    if (!fe_set_b32_limit(&rx, sig64)) return 0;
    sc_set_b32(&s, sig64 + 32, &overflow); if (overflow) return 0;
    if (!fe_set_b32_limit(&px, pk32) || !ge_set_xo(&pk, &px, 0)) return 0;
    sha256_init(&h); sha256_write(&h, (const unsigned char *)tag, sizeof(tag) - 1); sha256_final(&h, th);
    sha256_init(&h); sha256_write(&h, th, 32); sha256_write(&h, th, 32);
    sha256_write(&h, sig64, 32); fe_get_b32(buf, &pk.x); sha256_write(&h, buf, 32); sha256_write(&h, msg, msglen); sha256_final(&h, buf);
    sc_set_b32(&e, buf, NULL); sc_neg(&e, &e);
    gej_set_ge(&pkj, &pk); ecmult(&rj, &pkj, &e, &s);
    if (rj.inf) return 0;
    ge_set_gej(&ra, &rj);
    return !fe_is_odd(&ra.y) && fe_equal(&rx, &ra.x);
}

/* ======================================================= half-aggregated schnorr ======================================== */
/* secp256k1_schnorrsig_aggverify, modules/schnorrsig_halfagg/main_impl.h:108-198, statement for statement: the running
 * randomizer hash over r_i | pk_i | m_i with a finalised copy per item (:153-163), T_i = R_i + e_i P_i (:165-181),
 * rhs += z_i T_i with z_0 = 1 (:183-185), accept iff s G == rhs (:187-197).  pks32 = x-only serialisations. */
int zo_schnorrsig_aggverify(const unsigned char *pks32, const unsigned char *msgs32, size_t n, const unsigned char *aggsig, size_t aggsig_len) {
    static const char tag_agg[] = "HalfAgg/randomizer", tag_ch[] = "BIP0340/challenge";
    sha256 hash, copy, h; unsigned char th[32], tc[32], out[32]; gej rhs, lhs; sc s; size_t i; int overflow;
    if ((aggsig_len / 32) == 0 || (aggsig_len / 32) - 1 != n || (aggsig_len % 32) != 0) return 0;
    sha256_init(&h); sha256_write(&h, (const unsigned char *)tag_agg, sizeof(tag_agg) - 1); sha256_final(&h, th);
    sha256_init(&h); sha256_write(&h, (const unsigned char *)tag_ch, sizeof(tag_ch) - 1); sha256_final(&h, tc);
    sha256_init(&hash); sha256_write(&hash, th, 32); sha256_write(&hash, th, 32);      /* the midstate of :12-18 */
    memset(&rhs, 0, sizeof(rhs)); rhs.inf = 1;
    for (i = 0; i < n; i++) {
        fe rx, px; ge rp, pp; gej ppj, ti, t2; sc ei, zi;
        if (!fe_set_b32_limit(&px, pks32 + 32 * i) || !ge_set_xo(&pp, &px, 0)) return 0;       /* the caller's xonly_pubkey_parse */
        sha256_write(&hash, aggsig + 32 * i, 32); sha256_write(&hash, pks32 + 32 * i, 32); sha256_write(&hash, msgs32 + 32 * i, 32);
        copy = hash; sha256_final(&copy, out); sc_set_b32(&zi, out, NULL);
        if (!fe_set_b32_limit(&rx, aggsig + 32 * i)) return 0;
        if (!ge_set_xo(&rp, &rx, 0)) return 0;
        sha256_init(&h); sha256_write(&h, tc, 32); sha256_write(&h, tc, 32);
        sha256_write(&h, aggsig + 32 * i, 32); sha256_write(&h, pks32 + 32 * i, 32); sha256_write(&h, msgs32 + 32 * i, 32); sha256_final(&h, out);
        sc_set_b32(&ei, out, NULL);
        gej_set_ge(&ppj, &pp); ecmult(&ti, &ppj, &ei, NULL);
        gej_add_ge(&t2, &ti, &rp); ti = t2;
        if (i != 0) { ecmult(&t2, &ti, &zi, NULL); ti = t2; }
        gej_add(&t2, &rhs, &ti); rhs = t2;
    }
    sc_set_b32(&s, aggsig + 32 * n, &overflow); if (overflow) return 0;
    { gej gj, ng; memset(&gj, 0, sizeof(gj)); gj.inf = 1; ecmult(&lhs, &gj, NULL, &s); gej_neg(&ng, &lhs); gej_add(&lhs, &ng, &rhs); }
    return lhs.inf;
}

/* ========================================================== pedersen tally =============================================== */
/* secp256k1_pedersen_verify_tally, modules/generator/main_impl.h:371-396: acc = -(sum of the negative list), then add the positive
 * list, accept iff infinity.  Commitments as 33-byte serialisations, decoded like secp256k1_pedersen_commitment_load (:266-273);
 * returns -1 for an encoding secp256k1_pedersen_commitment_parse would refuse. */
int zo_pedersen_verify_tally(const unsigned char *pos33, size_t pcnt, const unsigned char *neg33, size_t ncnt) {
    gej acc, t; ge add; fe x; size_t i;
    gej_set_inf(&acc);
    for (i = 0; i < ncnt; i++) {
        if ((neg33[33 * i] & 0xFE) != 8 || !fe_set_b32_limit(&x, neg33 + 33 * i + 1) || !ge_set_xquad(&add, &x)) return -1;
        if (neg33[33 * i] & 1) ge_neg(&add, &add);
        gej_add_ge(&t, &acc, &add); acc = t;
    }
    gej_neg(&t, &acc); acc = t;
    for (i = 0; i < pcnt; i++) {
        if ((pos33[33 * i] & 0xFE) != 8 || !fe_set_b32_limit(&x, pos33 + 33 * i + 1) || !ge_set_xquad(&add, &x)) return -1;
        if (pos33[33 * i] & 1) ge_neg(&add, &add);
        gej_add_ge(&t, &acc, &add); acc = t;
    }
    return acc.inf;
}

/* ============================================================= bppp ===================================================== */
static int parse33(ge *p, const unsigned char *in) {                   /* eckey_impl.h:18-22 */
    fe x;
    if (in[0] != 2 && in[0] != 3) return 0;
    if (!fe_set_b32_limit(&x, in + 1)) return 0;
    return ge_set_xo(p, &x, in[0] == 3);
}
static int parse_ext(ge *p, const unsigned char *in33) {               /* secp256k1.c:895-903 */
    int i, nz = 0; for (i = 0; i < 33; i++) nz |= in33[i];
    if (!nz) { memset(p, 0, sizeof(*p)); p->inf = 1; return 1; }
    return parse33(p, in33);
}
static int parse_one_of_points(ge *p, const unsigned char *in65, int idx) {      /* bppp_util.h:30-46 */
    unsigned char tmp[33]; int i, nz = 0;
    memset(tmp, 0, 33);
    if (in65[0] > 3) return 0;
    for (i = 0; i < 32; i++) nz |= in65[1 + 32 * idx + i];
    if (nz) { tmp[0] = 2 | ((in65[0] & (2 - idx)) >> (1 - idx)); memcpy(tmp + 1, in65 + 1 + 32 * idx, 32); }
    else if (in65[0] & (2 - idx)) return 0;
    return parse_ext(p, tmp);
}
static size_t log2sz(size_t n) { size_t l = 0; while (((size_t)2 << l) <= n) l++; return l; }
/* bppp_norm_product_impl.h:425-552; transcript104 = the reference's secp256k1_sha256 object bytes */
int zo_bppp_norm_verify(const unsigned char *proof, size_t proof_len, const unsigned char *transcript104, const unsigned char *rho32,
                        const unsigned char *gens33, size_t n_gens, size_t g_len, const unsigned char *c_vec32, size_t c_len, const unsigned char *commit33) {
    sc rho, rho_f, mu_f, v, n, l, rho_inv, h_c, *gammas, *s_g, *s_h, *pw, *msc; ge *mpt, commit; gej res1, res2, d; sha256 tr;
    size_t i, log_g, log_h, n_rounds, h_len = c_len; int overflow, ok = 1;
    if (g_len == 0 || c_len == 0) return 0;
    log_g = log2sz(g_len); log_h = log2sz(c_len); n_rounds = log_g > log_h ? log_g : log_h;
    if (n_gens != h_len + g_len || proof_len != 65 * n_rounds + 64) return 0;
    if ((g_len & (g_len - 1)) || (h_len & (h_len - 1))) return 0;
    sc_set_b32(&n, proof + n_rounds * 65, &overflow); if (overflow) return 0;
    sc_set_b32(&l, proof + n_rounds * 65 + 32, &overflow); if (overflow) return 0;
    sc_set_b32(&rho, rho32, NULL); if (sc_is_zero(&rho)) return 0;
    if (!parse_ext(&commit, commit33)) return 0;
    memcpy(tr.s, transcript104, 32); memcpy(tr.buf, transcript104 + 32, 64); memcpy(&tr.bytes, transcript104 + 96, 8);
    gammas = (sc *)malloc((n_rounds + 1) * sizeof(sc)); s_g = (sc *)malloc(g_len * sizeof(sc)); s_h = (sc *)malloc(h_len * sizeof(sc)); pw = (sc *)malloc((log_g + 1) * sizeof(sc));
    sc_inverse(&rho_inv, &rho);
    if (log_g) { pw[0] = rho_inv; for (i = 1; i < log_g; i++) sc_mul(&pw[i], &pw[i - 1], &pw[i - 1]); }
    rho_f = rho; for (i = 0; i < log_g; i++) sc_mul(&rho_f, &rho_f, &rho_f);
    for (i = 0; i < n_rounds; i++) {
        sha256 c; unsigned char le[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dg[32];
        sha256_write(&tr, proof + i * 65, 65);
        c = tr; sha256_write(&c, le, 8); sha256_final(&c, dg); sc_set_b32(&gammas[i], dg, NULL);
    }
    sc_mul(&s_g[0], &n, &rho_f); sc_mul(&s_g[0], &s_g[0], &rho_inv);
    for (i = 1; i < g_len; i++) { size_t li = log2sz(i), p2 = (size_t)1 << li; sc_mul(&s_g[i], &s_g[i - p2], &gammas[li]); sc_mul(&s_g[i], &s_g[i], &pw[li]); }
    s_h[0] = l;
    for (i = 1; i < h_len; i++) { size_t li = log2sz(i), p2 = (size_t)1 << li; sc_mul(&s_h[i], &s_h[i - p2], &gammas[li]); }
    sc_set_int(&h_c, 0);
    for (i = 0; i < h_len; i++) { sc c, t; sc_set_b32(&c, c_vec32 + 32 * i, NULL); sc_mul(&t, &c, &s_h[i]); sc_add(&h_c, &h_c, &t); }
    sc_mul(&mu_f, &rho_f, &rho_f); sc_mul(&v, &n, &n); sc_mul(&v, &v, &mu_f); sc_add(&v, &v, &h_c);
    /* MSM 1 (:375-402, :531) */
    msc = (sc *)malloc((2 * n_rounds + 1 + g_len + h_len) * sizeof(sc)); mpt = (ge *)malloc((2 * n_rounds + 1 + g_len + h_len) * sizeof(ge));
    sc_set_int(&msc[0], 1); mpt[0] = commit;
    for (i = 0; i < n_rounds && ok; i++) {
        sc one, m1; sc_set_int(&one, 1); sc_neg(&m1, &one);
        msc[1 + 2 * i] = gammas[i]; ok &= parse_one_of_points(&mpt[1 + 2 * i], proof + 65 * i, 0);
        sc_mul(&msc[2 + 2 * i], &gammas[i], &gammas[i]); sc_add(&msc[2 + 2 * i], &msc[2 + 2 * i], &m1);
        if (ok) ok &= parse_one_of_points(&mpt[2 + 2 * i], proof + 65 * i, 1);
    }
    if (ok) {
        ecmult_multi(&res1, NULL, msc, mpt, 2 * n_rounds + 1);
        /* MSM 2 (:404-420, :543) */
        for (i = 0; i < g_len + h_len && ok; i++) { msc[i] = i < g_len ? s_g[i] : s_h[i - g_len]; ok &= parse33(&mpt[i], gens33 + 33 * i); }
        if (ok) {
            ecmult_multi(&res2, &v, msc, mpt, g_len + h_len);
            gej_neg(&d, &res1); gej_add(&d, &d, &res2);          /* gej_eq_var(res1, res2) (:551) */
            ok = d.inf;
        }
    }
    free(gammas); free(s_g); free(s_h); free(pw); free(msc); free(mpt);
    return ok;
}

/* ========================================================== surjection =================================================== */
/* secp256k1_surjectionproof_parse + _verify on the wire format (modules/surjection/main_impl.h:45-82,360-402;
 * surjection_impl.h:19-37 message, :66-95 public keys); tags are 64-byte generators x||y */
int zo_surjectionproof_verify(const unsigned char *proof, size_t plen, const unsigned char *in_tags64, size_t n_tags, const unsigned char *out_tag64) {
    size_t n_inputs, bm, n_used = 0, i, j = 0, rsizes[1]; const unsigned char *data; sha256 h; unsigned char m[32], t33[33];
    static gej pubs[256]; static sc s[256]; ge out; int overflow;
    if (plen < 2) return 0;
    n_inputs = ((size_t)proof[1] << 8) + proof[0];
    if (n_inputs > 256) return 0;
    bm = (n_inputs + 7) / 8;
    if (plen < 2 + bm) return 0;
    if (n_inputs % 8 != 0 && (proof[2 + bm - 1] & (unsigned char)(0xFFu << (n_inputs % 8)))) return 0;
    for (i = 0; i < bm; i++) { unsigned b = proof[2 + i]; while (b) { n_used += b & 1; b >>= 1; } }
    if (plen != 2 + bm + 32 * (1 + n_used)) return 0;
    if (n_used == 0 || n_used > n_inputs || n_inputs != n_tags) return 0;
    data = proof + 2 + bm;
    ge_from_b64(&out, out_tag64, 0);
    for (i = 0; i < n_inputs; i++) {
        if (proof[2 + i / 8] & (1 << (i % 8))) { ge tin; ge_from_b64(&tin, in_tags64 + 64 * i, 0); ge_neg(&tin, &tin); gej_set_ge(&pubs[j], &tin); gej_add_ge(&pubs[j], &pubs[j], &out); j++; }
    }
    for (i = 0; i < n_used; i++) { sc_set_b32(&s[i], data + 32 + 32 * i, &overflow); if (overflow) return 0; }
    sha256_init(&h);
    for (i = 0; i <= n_tags; i++) { const unsigned char *t = i < n_tags ? in_tags64 + 64 * i : out_tag64; t33[0] = 2 + (t[63] & 1); memcpy(t33 + 1, t, 32); sha256_write(&h, t33, 33); }
    sha256_final(&h, m);
    rsizes[0] = n_used;
    return borromean_verify(NULL, data, s, pubs, rsizes, 1, m);
}

/* ======================================================= byte-level exports ============================================= */
void zo_fe_mul(unsigned char *r, const unsigned char *a, const unsigned char *b) { fe x, y; fe_set_b32_mod(&x, a); fe_set_b32_mod(&y, b); fe_mul(&x, &x, &y); fe_get_b32(r, &x); }
void zo_fe_inv(unsigned char *r, const unsigned char *a) { fe x; fe_set_b32_mod(&x, a); fe_inv(&x, &x); fe_get_b32(r, &x); }
int zo_fe_sqrt(unsigned char *r, const unsigned char *a) { fe x, y; int ok; fe_set_b32_mod(&x, a); ok = fe_sqrt(&y, &x); fe_get_b32(r, &y); return ok; }
void zo_scalar_mul(unsigned char *r, const unsigned char *a, const unsigned char *b) { sc x, y; sc_set_b32(&x, a, NULL); sc_set_b32(&y, b, NULL); sc_mul(&x, &x, &y); sc_get_b32(r, &x); }
void zo_scalar_split_lambda(unsigned char *r1, unsigned char *r2, const unsigned char *k) { sc a, b, x; sc_set_b32(&x, k, NULL); sc_split_lambda(&a, &b, &x); sc_get_b32(r1, &a); sc_get_b32(r2, &b); }
int zo_ge_add(unsigned char *r64, const unsigned char *a64, int ainf, const unsigned char *b64, int binf) {
    ge a, b; gej j; ge_from_b64(&a, a64, ainf); ge_from_b64(&b, b64, binf); gej_set_ge(&j, &a); gej_add_ge(&j, &j, &b); return gej_to_b64(r64, &j);
}
int zo_ecmult(unsigned char *r64, const unsigned char *a64, int ainf, const unsigned char *na32, const unsigned char *ng32) {
    ge a; gej aj, rj; sc na, ng; ge_from_b64(&a, a64, ainf); gej_set_ge(&aj, &a); sc_set_b32(&na, na32, NULL); if (ng32) sc_set_b32(&ng, ng32, NULL);
    ecmult(&rj, &aj, &na, ng32 ? &ng : NULL); return gej_to_b64(r64, &rj);
}
int zo_ecmult_multi(unsigned char *r64, const unsigned char *g_sc32, const unsigned char *sc32, const unsigned char *pt64, const unsigned char *inf, size_t n) {
    sc *s = (sc *)malloc((n + 1) * sizeof(sc)), g; ge *p = (ge *)malloc((n + 1) * sizeof(ge)); gej r; size_t i; int ret;
    for (i = 0; i < n; i++) { sc_set_b32(&s[i], sc32 + 32 * i, NULL); ge_from_b64(&p[i], pt64 + 64 * i, inf ? inf[i] : 0); }
    if (g_sc32) sc_set_b32(&g, g_sc32, NULL);
    ecmult_multi(&r, g_sc32 ? &g : NULL, s, p, n);
    ret = gej_to_b64(r64, &r); free(s); free(p); return ret;
}
void zo_sha256(unsigned char *out32, const unsigned char *msg, size_t len) { sha256 h; sha256_init(&h); sha256_write(&h, msg, len); sha256_final(&h, out32); }
void zo_rangeproof_verify_many(int *results, uint64_t *min_v, uint64_t *max_v, const unsigned char *commits33, const unsigned char *proofs, size_t stride,
                               const size_t *plens, const unsigned char *gens64, size_t n, int threads) {
    long i; (void)threads;
#ifdef _OPENMP
    #pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
#endif
    for (i = 0; i < (long)n; i++) { min_v[i] = 0; max_v[i] = 0; results[i] = zo_rangeproof_verify(&min_v[i], &max_v[i], commits33 + 33 * i, proofs + stride * i, plens[i], NULL, 0, gens64 + 64 * i); }
}
