// group.h -- secp256k1 group law, one point per lane, all coordinates in VGPRs (fe.h limbs).
//
// Role of the reference's src/group_impl.h: gej_double (:468-501), gej_add_var (:534-596),
// gej_add_ge_var (:598-659), ge_set_gej_var (:177-196), ge_set_xquad (:347-355), ge_set_xo_var (:357-373),
// ge_mul_lambda (:925-932).  The formulas are the standard a=0 Jacobian ones; what is different from the
// reference is everything around them:
//   * the `_var` early-outs (a = inf, b = inf, same x) become per-lane flags + selects, so a wavefront never
//     branches on point data; the one case that needs *different arithmetic* (P + P) is reported to the caller
//     (`GEJ_ADD_NEEDS_DOUBLE`), which re-issues it through its single doubling site (ecmult.h);
//   * magnitudes follow the 9x29 contract of fe.h (product of input magnitudes <= 7), with weak
//     normalisations placed where the contract needs them rather than where the 5x52 code has them.
// Output contract: gej_double -> (x,y,z) magnitudes (5,3,1); gej_add_ge -> (1,3,1); both accept (5,3,1) inputs.
#pragma once
#include "fe.h"

struct ge  { fe x, y; };                 // affine, never infinity (callers carry a flag where needed)
struct gej { fe x, y, z; int inf; };     // Jacobian

#define GEJ_ADD_NEEDS_DOUBLE 1

// beta: cube root of unity in Fp, lambda*(x,y) = (beta*x, y)  (group_impl.h:925-932)
S2K_HD void fe_set_beta(fe& r) {
    const u32 b[9] = {0x119501EEu, 0x09CB6143u, 0x1D626570u, 0x0092EA25u, 0x034E99CFu, 0x03CF561Au, 0x1C41B991u, 0x056CAF80u, 0x007AE96Au};
#pragma unroll
    for (int i = 0; i < 9; i++) r.n[i] = b[i];
}

S2K_HD void gej_set_infinity(gej& r) { fe_set_zero(r.x); fe_set_zero(r.y); fe_set_zero(r.z); r.inf = 1; }
S2K_HD void gej_set_ge(gej& r, const ge& a) { r.x = a.x; r.y = a.y; fe_set_int(r.z, 1); r.inf = 0; }
S2K_HD void ge_neg(ge& r, const ge& a) { r.x = a.x; fe_neg(r.y, a.y, 1); fe_norm_weak(r.y); }

// r = 2a.  3M + 4S + half (cf. secp256k1_gej_double, group_impl.h:468-501: L = 3/2 X^2, S = Y^2, T = -X S,
// X3 = L^2 + 2T, Y3 = -(L (X3 + T) + S^2), Z3 = Y Z).  Input magnitudes up to (7,7,7); infinity stays infinity
// (Z3 = Y*0), the flag is carried through unchanged.
S2K_HD void gej_double(gej& r, const gej& a) {
    fe x = a.x, y = a.y, l, s, t;
    fe_norm_weak(x); fe_norm_weak(y);
    // the seven products go out as three lockstep pairs (fe_dual) and one single
    fe_mul_sqr(r.z, y, a.z, s, y);             // Z3 = Y*Z, S = Y^2    (1, 1)
    fe_mul_sqr(t, x, s, l, x);                 // X*S, X^2             (1, 1)
    fe_neg(t, t, 1);                           // T = -X*S             (2)
    fe_mul_int(l, 3); fe_half(l);              // L = 3/2 X^2         (<= 2.5)
    fe_norm_weak(l);                           //                     (1)
    fe_sqr2(r.x, l, s, s);                     // L^2, S^2            (1, 1)
    fe_add(r.x, t); fe_add(r.x, t);            // X3 = L^2 + 2T       (5)
    fe_add(t, r.x);                            // X3 + T              (7)
    fe_mul(r.y, t, l);                         // L*(X3+T)            (1)   7*1 <= 7
    fe_add(r.y, s);                            //                     (2)
    fe_neg(r.y, r.y, 2);                       // Y3                  (3)
    r.inf = a.inf;
}

// r = a + b, b affine and finite.  8M + 3S (cf. secp256k1_gej_add_ge_var, group_impl.h:598-659).
// Returns GEJ_ADD_NEEDS_DOUBLE when a == b (then r is set to b with Z = 1 and the caller must double it);
// a == -b gives r.inf = 1; a.inf gives r = b.  Inputs: a magnitudes up to (5,3,1), b up to (1,2).  Output (1,3,1).
S2K_HD int gej_add_ge(gej& r, const gej& a, const ge& b, fe* zr = nullptr) {
    fe z12, u2, s2, h, i, h2, h3, t, i2;
    fe_sqr(z12, a.z);
    fe_mul2(u2, b.x, z12, s2, b.y, z12);       // lockstep pairs of independent products (fe_dual)
    fe_neg(h, a.x, 5); fe_add(h, u2);          // h = u2 - X1         (7)
    fe_norm_seq(h);                            // exact limbs: needed for the zero test, and magnitude 1 for the products
    fe zz; fe_mul2(s2, s2, a.z, zz, a.z, h);   // s2 = Y2*Z1^3, Z3 = Z1*h
    fe_neg(i, a.y, 3); fe_add(i, s2);          // i = s2 - Y1         (5)
    fe_norm_seq(i);
    const int hz = fe_seq_is_zero(h), iz = fe_seq_is_zero(i);
    fe_sqr2(i2, i, h2, h);
    fe_mul2(h3, h, h2, t, a.x, h2);            // h^3, t = X1*h2      (6*1)
    if (zr) *zr = h;                           // Z3 / Z1, for global-Z table construction
    fe x3, y3, tn;
    fe_neg(x3, h3, 1);                         // -h3                 (2)
    fe_neg(tn, t, 1);                          // -t                  (2)
    fe_add(x3, tn); fe_add(x3, tn); fe_add(x3, i2);   // X3 = i2 - h3 - 2t   (7)
    fe_norm_weak(x3);                          //                     (1)
    fe_neg(tn, x3, 1); fe_add(tn, t);          // t - X3              (3)
    fe_mul2(y3, tn, i, h3, h3, a.y);           // i*(t - X3), Y1*h3   (3*1, 1*5... <= 7)
    fe_neg(h3, h3, 1);
    fe_add(y3, h3);                            // Y3                  (3)
    // case resolution (per lane, no branches)
    const int dbl = (!a.inf) & hz & iz;
    const int inf = (!a.inf) & hz & (!iz);
    const int take_b = a.inf | dbl;
    fe one; fe_set_int(one, 1);
    fe_select(r.x, b.x, x3, take_b);
    fe_select(r.y, b.y, y3, take_b);
    fe_select(r.z, one, zz, take_b);
    r.inf = inf;
    return dbl ? GEJ_ADD_NEEDS_DOUBLE : 0;
}

// ---- lean forms for the lock-step main loop (ecmult.h) ------------------------------------------------------------
// Same formulas, no case analysis, the Y3 line as one fused product pair (fe_muladd) and the weak normalisations placed so
// that none is repeated at the next operation's entry.  Contract (both directions): X magnitude 1, Y magnitude <= 2, Z
// magnitude 1, point finite.  What they do NOT handle is reported or excluded by the caller: gej_double_lean assumes a finite
// point (a doubled point of odd order is never infinity); gej_add_ge_lean returns 1 when the operands share their x
// coordinate (P + P or P - P) and the caller redoes that addition through gej_add_ge.
S2K_HD void gej_double_lean(gej& r, const gej& a) {
    fe l, s, t, nx, w;
    fe_mul_sqr(r.z, a.y, a.z, s, a.y);         // Z3 = Y*Z (2*1), S = Y^2 (mag 2 ok)          -> (1, 1)
    fe_neg(nx, a.x, 1);                        // -X                                           (2)
    fe_mul_sqr(t, nx, s, l, a.x);              // T = -X*S (2*1), X^2                          -> (1, 1)
    fe_mul_int(l, 3); fe_half(l);              // L = 3/2 X^2                                  (<= 2)
    fe_norm_weak(l);                           //                                              (1)
    fe_sqr(r.x, l);                            // L^2                                          (1)
    fe_add(r.x, t); fe_add(r.x, t);            // X3 = L^2 + 2T                                (3)
    fe_norm_weak(r.x);                         //                                              (1)
    fe_add2(w, r.x, t);                        // X3 + T                                       (2)
    fe_muladd<false, true>(r.y, l, w, s, s);   // L*(X3+T) + S^2 : 1*2 + 1*1 <= 7              (1)
    fe_neg(r.y, r.y, 1);                       // Y3                                           (2)
    r.inf = 0;
}
// r = a + b (b affine, magnitudes (1, <= 2)); returns 1 iff a and b have the same x (then r is meaningless).
S2K_HD int gej_add_ge_lean(gej& r, const gej& a, const ge& b) {
    fe z12, u2, s2, h, i, h2, h3, t, i2, ny, w;
    fe_sqr(z12, a.z);
    fe_mul2(u2, b.x, z12, s2, b.y, z12);       // U2 = x2*Z1^2, y2*Z1^2 (2*1)
    fe_neg(h, a.x, 1); fe_add(h, u2);          // h = U2 - X1                                  (3)
    fe_norm_seq(h);                            // exact limbs for the zero test, magnitude 1
    fe_mul2(s2, s2, a.z, r.z, a.z, h);         // S2 = y2*Z1^3, Z3 = Z1*h
    fe_neg(ny, a.y, 2);                        // -Y1                                          (3)
    fe_add2(i, s2, ny);                        // i = S2 - Y1                                  (4)
    fe_norm_weak(i);                           //                                              (1)
    const int hz = fe_seq_is_zero(h);
    fe_sqr2(i2, i, h2, h);
    fe_mul2(h3, h, h2, t, a.x, h2);            // h^3, t = X1*h^2
    fe_add2(w, t, t); fe_add(w, h3);           // h^3 + 2t                                     (3)
    fe_neg(w, w, 3);                           //                                              (4)
    fe_add2(r.x, i2, w);                       // X3 = i^2 - h^3 - 2t                          (5)
    fe_norm_weak(r.x);                         //                                              (1)
    fe_neg(w, r.x, 1); fe_add(w, t);           // t - X3                                       (3)
    fe_muladd<false, false>(r.y, i, w, ny, h3);    // Y3 = i*(t - X3) - Y1*h^3 : 1*3 + 3*1 <= 7    (1)
    r.inf = 0;
    return hz;
}

// r = a + b, both Jacobian.  12M + 4S (cf. secp256k1_gej_add_var, group_impl.h:534-596).  Complete: handles
// infinity, a == b (by doubling -- this function is only used in cold prologue/epilogue code) and a == -b.
// Inputs magnitudes up to (5,3,1).
S2K_HD void gej_add_var(gej& r, const gej& a, const gej& b) {
    fe z22, z12, u1, u2, s1, s2, h, i, h2, h3, t, i2;
    // all 16 products as lockstep pairs (fe_dual): this function runs in single-wave, latency-bound tails where the
    // dependent-MAC wait states are fully exposed
    fe_sqr2(z22, b.z, z12, a.z);
    fe_mul2(u1, a.x, z22, u2, b.x, z12);
    fe_mul2(s1, a.y, z22, s2, b.y, z12);
    fe_mul2(s1, s1, b.z, s2, s2, a.z);
    fe_neg(h, u1, 1); fe_add(h, u2);
    fe_neg(i, s1, 1); fe_add(i, s2);
    fe_norm_seq(h); fe_norm_seq(i);
    const int hz = fe_seq_is_zero(h), iz = fe_seq_is_zero(i);
    gej res;
    if ((!a.inf) & (!b.inf) & hz & iz) {
        gej_double(res, a);
    } else {
        fe zz;
        fe_mul_sqr(zz, a.z, b.z, h2, h);
        fe_mul_sqr(res.z, zz, h, i2, i);
        fe_mul2(h3, h, h2, t, u1, h2);
        fe x3, tn;
        fe_neg(x3, h3, 1); fe_neg(tn, t, 1);
        fe_add(x3, tn); fe_add(x3, tn); fe_add(x3, i2);
        fe_norm_weak(x3);
        fe_neg(tn, x3, 1); fe_add(tn, t);
        fe_mul2(res.y, tn, i, h3, h3, s1);
        fe_neg(h3, h3, 1);
        fe_add(res.y, h3);
        res.x = x3;
        res.inf = hz & (!iz);
        if (a.inf) res = b;
        else if (b.inf) res = a;
    }
    r = res;
}

// Jacobian -> affine: one inversion + 1S + 3M (cf. secp256k1_ge_set_gej_var :177-196).  Output normalised.
// Caller must handle a.inf.
S2K_HD void ge_set_gej(ge& r, const gej& a) {
    fe zi, zi2, zi3;
    fe_inv(zi, a.z);
    fe_sqr(zi2, zi); fe_mul(zi3, zi2, zi);
    fe_mul(r.x, a.x, zi2); fe_mul(r.y, a.y, zi3);
    fe_normalize(r.x); fe_normalize(r.y);
}

// y^2 = x^3 + 7
S2K_HD void ge_curve_rhs(fe& c, const fe& x) {
    fe x2; fe_sqr(x2, x); fe_mul(c, x2, x);
    c.n[0] += 7u;
}
// lift x to the point whose y is a quadratic residue (cf. secp256k1_ge_set_xquad :347-355).
// Returns 1 iff x is on the curve; y = (x^3+7)^((p+1)/4) is produced either way, as in the reference.
S2K_HD int ge_set_xquad(ge& r, const fe& x) {
    fe c; ge_curve_rhs(c, x);
    fe_norm_weak(c);
    r.x = x;
    return fe_sqrt(r.y, c);
}
// lift x with chosen parity (cf. secp256k1_ge_set_xo_var :357-373)
S2K_HD int ge_set_xo(ge& r, const fe& x, int odd) {
    if (!ge_set_xquad(r, x)) return 0;
    fe_normalize(r.y);
    if (fe_is_odd(r.y) != odd) { fe_neg(r.y, r.y, 1); fe_normalize(r.y); }
    return 1;
}
// y^2 == x^3 + 7 ?  (cf. secp256k1_ge_is_valid_var)
S2K_HD int ge_is_valid(const ge& a) {
    fe y2, c; fe_sqr(y2, a.y); ge_curve_rhs(c, a.x); fe_norm_weak(c);
    return fe_equal(y2, c);
}

// ---- extended Jacobian ("XYZZ") accumulators: bucket sums of the MSM (msm.h) and the table parts of the double multiplications (ecmult.h) ----
// Bucket accumulator in extended Jacobian ("XYZZ") coordinates: x = X/ZZ, y = Y/ZZZ with ZZ^3 = ZZZ^2.  Adding an affine point
// costs 8M + 2S (one squaring less than the Jacobian mixed addition) and all ten products go out as lockstep pairs.
// Magnitudes: X <= 1, Y <= 3, ZZ, ZZZ 1.
struct gez { fe x, y, zz, zzz; int inf; };
S2K_HD void gez_add_ge(gez& a, const ge& b) {
    if (a.inf) { a.x = b.x; a.y = b.y; fe_set_int(a.zz, 1); fe_set_int(a.zzz, 1); a.inf = 0; return; }
    fe u2, s2, p, r;
    fe_mul2(u2, b.x, a.zz, s2, b.y, a.zzz);
    fe_neg(p, a.x, 1); fe_add(p, u2);              // P = U2 - X1      (3)
    fe_neg(r, a.y, 3); fe_add(r, s2);              // R = S2 - Y1      (5)
    fe_norm_seq(p); fe_norm_seq(r);
    if (fe_seq_is_zero(p)) {
        if (!fe_seq_is_zero(r)) { a.inf = 1; return; }                 // b == -a
        gej t, d; gej_set_ge(t, b); gej_double(d, t);                  // b == a: 2b, back to XYZZ
        a.x = d.x; a.y = d.y; fe_norm_weak(a.x); fe_norm_weak(a.y);
        fe_sqr(a.zz, d.z); fe_mul(a.zzz, a.zz, d.z);
        return;
    }
    fe pp, rr, ppp, q;
    fe_sqr2(pp, p, rr, r);
    fe_mul2(ppp, p, pp, q, a.x, pp);
    fe x3, t1, nq, y3a, y3b;
    fe_neg(x3, ppp, 1); fe_neg(nq, q, 1);
    fe_add(x3, nq); fe_add(x3, nq); fe_add(x3, rr); // X3 = R^2 - PPP - 2Q (7)
    fe_norm_weak(x3);
    fe_neg(t1, x3, 1); fe_add(t1, q);              // Q - X3           (3)
    fe_mul2(y3a, r, t1, y3b, a.y, ppp);            // (1*3), (3*1)
    fe_mul2(a.zz, a.zz, pp, a.zzz, a.zzz, ppp);
    fe_neg(y3b, y3b, 1); fe_add(y3a, y3b);         // Y3               (3)
    a.x = x3; a.y = y3a;
}
S2K_HD void gej_set_gez(gej& r, const gez& a) {    // (X*ZZ, Y*ZZZ, ZZ) is the same point in Jacobian coordinates
    r.inf = a.inf;
    if (a.inf) { fe_set_zero(r.x); fe_set_zero(r.y); fe_set_zero(r.z); return; }
    fe_mul2(r.x, a.x, a.zz, r.y, a.y, a.zzz);
    r.z = a.zz;
}
// ---- the lean form of the same accumulation (round 5) --------------------------------------------------------------------------------
// No case analysis per addition.  An exceptional addition (the operand has the accumulator's x: P + P or P - P) makes P = U2 - X1 == 0,
// hence ZZ3 = ZZ * P^2 == 0 -- and ZZ then STAYS 0 through every later addition of the run (a product with a zero factor), while a run
// without one keeps ZZ != 0 (a product of non-zero field elements).  So one zero test of ZZ at the END of a run tells whether any of its
// additions was exceptional, and only then is the run summed again by the exact form above (adversarial inputs only: equal or opposite
// points in one bucket).  Besides the two zero tests this drops the sequential carry passes (P and R only need a weak normalisation as
// inputs of the squarings) and pays ONE reduction for Y3 = R (Q - X3) - Y1 PPP (fe_muladd).  Static count of round 1's loop body: 1 956 ->
// 1 5xx VALU instructions per bucket addition.
// Magnitudes: a.x, a.y, a.zz, a.zzz 1 on entry and on exit; b.x 1, b.y <= 2.
S2K_HD void gez_add_ge_lean(gez& a, const ge& b) {
    fe u2, s2, p, r;
    fe_mul2(u2, b.x, a.zz, s2, b.y, a.zzz);
    fe_neg(p, a.x, 1); fe_add(p, u2); fe_norm_weak(p);          // P = U2 - X1      (3 -> 1)
    fe_neg(r, a.y, 1); fe_add(r, s2); fe_norm_weak(r);          // R = S2 - Y1      (3 -> 1)
    fe pp, rr, ppp, q;
    fe_sqr2(pp, p, rr, r);
    fe_mul2(ppp, p, pp, q, a.x, pp);
    fe x3, nq, t1, ny;
    fe_neg(x3, ppp, 1); fe_neg(nq, q, 1);
    fe_add(x3, nq); fe_add(x3, nq); fe_add(x3, rr);             // X3 = R^2 - PPP - 2Q (7)
    fe_norm_weak(x3);
    fe_neg(t1, x3, 1); fe_add(t1, q);                           // Q - X3           (3)
    fe_neg(ny, a.y, 1);                                         // -Y1              (2)
    fe_muladd<false, false>(a.y, r, t1, ny, ppp);               // Y3 = R (Q - X3) - Y1 PPP: 1*3 + 2*1 = 5 <= 7, one reduction
    fe_mul2(a.zz, a.zz, pp, a.zzz, a.zzz, ppp);
    a.x = x3;
}
