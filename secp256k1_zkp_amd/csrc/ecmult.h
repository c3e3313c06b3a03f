// ecmult.h -- per-lane double multiplication  R = na*A + ng*G  (the reference's secp256k1_ecmult,
// src/ecmult_impl.h:365-375 -> strauss_wnaf :252-363) re-designed for a 64-wide SIMT machine.
//
// The reference interleaves wNAF(5) of the GLV halves of na with wNAF(15) of ng and walks 129 doublings; its
// per-call table (8 odd multiples + lambda copies) lives on the CPU stack.  On gfx950 a per-lane table cannot be
// indexed in registers and a sparse wNAF wastes lanes (an add slot costs the whole wavefront whenever *any* lane
// has a digit), so the schedule here is:
//
//   variable point:  GLV split (scalar.h) -> joint sparse form of (k1, k2)  -> 4-entry co-Z table
//                    {P, Q=lambda P, P+Q, P-Q} held in VGPRs and selected with v_cndmask (no memory, no LDS);
//                    the table shares one Z, so the loop runs on the isomorphic curve exactly like the
//                    reference's global-Z trick (ecmult_impl.h:289-320) and fixes Z once at the end;
//   generator:       no doublings at all: ng is cut into 32 bytes and each byte indexes a precomputed
//                    (window, byte) -> affine multiple table in HBM/L2 (gtable.h), 32 mixed adds;
//   control:         one loop whose body contains exactly ONE doubling site and ONE mixed-add site, driven by a
//                    per-lane state machine.  The only data-dependent *arithmetic* case (P + P inside an add)
//                    is turned into "take the operand and double it on the next trip", so exceptional inputs
//                    cost one extra iteration for that lane instead of a second copy of the doubling code.
//
// Results are identical to the reference as group elements (and therefore as serialised bytes).
#pragma once
#include "group.h"
#include "scalar.h"

// ---- generator table ---------------------------------------------------------------------------------
// gtab[(w*256 + b)*18 .. +18) = affine (x limbs[9], y limbs[9]) of  b * 256^w * G ,  b = 1..255, w = 0..31.
#define S2K_GTAB_WINDOWS 32
#define S2K_GTAB_ENTRY_WORDS 18
#define S2K_GTAB_WORDS (S2K_GTAB_WINDOWS * 256 * S2K_GTAB_ENTRY_WORDS)

S2K_HD void gtab_load(ge& r, const u32* gtab, u32 window, u32 byte) {
    const u32* p = gtab + (size_t)(window * 256u + byte) * S2K_GTAB_ENTRY_WORDS;
#pragma unroll
    for (int i = 0; i < 9; i++) { r.x.n[i] = p[i]; r.y.n[i] = p[9 + i]; }
}

// ---- joint sparse form ---------------------------------------------------------------------------------
// Digits (u1,u2) in {-1,0,1}^2 of two 129-bit magnitudes, least significant first (Solinas 2001).  Each digit pair
// is packed in a nibble  [bit0: u1 != 0, bit1: u1 < 0, bit2: u2 != 0, bit3: u2 < 0]  and pushed in at the TOP of a
// 17-word shift register, so that after S2K_JSF_LEN steps the most significant digit sits in the top nibble and
// the main loop can pop digits MSB-first with the same (wave-uniform) shift -- no indexed register access.
#define S2K_JSF_LEN 130
#define S2K_JSF_WORDS 17

struct jsf_digits { u32 w[S2K_JSF_WORDS]; };

S2K_HD void jsf_push_top(jsf_digits& d, u32 nib) {
#pragma unroll
    for (int i = 0; i < S2K_JSF_WORDS - 1; i++) d.w[i] = (d.w[i] >> 4) | (d.w[i + 1] << 28);
    d.w[S2K_JSF_WORDS - 1] = (d.w[S2K_JSF_WORDS - 1] >> 4) | (nib << 28);
}
S2K_HD u32 jsf_pop_top(jsf_digits& d) {
    const u32 nib = d.w[S2K_JSF_WORDS - 1] >> 28;
#pragma unroll
    for (int i = S2K_JSF_WORDS - 1; i > 0; i--) d.w[i] = (d.w[i] << 4) | (d.w[i - 1] >> 28);
    d.w[0] <<= 4;
    return nib;
}
S2K_HD void hs_shr1(u32 w[5]) {
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = (w[i] >> 1) | (w[i + 1] << 31);
    w[4] >>= 1;
}
S2K_HD void jsf_recode(jsf_digits& d, const half_scalar& k1, const half_scalar& k2) {
    u32 a[5], b[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { a[i] = k1.w[i]; b[i] = k2.w[i]; }
#pragma unroll
    for (int i = 0; i < S2K_JSF_WORDS; i++) d.w[i] = 0;
    u32 da = 0, db = 0;
    for (int j = 0; j < S2K_JSF_LEN; j++) {
        const u32 la = (a[0] + da) & 7u, lb = (b[0] + db) & 7u;
        int ua = 0, ub = 0;
        if (la & 1u) { ua = 2 - (int)(la & 3u); if ((la == 3u || la == 5u) && (lb & 3u) == 2u) ua = -ua; }
        if (lb & 1u) { ub = 2 - (int)(lb & 3u); if ((lb == 3u || lb == 5u) && (la & 3u) == 2u) ub = -ub; }
        if ((int)(2 * da) == 1 + ua) da = 1 - da;
        if ((int)(2 * db) == 1 + ub) db = 1 - db;
        hs_shr1(a); hs_shr1(b);
        const u32 nib = (ua != 0 ? 1u : 0u) | (ua < 0 ? 2u : 0u) | (ub != 0 ? 4u : 0u) | (ub < 0 ? 8u : 0u);
        jsf_push_top(d, nib);
    }
}

// ---- co-Z table {P, Q, P+Q, P-Q} ------------------------------------------------------------------------
// P = (X1,Y1,Z), Q = (X2,Y2,Z) with X1 != X2.  ZADDU + conjugate addition (Meloni 2007; Goundar, Joye, Miyaji 2010):
// all four points leave with the common  Z' = Z (X1 - X2).  7M + 3S.  Q is always lambda*P up to sign here, so
// only the sign of Y2 relative to Y1 is kept (q_ysign) instead of a ninth field element.
struct jsf_table {
    fe px, py;        // P  on the isomorphic curve (Z' implicit)
    fe qx;            // Q.x ;  Q.y = q_neg ? -py : py
    fe sx, sy;        // P + Q
    fe dx, dy;        // P - Q
    fe ziso;          // Z' : multiply the accumulator's Z by this when leaving the isomorphic curve
    int q_neg;
};

// Inputs: P finite Jacobian with magnitudes (<=1,<=1,1) after the normalisations below; negp/negq pick -P / -Q.
S2K_HD void jsf_table_build(jsf_table& t, const gej& P, int negp, int negq) {
    fe beta, x1 = P.x, y1 = P.y, x2, y2, d, c, w1, w2, a1, e, f, g;
    fe_norm_weak(x1); fe_norm_weak(y1);
    if (negp) { fe_neg(y1, y1, 1); fe_norm_weak(y1); }
    fe_set_beta(beta);
    fe_mul(x2, x1, beta);
    t.q_neg = (negp != negq);                 // sign of Q.y relative to (possibly negated) P.y
    fe_neg(y2, y1, 1); fe_norm_weak(y2);      // -y1
    fe_select(y2, y2, y1, t.q_neg);           // y2 = q_neg ? -y1 : y1
    fe_neg(d, x2, 1); fe_add(d, x1); fe_norm_weak(d);      // d = X1 - X2
    fe_sqr(c, d);
    fe_mul(w1, x1, c); fe_mul(w2, x2, c);
    fe_neg(e, w2, 1); fe_add(e, w1);                       // W1 - W2  (3)
    fe_mul(a1, y1, e);                                     // A1 = Y1 (W1 - W2) = P.y'
    fe_mul(t.ziso, P.z, d);
    // P + Q
    fe_neg(f, y2, 1); fe_add(f, y1); fe_norm_weak(f);      // Y1 - Y2
    fe_sqr(g, f);
    fe nw; fe_add2(nw, w1, w2); fe_neg(nw, nw, 2);         // -(W1 + W2)  (3)
    fe_add2(t.sx, g, nw); fe_norm_weak(t.sx);              // X3 = D - W1 - W2
    fe_neg(g, t.sx, 1); fe_add(g, w1);                     // W1 - X3  (3)
    fe_mul(t.sy, f, g);
    fe na1; fe_neg(na1, a1, 1);                            // -A1 (2)
    fe_add(t.sy, na1); fe_norm_weak(t.sy);
    // P - Q
    fe_add2(f, y1, y2); fe_norm_weak(f);                   // Y1 + Y2
    fe_sqr(g, f);
    fe_add2(t.dx, g, nw); fe_norm_weak(t.dx);
    fe_neg(g, t.dx, 1); fe_add(g, w1);
    fe_mul(t.dy, f, g);
    fe_add(t.dy, na1); fe_norm_weak(t.dy);
    t.px = w1; t.py = a1; t.qx = w2;
}

// operand for JSF nibble (nonzero): x from {px,qx,sx,dx}, y from {py,sy,dy} with sign.  Output magnitudes (1,2).
S2K_HD void jsf_select(ge& o, const jsf_table& t, u32 nib) {
    const int u1nz = nib & 1u, u1neg = (nib >> 1) & 1u, u2nz = (nib >> 2) & 1u, u2neg = (nib >> 3) & 1u;
    const int both = u1nz & u2nz;
    const int same = both & (u1neg == u2neg);      // +-(P+Q)
    const int diff = both & (u1neg != u2neg);      // +-(P-Q)
    const int only_q = u2nz & !u1nz;
    fe x, y;
    fe_select(x, t.qx, t.px, only_q);
    fe_cmov(x, t.sx, same);
    fe_cmov(x, t.dx, diff);
    y = t.py;
    fe_cmov(y, t.sy, same);
    fe_cmov(y, t.dy, diff);
    // sign: single P: u1neg; single Q: u2neg ^ q_neg; P+Q / P-Q: sign of u1
    const int neg = only_q ? (u2neg ^ t.q_neg) : u1neg;
    fe yn; fe_neg(yn, y, 1);
    fe_select(o.y, yn, y, neg);
    o.x = x;
}

// ---- wave-level predicates ------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#define S2K_WAVE_ANY(p) (__any(p))
#else
#define S2K_WAVE_ANY(p) (p)
#endif

// R = na*A + ng*G for this lane.  A is Jacobian (A.inf allowed), ng may be absent (has_ng = 0).
// gtab: generator table (see above).  R is returned on the real curve, magnitudes (<=6,<=3,1).
S2K_HD void ecmult_lane(gej& R, const gej& A, const scalar& na, const scalar& ng, int has_ng, const u32* gtab) {
    jsf_digits dig;
    jsf_table tab;
    const int p_active = (!A.inf) & (!sc_is_zero(na));
    const int g_active = has_ng & (!sc_is_zero(ng));
    {
        scalar k1s, k2s; half_scalar k1, k2;
        sc_split_lambda(k1s, k2s, na);
        sc_to_half(k1, k1s); sc_to_half(k2, k2s);
        jsf_recode(dig, k1, k2);
        jsf_table_build(tab, A, k1.neg, k2.neg);
    }
    gej_set_infinity(R);
    int phase = p_active ? 0 : (g_active ? 1 : 2);
    int pending = 0;          // the previous add hit P + P: R holds P (Z = 1), double it next
    int left = S2K_JSF_LEN;   // digits left in phase 0
    u32 win = 0;              // next generator window in phase 1
    u32 gw[8];                // ng, rotated right one byte per window (no indexed register access)
#pragma unroll
    for (int i = 0; i < 8; i++) gw[i] = ng.d[i];
    while (S2K_WAVE_ANY(phase != 2)) {
        int do_dbl = 0, do_add = 0;
        ge opnd;
        fe_set_zero(opnd.x); fe_set_zero(opnd.y);
        if (phase == 0) {
            if (pending) { do_dbl = 1; pending = 0; }
            else {
                do_dbl = !R.inf;
                const u32 nib = jsf_pop_top(dig);
                left--;
                if (nib) { do_add = 1; jsf_select(opnd, tab, nib); }
            }
        } else if (phase == 1) {
            if (pending) { do_dbl = 1; pending = 0; }
            else {
                const u32 byte = gw[0] & 0xFFu;
#pragma unroll
                for (int i = 0; i < 7; i++) gw[i] = (gw[i] >> 8) | (gw[i + 1] << 24);
                gw[7] >>= 8;
                if (byte) { do_add = 1; gtab_load(opnd, gtab, win, byte); }
                win++;
            }
        }
        if (S2K_WAVE_ANY(do_dbl)) {
            gej t; gej_double(t, R);
            if (do_dbl) R = t;
        }
        if (S2K_WAVE_ANY(do_add)) {
            gej t; const int f = gej_add_ge(t, R, opnd);
            if (do_add) { R = t; pending = (f == GEJ_ADD_NEEDS_DOUBLE); }
        }
        // phase transitions
        const int leave0 = (phase == 0) & (left == 0) & (!pending);
        if (S2K_WAVE_ANY(leave0)) {
            fe z; fe_mul(z, R.z, tab.ziso);
            if (leave0) { R.z = z; phase = g_active ? 1 : 2; }
        }
        if ((phase == 1) & (win == S2K_GTAB_WINDOWS) & (!pending)) phase = 2;
    }
}
