// gtable.h -- device-resident fixed-base table for the generator G (and, same layout, for a rangeproof generator).
//
// Role of the reference's secp256k1_pre_g / secp256k1_pre_g_128 (src/precomputed_ecmult.h:30-33, built by
// src/ecmult_compute_table_impl.h:14-46): odd multiples for wNAF(15).  Here the table is organised for a machine with
// 288 GB of HBM that would rather gather 64 bytes than execute doublings: entry (w, v) = v * 2^(D w) * G for every magnitude
// v = 1 .. 2^(D-1) of a SIGNED D-bit digit of every window of a scalar (ecmult.h "generator table"), as one aligned 64-byte sector of
// canonical words (affine x, y), so ng*G is W = ceil(256 / D) mixed additions and zero doublings.  D = 26: 10 x 2^25 entries x 64 B =
// 21.5 GB; D = 24 / 22 / 20: 5.9 / 1.6 / 0.44 GB with 11 / 12 / 13 additions (the engine falls back to those when the memory is not
// there: the table's header says which width it has).  The gathers are issued one addition ahead, so their HBM latency is covered.
// The table is *computed on the device* by the first call that needs it, never shipped as data.
//
// Construction (round 5; 645 ms per table until round 4, where every entry was its own 26-step double-and-add with its own inversion).
// A window's entries are the consecutive multiples v * B of its base, so with v = a * Kc +- b, b <= Kc / 2
//     E[a Kc +- b] = R[a] +- C[b],     R[a] = (a Kc) B  (the "row anchors"),  C[b] = b B  (the "column points"),  Kc ~ sqrt(2^(D-1)):
//   1. the W window bases                          (one lane each: D w doublings)                       k_gtab_base
//   2. the Kc/2 + 2^(D-1)/Kc seeds of every window (the generic double-and-add, one lane per seed)      k_gtab_seeds
//   3. every other entry as ONE AFFINE addition of two seeds with Montgomery's shared inversion: a lane owns a column b and a run of
//      GTAB_FILL_RUN rows, multiplies the run's denominators x(R[a]) - x(C[b]) up (prefix products in registers), inverts the product
//      once (modinv.h) and unwinds; R + C and R - C have the SAME denominator, so every share of the inversion yields two entries (round
//      6) -- 4.5 products + 1 squaring + 1/32 inversion per entry instead of ~400 products + 1 inversion.  The lanes of a wavefront hold
//      64 consecutive columns of one row run, so every step stores two 4 KB stretches of consecutive table (one upwards, one downwards
//      from the anchor) and reads one row anchor through a uniform address.                              k_gtab_fill
//   (R[a] +- C[b] is never exceptional: b <= Kc / 2 < a Kc and a Kc + b < n / 2, so the two points are neither equal nor opposite.)
// The reference's group_impl.h:236-287 (secp256k1_ge_set_all_gej_var) is the same shared inversion over Jacobian inputs.
#pragma once
#include "ecmult.h"

S2K_HD void ge_set_generator(ge& g) {
    const u32 gx[9] = {0x16F81798u, 0x0F940AD8u, 0x138A3656u, 0x17F9B65Bu, 0x10B07029u, 0x114AE743u, 0x0EB15681u, 0x0FDF3B97u, 0x0079BE66u};
    const u32 gy[9] = {0x1B10D4B8u, 0x023E847Fu, 0x01550667u, 0x0F68914Du, 0x108A8FD1u, 0x1DFE0708u, 0x11957693u, 0x0EE4D478u, 0x00483ADAu};
#pragma unroll
    for (int i = 0; i < 9; i++) { g.x.n[i] = gx[i]; g.y.n[i] = gy[i]; }
}
S2K_HD void gtab_store(u32* gtab, u32 D, u32 w, u32 v, const ge& a) {
    u32* p = gtab + gtab_slot(D, w, v) * S2K_GTAB_ENTRY_WORDS;
    u32 wx[8], wy[8];
    fe_to_words(wx, a.x); fe_to_words(wy, a.y);                     // `a` is normalised
    for (int i = 0; i < 8; i++) { p[i] = wx[i]; p[8 + i] = wy[i]; }
}
// step 1 (one thread per window w): base[w] = 2^(D w) * G, affine, stored as entry (w, 1).  `base`: any other point than G (the
// fixed-base tables of the rangeproof generators, rangeproof.h "shared-generator form", have exactly this layout).  Window 0's thread
// also writes the table's header.
S2K_HD void gtab_build_base(u32* gtab, u32 D, u32 w, const ge* base = nullptr) {
    ge g; if (base) g = *base; else ge_set_generator(g);
    gej j; gej_set_ge(j, g);
    for (u32 i = 0; i < D * w; i++) { gej t; gej_double(t, j); j = t; }
    ge a; ge_set_gej(a, j);
    gtab_store(gtab, D, w, 1, a);
}
// entry = v * base[w] by left-to-right double-and-add, its own inversion (the seeds; every entry until round 4)
S2K_HD void gtab_build_entry(u32* gtab, u32 D, u32 w, u32 v) {
    ge base; gtab_load(base, gtab, w, 1);
    gej acc; gej_set_infinity(acc);
    for (int bit = (int)D - 1; bit >= 0; bit--) {
        gej t; gej_double(t, acc); acc = t;
        if ((v >> bit) & 1u) {
            const int f = gej_add_ge(t, acc, base);
            acc = t;
            if (f == GEJ_ADD_NEEDS_DOUBLE) { gej_double(t, acc); acc = t; }
        }
    }
    ge a; ge_set_gej(a, acc);
    gtab_store(gtab, D, w, v, a);
}

// ---- the seeded construction ------------------------------------------------------------------------------------------------------------
struct gtab_fill_plan { u32 D, W, kc, Kc, NA, top_rows; };       // Kc = 2^kc: distance of the row anchors, NA = 2^(D-1) / Kc rows; the top window only needs its first top_rows rows
S2K_HD gtab_fill_plan gtab_make_fill_plan(u32 D) {
    gtab_fill_plan p; p.D = D; p.W = gtab_windows_for(D);
    p.kc = D / 2u; p.Kc = 1u << p.kc; p.NA = 1u << (D - 1u - p.kc);
    // the top window only ever sees the bits that are left of a 256-bit scalar, plus the carry of the recoding: v <= 2^TOP + 1
    const u32 vmax = (1u << gtab_top_bits_for(D)) + 1u;
    p.top_rows = vmax / p.Kc + 1u; if (p.top_rows > p.NA) p.top_rows = p.NA;
    return p;
}
// seeds of window w: t < Kc/2 -> column point C[t + 1] = entry (w, t + 1) (t = 0 is the base, already there); otherwise row anchor R[a],
// a = t - Kc/2 + 1 in 1 .. NA, = entry (w, a Kc)
S2K_HD u32 gtab_fill_cols(const gtab_fill_plan& p) { return p.Kc >> 1; }
S2K_HD u32 gtab_seeds_per_window(const gtab_fill_plan& p) { return gtab_fill_cols(p) + p.NA; }
S2K_HD void gtab_build_seed(u32* gtab, const gtab_fill_plan& p, u32 w, u32 t) {
    const u32 rows = (w + 1u < p.W) ? p.NA : p.top_rows, cols = gtab_fill_cols(p);
    u32 v;
    if (t < cols) v = t + 1u; else { const u32 a = t - cols + 1u; if (a > rows) return; v = a * p.Kc; }
    if (v >= 2u) gtab_build_entry(gtab, p.D, w, v);
}
#define GTAB_FILL_RUN 16
S2K_HD u32 gtab_fill_runs(const gtab_fill_plan& p) { return (p.NA + GTAB_FILL_RUN - 1u) / GTAB_FILL_RUN; }
S2K_HD void gtab_load_d(ge& r, const u32* gtab, u32 D, u32 w, u32 v) {       // gtab_load without the header read
    const u32* q = gtab + gtab_slot(D, w, v) * S2K_GTAB_ENTRY_WORDS;
    u32 t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = q[i];
    fe_from_words(r.x, t); fe_from_words(r.y, t + 8);
}
// one finished entry from lambda = (y_R -+ y_C) / (x_R - x_C): x3 = lambda^2 - x_R - x_C, y3 = lambda (x_C - x3) -+ y_C
S2K_HD void gtab_fill_finish(u32* gtab, u32 D, u32 w, u32 v, const fe& lam, const fe& nrx, const fe& ncx, const fe& cx, const fe& ycorr /* -+ y_C, magnitude <= 2 */) {
    fe x3; fe_sqr(x3, lam);
    fe_add(x3, nrx); fe_add(x3, ncx);                               // (5)
    fe_norm_weak(x3);
    fe t; fe_neg(t, x3, 1); fe_add(t, cx);                          // x_C - x3  (3)
    fe y3; fe_mul(y3, lam, t);
    fe_add(y3, ycorr);                                              // (3)
    ge o; o.x = x3; o.y = y3; fe_normalize(o.x); fe_normalize(o.y);
    gtab_store(gtab, D, w, v, o);
}
// lane (window w, column b in 1 .. Kc/2, run of rows a1 .. a1 + GTAB_FILL_RUN - 1, a1 >= 1): for every row a <= rows of the run the TWO
// entries  (w, a Kc + b) = R[a] + C[b]  and  (w, a Kc - b) = R[a] - C[b]  -- they share the denominator x(R[a]) - x(C[b]), so one share of
// the run's inversion and one pair of unwinding products serve both (round 6: 4.5 products + 1 squaring + 1/32 inversion per entry; until
// then one entry per denominator, 5 + 1 + 1/16).  Row `rows` only has its lower side (the upper one belongs to the next window, or is past
// anything the top window's digit can address); column Kc/2's lower side is the upper side of the row below.
S2K_HD void gtab_fill_run(u32* gtab, const gtab_fill_plan& p, u32 w, u32 b, u32 a1) {
    const u32 rows = (w + 1u < p.W) ? p.NA : p.top_rows, cols = gtab_fill_cols(p);
    ge c; gtab_load_d(c, gtab, p.D, w, b);
    fe ncx, ncy; fe_neg(ncx, c.x, 1); fe_neg(ncy, c.y, 1);          // -x(C), -y(C)   (2)
    fe pre[GTAB_FILL_RUN];                                          // (rolled loops: the prefix products live in the lane's scratch, 576 B that never leave the caches)
    fe run; fe_set_int(run, 1);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < 9; i++) S2K_OPAQUE(run.n[i]);               // (see msm_sum_refs_lean: a known constant start value pessimises the chain)
#endif
    // up: pre[j] = d_0 ... d_(j-1), d_j = x(R[a1 + j]) - x(C)   (rows beyond `rows`: d_j = 1)
#pragma unroll 1
    for (int j = 0; j < GTAB_FILL_RUN; j++) {
        const u32 a = a1 + (u32)j;
        pre[j] = run;
        if (a <= rows) {
            ge r; gtab_load_d(r, gtab, p.D, w, a * p.Kc);
            fe d = r.x; fe_add(d, ncx);                              // (3)
            fe_mul(run, run, d);
        }
    }
    fe inv; fe_inv(inv, run);
    // down: 1 / d_j = inv * pre[j]; inv <- inv * d_j
#pragma unroll 1
    for (int j = GTAB_FILL_RUN - 1; j >= 0; j--) {
        const u32 a = a1 + (u32)j;
        if (a <= rows) {
            ge r; gtab_load_d(r, gtab, p.D, w, a * p.Kc);
            fe d = r.x; fe_add(d, ncx);                              // (3)
            fe dinv; fe_mul(dinv, inv, pre[j]);
            fe_mul(inv, inv, d);
            fe nrx; fe_neg(nrx, r.x, 1);
            if (a < rows) {                                          // upper side: R + C
                fe dy = r.y; fe_add(dy, ncy);                        // y_R - y_C  (3)
                fe lam; fe_mul(lam, dy, dinv);
                gtab_fill_finish(gtab, p.D, w, a * p.Kc + b, lam, nrx, ncx, c.x, ncy);
            }
            if (b < cols) {                                          // lower side: R - C = R + (x_C, -y_C)
                fe dy = r.y; fe_add(dy, c.y);                        // y_R + y_C  (2)
                fe lam; fe_mul(lam, dy, dinv);
                gtab_fill_finish(gtab, p.D, w, a * p.Kc - b, lam, nrx, ncx, c.x, c.y);
            }
        }
    }
}
