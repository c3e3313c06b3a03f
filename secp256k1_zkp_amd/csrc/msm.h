// msm.h -- multi-scalar multiplication  r = g_sc*G + sum_i sc_i*P_i  (secp256k1_ecmult_multi_var,
// src/ecmult_impl.h:823-867; Pippenger: pippenger_wnaf :516-591, endo split :645-658, window choice :597-621).
//
// The reference walks one window at a time over one bucket array on one core.  On the GPU every (window, bucket)
// pair is its own lane and the windows are independent, so the whole bucket phase is three data-parallel passes:
//
//   msm_prep     1 lane / term   : byte decode, GLV split (scalar.h), signed c-bit digits of both 128-bit halves;
//                                  limb-form point (x, beta*x, y) written once; one histogram atomic per digit
//   (scan)       exclusive prefix sum of the (window,bucket) histogram
//   msm_scatter  1 lane / term   : counting-sort scatter of (term, half, sign) references into bucket order
//   msm_round1   1 lane / <=T refs: partial sums of runs of at most T consecutive references of one bucket (mixed additions).
//   msm_roundN   1 lane / <=T part.: the same on the partial sums, repeated until every bucket has at most one -- bucket
//                                  sizes are data dependent (the top window of a 129-bit half only has 2-3 live bits, and
//                                  equal scalars put every point in one bucket), so no lane ever owns a whole bucket
//   msm_finish   1 lane / bucket : the bucket's own weight  b * B_b  by a short double-and-add (replaces the reference's
//                                  serial running sum :581-588, which has no parallelism inside a window)
//   gej_reduce   tree sums       : per-window totals S_w
//   msm_combine  1 lane          : Horner over windows  r = sum_w 2^(c w) S_w   (c*W ~ 136 doublings)
//
// Any order of additions gives the same group element, so the atomics-driven bucket order does not affect the
// (bit-exact) serialised result.  Small inputs (n < MSM_SMALL_N) skip the bucket machinery: one full
// double-and-add per lane (ecmult.h) and a tree sum -- the analogue of the reference switching to Strauss below 88
// points (:55, :848-855).
#pragma once
#include "ecmult.h"

#define MSM_SMALL_N 192
#define MSM_MAX_WINDOWS 33          // c >= 4  ->  ceil(129/4)
#define MSM_TERM_WORDS 28           // x[9], beta*x[9], y[9], flags

struct msm_plan { u32 c; u32 windows; u32 nb; };   // nb = buckets per window = 2^(c-1) + 1 (bucket 0 unused)

static inline msm_plan msm_make_plan(size_t n_terms) {
    u32 lg = 0; while (((size_t)1 << (lg + 1)) <= n_terms) lg++;
    int c = (int)lg - 6; if (c < 4) c = 4; if (c > 13) c = 13;      // minimises W*(2n + 23*2^(c-1)) over the sizes of interest
    msm_plan p; p.c = (u32)c; p.windows = (129 + c - 1) / c; p.nb = (1u << (c - 1)) + 1u;
    return p;
}

// signed c-bit digits of a 129-bit magnitude: k = sum d_w 2^(c w), d_w in [-2^(c-1), 2^(c-1)]
S2K_HD int msm_digit(const u32 k[5], u32 w, u32 c, int& carry) {
    const u32 bit = w * c, word = bit >> 5, sh = bit & 31;
    u64 v = 0;
    if (word < 5) v = k[word];
    if (word + 1 < 5) v |= (u64)k[word + 1] << 32;
    u32 d = (u32)(v >> sh) & ((1u << c) - 1u);
    d += (u32)carry;
    if (d > (1u << (c - 1))) { carry = 1; return (int)d - (int)(1u << c); }
    carry = 0; return (int)d;
}

// ---- pass 1: per-term preparation ------------------------------------------------------------------------------
// term_data[i] : 28 words (x, beta x, y limbs, flags bit0 = active)
// keys[(2*i + h) * W + w] = bucket key (w*nb + |d|) << 1 | sign  (0 = no contribution)
S2K_HD void msm_prep(u32* term, u32* keys, u32* hist, const unsigned char* sc32, const unsigned char* pt64, int pt_inf, int is_g,
                     const msm_plan& pl) {
    scalar k; sc_set_b32(k, sc32, nullptr);
    ge P;
    if (is_g) { const u32 gx[9] = {0x16F81798u, 0x0F940AD8u, 0x138A3656u, 0x17F9B65Bu, 0x10B07029u, 0x114AE743u, 0x0EB15681u, 0x0FDF3B97u, 0x0079BE66u};
                const u32 gy[9] = {0x1B10D4B8u, 0x023E847Fu, 0x01550667u, 0x0F68914Du, 0x108A8FD1u, 0x1DFE0708u, 0x11957693u, 0x0EE4D478u, 0x00483ADAu};
                for (int i = 0; i < 9; i++) { P.x.n[i] = gx[i]; P.y.n[i] = gy[i]; } }
    else { fe_set_b32_mod(P.x, pt64); fe_set_b32_mod(P.y, pt64 + 32); fe_norm_weak(P.x); fe_norm_weak(P.y); }
    const int active = (!pt_inf) & (!sc_is_zero(k));
    fe beta, bx; fe_set_beta(beta); fe_mul(bx, P.x, beta);
    for (int i = 0; i < 9; i++) { term[i] = P.x.n[i]; term[9 + i] = bx.n[i]; term[18 + i] = P.y.n[i]; }
    term[27] = (u32)active;
    scalar k1s, k2s; half_scalar h[2];
    sc_split_lambda(k1s, k2s, k);
    sc_to_half(h[0], k1s); sc_to_half(h[1], k2s);
    for (int half = 0; half < 2; half++) {
        int carry = 0;
        for (u32 w = 0; w < pl.windows; w++) {
            int d = msm_digit(h[half].w, w, pl.c, carry);
            u32 key = 0;
            if (active && d != 0) {
                const int neg = (d < 0) ^ h[half].neg;
                const u32 mag = (u32)(d < 0 ? -d : d);
                key = ((w * pl.nb + mag) << 1) | (u32)neg;
#if defined(__HIP_DEVICE_COMPILE__)
                atomicAdd(&hist[w * pl.nb + mag], 1u);
#else
                hist[w * pl.nb + mag]++;
#endif
            }
            keys[(size_t)half * pl.windows + w] = key;
        }
    }
}

// ---- pass 3: partial sums of runs of references / partials ---------------------------------------------------------
// largest k with off[k] <= m   (off is a non-decreasing exclusive prefix array of nk+1 entries, m < off[nk])
S2K_HD u32 msm_find_key(const u32* off, u32 nk, u32 m) {
    u32 lo = 0, hi = nk;            // invariant: off[lo] <= m < off[hi]
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (off[mid] <= m) lo = mid; else hi = mid; }
    return lo;
}
// refs[j] = term_index << 2 | half << 1 | neg
S2K_HD void msm_sum_refs(gej& out, const u32* refs, u32 start, u32 end, const u32* term_data) {
    gej acc; gej_set_infinity(acc);
    for (u32 j = start; j < end; j++) {
        const u32 r = refs[j];
        const u32* t = term_data + (size_t)(r >> 2) * MSM_TERM_WORDS;
        ge p; const int half = (r >> 1) & 1, neg = r & 1;
        for (int i = 0; i < 9; i++) { p.x.n[i] = half ? t[9 + i] : t[i]; p.y.n[i] = t[18 + i]; }
        if (neg) { fe_neg(p.y, p.y, 1); }
        gej s; const int f = gej_add_ge(s, acc, p); acc = s;
        if (f == GEJ_ADD_NEEDS_DOUBLE) { gej_double(s, acc); acc = s; }
    }
    out = acc;
}
// weight * acc, weight < 2^16  (the bucket's index)
S2K_HD void msm_scale(gej& out, const gej& in, u32 weight) {
    gej r; gej_set_infinity(r);
    if (!in.inf) {
        gej acc = in;
        fe_norm_weak(acc.x); fe_norm_weak(acc.y);
        int top = 31; while (top > 0 && !((weight >> top) & 1u)) top--;
        for (int bit = top; bit >= 0; bit--) {
            gej s; gej_double(s, r); r = s;
            if ((weight >> bit) & 1u) { gej_add_var(s, r, acc); r = s; }
        }
    }
    out = r;
}

// ---- final combine: r = sum_w 2^(c w) S_w ---------------------------------------------------------------------------
S2K_HD void msm_combine(gej& r, const u32* window_sums28, const msm_plan& pl) {
    gej acc; gej_set_infinity(acc);
    for (int w = (int)pl.windows - 1; w >= 0; w--) {
        for (u32 k = 0; k < pl.c; k++) { gej s; gej_double(s, acc); acc = s; }
        gej sw;
        for (int i = 0; i < 9; i++) { sw.x.n[i] = window_sums28[28 * w + i]; sw.y.n[i] = window_sums28[28 * w + 9 + i]; sw.z.n[i] = window_sums28[28 * w + 18 + i]; }
        sw.inf = (int)window_sums28[28 * w + 27];
        gej s; gej_add_var(s, acc, sw); acc = s;
    }
    r = acc;
}
