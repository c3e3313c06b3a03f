// msm.h -- multi-scalar multiplication  r = g_sc*G + sum_i sc_i*P_i  (secp256k1_ecmult_multi_var,
// src/ecmult_impl.h:823-867; Pippenger: pippenger_wnaf :516-591, endo split :645-658, window choice :597-621).
//
// The reference walks one window at a time over one bucket array on one core.  On the GPU every (window, bucket)
// pair is its own lane and the windows are independent, so the whole bucket phase is three data-parallel passes:
//
//   msm_prep_term 1 lane / term  : byte decode, GLV split (scalar.h); limb-form point (x, beta*x, y) and the two 129-bit
//                                  half-scalar magnitudes written once
//   bin          1 workgroup / (window, chunk of terms): the signed c-bit digit of every half-scalar in the chunk for this
//                                  window (msm_digit_at needs no carry from the windows below), counted in an LDS
//                                  histogram; one global atomic per non-empty (workgroup, bucket) pair reserves the slots,
//                                  and the (term, half, sign) references go straight into fixed-capacity bucket regions.
//                                  A bucket that outgrows its region (adversarially equal scalars) sets a flag and the
//                                  launch falls back to the exact counting sort (scan of the same histogram + scatter)
//   msm_round1   1 lane / <=T refs: partial sums of runs of at most T consecutive references of one bucket (mixed additions).
//   msm_roundN   1 lane / <=T part.: the same on the partial sums, repeated until every bucket has at most one -- bucket
//                                  sizes are data dependent (the top window of a 129-bit half only has 2-3 live bits, and
//                                  equal scalars put every point in one bucket), so no lane ever owns a whole bucket
//   window sums                  : the reference's running sum (:581-588) has no parallelism inside a window.  Up to c = 13 the bucket
//                                  weights are taken apart into bits,  sum_b b B_b = sum_j 2^j (sum of the buckets whose weight has bit j),
//                                  as masked trees + a wavefront per bit in the wave-cooperative arithmetic (engine_msm.hip: k_msm_slices,
//                                  k_msm_window_sums; round 6); the widest plans keep a double-and-add by the weight per bucket (msm_scale)
//                                  and segmented trees (k_msm_finish, k_gej_reduce)
//   msm_combine  1 wavefront     : Horner over windows  r = sum_w 2^(c w) S_w   (c*W ~ 136 doublings, wave-cooperative: cofield.h), then the
//                                  affine result from the same launch
//
// Any order of additions gives the same group element, so the atomics-driven bucket order does not affect the
// (bit-exact) serialised result.  The reference switches to Strauss below 88 points (:55, :848-855); here NO size skips the bucket
// machinery since round 6: one double multiplication per lane is ~0.8 ms of latency on a single wavefront whatever the number of
// terms, the whole bucket pipeline 0.36-0.39 ms from one term up (profiles/r06an_msm_tiny.txt; until then sums below 32 terms took the
// bucket-free form, which is what the exact path of an overflowing launch still is: k_msm_direct).
#pragma once
#include "gtable.h"
#include "cofield.h"
#include <cstdlib>

#define MSM_MAX_WINDOWS 33          // c >= 4  ->  ceil(129/4)
// Term record (round 5): two aligned 64-byte sectors of canonical words, [ x | y ] and [ beta*x | y ] -- the operand record format of the
// rest of the engine (ecmult.h) -- so that a bucket reference gathers exactly ONE sector (the 28-word limb record of rounds 1-4 put the 72
// bytes an addition needs across two or three sectors: round 1's gathers were ~190 B per addition, and with the leaner addition of this
// round the kernel had become sensitive to them).  The y copy costs 32 B per term; the unpacking ~40 instructions per operand.
#define MSM_TERM_WORDS 32
S2K_HD void msm_store_term(u32* term, const ge& P, const fe& bx_in) {          // P.x, P.y, beta*x: any magnitude <= 2
    fe x = P.x, y = P.y, bx = bx_in; fe_normalize(x); fe_normalize(y); fe_normalize(bx);
    u32 wx[8], wy[8], wb[8]; fe_to_words(wx, x); fe_to_words(wy, y); fe_to_words(wb, bx);
#if defined(__HIP_DEVICE_COMPILE__)
    // (records are 128-byte aligned: eight 16-byte stores instead of 32 lone words -- one L2 request per quarter sector, round 6)
    typedef unsigned int msm_st4 __attribute__((ext_vector_type(4)));
    msm_st4* q = (msm_st4*)term;
    msm_st4 v;
    v.x = wx[0]; v.y = wx[1]; v.z = wx[2]; v.w = wx[3]; q[0] = v;  v.x = wx[4]; v.y = wx[5]; v.z = wx[6]; v.w = wx[7]; q[1] = v;
    v.x = wy[0]; v.y = wy[1]; v.z = wy[2]; v.w = wy[3]; q[2] = v; q[6] = v;  v.x = wy[4]; v.y = wy[5]; v.z = wy[6]; v.w = wy[7]; q[3] = v; q[7] = v;
    v.x = wb[0]; v.y = wb[1]; v.z = wb[2]; v.w = wb[3]; q[4] = v;  v.x = wb[4]; v.y = wb[5]; v.z = wb[6]; v.w = wb[7]; q[5] = v;
#else
    for (int i = 0; i < 8; i++) { term[i] = wx[i]; term[8 + i] = wy[i]; term[16 + i] = wb[i]; term[24 + i] = wy[i]; }
#endif
}

// nb = buckets per window = 2^(c-1) + 1 (bucket 0 unused).  [w0, w0 + wn) is the range of digit windows this launch owns:
// all of them on one GPU; a contiguous share when one MSM's bucket windows are spread over the GPUs of a node (SURVEY 8e,
// the reference's batching generalised, ecmult_impl.h:804-867).  A launch then returns sum_{w in range} 2^(c w) S_w, and
// the shares simply add up.
struct msm_plan { u32 c; u32 windows; u32 nb; u32 w0; u32 wn; };

// Fixed-capacity bucket regions (binning pass, engine_msm.hip): see k_msm_bin.  top_used: buckets 0..top_used-1 of the top window have a
// region; sub: power of two (every value of the top window is spread over `sub` buckets).
struct msm_layout { u32 cap, cap_top, top_used, sub; };     // top_used: buckets 0..top_used-1 of the top window have a region; sub: power of two
// bucket-region capacity of the fixed-capacity layout: the mean load plus ten standard deviations of a uniform digit
static inline u32 msm_cap_for(double mean) {
    double sd = 1.0; while (sd * sd < mean) sd += 1.0;
    size_t cap = (size_t)(mean + 10.0 * sd) + 8;
    return (u32)((cap + 7) & ~size_t(7));
}
static inline msm_layout msm_make_layout(size_t nt, const msm_plan& pl) {
    msm_layout L;
    const double mean = 2.0 * (double)nt / (double)(pl.nb - 1);
    const u32 top_bits = 128u - pl.c * (pl.windows - 1);              // live bits of the top window (0: only the carry reaches it)
    // c | 128: the top window only holds carries, and the window below it ends at bit 127, where the halves are NOT uniform (|k1| < 2^127.4,
    // |k2| < 2^126.9): its raw values stop at 0.66 * 2^c, so after the signed recoding the magnitudes above 0.34 * 2^c collect 1.29x the
    // uniform share (1.52x from k1, 1.07x from k2).  Those plans size every region for 1.5x.
    L.cap = msm_cap_for(top_bits == 0 ? 1.5 * mean : mean);
    const u32 top_vals = (top_bits >= pl.c - 1) ? (pl.nb - 1) : (1u << top_bits);
    // |k1| and |k2| stay below ~2^127.4 and ~2^126.9 (the GLV lattice bounds), so the top window's values are not uniform:
    // the low ones carry up to ~1.9x the uniform share.  4x (never more than every reference) leaves the same margin as below.
    double mean_top = 8.0 * (double)nt / (double)top_vals; if (mean_top > 2.0 * (double)nt) mean_top = 2.0 * (double)nt;
    // spread every value over `sub` buckets (see msm_layout) until its regions are about as full as the other windows', as far as the
    // window's nb - 1 bucket slots go
    u32 sub = 1;
    while (sub * 2 * top_vals <= pl.nb - 1 && mean_top / (double)sub > 1.5 * mean) sub *= 2;
    L.sub = sub;
    L.top_used = top_vals * sub + 1;
    L.cap_top = msm_cap_for(mean_top / (double)sub);
    return L;
}
static inline u32 msm_max_cap(const msm_plan& pl, const msm_layout& L) {
    const int has_top = (pl.w0 + pl.wn == pl.windows);
    const u32 a = pl.wn > (has_top ? 1u : 0u) ? L.cap : 0u, b = has_top ? L.cap_top : 0u;
    return a > b ? a : b;
}
S2K_HD msm_plan msm_plan_for(u32 c) {
    msm_plan p; p.c = c; p.windows = (129 + c - 1) / c; p.nb = (1u << (c - 1)) + 1u; p.w0 = 0; p.wn = p.windows;
    return p;
}
// Window width.  Large inputs: the width that minimises the work W*(2n + 23*2^(c-1)), capped by the binning pass's LDS histogram (13).
// Below 2^14 terms the call is a chain of latency-bound stages whatever c is, and what counts is their number: the smallest width (from
// 7) whose bucket regions are short enough for ONE round of partial sums (a lane per bucket walks its whole region: no counts / scan /
// second and third round), which also means fewer windows for the Horner tail.  In between: measured (profiles/r03c_msm_c_sweep.txt).
#define MSM_ONE_ROUND_CAP 40u
static inline msm_plan msm_make_plan(size_t n_terms, int c_override = 0) {
    u32 lg = 0; while (((size_t)1 << (lg + 1)) <= n_terms) lg++;
    int c = (int)lg - 6; if (c < 4) c = 4; if (c > 13) c = 13;
    if (lg <= 13) {
        // (no width of the one-round kind: measured, profiles/r06v_msm_small_plans.txt -- 13 bits, the fallback until round 6, cost 0.50-0.52 ms from
        //  3 000 to 12 000 terms where 11 / 12 / 10 bits take 0.45-0.50)
        c = lg <= 11 ? 11 : (lg == 12 ? 12 : 10);
        for (int t = 7; t <= 13; t++) { const msm_plan p = msm_plan_for((u32)t); if (msm_max_cap(p, msm_make_layout(n_terms, p)) <= MSM_ONE_ROUND_CAP) { c = t; break; } }
    } else if (lg <= 15) c = 10;
    else if (lg <= 17) c = 12;
    else if (lg >= 22) c = 16;      // 9 windows, the ninth only holds the carries: 8.2 bucket additions per half-scalar instead of 10 (k_msm_bin<1>)
    if (c_override >= 4 && c_override <= 17) c = c_override;      // diagnostic override (the engine reads $S2K_MSM_C once, in -DS2K_DIAG builds only)
    if (c > 13) {                   // the wide binning pass packs region offsets into 16 bits
        const msm_plan p = msm_plan_for((u32)c); const msm_layout L = msm_make_layout(n_terms, p);
        if (L.cap >= 32768u || L.cap_top >= 32768u) c = 13;
    }
    return msm_plan_for((u32)c);
}
// share `part` of `parts` of the windows (contiguous, sizes differ by at most one; parts > windows leaves some shares empty)
static inline void msm_plan_share(msm_plan& p, u32 part, u32 parts) {
    const u32 base = p.windows / parts, rem = p.windows % parts;
    p.w0 = part * base + (part < rem ? part : rem);
    p.wn = base + (part < rem ? 1u : 0u);
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

// Carry-free form of msm_digit: the carry into window w is 1 exactly when the bits below the window exceed the pattern
// "top bit of every lower window", so adding C_w = sum_{j<w} (2^(c-1) - 1) 2^(cj) to k produces it as an ordinary carry.
// (Induction over msm_digit's recurrence: t_w = raw_w + carry_w > H  <=>  (raw_w, lower bits) > (H, pattern_w).)
struct msm_wconst { u32 add[5]; u32 shift; };
S2K_HD void msm_window_const(msm_wconst& wc, u32 w, u32 c) {
    for (int i = 0; i < 5; i++) wc.add[i] = 0;
    const u32 hm1 = (1u << (c - 1)) - 1u;
    for (u32 j = 0; j < w; j++) {
        const u32 bit = j * c, word = bit >> 5, sh = bit & 31;
        const u64 v = (u64)hm1 << sh;
        if (word < 5) wc.add[word] |= (u32)v;
        if (word + 1 < 5) wc.add[word + 1] |= (u32)(v >> 32);
    }
    wc.shift = w * c;
}
S2K_HD int msm_digit_at(const u32 k[5], const msm_wconst& wc, u32 c) {
    u32 s[6]; u64 cy = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { cy += (u64)k[i] + wc.add[i]; s[i] = (u32)cy; cy >>= 32; }
    s[5] = 0;
    const u32 word = wc.shift >> 5, sh = wc.shift & 31;
    u64 v = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) if ((u32)i == word) v = (u64)s[i] | ((u64)s[i + 1] << 32);
    const u32 d = (u32)(v >> sh) & ((1u << c) - 1u);
    return d > (1u << (c - 1)) ? (int)d - (int)(1u << c) : (int)d;
}

// All windows of one half-scalar from ONE addition: C_w is the low part of C_full = C_(W-1) (the pattern in every window but the top
// one) and a carry into window w only depends on what lies below it, so s = k + C_full has, in window w, raw_w + carry_w plus the
// pattern's own (2^(c-1) - 1) there -- which comes off again mod 2^c (the top window has no pattern).  Same digits as msm_digit /
// msm_digit_at, bit for bit (tests/test_cpu_oracle.py::test_msm_digit_forms_agree).
struct msm_sfull { u32 s[6]; };
S2K_HD void msm_sum_full(msm_sfull& sf, const u32 k[5], u32 c, u32 windows) {
    msm_wconst wc; msm_window_const(wc, windows - 1u, c);
    u64 cy = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { cy += (u64)k[i] + wc.add[i]; sf.s[i] = (u32)cy; cy >>= 32; }
    sf.s[5] = (u32)cy;
}
S2K_HD int msm_digit_full(const msm_sfull& sf, u32 w, u32 c, u32 windows) {
    const u32 bit = w * c, word = bit >> 5, sh = bit & 31;
    u64 v = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) if ((u32)i == word) v = (u64)sf.s[i] | ((u64)sf.s[i + 1] << 32);
    const u32 mask = (1u << c) - 1u;
    u32 t = (u32)(v >> sh) & mask;
    if (w + 1u < windows) t = (t - ((1u << (c - 1)) - 1u)) & mask;
    return t > (1u << (c - 1)) ? (int)t - (int)(1u << c) : (int)t;
}
// half-scalar record of a term: k1 magnitude [5], k2 magnitude [5], flags (bit0 k1 negative, bit1 k2 negative, bit2 active), pad
#define MSM_HALF_WORDS 12
// GLV split of one term (no digits): term record as below, half-scalar record as above.  k: the scalar, reduced; P: the point (limbs, weakly
// normalised; ignored for the G term)
S2K_HD void msm_prep_term_kp(u32* term, u32* halves, const scalar& k, ge P, int pt_inf, int is_g) {
    if (is_g) ge_set_generator(P);
    const int active = (!pt_inf) & (!sc_is_zero(k));
    fe beta, bx; fe_set_beta(beta); fe_mul(bx, P.x, beta);
    msm_store_term(term, P, bx);
    scalar k1s, k2s; half_scalar h0, h1;
    sc_split_lambda(k1s, k2s, k);
    sc_to_half(h0, k1s); sc_to_half(h1, k2s);
    const u32 fl = (u32)h0.neg | ((u32)h1.neg << 1) | ((u32)active << 2);
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned int msm_st4 __attribute__((ext_vector_type(4)));
    msm_st4* q = (msm_st4*)halves;                                  // (48-byte records: 16-byte aligned)
    msm_st4 v;
    v.x = h0.w[0]; v.y = h0.w[1]; v.z = h0.w[2]; v.w = h0.w[3]; q[0] = v;
    v.x = h0.w[4]; v.y = h1.w[0]; v.z = h1.w[1]; v.w = h1.w[2]; q[1] = v;
    v.x = h1.w[3]; v.y = h1.w[4]; v.z = fl; v.w = 0u; q[2] = v;
#else
    for (int i = 0; i < 5; i++) { halves[i] = h0.w[i]; halves[5 + i] = h1.w[i]; }
    halves[10] = fl;
    halves[11] = 0;
#endif
}
// byte decode + the above
S2K_HD void msm_prep_term(u32* term, u32* halves, const unsigned char* sc32, const unsigned char* pt64, int pt_inf, int is_g) {
    scalar k; sc_set_b32(k, sc32, nullptr);
    ge P;
    if (is_g) ge_set_generator(P);
    else { fe_set_b32_mod(P.x, pt64); fe_set_b32_mod(P.y, pt64 + 32); fe_norm_weak(P.x); fe_norm_weak(P.y); }
    msm_prep_term_kp(term, halves, k, P, pt_inf, is_g);
}
#if defined(__HIPCC__)
// the same from 16-byte aligned inputs: two + four vector loads and byte swaps instead of 96 byte loads per term (round 6: the decode kernel
// was 100 us per 2^20 terms, most of it waiting for byte loads and lone word stores)
S2K_D void msm_prep_term_aligned(u32* term, u32* halves, const unsigned char* sc32, const unsigned char* pt64, int pt_inf, int is_g) {
    typedef unsigned int msm_ld4 __attribute__((ext_vector_type(4)));
    scalar k; ge P;
    {   const msm_ld4* q = (const msm_ld4*)sc32; const msm_ld4 a = q[0], b = q[1];
        const u32 be[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 8; i++) k.d[i] = __builtin_bswap32(be[7 - i]);
        const int o = sc_check_overflow(k.d); sc_reduce_once(k.d, o); }
    if (is_g) ge_set_generator(P);
    else {
        const msm_ld4* q = (const msm_ld4*)pt64; const msm_ld4 a = q[0], b = q[1], c = q[2], d = q[3];
        const u32 xb[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}, yb[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
        u32 xw[8], yw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { xw[i] = __builtin_bswap32(xb[7 - i]); yw[i] = __builtin_bswap32(yb[7 - i]); }
        fe_from_words(P.x, xw); fe_from_words(P.y, yw); fe_norm_weak(P.x); fe_norm_weak(P.y);
    }
    msm_prep_term_kp(term, halves, k, P, pt_inf, is_g);
}
#endif
// bucket key of (half-scalar record, half, window): (w*nb + |d|) << 1 | sign, 0 = no contribution
S2K_HD u32 msm_key_at(const u32* halves, int half, u32 w, const msm_wconst& wc, const msm_plan& pl) {
    const u32 fl = halves[10];
    if (!(fl & 4u)) return 0;
    const int d = msm_digit_at(halves + 5 * half, wc, pl.c);
    if (d == 0) return 0;
    const int neg = (d < 0) ^ (int)((fl >> half) & 1u);
    const u32 mag = (u32)(d < 0 ? -d : d);
    return ((w * pl.nb + mag) << 1) | (u32)neg;
}

// ---- pass 1 (exact counting-sort form, used by the host emulation and kept as the reference for msm_digit_at) ------------------------------------------------------------------------------
// term_data[i] : MSM_TERM_WORDS words (msm_store_term)
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
    msm_store_term(term, P, bx);
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
// (the XYZZ accumulator `gez`, its exact and its lean mixed addition live in group.h: the table parts of ecmult.h use them too)
// refs[j] = term_index << 2 | half << 1 | neg
S2K_HD void msm_ref_point(ge& p, u32 r, const u32* term_data);
S2K_HD void msm_sum_refs(gej& out, const u32* refs, size_t start, size_t end, const u32* term_data) {
    gez acc; acc.inf = 1; fe_set_zero(acc.x); fe_set_zero(acc.y); fe_set_zero(acc.zz); fe_set_zero(acc.zzz);
    for (size_t j = start; j < end; j++) {
        ge p; msm_ref_point(p, refs[j], term_data);
        gez_add_ge(acc, p);
    }
    gej_set_gez(out, acc);
}
// the operand a bucket reference names: (x or beta*x, +-y) of its term record
S2K_HD void msm_ref_point(ge& p, u32 r, const u32* term_data) {
    const u32* t = term_data + (size_t)(r >> 2) * MSM_TERM_WORDS + ((r >> 1) & 1u) * 16u;      // one aligned 64-byte sector: [ x | y ] or [ beta*x | y ]
    u32 w[16];
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned int msm_u32x4 __attribute__((ext_vector_type(4)));
    const msm_u32x4* q = (const msm_u32x4*)t;
#pragma unroll
    for (int k = 0; k < 4; k++) { const msm_u32x4 v = q[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
#else
    for (int k = 0; k < 16; k++) w[k] = t[k];
#endif
    fe_from_words(p.x, w); fe_from_words(p.y, w + 8);
    if (r & 1u) fe_neg(p.y, p.y, 1);
}
// returns 1 with the run's sum in `out`; 0 when the run met an exceptional addition (the caller sums it again with msm_sum_refs)
S2K_HD int msm_sum_refs_lean(gej& out, const u32* refs, size_t start, size_t end, const u32* term_data) {
    if (start >= end) { gej_set_infinity(out); return 1; }
    gez acc; acc.inf = 0;
    { ge p; msm_ref_point(p, refs[start], term_data); fe_norm_weak(p.y); acc.x = p.x; acc.y = p.y; fe_set_int(acc.zz, 1); fe_set_int(acc.zzz, 1);
      // (opaque to the optimiser: knowing that ZZ = ZZZ = 1 on entry, ROCm 7.2's clang specialises the loop's accumulator chains on the
      //  zero limbs of the first trip and the WHOLE loop body grows from 1 595 to 1 983 instructions -- 126 more multiply-accumulates
      //  and 270 more moves per addition, found with tools/static_count/loops.py)
#pragma unroll
      for (int i = 0; i < 9; i++) { S2K_OPAQUE(acc.zz.n[i]); S2K_OPAQUE(acc.zzz.n[i]); } }
    // the operand of addition j + 1 is requested before the arithmetic of addition j
    ge nxt; u32 have = 0;
    if (start + 1 < end) { msm_ref_point(nxt, refs[start + 1], term_data); have = 1; }
    for (size_t j = start + 1; j < end; j++) {
        const ge cur = nxt;
        if (j + 1 < end) msm_ref_point(nxt, refs[j + 1], term_data);
        gez_add_ge_lean(acc, cur);
    }
    (void)have;
    if (fe_normalizes_to_zero(acc.zz)) return 0;
    gej_set_gez(out, acc);
    return 1;
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

// ---- final combine: r = sum_{w in [w0, w0+wn)} 2^(c w) S_w ; window_sums28 holds the wn local sums --------------------------
// Horner: ~c * windows dependent doublings of one point -- the latency floor of an MSM.  On the device the runs of doublings
// and the additions between them run in the wave-cooperative form (cofield.h: limb l in lane l, ~2.2x fewer instructions on the
// critical path); every lane of the wavefront must call this with the same arguments (k_msm_combine runs all 64).
#if defined(__HIPCC__)
S2K_D void msm_combine_cooperative(gej& r, const u32* window_sums28, const msm_plan& pl) {
    cgej acc; int acc_inf = 1;
    acc.x.v = acc.y.v = acc.z.v = 0;
    // (the record of window w - 1 is requested before the doublings that precede its addition: a load in the chain is ~1-2 us exposed)
    cgej nx; int nx_inf = 1; nx.x.v = nx.y.v = nx.z.v = 0;
    if (pl.wn) nx_inf = cgej_load28(nx, window_sums28 + 28 * (pl.wn - 1));
    for (int w = (int)pl.wn - 1; w >= 0; w--) {
        const cgej sw = nx; const int sw_inf = nx_inf;
        if (w > 0) nx_inf = cgej_load28(nx, window_sums28 + 28 * (w - 1));
        if (!acc_inf) {
#pragma unroll 1
            for (u32 k = 0; k < pl.c; k++) cgej_double(acc);
        }
        if (!sw_inf) {
            if (acc_inf) { acc = sw; acc_inf = 0; }
            else acc_inf = cgej_add(acc, sw);
        }
    }
    if (!acc_inf) {
#pragma unroll 1
        for (u32 k = 0; k < pl.c * pl.w0; k++) cgej_double(acc);
        cgej_to_gej(r, acc);
    } else gej_set_infinity(r);
}
#endif
S2K_HD void msm_combine(gej& r, const u32* window_sums28, const msm_plan& pl) {
#if defined(__HIP_DEVICE_COMPILE__)
    msm_combine_cooperative(r, window_sums28, pl);
#else
    gej acc; gej_set_infinity(acc);
    for (int w = (int)pl.wn - 1; w >= 0; w--) {
        if (!acc.inf) for (u32 k = 0; k < pl.c; k++) { gej s; gej_double(s, acc); acc = s; }
        gej sw;
        for (int i = 0; i < 9; i++) { sw.x.n[i] = window_sums28[28 * w + i]; sw.y.n[i] = window_sums28[28 * w + 9 + i]; sw.z.n[i] = window_sums28[28 * w + 18 + i]; }
        sw.inf = (int)window_sums28[28 * w + 27];
        gej s; gej_add_var(s, acc, sw); acc = s;
    }
    if (!acc.inf) for (u32 k = 0; k < pl.c * pl.w0; k++) { gej s; gej_double(s, acc); acc = s; }
    r = acc;
#endif
}

// ---- the part of a scalar that a window share sees -------------------------------------------------------------------------
// k = k1 + lambda k2 with signed c-bit digits d_{h,w} of the two 129-bit magnitudes; a share [w0, w0+wn) of the windows
// contributes  sum_h s_h lambda^h sum_{w in share} d_{h,w} 2^(c w).  msm_share_scalar returns that residue, so that
// "k_share * P summed over the terms" is the share's result computed without any bucket -- the exact (slow) path a launch
// falls back to when an adversarial input overflows a bucket region.  For the full range it returns k itself.
S2K_HD void msm_low_part(u32 t[6], int& carry, const u32 k[5], u32 w, u32 c) {
    // T(w) = sum_{j<w} d_j 2^(c j) = (k mod 2^(c w)) - carry_w 2^(c w); returns k mod 2^(c w) and carry_w
    const u32 bits = w * c;
    msm_wconst wc; msm_window_const(wc, w, c);
    u32 low[6]; u64 cy = 0;
    for (int i = 0; i < 6; i++) {
        const u32 ki = i < 5 ? k[i] : 0u;
        const int lo_bit = 32 * i;
        u32 m = 0xFFFFFFFFu;
        if ((int)bits <= lo_bit) m = 0u; else if (bits < (u32)lo_bit + 32u) m = (1u << (bits - lo_bit)) - 1u;
        low[i] = ki & m; t[i] = low[i];
    }
    u32 s[6];
    for (int i = 0; i < 6; i++) { cy += (u64)low[i] + (i < 5 ? wc.add[i] : 0u); s[i] = (u32)cy; cy >>= 32; }
    carry = (bits < 192u) ? (int)((s[bits >> 5] >> (bits & 31)) & 1u) : 0;
}
S2K_HD void msm_add_bit(u32 x[6], u32 bit) {          // x += 2^bit (bit < 192)
    u64 cy = 0;
    for (int i = 0; i < 6; i++) {
        cy += (u64)x[i] + (((bit >> 5) == (u32)i) ? ((u64)1 << (bit & 31)) : 0);
        x[i] = (u32)cy; cy >>= 32;
    }
}
S2K_HD void msm_share_scalar(scalar& r, const scalar& k, const msm_plan& pl) {
    if (pl.w0 == 0 && pl.wn == pl.windows) { r = k; return; }
    scalar ks[2]; half_scalar h[2];
    sc_split_lambda(ks[0], ks[1], k);
    sc_to_half(h[0], ks[0]); sc_to_half(h[1], ks[1]);
    scalar part[2];
    for (int hf = 0; hf < 2; hf++) {
        // m = T(w1) - T(w0), as a 192-bit two's complement number
        u32 hi[6], lo[6]; int chi, clo;
        const u32 w1 = pl.w0 + pl.wn;
        msm_low_part(hi, chi, h[hf].w, w1, pl.c);
        if (w1 >= pl.windows) { for (int i = 0; i < 6; i++) hi[i] = i < 5 ? h[hf].w[i] : 0u; chi = 0; }      // the top window never carries out
        msm_low_part(lo, clo, h[hf].w, pl.w0, pl.c);
        // m = (hi + clo 2^(c w0)) - (lo + chi 2^(c w1))
        if (clo) msm_add_bit(hi, pl.c * pl.w0);
        if (chi) msm_add_bit(lo, pl.c * w1);
        u32 m[6]; u32 borrow = 0;
        for (int i = 0; i < 6; i++) {
            const u64 d = (u64)hi[i] - (u64)lo[i] - (u64)borrow;
            m[i] = (u32)d; borrow = (u32)(d >> 63);
        }
        const int neg = (m[5] >> 31) & 1;
        if (neg) { u64 cy = 1; for (int i = 0; i < 6; i++) { cy += (u64)(~m[i]); m[i] = (u32)cy; cy >>= 32; } }
        scalar mag; for (int i = 0; i < 8; i++) mag.d[i] = i < 6 ? m[i] : 0u;           // |m| < 2^131 < n
        if (neg ^ h[hf].neg) sc_negate(part[hf], mag); else part[hf] = mag;
    }
    scalar lam; sc_set_lambda(lam);
    scalar t; sc_mul(t, part[1], lam);
    sc_add(r, part[0], t);
}
