// cofield.h -- wave-cooperative field arithmetic for LATENCY chains: one field element spread over the lanes of a wavefront.
//
// The bucket MSM ends in chains nobody can parallelise across points: the Horner recombination of the window sums is ~120 dependent
// doublings of ONE point (msm.h, msm_combine).  A single wavefront issues one VALU instruction per ~7.6 cycles whatever the instruction
// level parallelism, so the serial doubling (fe.h / group.h: ~1 020 instructions, every lane redundantly doing the same
// 9-limb arithmetic) costs ~3.7 us, and the chain 0.4 ms.  Here limb l of an element lives in lane l instead:
//   * product: lane k accumulates column k = sum_i a_i b_(k-i):  a_i comes from v_readlane (a scalar operand of the multiply-
//     accumulate), b shifted by i lanes from one DPP `wave_shr:1` move per step -- 9 steps for all 17 columns;
//   * reduction: carries move one lane up with the same DPP move (three split-and-shift rounds instead of a 17-step ripple), the high
//     columns come down 9 and 8 lanes with ds_bpermute for the fold 2^261 == 2^37 + 31264 (mod p), twice, then the bits above 2^256
//     go through 2^256 == 2^32 + 977 exactly as in fe_norm_weak;
//   * additions, negations, small multiples: ONE instruction instead of nine.
// ~70 instructions per product instead of ~145; and three independent products share one pass (three lane groups, see below), so a
// doubling is three passes (~270 instructions) instead of ~1 020, an addition six.  Same representation (9 x 29-bit limbs, lazily
// reduced, top limb 24 bits) and the same magnitude contract as fe.h (product of input magnitudes <= 7), so elements move between the
// two forms limb by limb.  The reduction's bounds were checked against an integer model at the contract's limits before this was
// written for the device; tests/test_gpu_prims.py::test_cooperative_field_arithmetic compares every routine with its fe.h / group.h
// twin on the device.
// Device only: the host build (tests/host_emul) keeps the serial code.
#pragma once
#include "group.h"

#if defined(__HIPCC__)                      /* both passes of hipcc see the declarations; every routine is __device__ */
// Three products at a time.  The 64 lanes are three GROUPS of 21 (lane 63 idles); an element normally sits REPLICATED -- limb k in lanes
// k, 21 + k and 42 + k, zeros in between -- so that every lane-wise routine below works on all three copies at once and the old scalar
// operand fetch (v_readlane from group 0) stays valid.  A grouped product (cfe_mul3) takes operands that DIFFER per group (picked from
// replicated values with two selects, co_sel3) and leaves three different products, one per group; co_bcast replicates the one wanted.
// The multiplier limbs of a grouped product come through ds_bpermute (a per-lane source) instead of v_readlane.  A doubling is three
// grouped products instead of seven passes (its dependency depth), an addition six instead of sixteen.
struct cfe { u32 v; };                      // this lane's limb (lanes with k >= 9 hold 0: the products rely on it)

#define CO_STRIDE 21u
S2K_D u32 co_abs() { return (u32)(threadIdx.x & 63u); }
S2K_D u32 co_group() { const u32 l = co_abs(); return l >= 2u * CO_STRIDE ? 2u : (l >= CO_STRIDE ? 1u : 0u); }
S2K_D u32 co_lane() { return co_abs() - CO_STRIDE * co_group(); }                       // k: the limb index inside the group (lane 63: 21)
S2K_D u32 co_up1(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true); }     // lane l gets lane l-1 (lane 0: 0)
S2K_D u32 co_down(u32 x, u32 k) { return (u32)__builtin_amdgcn_ds_bpermute((int)(((co_abs() + k) & 63u) << 2), (int)x); }   // lane l gets lane l+k
S2K_D u32 co_get(u32 x, int l) { return (u32)__builtin_amdgcn_readlane((int)x, l); }                                        // group 0's lane l, as a scalar
S2K_D u32 co_kget(u32 x, u32 i) { return (u32)__builtin_amdgcn_ds_bpermute((int)((co_abs() - co_lane() + i) << 2), (int)x); }   // this group's lane i
S2K_D u32 co_sel3(u32 a0, u32 a1, u32 a2) { const u32 g = co_group(); return g == 0 ? a0 : (g == 1 ? a1 : a2); }
// every group gets group g's value (lane 63 keeps its zero)
S2K_D u32 co_bcast(u32 x, u32 g) {
    const u32 v = (u32)__builtin_amdgcn_ds_bpermute((int)((CO_STRIDE * g + co_lane()) << 2), (int)x);
    return co_abs() == 63u ? 0u : v;
}

S2K_D void cfe_from_fe(cfe& r, const fe& a) {        // `a` is the same in every lane (serial code runs redundantly in all of them)
    const u32 l = co_lane();
    u32 v = 0;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) v = (l == (u32)i) ? a.n[i] : v;
    r.v = v;
}
S2K_D void cfe_to_fe(fe& r, const cfe& a) {           // of a replicated element
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) r.n[i] = co_get(a.v, i);
}
S2K_D u32 cfe_p_limb() { const u32 l = co_lane(); return l == 0 ? FE_P0 : (l == 1 ? FE_P1 : (l < 8 ? FE_M : (l == 8 ? FE_TOPM : 0u))); }

S2K_D void cfe_add(cfe& r, const cfe& a) { r.v += a.v; }
S2K_D void cfe_add2(cfe& r, const cfe& a, const cfe& b) { r.v = a.v + b.v; }
S2K_D void cfe_neg(cfe& r, const cfe& a, u32 m) { r.v = (m + 1u) * cfe_p_limb() - a.v; }       // magnitude m -> m + 1
S2K_D void cfe_mul_int(cfe& r, u32 k) { r.v *= k; }
// any magnitude <= 7 -> 1 (fe_norm_weak); replicated elements
S2K_D void cfe_norm_weak(cfe& r) {
    const u32 l = co_lane();
    const u32 t = co_get(r.v, 8) >> 24;
    const u32 c = (l <= 7) ? (r.v >> FE_BITS) : 0u;
    u32 v = (l == 8) ? (r.v & FE_TOPM) : (r.v & FE_M);
    v += co_up1(c);
    v += (l == 0) ? t * 977u : 0u;
    v += (l == 1) ? (t << 3) : 0u;
    r.v = v;
}
// r/2 (fe_half): make it even with +p, then every limb takes the low bit of its upper neighbour; replicated elements
S2K_D void cfe_half(cfe& r) {
    const u32 l = co_lane();
    const u32 odd = 0u - (co_get(r.v, 0) & 1u);
    const u32 v = r.v + (cfe_p_limb() & odd);
    const u32 up = co_down(v, 1);
    r.v = (v >> 1) + ((l <= 7) ? ((up & 1u) << (FE_BITS - 1)) : 0u);
}

// columns of a*b added to acc: lane k of a group holds sum_{i+j=k} a_i b_j  (k = 0..16).  GROUPED: the operands differ per group
// (multiplier limbs through ds_bpermute); otherwise they are replicated (multiplier limbs as scalars from group 0).
template <bool GROUPED>
S2K_D void cfe_columns(u64& acc, const cfe& a, const cfe& b) {
    u32 bs = b.v;
    if (GROUPED) {
        u32 ai[FE_LIMBS];
#pragma unroll
        for (int i = 0; i < FE_LIMBS; i++) ai[i] = co_kget(a.v, (u32)i);
#pragma unroll
        for (int i = 0; i < FE_LIMBS; i++) {
            acc += (u64)ai[i] * bs; S2K_CHAIN(acc);
            if (i + 1 < FE_LIMBS) bs = co_up1(bs);
        }
    } else {
#pragma unroll
        for (int i = 0; i < FE_LIMBS; i++) {
            u32 ai = co_get(a.v, i);
            acc += (u64)ai * bs; S2K_CHAIN(acc);
            if (i + 1 < FE_LIMBS) bs = co_up1(bs);
        }
    }
}
// 17 columns (each < 2^64) -> magnitude-1 limbs.  Bounds in the comments: what the integer model measured at magnitude product 7.
// (All lane shifts are relative and stay inside a group: 19 of its 21 lanes are ever non-zero.)
template <bool GROUPED>
S2K_D void cfe_reduce(cfe& r, u64 acc) {
    const u32 l = co_lane();
    // round A: split at 29 bits, carries one lane up                                   (lanes 0..17, < 2^29 + 2^35)
    const u64 hi = acc >> FE_BITS;
    const u64 v = (u64)((u32)acc & FE_M) + ((u64)co_up1((u32)hi) | ((u64)co_up1((u32)(hi >> 32)) << 32));
    // round B                                                                          (lanes 0..18, < 2^29 + 2^7: 32 bits from here on)
    const u32 w = ((u32)v & FE_M) + co_up1((u32)(v >> FE_BITS));
    // fold 1: columns 9..18 have weight 2^261 * 2^(29 j) == (2^37 + 31264) 2^(29 j)     (lanes 0..10, < 2^45)
    u32 x = co_down(w, 9), y = co_down(w, 8);
    x = (l <= 9) ? x : 0u; y = (l >= 1 && l <= 10) ? y : 0u;
    S2K_OPAQUE(x); S2K_OPAQUE(y);
    u32 k256 = 256u; S2K_OPAQUE(k256);
    u64 f = (u64)((l <= 8) ? w : 0u);
    f += (u64)x * 31264u; f += (u64)y * k256;
    // round C                                                                          (lanes 0..11, < 2^29 + 2^16)
    const u32 g = ((u32)f & FE_M) + co_up1((u32)(f >> FE_BITS));
    // fold 2: what fold 1 left in lanes 9..11                                          (lanes 0..8, < 2^29 + 2^45)
    u32 x2 = co_down(g, 9), y2 = co_down(g, 8);
    x2 = (l <= 2) ? x2 : 0u; y2 = (l >= 1 && l <= 3) ? y2 : 0u;
    S2K_OPAQUE(x2); S2K_OPAQUE(y2);
    u64 h = (u64)((l <= 8) ? g : 0u);
    h += (u64)x2 * 31264u; h += (u64)y2 * k256;
    // round D                                                                          (lanes 0..9, < 2^29 + 2^16; lane 9: 0 or 1)
    const u32 q = ((u32)h & FE_M) + co_up1((u32)(h >> FE_BITS));
    // everything of weight >= 2^256 through 2^256 == 2^32 + 977                         (t <= 32)
    const u32 t = GROUPED ? (co_kget(q, 8) >> 24) + (co_kget(q, 9) << 5) : (co_get(q, 8) >> 24) + (co_get(q, 9) << 5);
    u32 o = (l <= 7) ? q : ((l == 8) ? (q & FE_TOPM) : 0u);
    o += (l == 0) ? t * 977u : 0u;
    o += (l == 1) ? (t << 3) : 0u;
    r.v = o;
}
S2K_D void cfe_mul(cfe& r, const cfe& a, const cfe& b) { u64 acc = 0; cfe_columns<false>(acc, a, b); cfe_reduce<false>(r, acc); }
S2K_D void cfe_sqr(cfe& r, const cfe& a) { cfe_mul(r, a, a); }
// a1*b1 + a2*b2 with one reduction (sum of the two magnitude products <= 7)
S2K_D void cfe_muladd(cfe& r, const cfe& a1, const cfe& b1, const cfe& a2, const cfe& b2) {
    u64 acc = 0; cfe_columns<false>(acc, a1, b1); cfe_columns<false>(acc, a2, b2); cfe_reduce<false>(r, acc);
}
// three products at once: group g gets (its a) * (its b); operands picked per group from replicated elements
S2K_D void cfe_mul3(cfe& r, const cfe& a0, const cfe& b0, const cfe& a1, const cfe& b1, const cfe& a2, const cfe& b2) {
    cfe a, b; a.v = co_sel3(a0.v, a1.v, a2.v); b.v = co_sel3(b0.v, b1.v, b2.v);
    u64 acc = 0; cfe_columns<true>(acc, a, b); cfe_reduce<true>(r, acc);
}
S2K_D void cfe_pick(cfe& r, const cfe& grouped, u32 g) { r.v = co_bcast(grouped.v, g); }

struct cgej { cfe x, y, z; };                 // a finite Jacobian point, cooperative form (replicated)
S2K_D void cgej_from_gej(cgej& r, const gej& a) { cfe_from_fe(r.x, a.x); cfe_from_fe(r.y, a.y); cfe_from_fe(r.z, a.z); }
S2K_D void cgej_to_gej(gej& r, const cgej& a) { cfe_to_fe(r.x, a.x); cfe_to_fe(r.y, a.y); cfe_to_fe(r.z, a.z); r.inf = 0; }
// the lean doubling of group.h (gej_double_lean), same formulas and magnitudes: X 1, Y <= 2, Z 1 in and out; a finite point of odd
// order never doubles to infinity.  Three grouped products: {Y Z, Y^2, X^2}, {-X S, L^2}, {-L (X3 + T), -S^2}.
S2K_D void cgej_double(cgej& p) {
    cfe m, z3, s, l, nx, t, x3, w, ns, y3, y3b;
    cfe_mul3(m, p.y, p.z, p.y, p.y, p.x, p.x);                 // (2, 4, 1)
    cfe_pick(z3, m, 0); cfe_pick(s, m, 1); cfe_pick(l, m, 2);  // Z3 = Y Z, S = Y^2, X^2
    cfe_mul_int(l, 3); cfe_half(l); cfe_norm_weak(l);          // L = 3/2 X^2
    cfe_neg(nx, p.x, 1);
    cfe_mul3(m, nx, s, l, l, l, l);                            // (2, 1, -)
    cfe_pick(t, m, 0); cfe_pick(x3, m, 1);                     // T = -X S, L^2
    cfe_add(x3, t); cfe_add(x3, t); cfe_norm_weak(x3);         // X3 = L^2 + 2T
    cfe_add2(w, x3, t); cfe_neg(w, w, 2);                      // -(X3 + T)       (3)
    cfe_neg(ns, s, 1);                                         // -S              (2)
    cfe_mul3(m, l, w, ns, s, ns, s);                           // (3, 2, -)
    cfe_pick(y3, m, 0); cfe_pick(y3b, m, 1);
    cfe_add(y3, y3b);                                          // Y3 = -(L (X3 + T) + S^2)   (2)
    p.x = x3; p.y = y3; p.z = z3;
}
// exact zero test of a cooperative element: through the serial sequential normalisation, every lane redundantly (cold: once or
// twice per addition)
S2K_D int cfe_is_zero(const cfe& a) { fe t; cfe_to_fe(t, a); return fe_normalizes_to_zero(t); }
// p <- p + q, both finite Jacobian points in cooperative form (12M + 4S, the formulas of gej_add_var, in six grouped products);
// complete: p == q doubles, p == -q returns 1 (the sum is the point at infinity, p is then meaningless).
// Magnitudes in: up to (5, 3, 1); out: (1, 1, 1).
S2K_D int cgej_add(cgej& p, const cgej& q) {
    cfe m, z22, z12, zz, z23, z13, u1, u2, s1, s2, h, i, h2, i2, z3, h3, t, x3, tn, nh3, y3, y3b;
    cfe_mul3(m, q.z, q.z, p.z, p.z, p.z, q.z);
    cfe_pick(z22, m, 0); cfe_pick(z12, m, 1); cfe_pick(zz, m, 2);
    cfe_mul3(m, z22, q.z, z12, p.z, p.x, z22);
    cfe_pick(z23, m, 0); cfe_pick(z13, m, 1); cfe_pick(u1, m, 2);
    cfe_mul3(m, q.x, z12, p.y, z23, q.y, z13);
    cfe_pick(u2, m, 0); cfe_pick(s1, m, 1); cfe_pick(s2, m, 2);
    cfe_neg(h, u1, 1); cfe_add(h, u2);                         // (3)
    cfe_neg(i, s1, 1); cfe_add(i, s2);                         // (3)
    if (cfe_is_zero(h)) {
        if (!cfe_is_zero(i)) return 1;                         // p == -q
        cfe_norm_weak(p.x); cfe_norm_weak(p.y);
        cgej_double(p);                                        // p == q
        return 0;
    }
    cfe_norm_weak(h); cfe_norm_weak(i);
    cfe_mul3(m, h, h, i, i, zz, h);
    cfe_pick(h2, m, 0); cfe_pick(i2, m, 1); cfe_pick(z3, m, 2);
    cfe_mul3(m, h, h2, u1, h2, u1, h2);
    cfe_pick(h3, m, 0); cfe_pick(t, m, 1);
    cfe_neg(x3, h3, 1); cfe_neg(tn, t, 1);
    cfe_add(x3, tn); cfe_add(x3, tn); cfe_add(x3, i2);         // X3 = i^2 - h^3 - 2t   (7)
    cfe_norm_weak(x3);
    cfe_neg(tn, x3, 1); cfe_add(tn, t);                        // t - X3                (3)
    cfe_neg(nh3, h3, 1);                                       // -h^3                  (2)
    cfe_mul3(m, tn, i, nh3, s1, nh3, s1);                      // (3, 2, -)
    cfe_pick(y3, m, 0); cfe_pick(y3b, m, 1);
    cfe_add(y3, y3b); cfe_norm_weak(y3);                       // Y3 = i (t - X3) - s1 h^3
    p.x = x3; p.y = y3; p.z = z3;
    return 0;
}
// the 28-word record of a Jacobian point (limbs x, y, z, infinity flag), straight into cooperative form: lane l loads limb l
S2K_D int cgej_load28(cgej& r, const u32* p28) {
    const u32 l = co_lane();
    r.x.v = (l < 9) ? p28[l] : 0u; r.y.v = (l < 9) ? p28[9 + l] : 0u; r.z.v = (l < 9) ? p28[18 + l] : 0u;
    cfe_norm_weak(r.x); cfe_norm_weak(r.y);
    return (int)p28[27];
}
// r <- 2^count * r for a serial point r that is the same in every lane of the wavefront (all 64 lanes must be here)
S2K_D void gej_double_n_cooperative(gej& r, u32 count) {
    if (r.inf || count == 0) return;
    fe_norm_weak(r.x); fe_norm_weak(r.y);
    cgej p; cgej_from_gej(p, r);
#pragma unroll 1
    for (u32 k = 0; k < count; k++) cgej_double(p);
    cgej_to_gej(r, p);
}
#endif
