// tests/gpu_prims/prims.hip -- TEST-ONLY kernels that run the per-lane device primitives (fe/scalar/group/sha256)
// one call per lane so the GPU test-suite can compare each of them with the reference at byte level
// (role of the reference's per-primitive unit tests, src/tests.c:3375-4213).  Not part of the product library.
#include "../../secp256k1_zkp_amd/csrc/gtable.h"
#include "../../secp256k1_zkp_amd/csrc/sha256.h"
#include "../../secp256k1_zkp_amd/csrc/cofield.h"
#include <hip/hip_runtime.h>

__global__ void k_prim(int op, unsigned char* out, int* flag, const unsigned char* a, const unsigned char* b, const unsigned char* c, const u32* gtab, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe x, y, z;
    switch (op) {
    case 0: fe_set_b32_mod(x, a + 32 * i); fe_set_b32_mod(y, b + 32 * i); fe_mul(z, x, y); fe_normalize(z); fe_get_b32(out + 32 * i, z); break;
    case 1: fe_set_b32_mod(x, a + 32 * i); fe_sqr(z, x); fe_normalize(z); fe_get_b32(out + 32 * i, z); break;
    case 2: fe_set_b32_mod(x, a + 32 * i); fe_inv(z, x); fe_normalize(z); fe_get_b32(out + 32 * i, z); break;
    case 3: fe_set_b32_mod(x, a + 32 * i); flag[i] = fe_sqrt(z, x); fe_normalize(z); fe_get_b32(out + 32 * i, z); break;
    case 4: fe_set_b32_mod(x, a + 32 * i); fe_half(x); fe_normalize(x); fe_get_b32(out + 32 * i, x); break;
    case 5: {
        ge p, q; gej j, t; fe_set_b32_mod(p.x, a + 64 * i); fe_set_b32_mod(p.y, a + 64 * i + 32); fe_set_b32_mod(q.x, b + 64 * i); fe_set_b32_mod(q.y, b + 64 * i + 32);
        gej_set_ge(j, p); int f = gej_add_ge(t, j, q); if (f == GEJ_ADD_NEEDS_DOUBLE) { gej u; gej_double(u, t); t = u; }
        flag[i] = t.inf; ge r; ge_set_gej(r, t); fe_get_b32(out + 64 * i, r.x); fe_get_b32(out + 64 * i + 32, r.y);
    } break;
    case 6: {
        ge p; gej j, t; fe_set_b32_mod(p.x, a + 64 * i); fe_set_b32_mod(p.y, a + 64 * i + 32);
        gej_set_ge(j, p); gej_double(t, j); flag[i] = t.inf; ge r; ge_set_gej(r, t); fe_get_b32(out + 64 * i, r.x); fe_get_b32(out + 64 * i + 32, r.y);
    } break;
    case 7: {   // Jacobian + Jacobian with random Z's (c holds two 32-byte z values per item)
        ge p, q; gej ja, jb, r; fe za, zb, z2, z3;
        fe_set_b32_mod(p.x, a + 64 * i); fe_set_b32_mod(p.y, a + 64 * i + 32); fe_set_b32_mod(q.x, b + 64 * i); fe_set_b32_mod(q.y, b + 64 * i + 32);
        fe_set_b32_mod(za, c + 64 * i); fe_set_b32_mod(zb, c + 64 * i + 32);
        gej_set_ge(ja, p); gej_set_ge(jb, q);
        fe_sqr(z2, za); fe_mul(z3, z2, za); fe_mul(ja.x, ja.x, z2); fe_mul(ja.y, ja.y, z3); ja.z = za;
        fe_sqr(z2, zb); fe_mul(z3, z2, zb); fe_mul(jb.x, jb.x, z2); fe_mul(jb.y, jb.y, z3); jb.z = zb;
        gej_add_var(r, ja, jb); flag[i] = r.inf; ge o; ge_set_gej(o, r); fe_get_b32(out + 64 * i, o.x); fe_get_b32(out + 64 * i + 32, o.y);
    } break;
    case 8: { scalar s, t; sc_set_b32(s, a + 32 * i, nullptr); sc_set_b32(t, b + 32 * i, nullptr); sc_mul(s, s, t); sc_get_b32(out + 32 * i, s); } break;
    case 9: { scalar s, r1, r2; sc_set_b32(s, a + 32 * i, nullptr); sc_split_lambda(r1, r2, s); sc_get_b32(out + 64 * i, r1); sc_get_b32(out + 64 * i + 32, r2); } break;
    case 10: { sha256_stream h; sha256_stream_init(h); sha256_stream_write(h, a + 100 * i, 100); sha256_stream_finalize(h, out + 32 * i); } break;
    case 11: { const u32 wv = ((const u32*)a)[i]; ge g; gtab_load(g, gtab, wv >> S2K_GTAB_MAX_BITS, wv & ((1u << S2K_GTAB_MAX_BITS) - 1u));      /* (window, magnitude) packed at 26 bits whatever the table's width */ fe_normalize(g.x); fe_normalize(g.y); fe_get_b32(out + 64 * i, g.x); fe_get_b32(out + 64 * i + 32, g.y); } break;
    case 20: {   // intermediates of gej_double for debugging: Z3, S, L, T, X3, S2, (X3+T), Y3pre
        ge p; fe_set_b32_mod(p.x, a + 64 * i); fe_set_b32_mod(p.y, a + 64 * i + 32);
        fe X = p.x, Y = p.y, l, s, t, rx, ry, rz, one; fe_set_int(one, 1);
        unsigned char* o = out + 256 * i;
        fe_mul(rz, Y, one); { fe q = rz; fe_normalize(q); fe_get_b32(o, q); }
        fe_sqr(s, Y); { fe q = s; fe_normalize(q); fe_get_b32(o + 32, q); }
        fe_sqr(l, X); fe_mul_int(l, 3); fe_half(l); fe_norm_weak(l); { fe q = l; fe_normalize(q); fe_get_b32(o + 64, q); }
        fe_mul(t, X, s); fe_neg(t, t, 1); { fe q = t; fe_normalize(q); fe_get_b32(o + 96, q); }
        fe_sqr(rx, l); fe_add(rx, t); fe_add(rx, t); { fe q = rx; fe_normalize(q); fe_get_b32(o + 128, q); }
        fe_sqr(s, s); { fe q = s; fe_normalize(q); fe_get_b32(o + 160, q); }
        fe_add(t, rx); { fe q = t; fe_normalize(q); fe_get_b32(o + 192, q); }
        fe_mul(ry, t, l); { fe q = ry; fe_normalize(q); fe_get_b32(o + 224, q); }
    } break;
    // lazy-limb inputs (9 raw uint32 limbs per operand, magnitudes up to the contract's limits): the products on the DEVICE at
    // the edge of the 64-bit column accumulators
    case 30: case 31: case 32: case 33: case 34: case 35: case 36: {
        fe u, v2, w; const u32* la = (const u32*)a + 9 * i; const u32* lb = b ? (const u32*)b + 9 * i : la; const u32* lc = c ? (const u32*)c + 9 * i : la;
        for (int k = 0; k < 9; k++) { u.n[k] = la[k]; v2.n[k] = lb[k]; w.n[k] = lc[k]; }
        fe r1, r2; fe_set_zero(r2);
        if (op == 30) fe_mul(r1, u, v2);
        else if (op == 31) fe_sqr(r1, u);
        else if (op == 32) fe_mul2(r1, u, v2, r2, w, v2);                       // a*b, c*b
        else if (op == 33) fe_muladd<false, false>(r1, u, v2, w, u);            // a*b + c*a
        else if (op == 34) fe_muladd<false, true>(r1, u, v2, w, w);             // a*b + c^2
        else if (op == 35) fe_mul_sqr(r1, u, v2, r2, w);                        // a*b, c^2
        else { fe_sqr2(r1, u, r2, w); }                                         // a^2, c^2
        fe_normalize(r1); fe_normalize(r2);
        fe_get_b32(out + 64 * i, r1); fe_get_b32(out + 64 * i + 32, r2);
    } break;
    case 39: {   // wave-cooperative field arithmetic (cofield.h) against the integers / the reference: one item per WAVEFRONT (n = 64 * items,
                 // 64-lane blocks).  a, b: 9 raw limbs each (lazy magnitudes chosen by the test), c: an affine point.
                 // out (288 bytes per item, written by lane 0): a*b | a*b + b*b | b - 3a/2 | 2^13 * c (x, y) | grouped product: a*b | b*b | a*a
        const int it = i >> 6;
        fe fa, fb; for (int k = 0; k < 9; k++) { fa.n[k] = ((const u32*)a)[9 * it + k]; fb.n[k] = ((const u32*)b)[9 * it + k]; }
        cfe ca, cb, r1, r2, r3; cfe_from_fe(ca, fa); cfe_from_fe(cb, fb);
        cfe_mul(r1, ca, cb);
        cfe_muladd(r2, ca, cb, cb, cb);
        r3 = ca; cfe_mul_int(r3, 3); cfe_half(r3); cfe_norm_weak(r3); cfe_neg(r3, r3, 1); cfe_add(r3, cb); cfe_norm_weak(r3);
        cfe m3, g0, g1, g2; cfe_mul3(m3, ca, cb, cb, cb, ca, ca); cfe_pick(g0, m3, 0); cfe_pick(g1, m3, 1); cfe_pick(g2, m3, 2);
        ge p; fe_set_b32_mod(p.x, c + 64 * it); fe_set_b32_mod(p.y, c + 64 * it + 32);
        gej j; gej_set_ge(j, p); gej_double_n_cooperative(j, 13);
        ge o; ge_set_gej(o, j);
        // cooperative addition: 2^13 c + c (generic), c + c (doubling branch), c + (-c) (infinity): flag bits 1..3
        int addok = 0;
        {
            cgej cj, cp, cn; gej t1, pj; gej_set_ge(pj, p);
            cgej_from_gej(cj, j); cgej_from_gej(cp, pj);
            const int inf1 = cgej_add(cj, cp); cgej_to_gej(t1, cj);
            gej want; gej_add_var(want, j, pj);
            ge g1, g2; ge_set_gej(g1, t1); ge_set_gej(g2, want);
            addok |= (!inf1 && fe_equal(g1.x, g2.x) && fe_equal(g1.y, g2.y)) ? 2 : 0;
            cgej_from_gej(cj, pj); const int inf2 = cgej_add(cj, cp); cgej_to_gej(t1, cj);
            gej d2; gej_double(d2, pj); ge_set_gej(g1, t1); ge_set_gej(g2, d2);
            addok |= (!inf2 && fe_equal(g1.x, g2.x) && fe_equal(g1.y, g2.y)) ? 4 : 0;
            gej nj = pj; fe_neg(nj.y, nj.y, 1); fe_norm_weak(nj.y); cgej_from_gej(cn, nj); cgej_from_gej(cj, pj);
            addok |= cgej_add(cj, cn) ? 8 : 0;
        }
        fe q1, q2, q3; cfe_to_fe(q1, r1); cfe_to_fe(q2, r2); cfe_to_fe(q3, r3);
        // lanes >= 9 must still hold zero
        fe q4, q5, q6; cfe_to_fe(q4, g0); cfe_to_fe(q5, g1); cfe_to_fe(q6, g2);
        // the three copies of a replicated element agree
        const u32 differ = (g0.v ^ co_bcast(g0.v, 1)) | (g1.v ^ co_bcast(g1.v, 2)) | (r1.v ^ co_bcast(r1.v, 2)) | (r3.v ^ co_bcast(r3.v, 1));
        const u32 stray = ((co_lane() >= 9) ? (r1.v | r2.v | r3.v | g0.v | g1.v | g2.v) : 0u) | differ;
        const int clean = !__any((int)(stray != 0u));
        if ((i & 63) == 0) {
            unsigned char* w = out + 288 * it;
            fe_normalize(q4); fe_get_b32(w + 192, q4); fe_normalize(q5); fe_get_b32(w + 224, q5); fe_normalize(q6); fe_get_b32(w + 256, q6);
            fe_normalize(q1); fe_get_b32(w, q1); fe_normalize(q2); fe_get_b32(w + 32, q2); fe_normalize(q3); fe_get_b32(w + 64, q3);
            fe_normalize(o.x); fe_normalize(o.y); fe_get_b32(w + 96, o.x); fe_get_b32(w + 128, o.y);
            flag[i] = clean | addok;
        }
    } break;
    case 38: {   // two-piece double multiplication (ecmult_lane_split) given A (random Z from c) and T = 2^64 A, with the caller's fallback to
                 // ecmult_lane; a = points, b = (na || ng) 64 bytes per item; flag = 2 * took_split + infinity
        __shared__ u32 s_dig[S2K_DIG_WORDS * 256];
        ge p; fe_set_b32_mod(p.x, a + 64 * i); fe_set_b32_mod(p.y, a + 64 * i + 32);
        gej A, T, R; gej_set_ge(A, p);
        { fe zz, z2, z3; fe_set_b32_mod(zz, c + 32 * i); fe_norm_weak(zz); fe_sqr(z2, zz); fe_mul(z3, z2, zz); fe_mul(A.x, A.x, z2); fe_mul(A.y, A.y, z3); A.z = zz; }
        T = A; for (int k = 0; k < 64; k++) { gej t; gej_double(t, T); T = t; }
        scalar na, ng; sc_set_b32(na, b + 64 * i, nullptr); sc_set_b32(ng, b + 64 * i + 32, nullptr);
        u32* ptab = (u32*)flag + n + (size_t)i * S2K_PTAB_WORDS;          // scratch behind the flags (the test allocates it)
        const lane_mem lm{ptab, S2K_LANE_DIG(s_dig)};
        const int done = ecmult_lane_split(R, A, T, na, ng, 1, gtab, lm);
        if (!S2K_WAVE_ALL(done)) ecmult_lane(R, A, na, ng, 1, gtab, lm);
        ge r; r.x = R.x; r.y = R.y; if (!R.inf) ge_set_gej(r, R);
        flag[i] = 2 * (S2K_WAVE_ALL(done) ? 1 : 0) + (R.inf ? 1 : 0);
        fe_normalize(r.x); fe_normalize(r.y); fe_get_b32(out + 64 * i, r.x); fe_get_b32(out + 64 * i + 32, r.y);
    } break;
    case 40: {   // the ring form (ecmult.h: ecmult_ring_tables + ecmult_ring_step) as the rangeproof rings use it: R = e*A + s*G + f*G (the table of G
                 // stands in for the generator's); a = points, b = (e || s || f) 96 bytes per item; flag = 2 * completed + infinity.  One wavefront
                 // per 64 items (64-lane workgroups); scratch behind the flags: n * S2K_RTAB_WORDS, then one parking area per wavefront, then
                 // n * S2K_PTAB_WORDS for the caller's fallback: a wavefront whose step returns 0 (an exceptional addition) takes the general
                 // form on the same point, as k_rp_rings_shared -> k_rp_rings does.
        __shared__ u32 s_dig[S2K_RING_DIG_WORDS * 256];
        ge p; fe_set_b32_mod(p.x, a + 64 * i); fe_set_b32_mod(p.y, a + 64 * i + 32); fe_norm_weak(p.x); fe_norm_weak(p.y);
        gej A, T, R; gej_set_ge(A, p);
        T = A; for (int k = 0; k < 64; k++) gej_double_lean(T, T);
        fe_norm_weak(T.y);
        scalar e, sg, f; sc_set_b32(e, b + 96 * i, nullptr); sc_set_b32(sg, b + 96 * i + 32, nullptr); sc_set_b32(f, b + 96 * i + 64, nullptr);
        u32* const scratch = (u32*)flag + n;
        u32* rtab = scratch + (size_t)i * S2K_RTAB_WORDS;
        u32* raw = scratch + (size_t)n * S2K_RTAB_WORDS + (size_t)(i >> 6) * S2K_RRAW_WAVE_WORDS + (i & 63);
        u32* ptab = scratch + (size_t)n * S2K_RTAB_WORDS + (size_t)((n + 63) >> 6) * S2K_RRAW_WAVE_WORDS + (size_t)i * S2K_PTAB_WORDS;
        ecmult_ring_tables(rtab, raw, A, T);
        const int done = S2K_WAVE_ALL(ecmult_ring_step(R, rtab, e, sg, f, 1, gtab, gtab, S2K_LANE_DIG(s_dig)));
        if (!done) {
            scalar sf; sc_add(sf, sg, f);
            const lane_mem lm{ptab, S2K_LANE_DIG(s_dig)};
            ecmult_lane(R, A, e, sf, 1, gtab, lm);
        } else R.inf = 0;
        ge r; fe_set_zero(r.x); fe_set_zero(r.y);
        if (!R.inf) ge_set_gej(r, R);
        flag[i] = 2 * done + (R.inf ? 1 : 0);
        fe_normalize(r.x); fe_normalize(r.y); fe_get_b32(out + 64 * i, r.x); fe_get_b32(out + 64 * i + 32, r.y);
    } break;
    case 37: {   // lean point operations (group.h) against the general ones: double, then add b, from an affine start
        ge p, q; fe_set_b32_mod(p.x, a + 64 * i); fe_set_b32_mod(p.y, a + 64 * i + 32); fe_set_b32_mod(q.x, b + 64 * i); fe_set_b32_mod(q.y, b + 64 * i + 32);
        gej j; gej_set_ge(j, p);
        gej_double_lean(j, j); gej_double_lean(j, j);
        gej t; const int same = gej_add_ge_lean(t, j, q);
        flag[i] = same;
        gej_double_lean(t, t);
        ge r; ge_set_gej(r, t); fe_get_b32(out + 64 * i, r.x); fe_get_b32(out + 64 * i + 32, r.y);
    } break;
    case 12: { scalar s; int o; sc_set_b32(s, a + 32 * i, &o); flag[i] = o; sc_negate(s, s); sc_get_b32(out + 32 * i, s); } break;
    case 13: { scalar s; sc_set_b32(s, a + 32 * i, nullptr); sc_inverse(s, s); sc_get_b32(out + 32 * i, s); } break;
    }
}
extern "C" __attribute__((visibility("default")))
int s2k_test_prim(int op, unsigned char* out, int* flag, const unsigned char* a, const unsigned char* b, const unsigned char* c, const void* gtab, int n) {
    hipLaunchKernelGGL(k_prim, dim3((n + 63) / 64), dim3(64), 0, 0, op, out, flag, a, b, c, (const u32*)gtab, n);
    return hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess;
}
