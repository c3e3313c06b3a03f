// modinv.h -- inverse modulo the field prime p and the group order n by division steps, organised for a 64-lane wavefront.
//
// Role of the reference's secp256k1_fe_inv_var / secp256k1_scalar_inverse_var (src/field_5x52_impl.h:491-522,
// src/scalar_4x64_impl.h, both on src/modinv64_impl.h).  The algorithm is Bernstein-Yang "safegcd" in the half-delta form
// whose iteration bound the reference's doc/safegcd_implementation.md (sections 1-5) derives: starting from delta = 1/2,
// 590 division steps bring any g below a 256-bit odd modulus f to 0; 20 batches of 30 are run.  The *organisation* is this
// engine's own:
//
//  * Batch kernel (`ds_batch`): the 30 steps of a batch only look at the low 32 bits of f and g.  A lane does not walk them one
//    by one: every loop trip first retires the whole run of even-g steps at once (one count-trailing-zeros, three variable
//    shifts) and then one odd-g step written with selects, so a trip advances a lane by 2 steps on average.  The trip loop is
//    closed by a wavefront vote -- it runs until the slowest lane has used its 30 steps (typically 18-21 trips), lanes that are
//    done ride along with all updates predicated off.  No lane ever branches on its data.
//  * Transition matrices are applied to (f, g) exactly and to (d, e) modulo m with 32x32+64 signed multiply-accumulates
//    (v_mad_i64_i32), nine 30-bit limbs per operand.  (d, e) are NOT renormalised per batch: each batch adds a multiple
//    k*m, 0 <= k < 2^30, chosen to clear the low 30 bits, which lets |d|, |e| grow by at most m per batch -- 21 m < 2^261
//    after 20 batches, well inside the 270-bit signed limbs -- and one short reduction at the very end (quotient estimate
//    from the top limb, one multiply-subtract, one conditional subtraction) brings the result to [0, m).
//
// Everything is branch-free per lane and the same for both moduli; the modulus is a compile-time constant, so p's zero limbs
// cost nothing.  Result: a^-1 mod m, and 0 for a = 0.
#pragma once
#include "s2k_common.h"

#define DS_LIMBS 9
#define DS_BITS 30
#define DS_MASK ((int32_t)0x3FFFFFFF)
#define DS_BATCHES 20

struct ds_int { int32_t w[DS_LIMBS]; };                 // value = sum w[i] 2^(30 i); w[0..7] in [0, 2^30), w[8] signed
struct ds_modulus { int32_t w[DS_LIMBS]; u32 inv; };    // modulus in that form (limbs may be negative: a sparse signed form), m^-1 mod 2^30

// p = 2^256 - 2^32 - 977 = 65536 * 2^240 - 4 * 2^30 - 977
#define DS_MOD_P ds_modulus{{-977, -4, 0, 0, 0, 0, 0, 0, 65536}, 0x2DDACACFu}
// n = group order of secp256k1
#define DS_MOD_N ds_modulus{{0x10364141, 0x3F497A33, 0x348A03BB, 0x2BB739AB, 0x3FFFFEBA, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0xFFFF}, 0x2A774EC1u}

#if defined(__HIP_DEVICE_COMPILE__)
#define DS_WAVE_ANY(p) (__any(p))
#define DS_WAVE_ALL(p) (__all(p))
#else
#define DS_WAVE_ANY(p) (p)
#define DS_WAVE_ALL(p) (p)
#endif

S2K_HD int ds_ctz32(u32 x) { return __builtin_ctz(x | 0x80000000u); }         // <= 31 also for x = 0

// 30 division steps on the low words.  zeta = -(delta + 1/2).  Returns the new zeta and the transition matrix
// t = (u v; q r) scaled by 2^30:  2^30 * (f', g') = t * (f, g).
S2K_HD int32_t ds_batch(int32_t zeta, u32 f, u32 g, int32_t t[4]) {
    int32_t u = 1, v = 0, q = 0, r = 1;
    int left = DS_BITS;
    while (DS_WAVE_ANY(left > 0)) {
        // the whole run of even g in one go: g/2^z, delta + z, f's row of the matrix catches up by 2^z
        int z = ds_ctz32(g); z = z < left ? z : left;
        g >>= z; u = (int32_t)((u32)u << z); v = (int32_t)((u32)v << z); zeta -= z; left -= z;
        // one step with g odd (lanes with steps left): delta > 0 swaps, (f, g) <- (g, (g - f)/2), else g <- (g + f)/2
        const int act = left > 0;
        const int sw = act & (zeta < 0);
        const u32 fm = sw ? 0u - f : f;
        const int32_t um = sw ? -u : u, vm = sw ? -v : v;
        const u32 g2 = (g + fm) >> 1;
        const int32_t q2 = q + um, r2 = r + vm;
        f = sw ? g : f; u = sw ? q : u; v = sw ? r : v;
        g = act ? g2 : g; q = act ? q2 : q; r = act ? r2 : r;
        zeta = sw ? -zeta - 2 : zeta - act;
        u = (int32_t)((u32)u << act); v = (int32_t)((u32)v << act); left -= act;
    }
    t[0] = u; t[1] = v; t[2] = q; t[3] = r;
    return zeta;
}

// (a, b) <- (t00 a + t01 b + ka m, t10 a + t11 b + kb m) / 2^30.  WITH_M = false: the exact (f, g) update (ka = kb = 0, the
// low 30 bits vanish by construction); WITH_M = true: the (d, e) update, ka / kb in [0, 2^30) clear the low 30 bits.
template <bool WITH_M>
S2K_HD void ds_apply(ds_int& a, ds_int& b, const int32_t t[4], const ds_modulus md) {
    const int64_t t00 = t[0], t01 = t[1], t10 = t[2], t11 = t[3];
    int64_t ca = t00 * a.w[0] + t01 * b.w[0], cb = t10 * a.w[0] + t11 * b.w[0];
    int64_t ka = 0, kb = 0;
    if (WITH_M) {
        ka = (int64_t)((0u - (u32)ca * md.inv) & (u32)DS_MASK);            // ka * m == -ca  (mod 2^30)
        kb = (int64_t)((0u - (u32)cb * md.inv) & (u32)DS_MASK);
        ca += ka * md.w[0]; cb += kb * md.w[0];
    }
    ca >>= DS_BITS; cb >>= DS_BITS;
#pragma unroll
    for (int i = 1; i < DS_LIMBS; i++) {
        ca += t00 * a.w[i] + t01 * b.w[i];
        cb += t10 * a.w[i] + t11 * b.w[i];
        if (WITH_M && md.w[i] != 0) { ca += ka * md.w[i]; cb += kb * md.w[i]; }
        a.w[i - 1] = (int32_t)ca & DS_MASK; ca >>= DS_BITS;
        b.w[i - 1] = (int32_t)cb & DS_MASK; cb >>= DS_BITS;
    }
    a.w[DS_LIMBS - 1] = (int32_t)ca; b.w[DS_LIMBS - 1] = (int32_t)cb;
}

// r <- r + c * m, carries propagated (|c| small)
S2K_HD void ds_add_multiple(ds_int& r, int32_t c, const ds_modulus md) {
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < DS_LIMBS - 1; i++) { acc += (int64_t)r.w[i] + (int64_t)c * md.w[i]; r.w[i] = (int32_t)acc & DS_MASK; acc >>= DS_BITS; }
    r.w[DS_LIMBS - 1] = (int32_t)(acc + r.w[DS_LIMBS - 1] + (int64_t)c * md.w[DS_LIMBS - 1]);
}

// w (8 little-endian 32-bit words, value < m) -> w^-1 mod m (0 for 0)
S2K_HD void ds_inverse_words(u32 o[8], const u32 w[8], const ds_modulus md) {
    ds_int f, g, d, e;
#pragma unroll
    for (int i = 0; i < DS_LIMBS; i++) {
        const int bit = DS_BITS * i, idx = bit >> 5, sh = bit & 31;
        u64 v = w[idx];
        if (idx + 1 < 8) v |= (u64)w[idx + 1] << 32;
        g.w[i] = (int32_t)((u32)(v >> sh) & (u32)DS_MASK);
        f.w[i] = md.w[i]; d.w[i] = 0; e.w[i] = 0;
    }
    e.w[0] = 1;
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < DS_BATCHES; it++) {
        int32_t t[4];
        zeta = ds_batch(zeta, (u32)f.w[0], (u32)g.w[0], t);
        ds_apply<true>(d, e, t, md);
        ds_apply<false>(f, g, t, md);
        // 590 steps is the worst case; random inputs are through after 501..531 of them (18 batches).  Once g = 0 a further batch leaves f
        // and d as they are (its matrix is diag(2^30, 1): d <- (2^30 d + 0 m) / 2^30), so stopping when EVERY lane of the wavefront has
        // g = 0 gives bit for bit the result of running all 20 batches.
        if (it >= DS_BATCHES - 5) {
            int32_t gz = 0;
#pragma unroll
            for (int i = 0; i < DS_LIMBS; i++) gz |= g.w[i];
            if (DS_WAVE_ALL(gz == 0)) break;
        }
    }
    // now g = 0, f = +-gcd = +-1 (or f = +-m when the input was 0, with d = 0), and d * input == f (mod m), |d| < 21 m.
    // r = d + 32 m > 0; quotient estimate q = r >> 256 (m = 2^256 - c, c < 2^129, so r - q m = (r mod 2^256) + q c < 2 m)
    ds_add_multiple(d, 32, md);
    const int32_t qest = d.w[DS_LIMBS - 1] >> 16;
    ds_add_multiple(d, -qest, md);
    { ds_int s = d; ds_add_multiple(s, -1, md); const int keep = s.w[DS_LIMBS - 1] >= 0;                 // one conditional subtraction
#pragma unroll
      for (int i = 0; i < DS_LIMBS; i++) d.w[i] = keep ? s.w[i] : d.w[i]; }
    // sign of f: a negative f means the inverse is -d = m - d (for d != 0)
    { const int fneg = f.w[DS_LIMBS - 1] < 0;
      int32_t nz = 0;
#pragma unroll
      for (int i = 0; i < DS_LIMBS; i++) nz |= d.w[i];
      ds_int s;
#pragma unroll
      for (int i = 0; i < DS_LIMBS; i++) s.w[i] = -d.w[i];
      // -d + m, limbs renormalised
      int64_t acc = 0;
#pragma unroll
      for (int i = 0; i < DS_LIMBS - 1; i++) { acc += (int64_t)s.w[i] + md.w[i]; s.w[i] = (int32_t)acc & DS_MASK; acc >>= DS_BITS; }
      s.w[DS_LIMBS - 1] = (int32_t)(acc + s.w[DS_LIMBS - 1] + md.w[DS_LIMBS - 1]);
      const int take = fneg & (nz != 0);
#pragma unroll
      for (int i = 0; i < DS_LIMBS; i++) d.w[i] = take ? s.w[i] : d.w[i]; }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int bit = 32 * j, i = bit / DS_BITS, sh = bit % DS_BITS;
        u64 v = (u64)(u32)d.w[i] >> sh;
        if (i + 1 < DS_LIMBS) v |= (u64)(u32)d.w[i + 1] << (DS_BITS - sh);
        if (i + 2 < DS_LIMBS) v |= (u64)(u32)d.w[i + 2] << (2 * DS_BITS - sh);
        o[j] = (u32)v;
    }
}
