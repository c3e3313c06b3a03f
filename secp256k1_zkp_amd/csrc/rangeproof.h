// rangeproof.h -- Borromean rangeproof verification (secp256k1_rangeproof_verify) as five device stages.
//
// Reference call stack (src/modules/rangeproof/): main_impl.h:54-71 -> rangeproof_impl.h:541-683
// (getheader_impl :487-538, pub_expand :20-51) -> borromean_impl.h:53-104.  One proof is 32 rings x 4 steps of
//      e_{j+1} = SHA256( ser33( s_j G + e_j P_j ) || m || i || j+1 )
// -- serial inside a ring, independent across rings and proofs.  The mapping chosen for gfx950:
//
//   K0  rp_prologue   1 lane / proof : header parse + structural checks, commitment / generator load, the message
//                                      hash m, min_value*H, and the chain of ring bases  B_i = -(4^i 10^exp H)
//   K1  rp_lift       1 lane / ring  : C_i = lift_x(proof x_i) (one field sqrt per lane) -> first public key of ring i
//   K2  rp_sum        1 lane / proof : last ring's key  = commit - min_value*H - sum C_i
//   K3  rp_rings      1 lane / ring  : the 4-step chain: double multiplication (ecmult.h), to-affine, serialise, SHA-256
//   K4  rp_final      1 lane / proof : e0 == SHA256(r_0 || ... || r_{rings-1} || m) and all flags
//
// K3 is >98% of the work (128 double multiplications per 64-bit proof); K0/K2/K4 run one lane per proof but are
// two orders of magnitude cheaper, so their low parallelism does not matter.  Stages communicate through a
// per-proof scratch record in HBM (layout below); nothing is read back by the host between stages.
#pragma once
#include "gtable.h"
#include "sha256.h"

#define RP_MAX_RINGS 32
#define RP_GEJ_WORDS 28
#define RP_RING_OUT_BYTES (RP_MAX_RINGS * 33 + 32)      /* 1088 = 17 SHA-256 blocks for a full-size proof */
#define RP_GSLOT_NONE 0xFFFFFFFFu

struct rp_rec {
    u32 ok;            // structural checks passed (K0); cleared by later stages on failure
    u32 rings;         // 1..32
    u32 last_rsize;    // size of the last ring: 4, 2 (odd mantissa) or 1 (mantissa 0)
    u32 off_signs;     // byte offsets inside the proof
    u32 off_pts;
    u32 off_e0;
    u32 off_s;
    u32 hdr;           // bit 0: header and structure parsed (K0a); bits 8..15: exp + 1
    u32 gslot;         // slot of the proof's generator in the engine's table cache, RP_GSLOT_NONE when it has none (shared-generator form of K3)
    u32 m[8];          // message hash, big-endian words
    u32 commit[18];    // commitment, affine limbs (x, y)
    u32 accj[RP_GEJ_WORDS];   // min_value * H (infinity when min_value == 0)
};

struct rp_ws {           // device pointers into the engine workspace
    rp_rec* rec;         // [n]
    u32* bases;          // [n][32][28]
    u32* pub0;           // [n][32][28]
    u32* dbases;         // [n][32][28]  2^64 * B_i : the step of T = 2^64 * key from one ring position to the next (ecmult_lane_split)
    u32* tcur;           // [n][32][28]  2^64 * (current key of the ring)
    unsigned char* lift_ok;   // [n][32]
    unsigned char* ring_out;  // [n][RP_RING_OUT_BYTES]: the rings' 33-byte outputs back to back, then room for m
    unsigned char* ring_ok;   // [n][32]
    u32* plan;                // work lists of K3 (k_rp_plan): [0] groups in mapF, [1] rings in mapG
    u32* mapF;                // [n * 32 / K] proof | ring group << 20   (shared-generator form)
    u32* mapG;                // [n * 32]     proof | ring << 20         (general form)
};

S2K_HD void gej_store28_h(u32* p, const gej& a) {
    fe x = a.x, y = a.y, z = a.z;
    fe_norm_weak(x); fe_norm_weak(y); fe_norm_weak(z);
    for (int i = 0; i < 9; i++) { p[i] = x.n[i]; p[9 + i] = y.n[i]; p[18 + i] = z.n[i]; }
    p[27] = (u32)a.inf;
}
S2K_HD void gej_load28_h(gej& a, const u32* p) {
    for (int i = 0; i < 9; i++) { a.x.n[i] = p[i]; a.y.n[i] = p[9 + i]; a.z.n[i] = p[18 + i]; }
    a.inf = (int)p[27];
}

// ---- K0 -------------------------------------------------------------------------------------------------
// cf. secp256k1_rangeproof_getheader_impl (rangeproof_impl.h:487-538).  min/max are written exactly where the
// reference writes through its pointers, so that a failed parse leaves the same partial values behind.
S2K_HD int rp_getheader(u32& offset, int& exp, int& mantissa, u64& scale, u64* min_value, u64* max_value, const unsigned char* proof, u64 plen) {
    offset = 0;
    if (plen < 65 || ((proof[0] & 128) != 0)) return 0;
    const int has_nz_range = proof[0] & 64, has_min = proof[0] & 32;
    exp = -1; mantissa = 0;
    if (has_nz_range) {
        exp = proof[0] & 31;
        offset += 1;
        if (exp > 18) return 0;
        mantissa = proof[offset] + 1;
        if (mantissa > 64) return 0;
        *max_value = 0xFFFFFFFFFFFFFFFFull >> (64 - mantissa);
    } else {
        *max_value = 0;
    }
    offset += 1;
    scale = 1;
    for (int i = 0; i < exp; i++) {
        if (*max_value > 0xFFFFFFFFFFFFFFFFull / 10) return 0;
        *max_value *= 10; scale *= 10;
    }
    *min_value = 0;
    if (has_min) {
        if (plen - offset < 8) return 0;
        u64 v = 0;
        for (int i = 0; i < 8; i++) v = (v << 8) | proof[offset + i];
        *min_value = v;
        offset += 8;
    }
    if (*max_value > 0xFFFFFFFFFFFFFFFFull - *min_value) return 0;
    *max_value += *min_value;
    return 1;
}

// K0a: header and structure (rangeproof_impl.h:541-608 up to the point where curve arithmetic starts).  Cheap; K1 (lift) only
// needs this part, so the expensive part below (K0b) can run next to K1.
S2K_HD void rp_header(rp_rec& rec, u64* min_value, u64* max_value, const unsigned char* proof, u64 plen) {
    rec.ok = 0; rec.rings = 0; rec.last_rsize = 0; rec.hdr = 0; rec.gslot = RP_GSLOT_NONE;
    rec.off_signs = 0; rec.off_pts = 0; rec.off_e0 = 0; rec.off_s = 0;
    *min_value = 0; *max_value = 0;
    u32 offset; int exp, mantissa; u64 scale;
    if (!rp_getheader(offset, exp, mantissa, scale, min_value, max_value, proof, plen)) return;
    u32 rings = 1, npub = 1, last_rsize = 1;
    if (mantissa != 0) {
        rings = (u32)mantissa >> 1; npub = rings * 4; last_rsize = 4;
        if (mantissa & 1) { npub += 2; rings++; last_rsize = 2; }
    }
    if (plen - offset < (u64)32 * (npub + rings - 1) + 32 + ((rings + 6) >> 3)) return;
    rec.rings = rings; rec.last_rsize = last_rsize;
    rec.off_signs = offset;
    offset += (rings + 6) >> 3;
    if ((rings - 1) & 7) {
        if ((proof[offset - 1] >> ((rings - 1) & 7)) != 0) return;
    }
    rec.off_pts = offset;
    rec.off_e0 = offset + 32 * (rings - 1);
    rec.off_s = rec.off_e0 + 32;
    if ((u64)rec.off_s + (u64)32 * npub != plen) return;        // "Extra data found, reject" (:643-646); too-short was caught above
    rec.hdr = 1u | ((u32)(exp + 1) << 8);
}
// K0b: the per-proof point work (:588-651, pub_expand :20-51) in three independent parts, so that the kernel can give each part
// its own wave (one lane per proof in each): A commitment load + min_value*H, B generator flag + message hash, C ring bases.
S2K_HD void rp_load_generator(ge& g, const unsigned char* gen64) {                  // generator/main_impl.h:40-49
    fe_set_b32_mod(g.x, gen64); fe_set_b32_mod(g.y, gen64 + 32);
    fe_normalize(g.x); fe_normalize(g.y);
}
// A: commitment: x = b32 mod p, y = sqrt(x^3+7), negated when bit 0 of the prefix is set (generator/main_impl.h:266-273);
//    accj = min_value * H (pedersen_ecmult_small, generator/pedersen_impl.h:33-38): plain double-and-add, the same group
//    element as the reference's ecmult_const
// Returns whether the 33 bytes are an encoding secp256k1_pedersen_commitment_parse accepts (generator/main_impl.h:281-297: prefix 8 or 9,
// x below p, x on the curve).  The reference's verifier only ever sees parsed objects, so for a caller that hands over serialised bytes
// a refused encoding must end as "invalid" here -- the load below would otherwise read just bit 0 of the prefix and lift whatever x says
// (found by the differential fuzz, round 4: a flipped bit 7 of the prefix on an otherwise valid proof was accepted).
S2K_HD int rp_pp_commit(rp_rec& rec, u64 min_value, const unsigned char* commit33, const unsigned char* gen64) {
    if (!(rec.hdr & 1u)) return 0;
    ge c;
    int valid = (commit33[0] & 0xFEu) == 8u;
    {
        fe x; valid &= fe_set_b32_limit(x, commit33 + 1);
        valid &= ge_set_xquad(c, x);
        fe_normalize(c.x); fe_normalize(c.y);
        if (commit33[0] & 1) { fe_neg(c.y, c.y, 1); fe_normalize(c.y); }
    }
    for (int i = 0; i < 9; i++) { rec.commit[i] = c.x.n[i]; rec.commit[9 + i] = c.y.n[i]; }
    gej acc; gej_set_infinity(acc);
    if (min_value) {
        ge g; rp_load_generator(g, gen64);
        for (int bit = 63; bit >= 0; bit--) {
            gej t; gej_double(t, acc); acc = t;
            if ((min_value >> bit) & 1) {
                const int f = gej_add_ge(t, acc, g); acc = t;
                if (f == GEJ_ADD_NEEDS_DOUBLE) { gej_double(t, acc); acc = t; }
            }
        }
    }
    gej_store28_h(rec.accj, acc);
    return valid;
}
// B: m = SHA256( ser(commit) || ser(gen) || proof[0..off_hdr) || (sign_i || x_i)_{i<rings-1} || extra )   (:588-651)
//    ser(point) = [ !is_square(y) ] || x   (rangeproof_serialize_point :53-59)
S2K_HD void rp_pp_hash(rp_rec& rec, const unsigned char* commit33, const unsigned char* proof, const unsigned char* extra, u64 extra_len,
                       const unsigned char* gen64) {
    if (!(rec.hdr & 1u)) return;
    const u32 rings = rec.rings, off_hdr = rec.off_signs;
    ge g; rp_load_generator(g, gen64);
    sha256_stream h; sha256_stream_init(h);
    unsigned char buf[32];
    // the commitment's y is sqrt(x^3+7) (a square) unless the prefix bit negates it; is_square(0) = 1, and y = 0 would need
    // x^3 + 7 = 0 (checked here without the square root; no such x exists on a curve of odd order, but the rule is kept exact)
    fe cx; fe_set_b32_mod(cx, commit33 + 1); fe_normalize(cx);
    fe rhs; ge_curve_rhs(rhs, cx);
    const int cy_zero = fe_normalizes_to_zero(rhs);
    sha256_stream_put(h, (unsigned char)((commit33[0] & 1) && !cy_zero ? 1 : 0));
    fe_get_b32(buf, cx); sha256_stream_write(h, buf, 32);
    fe r; const int gsq = fe_sqrt(r, g.y);
    sha256_stream_put(h, (unsigned char)(!gsq));
    fe_get_b32(buf, g.x); sha256_stream_write(h, buf, 32);
    sha256_stream_write(h, proof, off_hdr);
    for (u32 i = 0; i + 1 < rings; i++) {
        const unsigned char sign = (proof[rec.off_signs + (i >> 3)] & (1u << (i & 7))) != 0;
        sha256_stream_put(h, sign);
        sha256_stream_write(h, proof + rec.off_pts + 32 * i, 32);
    }
    if (extra) sha256_stream_write(h, extra, (size_t)extra_len);
    unsigned char m[32];
    sha256_stream_finalize(h, m);
    for (int i = 0; i < 8; i++) rec.m[i] = s2k_load_be32(m + 4 * i);
}
// C: ring bases: base_0 = -(10^exp) H, base_{i+1} = 4 base_i   (pub_expand, rangeproof_impl.h:20-51)
S2K_HD void rp_pp_bases(const rp_rec& rec, u32* bases /*[32][28]*/, const unsigned char* gen64, u32* dbases = nullptr /*[32][28]*/) {
    if (!(rec.hdr & 1u)) return;
    const u32 rings = rec.rings;
    const int exp = (int)((rec.hdr >> 8) & 0xFFu) - 1;
    ge g; rp_load_generator(g, gen64);
    gej base; ge ng; ng.x = g.x; fe_neg(ng.y, g.y, 1); fe_norm_weak(ng.y);
    gej_set_ge(base, ng);
    for (int e = 0; e < exp; e++) {          // multiplication by 10 = 8x + 2x
        gej t2, t8, s;
        gej_double(t2, base); gej_double(t8, t2); gej_double(s, t8);
        gej_add_var(base, s, t2);
    }
    // One chain: base_{i+1} = 4 base_i, and 2^64 * base_i is simply base_{i+32} -- the steps of the keys' 2^64-multiples (dbases) are
    // the continuation of the same chain, 2 * (rings + 31) doublings in all.
    fe_norm_weak(base.x); fe_norm_weak(base.y);
    const u32 total = dbases ? rings + 32 : rings;
    for (u32 i = 0; i < total; i++) {
        if (i < rings) gej_store28_h(bases + RP_GEJ_WORDS * i, base);
        if (dbases && i >= 32) gej_store28_h(dbases + RP_GEJ_WORDS * (i - 32), base);
        if (i + 1 < total) { gej_double_lean(base, base); gej_double_lean(base, base); }
    }
}
S2K_HD void rp_prologue_points(rp_rec& rec, u32* bases /*[32][28]*/, u64 min_value, const unsigned char* commit33,
                               const unsigned char* proof, const unsigned char* extra, u64 extra_len, const unsigned char* gen64, u32* dbases = nullptr) {
    if (!(rec.hdr & 1u)) return;
    const int commit_ok = rp_pp_commit(rec, min_value, commit33, gen64);
    rp_pp_hash(rec, commit33, proof, extra, extra_len, gen64);
    rp_pp_bases(rec, bases, gen64, dbases);
    rec.ok = (u32)commit_ok;
}

// both halves back to back (host emulation, tests)
S2K_HD void rp_prologue(rp_rec& rec, u32* bases, u64* min_value, u64* max_value, const unsigned char* commit33,
                        const unsigned char* proof, u64 plen, const unsigned char* extra, u64 extra_len, const unsigned char* gen64, u32* dbases = nullptr) {
    rp_header(rec, min_value, max_value, proof, plen);
    rp_prologue_points(rec, bases, *min_value, commit33, proof, extra, extra_len, gen64, dbases);
}

// ---- K1: lift the ring commitments (rangeproof_impl.h:609-626) ------------------------------------------------
S2K_HD void rp_lift(const rp_rec& rec, u32* pub0_ring, unsigned char* lift_ok, const unsigned char* proof, u32 ring) {
    fe x; ge c;
    const unsigned char* px = proof + rec.off_pts + 32 * ring;
    int ok = fe_set_b32_limit(x, px);
    ok &= ge_set_xquad(c, x);
    const int sign = (proof[rec.off_signs + (ring >> 3)] >> (ring & 7)) & 1;
    fe ny; fe_neg(ny, c.y, 1);
    fe_select(c.y, ny, c.y, sign);
    gej j; gej_set_ge(j, c);
    gej_store28_h(pub0_ring, j);
    *lift_ok = (unsigned char)ok;
}

// ---- K2: key of the last ring (rangeproof_impl.h:619-631) ------------------------------------------------------
S2K_HD void rp_sum(rp_rec& rec, u32* pub0 /*[32][28]*/, const unsigned char* lift_ok /*[32]*/) {
    if (!rec.ok) return;
    gej acc; gej_load28_h(acc, rec.accj);
    int ok = 1;
    for (u32 i = 0; i + 1 < rec.rings; i++) {
        ok &= lift_ok[i];
        gej c; gej_load28_h(c, pub0 + RP_GEJ_WORDS * i);
        ge ca; ca.x = c.x; ca.y = c.y;                        // lifted points carry Z = 1
        gej t; const int f = gej_add_ge(t, acc, ca); acc = t;
        if (f == GEJ_ADD_NEEDS_DOUBLE) { gej_double(t, acc); acc = t; }
    }
    // pubs[last] = commit - acc
    fe ny; fe_neg(ny, acc.y, 3); fe_norm_weak(ny); acc.y = ny;
    fe_norm_weak(acc.x);
    ge cm;
    for (int i = 0; i < 9; i++) { cm.x.n[i] = rec.commit[i]; cm.y.n[i] = rec.commit[9 + i]; }
    gej last; const int f = gej_add_ge(last, acc, cm);
    if (f == GEJ_ADD_NEEDS_DOUBLE) { gej t; gej_double(t, last); last = t; }
    if (last.inf) ok = 0;
    else { ge a; ge_set_gej(a, last); gej_set_ge(last, a); }          // every ring key leaves this stage with Z = 1 (rp_ring_shared relies on it)
    gej_store28_h(pub0 + RP_GEJ_WORDS * (rec.rings - 1), last);
    if (!ok) rec.ok = 0;
}

// ---- K3: one ring (borromean_impl.h:70-98) ------------------------------------------------------------------------
S2K_HD void rp_words_to_scalar(scalar& s, int& overflow, const u32 w[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) s.d[i] = w[7 - i];
    overflow = sc_check_overflow(s.d);
    sc_reduce_once(s.d, overflow);
}
// H(e0 || m || ring || 0)   (secp256k1_borromean_hash with elen = 32, :23-37)
S2K_HD void rp_hash_e0(u32 out[8], const u32 e0[8], const u32 m[8], u32 ring) {
    u32 st[8], w[16];
    sha256_init(st);
#pragma unroll
    for (int i = 0; i < 8; i++) { w[i] = e0[i]; w[8 + i] = m[i]; }
    sha256_compress(st, w);
    w[0] = ring; w[1] = 0; w[2] = 0x80000000u;
#pragma unroll
    for (int i = 3; i < 15; i++) w[i] = 0;
    w[15] = 72 * 8;
    sha256_compress(st, w);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = st[i];
}
// H(ser33 || m || ring || epos), ser33 = prefix byte || x (8 big-endian words)
S2K_HD void rp_hash_step(u32 out[8], u32 prefix, const u32 x[8], const u32 m[8], u32 ring, u32 epos) {
    u32 st[8], w[16];
    sha256_init(st);
    w[0] = (prefix << 24) | (x[0] >> 8);
#pragma unroll
    for (int i = 1; i < 8; i++) w[i] = (x[i - 1] << 24) | (x[i] >> 8);
    w[8] = (x[7] << 24) | (m[0] >> 8);
#pragma unroll
    for (int i = 1; i < 8; i++) w[8 + i] = (m[i - 1] << 24) | (m[i] >> 8);
    sha256_compress(st, w);
    w[0] = (m[7] << 24) | (ring >> 8);
    w[1] = (ring << 24) | (epos >> 8);
    w[2] = (epos << 24) | 0x00800000u;
#pragma unroll
    for (int i = 3; i < 15; i++) w[i] = 0;
    w[15] = 73 * 8;
    sha256_compress(st, w);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = st[i];
}

// ev_out (optional): the challenge e of each ring position, 4 x 8 big-endian words per ring -- the reference's `evalues`
// (borromean_impl.h:80-83), which only rewinding needs (rangeproof_rewind.h)
// dbase28 / t28 (optional, both or neither): 2^64 * base and a scratch record for 2^64 * key, which let the double multiplication
// run in its two-piece form (ecmult_lane_split: half the doublings).  2^64 * (first key) costs one 64-doubling chain per ring here.
S2K_HD void rp_ring(const rp_rec& rec, const u32* base28, u32* pub28, unsigned char* ring_out33, unsigned char* ring_ok,
                    const unsigned char* proof, u32 ring, int live, const u32* gtab, const lane_mem& lm, u32* ev_out = nullptr,
                    const u32* dbase28 = nullptr, u32* t28 = nullptr) {
    const u32 rsize = (ring + 1 == rec.rings) ? rec.last_rsize : 4u;
    int ok = live & (int)rec.ok;
    u32 e[8];
    {
        u32 m[8];
#pragma unroll
        for (int i = 0; i < 8; i++) m[i] = rec.m[i];
        u32 e0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) {                                   // never touch proof bytes of a proof that failed its structural checks
            const unsigned char* pe0 = proof + rec.off_e0;
#pragma unroll
            for (int i = 0; i < 8; i++) e0[i] = s2k_load_be32(pe0 + 4 * i);
        }
        rp_hash_e0(e, e0, m, ring);
    }
    // Nothing but flags and pointers stays live across ecmult_lane: the ring key is re-read from its scratch record and the
    // next key (key + base, pub_expand :43-45) is written back there before the multiplication starts.
    u32 outx[8] = {0, 0, 0, 0, 0, 0, 0, 0}; u32 outp = 0;
    S2K_PROF_DECL;
    if (t28) {
        gej t; gej_load28_h(t, pub28);
        const int tinf = t.inf; t.inf = 0;
#pragma unroll 1
        for (int k = 0; k < 64; k++) gej_double_lean(t, t);
        fe_norm_weak(t.y);
        t.inf = tinf;
        if (live) gej_store28_h(t28, t);
    }
    S2K_PROF_MARK(11);
#pragma unroll 1
    for (u32 j = 0; j < 4; j++) {
        const int step_live = ok & (j < rsize);
        scalar ens, s; int ov_e, ov_s = 0;
        if (ev_out && live) { for (int i = 0; i < 8; i++) ev_out[8 * j + i] = e[i]; }
        rp_words_to_scalar(ens, ov_e, e);
        sc_set_zero(s);
        if (step_live) sc_set_b32(s, proof + rec.off_s + 32 * (4 * ring + j), &ov_s);
        gej pub; gej_load28_h(pub, pub28);
        int good = step_live & !ov_e & !ov_s & !sc_is_zero(s) & !sc_is_zero(ens) & !pub.inf;
        if (!good) { sc_set_zero(ens); sc_set_zero(s); }            // dead lanes ride along with empty work
        gej T; if (t28) gej_load28_h(T, t28);
        if (j + 1 < rsize) {
            gej base, nxt; gej_load28_h(base, base28);
            gej_add_var(nxt, pub, base);
            if (live) gej_store28_h(pub28, nxt);
            if (t28) { gej_load28_h(base, dbase28); gej_add_var(nxt, T, base); if (live) gej_store28_h(t28, nxt); }
        }
        gej R;
        S2K_PROF_MARK(0);
        int split_done = 0;
        if (t28) split_done = ecmult_lane_split(R, pub, T, ens, s, 1, gtab, lm);
        if (!S2K_WAVE_ALL(split_done)) ecmult_lane(R, pub, ens, s, 1, gtab, lm);
        S2K_PROF_RESET;
        good &= !R.inf;
        ge a; ge_set_gej(a, R);
        S2K_PROF_MARK(4);
        u32 xw[8]; fe_to_words(xw, a.x);
        u32 xb[8];
#pragma unroll
        for (int i = 0; i < 8; i++) xb[i] = xw[7 - i];
        const u32 prefix = 2u | (u32)fe_is_odd(a.y);
        if (step_live) ok &= good;
        if (j + 1 < rsize) {
            u32 m[8];
#pragma unroll
            for (int i = 0; i < 8; i++) m[i] = rec.m[i];
            rp_hash_step(e, prefix, xb, m, ring, j + 1);
        } else if (j + 1 == rsize) {
#pragma unroll
            for (int i = 0; i < 8; i++) outx[i] = xb[i];
            outp = prefix;
        }
        S2K_PROF_MARK(5);
    }
    if (live) {
        ring_out33[0] = (unsigned char)outp;
        for (int i = 0; i < 8; i++) s2k_store_be32(ring_out33 + 1 + 4 * i, outx[i]);
        *ring_ok = (unsigned char)ok;
    }
}

// ---- the generator-table cache as the kernels see it ---------------------------------------------------------------------------
#define RP_GEN_SLOTS 8
#define RP_GEN_MBOX 4
struct rp_gen_dev {                       // by value to the kernels: the cache at the time of the launch
    const u32* tab[RP_GEN_SLOTS];         // fixed-base table (layout of gtab), nullptr for an empty slot
    const u32* xmul[RP_GEN_SLOTS];        // RP_XMUL_WORDS
    const unsigned char* keys;            // [RP_GEN_SLOTS][64] generator bytes
    u32 valid;                            // bit i: slot i holds a table
    u32 any;                              // index of some valid slot (idle lanes read its x-table)
};
// Generators that had no table, reported by the LAST stage (k_rp_final) for proofs that VERIFIED only -- an attacker cannot make the
// engine spend 0.6 s and 21.5 GB on a table by sending junk proofs that merely name a generator (the first few distinct generators of
// a call, with the number of valid proofs that carried them).  A slot is claimed by a 64-bit tag of the generator bytes (one
// compare-and-swap; lanes with the same generator then only count), so nothing on the device ever waits for another lane's key bytes
// -- the host reads those after the call and ignores a slot whose bytes do not hash to its tag.  Two generators with the same tag
// would share a count: harmless, the count only decides whether a table is worth building.
// hits[s]: valid proofs served by cached slot s in this call (the host refreshes the slot's least-recently-used stamp from it).
struct rp_gen_mbox { unsigned long long tag[RP_GEN_MBOX]; u32 count[RP_GEN_MBOX]; u32 hits[RP_GEN_SLOTS]; unsigned char key[RP_GEN_MBOX][64]; };
S2K_HD unsigned long long rp_gen_tag(const unsigned char* gen64) {
    unsigned long long h = 0xCBF29CE484222325ull;
    for (int k = 0; k < 64; k++) { h ^= gen64[k]; h *= 0x100000001B3ull; }
    return h ? h : 1ull;
}
S2K_HD u32 rp_gen_lookup(const rp_gen_dev& gc, const unsigned char* gen64) {
    u32 slot = RP_GSLOT_NONE;
    for (u32 i = 0; i < RP_GEN_SLOTS; i++) {
        if (!((gc.valid >> i) & 1u)) continue;
        int same = 1;
        for (int k = 0; k < 64; k++) same &= (gc.keys[64 * i + k] == gen64[k]);
        if (same) slot = i;
    }
    return slot;
}
#if defined(__HIPCC__) || defined(__HIP__)
// called by every lane of a (64-lane) wavefront; `report`: this lane holds a valid proof whose generator has no table
__device__ __forceinline__ void rp_gen_report_miss(rp_gen_mbox* mb, const unsigned char* gen64, int report) {
    const unsigned long long mask = __ballot(report);
    if (!mask) return;
    // a batch in which every proof has its own generator makes every lane come here: a slot that is taken by another tag is skipped
    // on a plain load (no atomic); a wavefront whose reporting lanes all name the same generator (the common case) counts once
    const int leader = __ffsll((long long)mask) - 1;
    const unsigned long long h = report ? rp_gen_tag(gen64) : 0ull;
    const unsigned long long h0 = __shfl(h, leader);
    const int uniform = __all(!report || h == h0);
    u32 weight = 1u;
    if (uniform) { if ((int)(threadIdx.x & 63u) != leader) return; weight = (u32)__popcll(mask); }
    else if (!report) return;
    for (int m = 0; m < RP_GEN_MBOX; m++) {
        unsigned long long old = *(volatile unsigned long long*)&mb->tag[m];
        if (old != 0ull && old != h) continue;
        if (old == 0ull) {
            old = atomicCAS(&mb->tag[m], 0ull, h);
            if (old == 0ull) { for (int k = 0; k < 64; k++) mb->key[m][k] = gen64[k]; }
            else if (old != h) continue;
        }
        atomicAdd(&mb->count[m], weight);
        return;
    }
}
#endif

// ---- K3, shared-generator form --------------------------------------------------------------------------------------
// The four keys of a ring are P_j = C + j*B with B = -(4^ring 10^exp)*H (pub_expand :19-51), so every step is
//     e_j*P_j + s_j*G = e_j*C + s_j*G + f_j*H ,   f_j = -(j 4^ring 10^exp) e_j mod n ,
// one variable point for the whole ring: its tables and the 2^64 chain are built once (ecmult.h, "the ring form") and f_j*H comes from a
// fixed-base table of the proof's generator (`htab`, same layout as the table of G; the engine caches one per generator).
// Bit-exactness of the accept/reject decision needs two things the reference does with the keys themselves:
//   * it rejects a key that is the point at infinity (borromean_impl.h:78): P_j = inf <=> C = -j*B, impossible for an honest prover but
//     a choice an adversarial one has.  `xmul` holds the affine x of j*(4^ring 10^exp)*H for j = 1..3 (a per-generator table, see
//     RP_XMUL_*): a ring whose C has one of these x coordinates is `suspect` and the caller takes the whole wavefront through rp_ring;
//   * exceptional additions inside a step (an operand with the accumulator's own x): the ring is handed back as well.
// Idle lanes (a proof that failed earlier, a ring beyond the proof's count, a step beyond the ring's size, a zero or overflowing
// scalar) ride along on a dummy point / dummy scalars so that the wavefront stays in lock step; their results are discarded.
#define RP_XMUL_EXPS 19
#define RP_XMUL_WORDS (RP_XMUL_EXPS * RP_MAX_RINGS * 3 * 8)       /* [exp][ring][j-1][8 canonical words, least significant first] */
S2K_HD void rp_ring_const(scalar& c, int exp, u32 ring) {         // 4^ring * 10^exp  (< 2^124)
    u64 scale = 1;
    for (int i = 0; i < exp; i++) scale *= 10;
    const u32 sh = 2 * ring;
    const u64 lo = scale << sh, hi = sh ? (scale >> (64 - sh)) : 0;
    sc_set_zero(c);
    c.d[0] = (u32)lo; c.d[1] = (u32)(lo >> 32); c.d[2] = (u32)hi; c.d[3] = (u32)(hi >> 32);
}
// 1 when C (affine, Z = 1 record) shares its x with one of B, 2B, 3B
S2K_HD int rp_ring_suspect(const gej& C, const u32* xmul /* this (exp, ring)'s 3 x 8 words */) {
    int hit = 0;
    for (int j = 0; j < 3; j++) {
        u32 w[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = xmul[8 * j + i];
        fe x; fe_from_words(x, w);
        hit |= fe_equal(x, C.x);
    }
    return hit;
}
// K rings per lane (consecutive rings ring0 .. ring0+K-1 of one proof).  Why several: every ring position ends in a field inversion
// (to-affine before the point is hashed), ~19 000 instructions of the wavefront whatever its lanes hold, and the K points of a lane
// share ONE inversion (Montgomery's trick, 3 (K - 1) products).  The K rings of a lane advance position by position: for each position
// j the K multiplications run one after the other (one copy of the code, `q` loop), each result is parked (27 words), then one
// inversion, then K x (affine, serialise, hash).  What a ring carries from one position to the next -- the challenge e and its ok
// flag -- is parked next to the point.
// Memory of a lane (rp_shared_mem): rtab = K x S2K_RTAB_WORDS of its own (finished tables); raw = its column of the wavefront's
// table-construction parking area; park = its column of the wavefront's K x RP_PARK_WORDS area (word stride S2K_RAW_WS like raw).
// Returns RP_SHARED_SERVED, or -- the wavefront then has to take rp_ring instead, nothing this call wrote is final -- RP_SHARED_SUSPECT (a ring
// key that may be infinity) / RP_SHARED_EXCEPTIONAL (an exceptional addition inside a step).
// pub28: the key records of the proof's rings from ring0 on (Z = 1: rp_lift; rp_sum brings the last key to affine).
#ifndef S2K_RP_K
#define S2K_RP_K 1                      /* rings per lane in the engine's shared-generator kernel; K > 1 shares the inversion of a ring position between K
                                           rings but keeps K times the tables alive: measured on MI355X 15.5 (K=1) / 15.75 (2) / 15.85 ms (4) per 2^14 proofs */
#endif
#define RP_SHARED_SUSPECT 0
#define RP_SHARED_SERVED 1
#define RP_SHARED_EXCEPTIONAL 2
#define RP_PARK_WORDS 40                /* 0..26 point (x, y, z or 1/z), 27..34 challenge e, 35 ok, 36 good */
struct rp_shared_mem { u32* rtab; u32* raw; u32* park; s2k_lds_ptr dig; };
template <int K>
S2K_HD int rp_rings_shared(const rp_rec& rec, const u32* pub28, unsigned char* ring_out /* the proof's */, unsigned char* ring_ok /* the proof's [32] */,
                           const unsigned char* proof, u32 ring0, int live, const u32* gtab, const u32* htab, const u32* xmul, const rp_shared_mem& M,
                           u32* ev_out = nullptr /* the proof's [32][32] */, u32 dbg = 0) {
    const int pok = live & (int)rec.ok;
    const int exp = pok ? (int)((rec.hdr >> 8) & 0xFFu) - 1 : 0;
    const int e_idx = exp < 0 ? 0 : exp;
    const u32 nrings = pok ? rec.rings : 0u;
    auto pk = [&](int q, int k) -> u32& { return M.park[(size_t)(q * RP_PARK_WORDS + k) * S2K_RAW_WS]; };
    // ---- a ring whose key collides with a multiple of its base goes back to the caller (with its whole wavefront) before any work is done
    {
        int suspect = 0;
#pragma unroll 1
        for (int q = 0; q < K; q++) {
            const u32 ring = ring0 + q;
            gej C; gej_load28_h(C, pub28 + RP_GEJ_WORDS * q);
            const int ok = (ring < nrings) & !C.inf;
            if (ok) suspect |= rp_ring_suspect(C, xmul + ((size_t)e_idx * RP_MAX_RINGS + ring) * 24);
        }
        if (S2K_WAVE_ANY(suspect)) return RP_SHARED_SUSPECT;
    }
    S2K_PROF_DECL;
    // ---- per ring: key (a dummy for an idle ring), first challenge, 2^64 chain, both tables
#pragma unroll 1
    for (int q = 0; q < K; q++) {
        const u32 ring = ring0 + q;
        gej C; gej_load28_h(C, pub28 + RP_GEJ_WORDS * q);
        int ok = (ring < nrings) & !C.inf;
        if (!ok) { ge g; ge_set_generator(g); gej_set_ge(C, g); }
        u32 e[8];
        {
            u32 m[8];
#pragma unroll
            for (int i = 0; i < 8; i++) m[i] = rec.m[i];
            u32 e0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) {                                   // never touch proof bytes of a proof that failed its structural checks
                const unsigned char* pe0 = proof + rec.off_e0;
#pragma unroll
                for (int i = 0; i < 8; i++) e0[i] = s2k_load_be32(pe0 + 4 * i);
            }
            rp_hash_e0(e, e0, m, ring);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) pk(q, 27 + i) = e[i];
        pk(q, 35) = (u32)ok;
        S2K_PROF_RESET;
        gej T = C;
        // dbg (diagnostic launches only, $S2K_RP_DEBUG; results are then meaningless): bit 0 = no chain, bit 1 = no tables, bit 2 = no steps
#pragma unroll 1
        for (int k = 0; k < ((dbg & 1u) ? 0 : 64); k++) gej_double_lean(T, T);
        fe_norm_weak(T.y);
        S2K_PROF_MARK(11);
        if (!(dbg & 2u)) ecmult_ring_tables(M.rtab + (size_t)q * S2K_RTAB_WORDS, M.raw, C, T);
        S2K_PROF_MARK(8);
    }
    if (dbg & 4u) return RP_SHARED_SERVED;
    // dummy scalars of idle rings / idle positions (any fixed nonzero values)
    const scalar dummy_e = {{0x9E3779B9u, 0x7F4A7C15u, 0xF39CC060u, 0x5CEDC834u, 0x1082276Bu, 0xF3A27251u, 0xF86C6A11u, 0x0D5A2B4Fu}};
    const scalar dummy_s = {{0x2545F491u, 0x4F6CDD1Du, 0x6C078965u, 0x5851F42Du, 0x14057B7Eu, 0xF767814Fu, 0x9FB21C65u, 0x1E35A7BDu}};
#pragma unroll 1
    for (u32 j = 0; j < 4; j++) {
        // ---- the K multiplications of this position
#pragma unroll 1
        for (int q = 0; q < K; q++) {
            const u32 ring = ring0 + q;
            const u32 rsize = (ring + 1 == nrings) ? rec.last_rsize : 4u;
            u32 e[8];
#pragma unroll
            for (int i = 0; i < 8; i++) e[i] = pk(q, 27 + i);
            const int ok = (int)pk(q, 35);
            const int step_live = ok & (j < rsize);
            scalar ens, s; int ov_e, ov_s = 0;
            if (ev_out && ok) { for (int i = 0; i < 8; i++) ev_out[(size_t)ring * 32 + 8 * j + i] = e[i]; }
            rp_words_to_scalar(ens, ov_e, e);
            sc_set_zero(s);
            if (step_live) sc_set_b32(s, proof + rec.off_s + 32 * (4 * ring + j), &ov_s);
            const int good = step_live & !ov_e & !ov_s & !sc_is_zero(s) & !sc_is_zero(ens);
            if (!good) { ens = dummy_e; s = dummy_s; }
            scalar f; sc_set_zero(f);
            if (j > 0) {
                scalar cring; rp_ring_const(cring, e_idx, ring);
                scalar k = cring;
                for (u32 t = 1; t < j; t++) sc_add(k, k, cring);
                sc_mul(f, ens, k); sc_negate(f, f);
            }
            gej R;
            S2K_PROF_MARK(0);
            // an exceptional addition somewhere in the wavefront (an operand with the accumulator's own x: adversarial inputs only): everything
            // goes back to the caller, i.e. to the general form, which starts the rings over on the keys themselves
            if (!S2K_WAVE_ALL(ecmult_ring_step(R, M.rtab + (size_t)q * S2K_RTAB_WORDS, ens, s, f, j > 0, gtab, htab, M.dig))) return RP_SHARED_EXCEPTIONAL;
            S2K_PROF_RESET;
            fe_norm_weak(R.x); fe_norm_weak(R.y);
#pragma unroll
            for (int i = 0; i < 9; i++) { pk(q, i) = R.x.n[i]; pk(q, 9 + i) = R.y.n[i]; pk(q, 18 + i) = R.z.n[i]; }
            pk(q, 36) = (u32)good;
        }
        // ---- one inversion for the K points (every R is finite here: all additions of the step had operands with different x)
        {
            fe pre[K], z;
#pragma unroll
            for (int q = 0; q < K; q++) {
#pragma unroll
                for (int i = 0; i < 9; i++) z.n[i] = pk(q, 18 + i);
                if (q == 0) pre[0] = z; else fe_mul(pre[q], pre[q - 1], z);
            }
            fe inv; fe_inv(inv, pre[K - 1]);
#pragma unroll
            for (int q = K - 1; q >= 1; q--) {
#pragma unroll
                for (int i = 0; i < 9; i++) z.n[i] = pk(q, 18 + i);
                fe zi; fe_mul2(zi, inv, pre[q - 1], inv, inv, z);          // 1/z_q ; inv <- 1/(z_0 .. z_{q-1})
#pragma unroll
                for (int i = 0; i < 9; i++) pk(q, 18 + i) = zi.n[i];
            }
#pragma unroll
            for (int i = 0; i < 9; i++) pk(0, 18 + i) = inv.n[i];
        }
        S2K_PROF_MARK(4);
        // ---- K x (affine, serialise, next challenge / ring output)
#pragma unroll 1
        for (int q = 0; q < K; q++) {
            const u32 ring = ring0 + q;
            const u32 rsize = (ring + 1 == nrings) ? rec.last_rsize : 4u;
            ge a;
            {
                fe x, y, zi, zi2, zi3;
#pragma unroll
                for (int i = 0; i < 9; i++) { x.n[i] = pk(q, i); y.n[i] = pk(q, 9 + i); zi.n[i] = pk(q, 18 + i); }
                fe_sqr(zi2, zi); fe_mul(zi3, zi2, zi);
                fe_mul2(a.x, x, zi2, a.y, y, zi3);
                fe_normalize(a.x); fe_normalize(a.y);
            }
            int ok = (int)pk(q, 35);
            const int good = (int)pk(q, 36);
            const int step_live = ok & (j < rsize);
            u32 xw[8]; fe_to_words(xw, a.x);
            u32 xb[8];
#pragma unroll
            for (int i = 0; i < 8; i++) xb[i] = xw[7 - i];
            const u32 prefix = 2u | (u32)fe_is_odd(a.y);
            if (step_live) ok &= good;
            pk(q, 35) = (u32)ok;
            if (j + 1 < rsize) {
                u32 m[8], e[8];
#pragma unroll
                for (int i = 0; i < 8; i++) m[i] = rec.m[i];
                rp_hash_step(e, prefix, xb, m, ring, j + 1);
#pragma unroll
                for (int i = 0; i < 8; i++) pk(q, 27 + i) = e[i];
            } else if (j + 1 == rsize && ring < nrings) {
                unsigned char* o = ring_out + 33 * ring;
                o[0] = (unsigned char)prefix;
                for (int i = 0; i < 8; i++) s2k_store_be32(o + 1 + 4 * i, xb[i]);
            }
            if (j == 3 && ring < nrings) ring_ok[ring] = (unsigned char)ok;
        }
        S2K_PROF_MARK(5);
    }
    return RP_SHARED_SERVED;
}

// ---- K4: close the loop (borromean_impl.h:100-103) ---------------------------------------------------------------------
S2K_HD int rp_final(const rp_rec& rec, unsigned char* ring_out /*[RP_RING_OUT_BYTES], 4-byte aligned*/, const unsigned char* ring_ok, const unsigned char* proof) {
    if (!rec.ok) return 0;
    int ok = 1;
    for (u32 i = 0; i < rec.rings; i++) ok &= ring_ok[i];
    // e0' = SHA256( r_0 | ... | r_{rings-1} | m ): the ring outputs already lie back to back, m is appended and the buffer is
    // hashed a block (16 aligned words) at a time
    const u32 len = 33 * rec.rings + 32;
    for (int i = 0; i < 8; i++) s2k_store_be32(ring_out + 33 * rec.rings + 4 * i, rec.m[i]);
    u32 st[8]; sha256_init(st);
    u32 off = 0;
    for (; off + 64 <= len; off += 64) {
        u32 w[16];
        for (int i = 0; i < 16; i++) w[i] = s2k_load_be32(ring_out + off + 4 * i);
        sha256_compress(st, w);
    }
    {
        u32 w[16];
        for (int i = 0; i < 16; i++) w[i] = 0;
        const u32 rem = len - off;                                   // < 64
        for (u32 k = 0; k < rem; k++) w[k >> 2] |= (u32)ring_out[off + k] << (24 - 8 * (k & 3));
        w[rem >> 2] |= 0x80u << (24 - 8 * (rem & 3));
        if (rem >= 56) { sha256_compress(st, w); for (int i = 0; i < 16; i++) w[i] = 0; }
        w[15] = len * 8;
        sha256_compress(st, w);
    }
    unsigned char d[32];
    for (int i = 0; i < 8; i++) s2k_store_be32(d + 4 * i, st[i]);
    int diff = 0;
    for (int i = 0; i < 32; i++) diff |= d[i] ^ proof[rec.off_e0 + i];
    return ok & (diff == 0);
}
