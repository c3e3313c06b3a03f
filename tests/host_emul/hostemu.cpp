// tests/host_emul/hostemu.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the per-lane device arithmetic headers (secp256k1_zkp_amd/csrc/*.h) for the *host* with S2K_VERIFY
// on, and exposes byte-level entry points mirroring oracle/ref_shim.c, so that the GPU-less CI container can
// check every field/scalar/group/ecmult primitive against the reference before any GPU time is spent.
// This library is never loaded by the product; the shipped path is HIP only.
#define S2K_VERIFY 1
static unsigned long long g_split_done = 0;
#define S2K_ON_SPLIT_DONE() (g_split_done++)
static unsigned long long g_ring_steps_done = 0;
#define S2K_ON_RING_STEP_DONE() (g_ring_steps_done++)
#include "../../secp256k1_zkp_amd/csrc/gtable.h"
#include "../../secp256k1_zkp_amd/csrc/sha256.h"
#include "../../secp256k1_zkp_amd/csrc/rangeproof.h"
#include "../../secp256k1_zkp_amd/csrc/rangeproof_rewind.h"
#include "../../secp256k1_zkp_amd/csrc/schnorr.h"
#include "../../secp256k1_zkp_amd/csrc/msm.h"
#include "../../secp256k1_zkp_amd/csrc/bppp.h"
#include "../../secp256k1_zkp_amd/csrc/surjection.h"
#include "../../secp256k1_zkp_amd/csrc/halfagg.h"
#include <string.h>
#include <vector>

static void fe_from_b32(fe& r, const unsigned char* b) { fe_set_b32_mod(r, b); }
static void fe_to_b32(unsigned char* b, const fe& a) { fe t = a; fe_normalize(t); fe_get_b32(b, t); }

extern "C" {
void emu_fe_mul(unsigned char* r, const unsigned char* a, const unsigned char* b) { fe x, y, z; fe_from_b32(x, a); fe_from_b32(y, b); fe_mul(z, x, y); fe_to_b32(r, z); }
void emu_fe_sqr(unsigned char* r, const unsigned char* a) { fe x, z; fe_from_b32(x, a); fe_sqr(z, x); fe_to_b32(r, z); }
void emu_fe_add(unsigned char* r, const unsigned char* a, const unsigned char* b) { fe x, y; fe_from_b32(x, a); fe_from_b32(y, b); fe_add(x, y); fe_to_b32(r, x); }
void emu_fe_negate(unsigned char* r, const unsigned char* a) { fe x, y; fe_from_b32(x, a); fe_neg(y, x, 1); fe_to_b32(r, y); }
void emu_fe_inv_fermat(unsigned char* r, const unsigned char* a) { fe x, z; fe_from_b32(x, a); fe_inv_fermat(z, x); fe_to_b32(r, z); }
void emu_fe_inv(unsigned char* r, const unsigned char* a) { fe x, z; fe_from_b32(x, a); fe_inv(z, x); fe_to_b32(r, z); }
int emu_fe_sqrt(unsigned char* r, const unsigned char* a) { fe x, z; fe_from_b32(x, a); int ok = fe_sqrt(z, x); fe_to_b32(r, z); return ok; }
void emu_fe_half(unsigned char* r, const unsigned char* a) { fe x; fe_from_b32(x, a); fe_half(x); fe_to_b32(r, x); }
int emu_fe_set_b32_limit(const unsigned char* a) { fe x; return fe_set_b32_limit(x, a); }
// stress: a lazily accumulated expression  ((a+b)*3 - c) * (a - b)  exercising magnitudes
void emu_fe_lazy(unsigned char* r, const unsigned char* a, const unsigned char* b, const unsigned char* c) {
    fe x, y, z, t, u; fe_from_b32(x, a); fe_from_b32(y, b); fe_from_b32(z, c);
    fe_add2(t, x, y); fe_mul_int(t, 3); fe_neg(u, z, 1); fe_add(t, u);      // mag 8? no: (1+1)*3=6, +2 = 8 -> too big; renormalise
    fe_norm_weak(t);
    fe_neg(u, y, 1); fe_add(u, x);                                          // mag 3
    fe_mul(t, t, u); fe_to_b32(r, t);
}

int emu_scalar_set_b32(unsigned char* r, const unsigned char* a) { scalar s; int o; sc_set_b32(s, a, &o); sc_get_b32(r, s); return o; }
void emu_scalar_mul(unsigned char* r, const unsigned char* a, const unsigned char* b) { scalar x, y; sc_set_b32(x, a, 0); sc_set_b32(y, b, 0); sc_mul(x, x, y); sc_get_b32(r, x); }
void emu_scalar_add(unsigned char* r, const unsigned char* a, const unsigned char* b) { scalar x, y; sc_set_b32(x, a, 0); sc_set_b32(y, b, 0); sc_add(x, x, y); sc_get_b32(r, x); }
void emu_scalar_negate(unsigned char* r, const unsigned char* a) { scalar x; sc_set_b32(x, a, 0); sc_negate(x, x); sc_get_b32(r, x); }
void emu_scalar_inverse(unsigned char* r, const unsigned char* a) { scalar x; sc_set_b32(x, a, 0); sc_inverse(x, x); sc_get_b32(r, x); }
void emu_scalar_split_lambda(unsigned char* r1, unsigned char* r2, const unsigned char* k) { scalar a, b, x; sc_set_b32(x, k, 0); sc_split_lambda(a, b, x); sc_get_b32(r1, a); sc_get_b32(r2, b); }

static void ge_from_b64(ge& g, const unsigned char* b) { fe_from_b32(g.x, b); fe_from_b32(g.y, b + 32); }
static int gej_to_b64(unsigned char* b, const gej& j) {
    if (j.inf) { memset(b, 0, 64); return 1; }
    ge a; ge_set_gej(a, j); fe_get_b32(b, a.x); fe_get_b32(b + 32, a.y); return 0;
}
int emu_ge_add(unsigned char* r64, const unsigned char* a64, int ainf, const unsigned char* b64, int binf) {
    ge a, b; gej j, t; ge_from_b64(a, a64); ge_from_b64(b, b64);
    if (ainf) gej_set_infinity(j); else gej_set_ge(j, a);
    if (binf) return gej_to_b64(r64, j);
    int f = gej_add_ge(t, j, b);
    if (f == GEJ_ADD_NEEDS_DOUBLE) { gej u; gej_double(u, t); t = u; }
    return gej_to_b64(r64, t);
}
int emu_gej_add_var(unsigned char* r64, const unsigned char* a64, int ainf, const unsigned char* b64, int binf, const unsigned char* za32, const unsigned char* zb32) {
    // a, b given affine; rescaled by za, zb to exercise the Jacobian+Jacobian path
    ge a, b; gej ja, jb, r; fe za, zb, z2, z3; ge_from_b64(a, a64); ge_from_b64(b, b64);
    fe_from_b32(za, za32); fe_from_b32(zb, zb32);
    gej_set_ge(ja, a); gej_set_ge(jb, b);
    fe_sqr(z2, za); fe_mul(z3, z2, za); fe_mul(ja.x, ja.x, z2); fe_mul(ja.y, ja.y, z3); ja.z = za; fe_norm_weak(ja.z);
    fe_sqr(z2, zb); fe_mul(z3, z2, zb); fe_mul(jb.x, jb.x, z2); fe_mul(jb.y, jb.y, z3); jb.z = zb; fe_norm_weak(jb.z);
    if (ainf) gej_set_infinity(ja);
    if (binf) gej_set_infinity(jb);
    gej_add_var(r, ja, jb);
    return gej_to_b64(r64, r);
}
int emu_ge_double(unsigned char* r64, const unsigned char* a64, int ainf) {
    ge a; gej j, t; ge_from_b64(a, a64);
    if (ainf) gej_set_infinity(j); else gej_set_ge(j, a);
    gej_double(t, j);
    return gej_to_b64(r64, t);
}
int emu_ge_set_xquad(unsigned char* r64, const unsigned char* x32) {
    fe x; ge g; fe_from_b32(x, x32); int ok = ge_set_xquad(g, x); fe_to_b32(r64, g.x); fe_to_b32(r64 + 32, g.y); return ok;
}

static std::vector<u32> g_gtab;
static u32 g_ptab[S2K_PTAB_WORDS];          // (two tables' worth: the split form of the double multiplication uses both)
static u32 g_dig[S2K_DIG_WORDS];
static const lane_mem g_lm{g_ptab, g_dig};
// Host-only construction of the window table (this library is compiled with -DS2K_GTAB_BITS=12 to keep it small): same entries as gtable.h's device kernels, but built by running
// sums + Montgomery batch inversion so that a CPU test does not spend a minute on a million inversions.
#ifndef S2K_EMU_GTAB_BITS
#define S2K_EMU_GTAB_BITS 12
#endif
static void table_host(std::vector<u32>& g_gtab, const ge* point) {
    {
        const u32 D = S2K_EMU_GTAB_BITS, W = gtab_windows_for(D), HALF = 1u << (D - 1u);
        g_gtab.assign(gtab_words_for(D), 0);
        gtab_write_header(g_gtab.data(), D);
        const u32 NV = HALF + 1u; std::vector<gej> acc(NV); std::vector<fe> pre(NV);      // magnitudes 1 .. 2^(D-1) of a signed digit
        for (u32 w = 0; w < W; w++) {
            gtab_build_base(g_gtab.data(), D, w, point);
            ge base; gtab_load(base, g_gtab.data(), w, 1);
            gej_set_ge(acc[1], base);
            for (u32 v = 2; v < NV; v++) {
                gej t; int f = gej_add_ge(t, acc[v - 1], base);
                if (f == GEJ_ADD_NEEDS_DOUBLE) { gej u; gej_double(u, t); t = u; }
                fe_norm_weak(t.y); acc[v] = t;
            }
            fe run; fe_set_int(run, 1);
            for (u32 v = 1; v < NV; v++) { pre[v] = run; fe_mul(run, run, acc[v].z); }
            fe inv; fe_inv(inv, run);
            for (u32 v = NV - 1; v >= 2; v--) {
                fe zi, zi2, zi3; fe_mul(zi, inv, pre[v]); fe_mul(inv, inv, acc[v].z);
                fe_sqr(zi2, zi); fe_mul(zi3, zi2, zi);
                ge a; fe_mul(a.x, acc[v].x, zi2); fe_mul(a.y, acc[v].y, zi3); fe_normalize(a.x); fe_normalize(a.y);
                gtab_store(g_gtab.data(), D, w, v, a);
            }
        }
    }
}
// the device's table construction (gtable.h: window bases, seeds, one affine addition per remaining entry with a shared inversion per run
// of rows) run sequentially for a table of width D, against the running-sum construction above: returns the number of entries that differ
// (-1: an entry the device construction leaves out is one a digit can address)
int emu_gtab_seeded_construction(unsigned D, const unsigned char* point64) {
    if (!gtab_bits_ok(D) || D > 16) return -2;
    ge pt; const ge* point = nullptr;
    if (point64) { ge_from_b64(pt, point64); fe_norm_weak(pt.x); fe_norm_weak(pt.y); point = &pt; }
    const gtab_fill_plan p = gtab_make_fill_plan(D);
    std::vector<u32> tab(gtab_words_for(D), 0xFFFFFFFFu);
    gtab_write_header(tab.data(), D);
    for (u32 w = 0; w < p.W; w++) gtab_build_base(tab.data(), D, w, point);
    for (u32 w = 0; w < p.W; w++) for (u32 t = 0; t < gtab_seeds_per_window(p); t++) gtab_build_seed(tab.data(), p, w, t);
    const u32 runs = gtab_fill_runs(p);
    for (u32 w = 0; w < p.W; w++) for (u32 run = 0; run < runs; run++) for (u32 b = 1; b <= gtab_fill_cols(p); b++) gtab_fill_run(tab.data(), p, w, b, 1u + run * GTAB_FILL_RUN);
    // reference: v * base by repeated addition
    int bad = 0;
    for (u32 w = 0; w < p.W; w++) {
        ge base; gtab_load(base, tab.data(), w, 1);
        gej acc; gej_set_ge(acc, base);
        const u32 vmax = (w + 1 < p.W) ? (1u << (D - 1)) : (1u << gtab_top_bits_for(D)) + 1u;
        for (u32 v = 1; v <= vmax; v++) {
            if (v > 1) { gej t; int f = gej_add_ge(t, acc, base); if (f == GEJ_ADD_NEEDS_DOUBLE) { gej u; gej_double(u, t); t = u; } acc = t; }
            ge a; ge_set_gej(a, acc);
            u32 wx[8], wy[8]; fe_to_words(wx, a.x); fe_to_words(wy, a.y);
            const u32* q = tab.data() + gtab_slot(D, w, v) * S2K_GTAB_ENTRY_WORDS;
            int same = 1, unset = 1;
            for (int i = 0; i < 8; i++) { same &= (q[i] == wx[i]) & (q[8 + i] == wy[i]); unset &= (q[i] == 0xFFFFFFFFu) & (q[8 + i] == 0xFFFFFFFFu); }
            if (unset) return -1;
            bad += !same;
        }
    }
    return bad;
}
static const u32* gtab_host() {
    if (g_gtab.empty()) table_host(g_gtab, nullptr);
    return g_gtab.data();
}
// spot-check entry (w, v) of the host table against the device construction path
int emu_gtab_entry(unsigned char* r64, unsigned w, unsigned v) {
    ge g; gtab_load(g, gtab_host(), w, v); fe_normalize(g.x); fe_normalize(g.y); fe_get_b32(r64, g.x); fe_get_b32(r64 + 32, g.y); return 0;
}
// z32 != NULL: present A in Jacobian form with that Z
int emu_ecmult(unsigned char* r64, const unsigned char* a64, int ainf, const unsigned char* na32, const unsigned char* ng32, const unsigned char* z32) {
    ge a; gej A, R; scalar na, ng; ge_from_b64(a, a64);
    if (ainf) gej_set_infinity(A); else gej_set_ge(A, a);
    if (z32 && !ainf) { fe z, z2, z3; fe_from_b32(z, z32); fe_norm_weak(z); fe_sqr(z2, z); fe_mul(z3, z2, z); fe_mul(A.x, A.x, z2); fe_mul(A.y, A.y, z3); A.z = z; }
    sc_set_b32(na, na32, 0);
    if (ng32) sc_set_b32(ng, ng32, 0); else sc_set_zero(ng);
    u32 dig[S2K_DIG_WORDS]; const lane_mem lm{g_ptab, dig};
    ecmult_lane(R, A, na, ng, ng32 != 0, gtab_host(), lm);
    return gej_to_b64(r64, R);
}
// the two-piece form (ecmult_lane_split) given A and T = 2^64 A, falling back to ecmult_lane exactly as rangeproof.h does; *took_split = 1
// when the two-piece form produced the result
int emu_ecmult_split(unsigned char* r64, int* took_split, const unsigned char* a64, const unsigned char* na32, const unsigned char* ng32, const unsigned char* z32) {
    ge a; gej A, T, R; scalar na, ng; ge_from_b64(a, a64);
    gej_set_ge(A, a);
    if (z32) { fe z, z2, z3; fe_from_b32(z, z32); fe_norm_weak(z); fe_sqr(z2, z); fe_mul(z3, z2, z); fe_mul(A.x, A.x, z2); fe_mul(A.y, A.y, z3); A.z = z; }
    T = A; for (int k = 0; k < 64; k++) { gej t; gej_double(t, T); T = t; }
    sc_set_b32(na, na32, 0);
    if (ng32) sc_set_b32(ng, ng32, 0); else sc_set_zero(ng);
    u32 dig[S2K_DIG_WORDS]; const lane_mem lm{g_ptab, dig};
    const int done = ecmult_lane_split(R, A, T, na, ng, ng32 != 0, gtab_host(), lm);
    if (!done) ecmult_lane(R, A, na, ng, ng32 != 0, gtab_host(), lm);
    *took_split = done;
    return gej_to_b64(r64, R);
}
void emu_sha256(unsigned char* out32, const unsigned char* msg, size_t len) {
    sha256_stream c; sha256_stream_init(c); sha256_stream_write(c, msg, len); sha256_stream_finalize(c, out32);
}

unsigned long long emu_split_count(void) { return g_split_done; }

// the five rangeproof stages of rangeproof.h run back to back for one proof
int emu_rangeproof_verify(unsigned long long* min_value, unsigned long long* max_value, const unsigned char* commit33, const unsigned char* proof, size_t plen,
                          const unsigned char* extra, size_t extra_len, const unsigned char* gen64) {
    rp_rec rec; std::vector<u32> bases(32 * 28, 0), pub0(32 * 28, 0), dbases(32 * 28, 0), tcur(32 * 28, 0);
    unsigned char lift_ok[32] = {0}, ring_out[RP_RING_OUT_BYTES] = {0}, ring_ok[32] = {0};
    u64 mn, mx;
    rp_prologue(rec, bases.data(), &mn, &mx, commit33, proof, plen, extra_len ? extra : nullptr, extra_len, gen64, dbases.data());
    *min_value = mn; *max_value = mx;
    if (rec.ok) for (u32 i = 0; i + 1 < rec.rings; i++) rp_lift(rec, pub0.data() + 28 * i, lift_ok + i, proof, i);
    rp_sum(rec, pub0.data(), lift_ok);
    for (u32 i = 0; i < 32; i++) rp_ring(rec, bases.data() + 28 * i, pub0.data() + 28 * i, ring_out + 33 * i, ring_ok + i, proof, i, i < rec.rings, gtab_host(), g_lm, nullptr,
                                         dbases.data() + 28 * i, tcur.data() + 28 * i);
    return rp_final(rec, ring_out, ring_ok, proof);
}

// the same with stage K3 in its shared-generator form (rp_ring_shared): a fixed-base table and the j*B x-table of the proof's generator
// are built here the way the engine's cache builds them; *fast_rings counts the rings that completed in that form
static std::vector<u32> g_htab, g_xmul; static unsigned char g_hkey[64]; static int g_hkey_valid = 0;
static void htab_host(const unsigned char* gen64) {
    if (g_hkey_valid && !memcmp(g_hkey, gen64, 64)) return;
    ge g; rp_load_generator(g, gen64);
    table_host(g_htab, &g);
    g_xmul.assign(RP_XMUL_WORDS, 0);
    gej A; gej_set_ge(A, g);
    for (int e = 0; e < RP_XMUL_EXPS; e++) for (u32 ring = 0; ring < RP_MAX_RINGS; ring++) for (u32 j = 1; j <= 3; j++) {
        scalar c, k, z; rp_ring_const(c, e, ring); k = c; for (u32 t = 1; t < j; t++) sc_add(k, k, c);
        sc_set_zero(z);
        gej R; u32 dig[S2K_DIG_WORDS]; const lane_mem lm{g_ptab, dig};
        ecmult_lane(R, A, k, z, 0, gtab_host(), lm);
        ge a; ge_set_gej(a, R);
        fe_to_words(g_xmul.data() + (((size_t)e * RP_MAX_RINGS + ring) * 3 + (j - 1)) * 8, a.x);
    }
    memcpy(g_hkey, gen64, 64); g_hkey_valid = 1;
}
}   // extern "C"
template <int K>
static int emu_rp_shared_k(rp_rec& rec, std::vector<u32>& bases, std::vector<u32>& pub0, std::vector<u32>& dbases, std::vector<u32>& tcur, unsigned char* ring_out, unsigned char* ring_ok,
                           const unsigned char* proof) {
    std::vector<u32> rtab((size_t)K * S2K_RTAB_WORDS, 0), rraw(2 * S2K_RING_ENTRIES * 27, 0), park((size_t)K * RP_PARK_WORDS, 0);
    u32 dig[S2K_RING_DIG_WORDS];
    int fast = 0;
    for (u32 g = 0; g * K < 32; g++) {
        const u32 r0 = g * K;
        int did = 0;
        if (rec.ok && r0 < rec.rings) {
            const rp_shared_mem M{rtab.data(), rraw.data(), park.data(), dig};
            did = rp_rings_shared<K>(rec, pub0.data() + 28 * r0, ring_out, ring_ok, proof, r0, 1, gtab_host(), g_htab.data(), g_xmul.data(), M);
        }
        if (did == RP_SHARED_SERVED) { fast += (int)((rec.rings - r0 < (u32)K) ? rec.rings - r0 : (u32)K); continue; }
        for (u32 i = r0; i < r0 + K; i++)
            rp_ring(rec, bases.data() + 28 * i, pub0.data() + 28 * i, ring_out + 33 * i, ring_ok + i, proof, i, i < rec.rings, gtab_host(), g_lm, nullptr,
                    dbases.data() + 28 * i, tcur.data() + 28 * i);
    }
    return fast;
}
extern "C" {
// rings_per_lane: 1, 2 or 4 (the engine's S2K_RP_K); *fast_rings = rings that completed in the shared-generator form
int emu_rangeproof_verify_shared(unsigned long long* min_value, unsigned long long* max_value, const unsigned char* commit33, const unsigned char* proof, size_t plen,
                                 const unsigned char* extra, size_t extra_len, const unsigned char* gen64, int* fast_rings, int rings_per_lane) {
    rp_rec rec; std::vector<u32> bases(32 * 28, 0), pub0(32 * 28, 0), dbases(32 * 28, 0), tcur(32 * 28, 0);
    unsigned char lift_ok[32] = {0}, ring_out[RP_RING_OUT_BYTES] = {0}, ring_ok[32] = {0};
    u64 mn, mx;
    htab_host(gen64);
    rp_prologue(rec, bases.data(), &mn, &mx, commit33, proof, plen, extra_len ? extra : nullptr, extra_len, gen64, dbases.data());
    *min_value = mn; *max_value = mx;
    if (rec.ok) for (u32 i = 0; i + 1 < rec.rings; i++) rp_lift(rec, pub0.data() + 28 * i, lift_ok + i, proof, i);
    rp_sum(rec, pub0.data(), lift_ok);
    int fast = 0;
    if (rings_per_lane == 4) fast = emu_rp_shared_k<4>(rec, bases, pub0, dbases, tcur, ring_out, ring_ok, proof);
    else if (rings_per_lane == 2) fast = emu_rp_shared_k<2>(rec, bases, pub0, dbases, tcur, ring_out, ring_ok, proof);
    else fast = emu_rp_shared_k<1>(rec, bases, pub0, dbases, tcur, ring_out, ring_ok, proof);
    if (fast_rings) *fast_rings = fast;
    return rp_final(rec, ring_out, ring_ok, proof);
}
unsigned long long emu_ring_step_count(void) { return g_ring_steps_done; }

// verification with the challenges kept, then rangeproof_rewind.h and the commitment check of k_rp_rewind (engine_rangeproof.hip)
int emu_rangeproof_rewind(unsigned char* blind_out, unsigned long long* value_out, unsigned char* msg_out, unsigned long long* outlen, const unsigned char* nonce32,
                          unsigned long long* min_value, unsigned long long* max_value, const unsigned char* commit33, const unsigned char* proof, size_t plen,
                          const unsigned char* gen64) {
    rp_rec rec; std::vector<u32> bases(32 * 28, 0), pub0(32 * 28, 0), ev(128 * 8, 0), prep(128 * 8, 0), secs(32 * 8, 0);
    unsigned char lift_ok[32] = {0}, ring_out[RP_RING_OUT_BYTES] = {0}, ring_ok[32] = {0};
    u64 mn, mx;
    rp_prologue(rec, bases.data(), &mn, &mx, commit33, proof, plen, nullptr, 0, gen64);
    *min_value = mn; *max_value = mx;
    if (rec.ok) for (u32 i = 0; i + 1 < rec.rings; i++) rp_lift(rec, pub0.data() + 28 * i, lift_ok + i, proof, i);
    rp_sum(rec, pub0.data(), lift_ok);
    for (u32 i = 0; i < 32; i++) rp_ring(rec, bases.data() + 28 * i, pub0.data() + 28 * i, ring_out + 33 * i, ring_ok + i, proof, i, i < rec.rings, gtab_host(), g_lm, ev.data() + 32 * i);
    if (!rp_final(rec, ring_out, ring_ok, proof)) return 0;
    scalar blind; u64 value = 0, mlen = (msg_out && outlen) ? *outlen : 0;
    if (!rp_rewind(blind, value, msg_out, &mlen, rec, proof, nonce32, gen64, ev.data(), prep.data(), secs.data())) { if (outlen) *outlen = 0; return 0; }
    u32 off; int exp, mant; u64 scale, a, b;
    rp_getheader(off, exp, mant, scale, &a, &b, proof, plen);
    const u64 vv = value * scale + mn;
    gej A, R; ge g; scalar sv;
    fe_set_b32_mod(g.x, gen64); fe_set_b32_mod(g.y, gen64 + 32); fe_norm_weak(g.x); fe_norm_weak(g.y); gej_set_ge(A, g);
    sc_set_u64(sv, vv);
    ecmult_lane(R, A, sv, blind, 1, gtab_host(), g_lm);
    if (R.inf) return 0;
    ge af; ge_set_gej(af, R);
    fe cx, cy, d;
    for (int i = 0; i < 9; i++) { cx.n[i] = rec.commit[i]; cy.n[i] = rec.commit[9 + i]; }
    fe_neg(d, af.x, 1); fe_add(d, cx); if (!fe_normalizes_to_zero(d)) return 0;
    fe_neg(d, af.y, 1); fe_add(d, cy); if (!fe_normalizes_to_zero(d)) return 0;
    sc_get_b32(blind_out, blind); *value_out = vv; if (outlen) *outlen = mlen;
    return 1;
}

int emu_schnorr_verify(const unsigned char* sig64, const unsigned char* msg, size_t msglen, const unsigned char* pk, int pk_format) {
    schnorr_midstate mid; schnorr_tag_midstate(mid);
    return schnorr_verify_lane(mid, sig64, msg, msglen, pk, pk_format, 1, gtab_host(), g_lm);
}

// the bucket MSM of msm.h run sequentially (force_c > 0 overrides the window width) for share `part` of `parts` of the windows;
// leaves the share's Jacobian partial in out28
static unsigned long g_msm_lean_refused = 0;       // runs the lean accumulation handed back (exceptional additions) since the library was loaded
unsigned long emu_msm_lean_refused(void) { return g_msm_lean_refused; }
static int emu_msm_core(u32* out28, const unsigned char* g_sc, const unsigned char* sc, const unsigned char* pt, const unsigned char* inf, size_t n, int force_c,
                        u32 part, u32 parts) {
    const size_t nt = n + (g_sc ? 1 : 0);
    msm_plan pl = msm_make_plan(nt ? nt : 1);
    if (force_c > 0) { pl.c = force_c; pl.windows = (129 + pl.c - 1) / pl.c; pl.nb = (1u << (pl.c - 1)) + 1u; pl.w0 = 0; pl.wn = pl.windows; }
    msm_plan full = pl;
    msm_plan_share(pl, part, parts);
    const size_t nk = (size_t)full.windows * full.nb;
    std::vector<u32> term(nt * MSM_TERM_WORDS + 1), keys(nt * 2 * full.windows + 1), hist(nk + 1, 0), off(nk + 1, 0), cur(nk + 1, 0), refs(nt * 2 * full.windows + 1);
    for (size_t i = 0; i < nt; i++) {
        const int isg = (g_sc && i == n);
        msm_prep(term.data() + i * MSM_TERM_WORDS, keys.data() + i * 2 * full.windows, hist.data(), isg ? g_sc : sc + 32 * i, isg ? sc : pt + 64 * i,
                 isg ? 0 : (inf ? inf[i] : 0), isg, full);
    }
    // the carry-free digit form the binning kernel uses (msm_prep_term + msm_key_at) must give exactly the same keys
    for (size_t i = 0; i < nt; i++) {
        const int isg = (g_sc && i == n);
        u32 t2[MSM_TERM_WORDS], hv[MSM_HALF_WORDS];
        msm_prep_term(t2, hv, isg ? g_sc : sc + 32 * i, isg ? sc : pt + 64 * i, isg ? 0 : (inf ? inf[i] : 0), isg);
        for (int q = 0; q < MSM_TERM_WORDS; q++) if (t2[q] != term[i * MSM_TERM_WORDS + q]) return -1;
        for (u32 w = 0; w < full.windows; w++) {
            msm_wconst wc; msm_window_const(wc, w, full.c);
            for (int half = 0; half < 2; half++) if (msm_key_at(hv, half, w, wc, full) != keys[i * 2 * full.windows + half * full.windows + w]) return -2;
        }
    }
    for (size_t k = 0; k < nk; k++) off[k + 1] = off[k] + hist[k];
    cur = off;
    for (size_t i = 0; i < nt; i++) for (int half = 0; half < 2; half++) for (u32 w = 0; w < full.windows; w++) {
        const u32 key = keys[i * 2 * full.windows + half * full.windows + w];
        if (key) refs[cur[key >> 1]++] = (u32)(i << 2) | (half << 1) | (key & 1);
    }
    std::vector<u32> wsum((pl.wn + 1) * 28);
    for (u32 wl = 0; wl < pl.wn; wl++) {
        const u32 w = pl.w0 + wl;
        gej s; gej_set_infinity(s);
        for (u32 b = 1; b < pl.nb; b++) {
            gej v, o;
            // the lean form first, the exact form when it reports an exceptional addition -- as k_msm_round1 does; and the two must agree
            // on every run the lean form accepts
            const int lean_ok = msm_sum_refs_lean(v, refs.data(), off[w * pl.nb + b], off[w * pl.nb + b + 1], term.data());
            { gej v2; msm_sum_refs(v2, refs.data(), off[w * pl.nb + b], off[w * pl.nb + b + 1], term.data());
              if (lean_ok) { gej nv2 = v2, d; if (!v2.inf) { fe_norm_weak(nv2.y); fe_neg(nv2.y, nv2.y, 1); fe_norm_weak(nv2.y); } gej_add_var(d, v, nv2); if (!d.inf || v.inf != v2.inf) return -3; } else { g_msm_lean_refused++; v = v2; } }
            msm_scale(o, v, b);
            gej t; gej_add_var(t, s, o); s = t;
        }
        gej_store28_h(wsum.data() + 28 * wl, s);
    }
    gej r; msm_combine(r, wsum.data(), pl);
    gej_store28_h(out28, r);
    return 0;
}
int emu_msm(unsigned char* r64, const unsigned char* g_sc, const unsigned char* sc, const unsigned char* pt, const unsigned char* inf, size_t n, int force_c) {
    u32 o[28];
    const int rc = emu_msm_core(o, g_sc, sc, pt, inf, n, force_c, 0, 1);
    if (rc < 0) return rc;
    gej r; gej_load28_h(r, o);
    return gej_to_b64(r64, r);
}
// the three forms of the signed window digits of a 129-bit magnitude: the carry recurrence, one window on its own (per-window constant),
// all windows from one addition; out: 3 x windows ints
int emu_msm_digit_forms(int* out, const u32* k5, unsigned c) {
    const msm_plan pl = msm_plan_for(c);
    int carry = 0;
    msm_sfull sf; msm_sum_full(sf, k5, pl.c, pl.windows);
    for (u32 w = 0; w < pl.windows; w++) {
        out[w] = msm_digit(k5, w, pl.c, carry);
        msm_wconst wc; msm_window_const(wc, w, pl.c);
        out[pl.windows + w] = msm_digit_at(k5, wc, pl.c);
        out[2 * pl.windows + w] = msm_digit_full(sf, w, pl.c, pl.windows);
    }
    return (int)pl.windows;
}
// a window share's partial by the bucket path (direct = 0) or by the bucket-free exact path of k_msm_direct (direct = 1)
int emu_msm_window_partial(u32* out28, const unsigned char* g_sc, const unsigned char* sc, const unsigned char* pt, const unsigned char* inf, size_t n,
                           unsigned part, unsigned parts, int direct, int force_c) {
    if (!direct) return emu_msm_core(out28, g_sc, sc, pt, inf, n, force_c, part, parts);
    const size_t nt = n + (g_sc ? 1 : 0);
    msm_plan pl = msm_make_plan(nt ? nt : 1);
    if (force_c > 0) { pl.c = force_c; pl.windows = (129 + pl.c - 1) / pl.c; pl.nb = (1u << (pl.c - 1)) + 1u; pl.w0 = 0; pl.wn = pl.windows; }
    msm_plan_share(pl, part, parts);
    gej acc; gej_set_infinity(acc);
    for (size_t i = 0; i < nt && pl.wn; i++) {
        gej A, R; scalar k, g; gej_set_infinity(A); sc_set_zero(k); sc_set_zero(g);
        if (i < n) { ge a; ge_from_b64(a, pt + 64 * i); fe_norm_weak(a.x); fe_norm_weak(a.y); gej_set_ge(A, a); A.inf = inf ? inf[i] : 0;
                     scalar kk; sc_set_b32(kk, sc + 32 * i, 0); msm_share_scalar(k, kk, pl); }
        else { scalar gg; sc_set_b32(gg, g_sc, 0); msm_share_scalar(g, gg, pl); }
        u32 dig[S2K_DIG_WORDS]; const lane_mem lm{g_ptab, dig};
        ecmult_lane(R, A, k, g, 1, gtab_host(), lm);
        gej s2; gej_add_var(s2, acc, R); acc = s2;
    }
    gej_store28_h(out28, acc);
    return 0;
}

int emu_bppp_verify(const unsigned char* proof, size_t proof_len, const unsigned char* transcript104, const unsigned char* rho32, const unsigned char* gens33,
                    size_t n_gens, size_t g_len, const unsigned char* c_vec32, size_t c_len, const unsigned char* commit33) {
    bp_shape sh;
    if (!bp_make_shape(sh, g_len, c_len, n_gens, proof_len)) return 0;
    std::vector<u32> gens18(n_gens * 18), term_sc(sh.n_terms * 8);
    int ok = 1;
    for (size_t i = 0; i < n_gens; i++) { ge p; ok &= bp_parse33(p, gens33 + 33 * i); fe_norm_weak(p.x); fe_norm_weak(p.y);
        for (int k = 0; k < 9; k++) { gens18[18 * i + k] = p.x.n[k]; gens18[18 * i + 9 + k] = p.y.n[k]; } }
    // both forms of the s_g vector: the serial recurrence inside bp_prologue and the per-entry products of the device path (k_bp_sg)
    std::vector<u32> term_sc2(sh.n_terms * 8), fac(8 * BP_MAX_LOG_G);
    ok &= bp_prologue(term_sc.data(), sh, proof, transcript104, rho32, c_vec32);
    if (!ok) return 0;
    if (!bp_prologue(term_sc2.data(), sh, proof, transcript104, rho32, c_vec32, fac.data())) return 0;
    for (u32 i = 1; i < sh.g_len; i++) bp_sg_entry(term_sc2.data(), fac.data(), sh, i);
    if (term_sc2 != term_sc) return 0;
    gej sum; gej_set_infinity(sum);
    for (u32 t = 0; t < sh.n_terms; t++) {
        gej o; ok &= bp_term(o, sh, t, term_sc.data(), gens18.data(), proof, commit33, 1, gtab_host(), g_lm);
        gej s; gej_add_var(s, sum, o); sum = s;
    }
    return ok & sum.inf;
}

// multi-GPU path pieces for the gloo test: a shard's Jacobian partial (28 words) and the final sum of gathered partials
void emu_msm_partial(u32* out28, const unsigned char* g_sc, const unsigned char* sc, const unsigned char* pt, const unsigned char* inf, size_t n) {
    gej acc; gej_set_infinity(acc);
    for (size_t i = 0; i < n + (g_sc ? 1 : 0); i++) {
        gej A, R; scalar k, g; gej_set_infinity(A); sc_set_zero(k); sc_set_zero(g);
        if (i < n) { ge a; ge_from_b64(a, pt + 64 * i); fe_norm_weak(a.x); fe_norm_weak(a.y); gej_set_ge(A, a); A.inf = inf ? inf[i] : 0; sc_set_b32(k, sc + 32 * i, 0); }
        else sc_set_b32(g, g_sc, 0);
        u32 dig[S2K_DIG_WORDS]; const lane_mem lm{g_ptab, dig};
        ecmult_lane(R, A, k, g, 1, gtab_host(), lm);
        gej s; gej_add_var(s, acc, R); acc = s;
    }
    gej_store28_h(out28, acc);
}
int emu_gej_sum(unsigned char* r64, const u32* gej28, size_t count) {
    gej acc; gej_set_infinity(acc);
    for (size_t i = 0; i < count; i++) { gej v, s; gej_load28_h(v, gej28 + 28 * i); gej_add_var(s, acc, v); acc = s; }
    return gej_to_b64(r64, acc);
}

int emu_surjection_verify(const unsigned char* proof, size_t plen, const unsigned char* in_tags64, size_t n_tags, const unsigned char* out_tag64) {
    return sj_verify_lane(proof, plen, in_tags64, n_tags, out_tag64, 1, gtab_host(), g_lm);
}

// the kernel sequence of secp256k1_schnorrsig_aggverify_amd (engine_halfagg.hip) run sequentially: points, schedules, chain, scalars, MSM
int emu_halfagg_verify(const unsigned char* pks, int pk_format, const unsigned char* msgs32, size_t n, const unsigned char* aggsig, size_t aggsig_len) {
    if ((aggsig_len / 32) == 0 || (aggsig_len / 32) - 1 != n || (aggsig_len % 32) != 0) return 0;
    const size_t nblocks = (3 * n) >> 1;
    std::vector<unsigned char> pts(128 * n + 1), pkx(32 * n + 1), sc(64 * n + 1);
    std::vector<u32> wk(64 * nblocks + 1), states(8 * nblocks + 1);
    int ok = 1;
    for (size_t i = 0; i < n; i++) ok &= ha_points(pts.data() + 128 * i, pkx.data() + 32 * i, aggsig + 32 * i, pks + (pk_format ? 64 : 32) * i, pk_format);
    for (size_t j = 0; j < nblocks; j++) ha_schedule(wk.data() + 64 * j, aggsig, pkx.data(), msgs32, j);
    u32 st[8]; ha_tag_midstate(st);
    for (size_t j = 0; j < nblocks; j++) { ha_rounds(st, wk.data() + 64 * j); for (int k = 0; k < 8; k++) states[8 * j + k] = st[k]; }
    schnorr_midstate mid; schnorr_tag_midstate(mid);
    for (size_t i = 0; i < n; i++) ha_scalars(sc.data() + 64 * i, states.data(), mid, aggsig, pkx.data(), msgs32, i);
    unsigned char g32[32], r64[64];
    ok &= ha_gscalar(g32, aggsig + 32 * n);
    const int inf = emu_msm(r64, g32, sc.data(), pts.data(), nullptr, 2 * n, 0);
    return ok && inf;
}
}
