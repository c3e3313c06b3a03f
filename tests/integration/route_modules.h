/* route_modules.h -- TEST INFRASTRUCTURE: the reference's own test program as a differential driver of the MI355X engine.
 *
 * oracle/Makefile (target _ref/ref_tests_routed) compiles the reference's src/tests.c -- which #includes src/secp256k1.c -- from the
 * sources where they lie, with two one-line insertions made on throw-away copies:
 *   - THIS file, in secp256k1.c between the core (field/scalar/group/ecmult) and the modules' main_impl.h includes (src/secp256k1.c:905),
 *     so that every call of the two static seams of the hot path,
 *         secp256k1_ecmult            (src/ecmult.h:47)   -- the double multiplication Borromean / BIP-340 / whitelist / ... run on
 *         secp256k1_ecmult_multi_var  (src/ecmult.h:62)   -- the multi-scalar multiplication BP++ and MuSig key aggregation run on
 *     made by the modules AND by the reference's tests (run_ecmult_*, test_ecmult_multi with its infinities, zero scalars, cancelling
 *     terms and failing callbacks) first runs the reference's code and then the engine on the same operands, and aborts on any difference;
 *   - route_api.h, in tests.c after the include of secp256k1.c, which does the same for the public verifiers.
 * The reference's return values and outputs are what the tests see: the test program's own assertions are untouched, the engine is an
 * observer whose disagreement is fatal.  Nothing here is part of the product or of the reference-side hook (integration/).
 *
 * Environment: S2K_RT_OFF=1 -- no engine, count only (CPU containers); S2K_RT_ECMULT_EVERY=k -- check every k-th secp256k1_ecmult call
 * (default 8; the multi-scalar seam and the public verifiers are always checked). */
#ifndef S2K_ROUTE_MODULES_H
#define S2K_ROUTE_MODULES_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "secp256k1_zkp_amd.h"

enum { S2K_RT_ECMULT, S2K_RT_MSM, S2K_RT_RANGEPROOF, S2K_RT_REWIND, S2K_RT_SCHNORR, S2K_RT_TALLY, S2K_RT_SURJECTION, S2K_RT_HALFAGG, S2K_RT_BPPP, S2K_RT_KINDS };
static const char *const s2k_rt_names[S2K_RT_KINDS] = {"ecmult", "ecmult_multi_var", "rangeproof_verify", "rangeproof_rewind", "schnorrsig_verify",
                                                       "pedersen_verify_tally", "surjectionproof_verify", "schnorrsig_aggverify", "bppp_norm_product_verify"};
static struct {
    int ready, off;
    s2k_engine *e;
    unsigned long every;
    unsigned long calls[S2K_RT_KINDS], checked[S2K_RT_KINDS], accepted[S2K_RT_KINDS];
    unsigned long msm_terms, msm_max;
} s2k_rt;

static void s2k_rt_report(void) {
    int k;
    fprintf(stderr, "s2k-route: engine %s\n", s2k_rt.off ? "OFF (count only)" : "on");
    for (k = 0; k < S2K_RT_KINDS; k++)
        fprintf(stderr, "s2k-route: %-26s calls %9lu  checked %9lu  reference-accepted %9lu\n", s2k_rt_names[k], s2k_rt.calls[k], s2k_rt.checked[k], s2k_rt.accepted[k]);
    fprintf(stderr, "s2k-route: ecmult_multi_var terms checked %lu, largest call %lu\n", s2k_rt.msm_terms, s2k_rt.msm_max);
}
static void s2k_rt_die(const char *what, const char *why) {
    fprintf(stderr, "s2k-route: MISMATCH in %s: %s (engine status %d: %s)\n", what, why, s2k_last_status(), s2k_last_error());
    s2k_rt_report();
    abort();
}
static int s2k_rt_on(void) {
    if (!s2k_rt.ready) {
        const char *ev = getenv("S2K_RT_ECMULT_EVERY");
        s2k_rt.ready = 1;
        s2k_rt.off = getenv("S2K_RT_OFF") != NULL;
        s2k_rt.every = ev != NULL ? strtoul(ev, NULL, 10) : 8;
        if (s2k_rt.every == 0) s2k_rt.every = 1;
        atexit(s2k_rt_report);
        if (!s2k_rt.off) {
            s2k_rt.e = s2k_engine_create(0);
            if (s2k_rt.e == NULL) { fprintf(stderr, "s2k-route: no engine: %s\n", s2k_last_error()); abort(); }    /* never a silent CPU-only pass */
        }
    }
    return !s2k_rt.off;
}

static void s2k_rt_ge_bytes(unsigned char *xy64, unsigned char *inf, const secp256k1_ge *p) {
    *inf = (unsigned char)secp256k1_ge_is_infinity(p);
    if (*inf) memset(xy64, 0, 64);
    else {
        secp256k1_fe x = p->x, y = p->y;
        secp256k1_fe_normalize_var(&x); secp256k1_fe_normalize_var(&y);
        secp256k1_fe_get_b32(xy64, &x); secp256k1_fe_get_b32(xy64 + 32, &y);
    }
}
static void s2k_rt_gej_bytes(unsigned char *xy64, unsigned char *inf, const secp256k1_gej *p) {
    secp256k1_ge a;
    if (secp256k1_gej_is_infinity(p)) { *inf = 1; memset(xy64, 0, 64); return; }
    { secp256k1_gej t = *p; secp256k1_ge_set_gej_var(&a, &t); }
    s2k_rt_ge_bytes(xy64, inf, &a);
}

/* ---- secp256k1_ecmult: r = na*a + ng*G; a may be infinity, ng may be NULL, r may alias a (src/ecmult.h:41-47) ---- */
static void s2k_rt_ecmult(secp256k1_gej *r, const secp256k1_gej *a, const secp256k1_scalar *na, const secp256k1_scalar *ng) {
    unsigned char axy[64], ainf, sna[32], sng[32], want[64], winf, got[64];
    int32_t ginf = 0;
    int check;
    s2k_rt.calls[S2K_RT_ECMULT]++;
    check = s2k_rt_on() && (s2k_rt.calls[S2K_RT_ECMULT] % s2k_rt.every) == 0;
    if (check) {
        if (a != NULL) s2k_rt_gej_bytes(axy, &ainf, a);
        else { ainf = 1; memset(axy, 0, 64); }      /* the reference's ellswift tests pass a == NULL with na == 0 (never read then: ecmult_impl.h:268) */
        secp256k1_scalar_get_b32(sna, na);
        if (ng != NULL) secp256k1_scalar_get_b32(sng, ng);
    }
    secp256k1_ecmult(r, a, na, ng);
    if (!check) return;
    s2k_rt_gej_bytes(want, &winf, r);
    if (!s2k_ecmult_batch(s2k_rt.e, got, &ginf, axy, &ainf, sna, ng != NULL ? sng : NULL, 1)) s2k_rt_die("secp256k1_ecmult", "the engine call failed");
    if ((ginf != 0) != (winf != 0) || memcmp(got, want, 64) != 0) s2k_rt_die("secp256k1_ecmult", "different point");
    s2k_rt.checked[S2K_RT_ECMULT]++;
}

/* ---- secp256k1_ecmult_multi_var: r = inp_g_sc*G + sum sc_i*pt_i through the pull callback (src/ecmult.h:49-62, ecmult_impl.h:823-867).
 * The callback is drained a second time for the engine: every callback of the reference and of its tests is a function of (idx, data). */
static int s2k_rt_ecmult_multi_var(const secp256k1_callback *error_callback, secp256k1_scratch *scratch, secp256k1_gej *r,
                                   const secp256k1_scalar *inp_g_sc, secp256k1_ecmult_multi_callback cb, void *cbdata, size_t n) {
    const int ret = secp256k1_ecmult_multi_var(error_callback, scratch, r, inp_g_sc, cb, cbdata, n);
    s2k_rt.calls[S2K_RT_MSM]++;
    if (ret) s2k_rt.accepted[S2K_RT_MSM]++;
    if (ret && s2k_rt_on()) {                      /* a 0 is "the callback refused": no result to compare (ecmult_impl.h:747-750) */
        unsigned char *sc = (unsigned char*)malloc(32 * n + 1), *pt = (unsigned char*)malloc(64 * n + 1), *inf = (unsigned char*)malloc(n + 1);
        unsigned char g32[32], want[64], winf, got[64];
        int32_t ginf = 0;
        size_t i;
        if (sc == NULL || pt == NULL || inf == NULL) s2k_rt_die("secp256k1_ecmult_multi_var", "out of host memory");
        for (i = 0; i < n; i++) {
            secp256k1_scalar s; secp256k1_ge p;
            if (!cb(&s, &p, i, cbdata)) s2k_rt_die("secp256k1_ecmult_multi_var", "the callback answered differently the second time");
            secp256k1_scalar_get_b32(sc + 32 * i, &s);
            s2k_rt_ge_bytes(pt + 64 * i, inf + i, &p);
        }
        if (inp_g_sc != NULL) secp256k1_scalar_get_b32(g32, inp_g_sc);
        s2k_rt_gej_bytes(want, &winf, r);
        if (!s2k_ecmult_multi(s2k_rt.e, got, &ginf, inp_g_sc != NULL ? g32 : NULL, sc, pt, inf, n)) s2k_rt_die("secp256k1_ecmult_multi_var", "the engine call failed");
        if ((ginf != 0) != (winf != 0) || memcmp(got, want, 64) != 0) s2k_rt_die("secp256k1_ecmult_multi_var", "different point");
        free(sc); free(pt); free(inf);
        s2k_rt.checked[S2K_RT_MSM]++;
        s2k_rt.msm_terms += n;
        if (n > s2k_rt.msm_max) s2k_rt.msm_max = n;
    }
    return ret;
}

#define secp256k1_ecmult s2k_rt_ecmult
#define secp256k1_ecmult_multi_var s2k_rt_ecmult_multi_var
#endif
