/* tests/integration/hooked_tu.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Builds oracle/_ref/libsecp256k1_hooked.so: the unmodified reference translation unit (through oracle/ref_shim.c, which
 * #includes src/secp256k1.c from where it lies) + integration/secp256k1_amd_hook.c, wired the way a maintainer would wire
 * it: every call site of the static secp256k1_ecmult_multi_var *after* its own definition -- i.e. the modules (BP++ norm
 * argument, bppp_norm_product_impl.h:386,397,543) and the shim's ref_ecmult_multi -- goes through the adapter
 * secp256k1_ecmult_multi_var_amd.  The redirect is a one-line macro placed after ecmult_impl.h has been included under its
 * real name (same include prefix as src/secp256k1.c:18-31), so no reference source is edited or copied.
 */
#define SECP256K1_BUILD
#include "../include/secp256k1.h"
#include "../include/secp256k1_preallocated.h"
#include "assumptions.h"
#include "checkmem.h"
#include "util.h"
#include "field_impl.h"
#include "scalar_impl.h"
#include "group_impl.h"
#include "ecmult_impl.h"

static int secp256k1_ecmult_multi_var_amd(const secp256k1_callback *error_callback, secp256k1_scratch *scratch, secp256k1_gej *r,
        const secp256k1_scalar *inp_g_sc, secp256k1_ecmult_multi_callback cb, void *cbdata, size_t n);
#define secp256k1_ecmult_multi_var secp256k1_ecmult_multi_var_amd
#include "ref_shim.c"
#undef secp256k1_ecmult_multi_var

#include "secp256k1_amd_hook.c"

/* ---- drivers for tests/test_cpu_hook.py ---- */
typedef struct { const unsigned char *sc; const unsigned char *pt; const unsigned char *inf; long fail_at; size_t calls; } hook_cbdata;
static int hook_cb(secp256k1_scalar *sc, secp256k1_ge *pt, size_t idx, void *data) {
    hook_cbdata *d = (hook_cbdata *)data;
    d->calls++;
    if (d->fail_at >= 0 && (size_t)d->fail_at == idx) return 0;
    ref_scalar_from_b32(sc, d->sc + 32 * idx);
    ref_ge_from_b64(pt, d->pt + 64 * idx, d->inf ? d->inf[idx] : 0);
    return 1;
}
/* returns -1 when the adapter returned 0, else the infinity flag; *calls = how often the callback ran */
REF_EXPORT int hook_test_ecmult_multi(unsigned char *r64, const unsigned char *g_sc32, const unsigned char *sc32, const unsigned char *pt64,
                                      const unsigned char *inf, size_t n, long fail_at, size_t *calls) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_scratch *scratch = secp256k1_scratch_space_create(ctx, 64 * 1000 * 1000);
    secp256k1_scalar g; secp256k1_gej rj; hook_cbdata d; int ok;
    d.sc = sc32; d.pt = pt64; d.inf = inf; d.fail_at = fail_at; d.calls = 0;
    if (g_sc32) ref_scalar_from_b32(&g, g_sc32);
    ok = secp256k1_ecmult_multi_var_amd(&ctx->error_callback, scratch, &rj, g_sc32 ? &g : NULL, hook_cb, &d, n);
    if (calls) *calls = d.calls;
    secp256k1_scratch_space_destroy(ctx, scratch);
    secp256k1_context_destroy(ctx);
    if (!ok) return -1;
    return ref_gej_to_b64(r64, &rj);
}

/* r[i] = na[i]*a[i] + ng[i]*G through the batch adapter secp256k1_ecmult_batch_amd (points as 64-byte x||y + infinity flags, a random
 * Jacobian Z applied here so that the adapter sees real Jacobian inputs); returns the adapter's return value */
REF_EXPORT int hook_test_ecmult_batch(unsigned char *r64, int *rinf, const unsigned char *a64, const unsigned char *ainf, const unsigned char *na32,
                                      const unsigned char *ng32, size_t n) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_gej *a = (secp256k1_gej*)malloc(sizeof(secp256k1_gej) * (n ? n : 1)), *r = (secp256k1_gej*)malloc(sizeof(secp256k1_gej) * (n ? n : 1));
    secp256k1_scalar *na = (secp256k1_scalar*)malloc(sizeof(secp256k1_scalar) * (n ? n : 1)), *ng = (secp256k1_scalar*)malloc(sizeof(secp256k1_scalar) * (n ? n : 1));
    size_t i; int ok;
    for (i = 0; i < n; i++) {
        secp256k1_ge p; secp256k1_fe z;
        ref_ge_from_b64(&p, a64 + 64 * i, ainf ? ainf[i] : 0);
        secp256k1_gej_set_ge(&a[i], &p);
        if (!secp256k1_gej_is_infinity(&a[i])) { secp256k1_fe_set_int(&z, (int)(3 + (i % 11))); secp256k1_gej_rescale(&a[i], &z); }
        ref_scalar_from_b32(&na[i], na32 + 32 * i);
        if (ng32) ref_scalar_from_b32(&ng[i], ng32 + 32 * i);
    }
    ok = secp256k1_ecmult_batch_amd(&ctx->error_callback, r, a, na, ng32 ? ng : NULL, n);
    for (i = 0; i < n; i++) rinf[i] = ref_gej_to_b64(r64 + 64 * i, &r[i]);
    free(a); free(r); free(na); free(ng);
    secp256k1_context_destroy(ctx);
    return ok;
}
#ifdef ENABLE_MODULE_BPPP
/* n norm-argument proofs of one length over one generator set through secp256k1_amd_bppp_norm_product_verify_batch (byte-level inputs as in
 * ref_bppp_norm_verify: transcripts n x 104 bytes of SHA-256 state, rho n x 32, c_vec n x c_len x 32, commits n x 33 in the extended format) */
REF_EXPORT int hook_test_bppp_batch(int *results, const unsigned char *proofs, size_t plen, const unsigned char *transcripts, const unsigned char *rho32,
                                    const unsigned char *gens33, size_t n_gens, size_t g_len, const unsigned char *c_vec32, size_t c_len,
                                    const unsigned char *commits33, size_t n) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_scratch *scratch = secp256k1_scratch_space_create(ctx, 4 * 1000 * 1000);
    secp256k1_bppp_generators *g = ref_gens_parse(gens33, n_gens);
    const unsigned char **pp = (const unsigned char**)malloc(sizeof(*pp) * (n ? n : 1));
    secp256k1_sha256 *tr = (secp256k1_sha256*)malloc(sizeof(secp256k1_sha256) * (n ? n : 1));
    secp256k1_scalar *rho = (secp256k1_scalar*)malloc(sizeof(secp256k1_scalar) * (n ? n : 1));
    secp256k1_scalar *cv = (secp256k1_scalar*)malloc(sizeof(secp256k1_scalar) * (n * c_len + 1));
    const secp256k1_scalar **cvp = (const secp256k1_scalar**)malloc(sizeof(*cvp) * (n ? n : 1));
    secp256k1_ge *cm = (secp256k1_ge*)malloc(sizeof(secp256k1_ge) * (n ? n : 1));
    size_t i, k; int ok = 0, overflow, parsed = g != NULL;
    for (i = 0; parsed && i < n; i++) {
        pp[i] = proofs + plen * i;
        memcpy(&tr[i], transcripts + 104 * i, sizeof(secp256k1_sha256));
        secp256k1_scalar_set_b32(&rho[i], rho32 + 32 * i, &overflow);
        for (k = 0; k < c_len; k++) secp256k1_scalar_set_b32(&cv[c_len * i + k], c_vec32 + 32 * (c_len * i + k), &overflow);
        cvp[i] = cv + c_len * i;
        if (!secp256k1_ge_parse_ext(&cm[i], commits33 + 33 * i)) parsed = 0;
    }
    if (parsed) ok = secp256k1_amd_bppp_norm_product_verify_batch(ctx, scratch, results, pp, plen, tr, rho, g, g_len, cvp, c_len, cm, n);
    if (g) { free(g->gens); free(g); }
    free(pp); free(tr); free(rho); free(cv); free(cvp); free(cm);
    secp256k1_scratch_space_destroy(ctx, scratch); secp256k1_context_destroy(ctx);
    return ok;
}
#endif
