/* secp256k1_amd_hook.c -- reference-side adapters between secp256k1-zkp's own types and the engine's C ABI.
 *
 * Meant to be #included at the end of the library's translation unit (the way src/secp256k1.c includes each module's
 * main_impl.h), because -- like the modules -- it uses the library's internal types: secp256k1_scalar, secp256k1_ge,
 * secp256k1_gej, secp256k1_callback, the static secp256k1_ecmult_multi_var (reference src/ecmult.h:49,62) and
 * checked_malloc (src/util.h:162-168).  See secp256k1_amd_hook.h for the rules.  Test tier only (oracle/Makefile `hooked`).
 */
#include "secp256k1_amd_hook.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

/* The installed table.  Every adapter reads ONE pointer (secp256k1_amd_cur) once per call and works on that table to the end, so a
 * concurrent secp256k1_amd_set_backend never shows it a half-written table: the new table is written into the slot that is not
 * current, then published by a single pointer store with release order (acquire on the reading side).  Two slots suffice for the
 * documented use (install at start-up, maybe swap later); a caller that installs tables back to back while verifications run must
 * leave a verification's duration between two installs. */
static secp256k1_amd_backend secp256k1_amd_slots[2];        /* all-NULL: CPU library */
static const secp256k1_amd_backend *secp256k1_amd_cur = &secp256k1_amd_slots[0];
/* Statistics, updated by concurrent verifier threads: atomic increments (GCC/Clang builtins, which the reference's supported compilers
 * provide -- cf. its own use of __builtin_* in src/util.h; a compiler without them gets plain increments: the counters are diagnostics). */
static size_t secp256k1_amd_served = 0, secp256k1_amd_fell_back = 0;
#if defined(__GNUC__) || defined(__clang__)
#define SECP256K1_AMD_COUNT(x) ((void)__atomic_fetch_add(&(x), 1, __ATOMIC_RELAXED))
#define SECP256K1_AMD_READ(x) __atomic_load_n(&(x), __ATOMIC_RELAXED)
#else
#define SECP256K1_AMD_COUNT(x) ((void)(x)++)
#define SECP256K1_AMD_READ(x) (x)
#endif
/* MSMs shorter than this stay on the CPU: one engine round trip costs ~0.65 ms whatever the size (profiles/r03*_msm_sweep.txt), the
 * reference's Strauss / Pippenger ~3-6 us per term on one core, so the crossover sits near 200 terms */
#define SECP256K1_AMD_MSM_MIN_TERMS_DEFAULT 256
static size_t secp256k1_amd_msm_min_terms = SECP256K1_AMD_MSM_MIN_TERMS_DEFAULT;
#if defined(__GNUC__)
#define SECP256K1_AMD_LOAD_BE() ((const secp256k1_amd_backend*)__atomic_load_n(&secp256k1_amd_cur, __ATOMIC_ACQUIRE))
#define SECP256K1_AMD_STORE_BE(p) __atomic_store_n(&secp256k1_amd_cur, (p), __ATOMIC_RELEASE)
#else
#define SECP256K1_AMD_LOAD_BE() (secp256k1_amd_cur)
#define SECP256K1_AMD_STORE_BE(p) (secp256k1_amd_cur = (p))
#endif
/* pk_format 1 hands secp256k1_xonly_pubkey objects over as they lie in memory: 64 bytes = secp256k1_ge_storage (x, y as 4 x 64-bit
 * little-endian words each: the engine decodes exactly that).  Checked once, at install time. */
static int secp256k1_amd_layout_ok(void) {
    const unsigned int one = 1;
    return sizeof(secp256k1_ge_storage) == 64 && sizeof(((secp256k1_xonly_pubkey*)0)->data) == 64 && *(const unsigned char*)&one == 1;
}

SECP256K1_AMD_API void secp256k1_amd_set_backend(const secp256k1_amd_backend *backend) {
    const secp256k1_amd_backend *cur = SECP256K1_AMD_LOAD_BE();
    secp256k1_amd_backend *next = (cur == &secp256k1_amd_slots[0]) ? &secp256k1_amd_slots[1] : &secp256k1_amd_slots[0];
    if (backend == NULL) memset(next, 0, sizeof(*next));
    else {
        *next = *backend;
        if (!secp256k1_amd_layout_ok()) { next->schnorrsig_verify_batch = NULL; next->schnorrsig_aggverify = NULL; }       /* those two pass objects as raw memory */
    }
    SECP256K1_AMD_STORE_BE(next);
}
SECP256K1_AMD_API void secp256k1_amd_stats(size_t *served, size_t *fell_back) {
    if (served != NULL) *served = SECP256K1_AMD_READ(secp256k1_amd_served);
    if (fell_back != NULL) *fell_back = SECP256K1_AMD_READ(secp256k1_amd_fell_back);
}
SECP256K1_AMD_API void secp256k1_amd_set_msm_min_terms(size_t n) { secp256k1_amd_msm_min_terms = n; }

/* ---------------------------------------------------------------------------------------------------------------
 * Batch form of secp256k1_rangeproof_verify (reference include/secp256k1_rangeproof.h:70-80).
 * results[i], min_value[i], max_value[i] are what the single call returns / writes for item i.  Returns 1 when the batch
 * was processed (on the engine or on the CPU), 0 only for illegal arguments.
 * --------------------------------------------------------------------------------------------------------------- */
/* Asynchronous pair for callers with a stream of batches: `_submit` hands the batch to the engine and returns at once with a ticket (the
 * engine has gathered the inputs by then: they may be reused; results / min_value / max_value must stay valid), `_wait` blocks until the
 * verdicts are in them.  At most two submissions in flight.  Without a backend -- or when the engine refuses the submission -- `_submit`
 * runs the library's own loop then and there and hands out ticket 0, which `_wait` accepts as "done".  `_wait` returning 0 (the device failed
 * underneath a batch it had accepted) leaves results all 0 = nothing verified: verify those items again through the synchronous call. */
#if INT_MAX != 0x7fffffff
#error "the asynchronous adapters hand `int *results` to the engine as int32_t"
#endif
SECP256K1_AMD_API int secp256k1_amd_rangeproof_verify_batch_submit(const secp256k1_context *ctx, uint64_t *ticket, int *results, uint64_t *min_value, uint64_t *max_value,
        const secp256k1_pedersen_commitment *const *commits, const unsigned char *const *proofs, const size_t *plens,
        const unsigned char *const *extra_commits, const size_t *extra_commit_lens, const secp256k1_generator *const *gens, size_t n) {
    const secp256k1_amd_backend *be = SECP256K1_AMD_LOAD_BE();
    size_t i;
    VERIFY_CHECK(ctx != NULL);
    ARG_CHECK(ticket != NULL);
    *ticket = 0;
    ARG_CHECK(results != NULL);
    ARG_CHECK(min_value != NULL);
    ARG_CHECK(max_value != NULL);
    ARG_CHECK(commits != NULL);
    ARG_CHECK(proofs != NULL);
    ARG_CHECK(plens != NULL);
    ARG_CHECK(gens != NULL);
    ARG_CHECK(extra_commits == NULL || extra_commit_lens != NULL);
    for (i = 0; i < n; i++) {
        ARG_CHECK(commits[i] != NULL);
        ARG_CHECK(proofs[i] != NULL);
        ARG_CHECK(gens[i] != NULL);
        ARG_CHECK(extra_commits == NULL || extra_commits[i] != NULL || extra_commit_lens[i] == 0);
    }
    if (n == 0) return 1;
    if (be->rangeproof_verify_batch_ptrs_submit != NULL && be->rangeproof_verify_batch_wait != NULL) {
        if (be->rangeproof_verify_batch_ptrs_submit(be->engine, ticket, (int32_t*)results, min_value, max_value, (const void *const *)commits, proofs, plens,
                                                    extra_commits, extra_commit_lens, (const void *const *)gens, n)) { SECP256K1_AMD_COUNT(secp256k1_amd_served); return 1; }
        *ticket = 0;
        SECP256K1_AMD_COUNT(secp256k1_amd_fell_back);
    }
    for (i = 0; i < n; i++) {
        results[i] = secp256k1_rangeproof_verify(ctx, &min_value[i], &max_value[i], commits[i], proofs[i], plens[i],
                                                 extra_commits != NULL ? extra_commits[i] : NULL, extra_commits != NULL ? extra_commit_lens[i] : 0, gens[i]);
    }
    return 1;
}
SECP256K1_AMD_API int secp256k1_amd_rangeproof_verify_batch_wait(const secp256k1_context *ctx, uint64_t ticket) {
    const secp256k1_amd_backend *be = SECP256K1_AMD_LOAD_BE();
    VERIFY_CHECK(ctx != NULL);
    if (ticket == 0) return 1;
    ARG_CHECK(be->rangeproof_verify_batch_wait != NULL);
    return be->rangeproof_verify_batch_wait(be->engine, ticket);
}
SECP256K1_AMD_API int secp256k1_amd_rangeproof_verify_batch(const secp256k1_context *ctx, int *results, uint64_t *min_value, uint64_t *max_value,
        const secp256k1_pedersen_commitment *const *commits, const unsigned char *const *proofs, const size_t *plens,
        const unsigned char *const *extra_commits, const size_t *extra_commit_lens, const secp256k1_generator *const *gens, size_t n) {
    const secp256k1_amd_backend *be = SECP256K1_AMD_LOAD_BE();
    size_t i;
    VERIFY_CHECK(ctx != NULL);
    ARG_CHECK(results != NULL);
    ARG_CHECK(min_value != NULL);
    ARG_CHECK(max_value != NULL);
    ARG_CHECK(commits != NULL);
    ARG_CHECK(proofs != NULL);
    ARG_CHECK(plens != NULL);
    ARG_CHECK(gens != NULL);
    ARG_CHECK(extra_commits == NULL || extra_commit_lens != NULL);
    for (i = 0; i < n; i++) {
        ARG_CHECK(commits[i] != NULL);
        ARG_CHECK(proofs[i] != NULL);
        ARG_CHECK(gens[i] != NULL);
        ARG_CHECK(extra_commits == NULL || extra_commits[i] != NULL || extra_commit_lens[i] == 0);
    }
    if (n == 0) return 1;
    if (be->rangeproof_verify_batch_ptrs != NULL) {
        /* the engine gathers straight from the library's objects into its pinned staging memory (one pass over the data, several host
         * threads, the copies to the device underneath): this side only converts the result type */
        int32_t *res32 = (int32_t*)checked_malloc(&ctx->error_callback, sizeof(int32_t) * n);
        int ok = res32 != NULL;
        if (ok) {
            ok = be->rangeproof_verify_batch_ptrs(be->engine, res32, min_value, max_value, (const void *const *)commits, proofs, plens,
                                                               extra_commits, extra_commit_lens, (const void *const *)gens, n);
            if (ok) for (i = 0; i < n; i++) results[i] = res32[i] != 0;
        }
        free(res32);
        if (ok) { SECP256K1_AMD_COUNT(secp256k1_amd_served); return 1; }
        SECP256K1_AMD_COUNT(secp256k1_amd_fell_back);
    } else if (be->rangeproof_verify_batch != NULL) {
        size_t pbytes = 0, ebytes = 0, po = 0, eo = 0;
        unsigned char *c33, *pbuf, *ebuf, *g64;
        uint64_t *poff, *eoff;
        int32_t *res32;
        int ok;
        for (i = 0; i < n; i++) { pbytes += plens[i]; if (extra_commits != NULL) ebytes += extra_commit_lens[i]; }
        c33 = (unsigned char*)checked_malloc(&ctx->error_callback, 33 * n);
        g64 = (unsigned char*)checked_malloc(&ctx->error_callback, 64 * n);
        pbuf = (unsigned char*)checked_malloc(&ctx->error_callback, pbytes + 1);
        ebuf = (unsigned char*)checked_malloc(&ctx->error_callback, ebytes + 1);
        poff = (uint64_t*)checked_malloc(&ctx->error_callback, sizeof(uint64_t) * (n + 1));
        eoff = (uint64_t*)checked_malloc(&ctx->error_callback, sizeof(uint64_t) * (n + 1));
        res32 = (int32_t*)checked_malloc(&ctx->error_callback, sizeof(int32_t) * n);
        ok = c33 != NULL && g64 != NULL && pbuf != NULL && ebuf != NULL && poff != NULL && eoff != NULL && res32 != NULL;
        if (ok) {
            for (i = 0; i < n; i++) {
                /* the first 33 bytes of the 64-byte commitment object are its serialisation (generator/main_impl.h:266-279),
                 * the generator object is 64 bytes x||y (generator/main_impl.h:40-56) */
                memcpy(c33 + 33 * i, commits[i]->data, 33);
                memcpy(g64 + 64 * i, gens[i]->data, 64);
                poff[i] = po; memcpy(pbuf + po, proofs[i], plens[i]); po += plens[i];
                eoff[i] = eo;
                if (extra_commits != NULL && extra_commit_lens[i] != 0) { memcpy(ebuf + eo, extra_commits[i], extra_commit_lens[i]); eo += extra_commit_lens[i]; }
                res32[i] = 0;
            }
            poff[n] = po; eoff[n] = eo;
            ok = be->rangeproof_verify_batch(be->engine, res32, min_value, max_value, c33, pbuf, poff,
                                                          extra_commits != NULL ? ebuf : NULL, extra_commits != NULL ? eoff : NULL, g64, n);
            if (ok) for (i = 0; i < n; i++) results[i] = res32[i] != 0;
        }
        free(c33); free(g64); free(pbuf); free(ebuf); free(poff); free(eoff); free(res32);
        if (ok) { SECP256K1_AMD_COUNT(secp256k1_amd_served); return 1; }
        SECP256K1_AMD_COUNT(secp256k1_amd_fell_back);              /* engine-level failure: the whole batch takes the library's own path */
    }
    for (i = 0; i < n; i++) {
        results[i] = secp256k1_rangeproof_verify(ctx, &min_value[i], &max_value[i], commits[i], proofs[i], plens[i],
                                                 extra_commits != NULL ? extra_commits[i] : NULL, extra_commits != NULL ? extra_commit_lens[i] : 0, gens[i]);
    }
    return 1;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Batch form of secp256k1_rangeproof_rewind (reference include/secp256k1_rangeproof.h:102-130): wallet-side scanning.
 * Per item i: results[i] = the single call's return value; for results[i] == 1, blind_out[32 i ..], value_out[i],
 * min_value[i], max_value[i] and -- when message_out is given -- message_out[i][0 .. outlen[i]) are what the single call
 * writes (outlen[i] in: capacity of message_out[i], out: bytes recovered).  For results[i] == 0 the reference leaves its
 * outputs unspecified; this form zeroes blind_out / value_out and sets outlen[i] = 0 on both paths.
 * --------------------------------------------------------------------------------------------------------------- */
SECP256K1_AMD_API int secp256k1_amd_rangeproof_rewind_batch(const secp256k1_context *ctx, int *results, unsigned char *blind_out, uint64_t *value_out,
        unsigned char *const *message_out, size_t *outlen, const unsigned char *const *nonces, uint64_t *min_value, uint64_t *max_value,
        const secp256k1_pedersen_commitment *const *commits, const unsigned char *const *proofs, const size_t *plens,
        const unsigned char *const *extra_commits, const size_t *extra_commit_lens, const secp256k1_generator *const *gens, size_t n) {
    const secp256k1_amd_backend *be = SECP256K1_AMD_LOAD_BE();
    size_t i;
    VERIFY_CHECK(ctx != NULL);
    ARG_CHECK(results != NULL);
    ARG_CHECK(blind_out != NULL);
    ARG_CHECK(value_out != NULL);
    ARG_CHECK(message_out == NULL || outlen != NULL);
    ARG_CHECK(nonces != NULL);
    ARG_CHECK(min_value != NULL);
    ARG_CHECK(max_value != NULL);
    ARG_CHECK(commits != NULL);
    ARG_CHECK(proofs != NULL);
    ARG_CHECK(plens != NULL);
    ARG_CHECK(gens != NULL);
    ARG_CHECK(extra_commits == NULL || extra_commit_lens != NULL);
    for (i = 0; i < n; i++) {
        ARG_CHECK(nonces[i] != NULL);
        ARG_CHECK(commits[i] != NULL);
        ARG_CHECK(proofs[i] != NULL);
        ARG_CHECK(gens[i] != NULL);
        ARG_CHECK(message_out == NULL || message_out[i] != NULL || outlen[i] == 0);
        ARG_CHECK(extra_commits == NULL || extra_commits[i] != NULL || extra_commit_lens[i] == 0);
    }
    if (n == 0) return 1;
    if (be->rangeproof_rewind_batch != NULL) {
        size_t pbytes = 0, ebytes = 0, po = 0, eo = 0, stride = 0;
        unsigned char *c33, *pbuf, *ebuf, *g64, *nn, *msg = NULL;
        uint64_t *poff, *eoff, *olen = NULL;
        int32_t *res32;
        int ok;
        for (i = 0; i < n; i++) {
            pbytes += plens[i];
            if (extra_commits != NULL) ebytes += extra_commit_lens[i];
            if (message_out != NULL && outlen[i] > stride) stride = outlen[i];
        }
        c33 = (unsigned char*)checked_malloc(&ctx->error_callback, 33 * n);
        g64 = (unsigned char*)checked_malloc(&ctx->error_callback, 64 * n);
        nn = (unsigned char*)checked_malloc(&ctx->error_callback, 32 * n);
        pbuf = (unsigned char*)checked_malloc(&ctx->error_callback, pbytes + 1);
        ebuf = (unsigned char*)checked_malloc(&ctx->error_callback, ebytes + 1);
        poff = (uint64_t*)checked_malloc(&ctx->error_callback, sizeof(uint64_t) * (n + 1));
        eoff = (uint64_t*)checked_malloc(&ctx->error_callback, sizeof(uint64_t) * (n + 1));
        res32 = (int32_t*)checked_malloc(&ctx->error_callback, sizeof(int32_t) * n);
        ok = c33 != NULL && g64 != NULL && nn != NULL && pbuf != NULL && ebuf != NULL && poff != NULL && eoff != NULL && res32 != NULL;
        if (ok && message_out != NULL) {
            msg = (unsigned char*)checked_malloc(&ctx->error_callback, stride * n + 1);
            olen = (uint64_t*)checked_malloc(&ctx->error_callback, sizeof(uint64_t) * n);
            ok = msg != NULL && olen != NULL;
        }
        if (ok) {
            for (i = 0; i < n; i++) {
                memcpy(c33 + 33 * i, commits[i]->data, 33);
                memcpy(g64 + 64 * i, gens[i]->data, 64);
                memcpy(nn + 32 * i, nonces[i], 32);
                poff[i] = po; memcpy(pbuf + po, proofs[i], plens[i]); po += plens[i];
                eoff[i] = eo;
                if (extra_commits != NULL && extra_commit_lens[i] != 0) { memcpy(ebuf + eo, extra_commits[i], extra_commit_lens[i]); eo += extra_commit_lens[i]; }
                res32[i] = 0;
                if (olen != NULL) olen[i] = (uint64_t)outlen[i];
            }
            poff[n] = po; eoff[n] = eo;
            ok = be->rangeproof_rewind_batch(be->engine, res32, blind_out, value_out, msg, olen, stride, nn, min_value, max_value,
                                                          c33, pbuf, poff, extra_commits != NULL ? ebuf : NULL, extra_commits != NULL ? eoff : NULL, g64, n);
            if (ok) {
                for (i = 0; i < n; i++) {
                    results[i] = res32[i] != 0;
                    if (message_out != NULL) {
                        const size_t got = results[i] ? (size_t)olen[i] : 0;
                        if (got != 0) memcpy(message_out[i], msg + stride * i, got);
                        outlen[i] = got;
                    }
                }
            }
        }
        if (nn != NULL) memset(nn, 0, 32 * n);                      /* nonces are secrets */
        free(c33); free(g64); free(nn); free(pbuf); free(ebuf); free(poff); free(eoff); free(res32); free(msg); free(olen);
        if (ok) { SECP256K1_AMD_COUNT(secp256k1_amd_served); return 1; }
        SECP256K1_AMD_COUNT(secp256k1_amd_fell_back);
    }
    for (i = 0; i < n; i++) {
        size_t ol = message_out != NULL ? outlen[i] : 0;
        results[i] = secp256k1_rangeproof_rewind(ctx, blind_out + 32 * i, &value_out[i], message_out != NULL ? message_out[i] : NULL,
                                                 (message_out != NULL && message_out[i] != NULL) ? &ol : NULL,       /* the reference wants outlen == NULL with a NULL buffer */
                                                 nonces[i], &min_value[i], &max_value[i], commits[i], proofs[i], plens[i],
                                                 extra_commits != NULL ? extra_commits[i] : NULL, extra_commits != NULL ? extra_commit_lens[i] : 0, gens[i]);
        if (!results[i]) { memset(blind_out + 32 * i, 0, 32); value_out[i] = 0; ol = 0; }
        if (message_out != NULL) outlen[i] = ol;
    }
    return 1;
}

/* ---------------------------------------------------------------------------------------------------------------
 * The MSM seam: same signature and contract as the static secp256k1_ecmult_multi_var (reference src/ecmult.h:62,
 * ecmult_impl.h:823-867): R = inp_g_sc*G + sum sc_i*pt_i; inp_g_sc may be NULL; a callback that returns 0 makes the call
 * return 0; the result may be infinity.  The pull-callback (src/ecmult.h:49) is drained into arrays on the host.
 * --------------------------------------------------------------------------------------------------------------- */
static int secp256k1_ecmult_multi_var_amd(const secp256k1_callback *error_callback, secp256k1_scratch *scratch, secp256k1_gej *r,
        const secp256k1_scalar *inp_g_sc, secp256k1_ecmult_multi_callback cb, void *cbdata, size_t n) {
    const secp256k1_amd_backend *be = SECP256K1_AMD_LOAD_BE();
    if (be->ecmult_multi != NULL && n >= secp256k1_amd_msm_min_terms && n > 0) {
        unsigned char *sc = (unsigned char*)checked_malloc(error_callback, 32 * n);
        unsigned char *pt = (unsigned char*)checked_malloc(error_callback, 64 * n);
        unsigned char *inf = (unsigned char*)checked_malloc(error_callback, n);
        unsigned char g32[32], out[64];
        int32_t rinf = 0;
        int ok = sc != NULL && pt != NULL && inf != NULL;
        size_t i;
        for (i = 0; ok && i < n; i++) {
            secp256k1_scalar s; secp256k1_ge p;
            if (!cb(&s, &p, i, cbdata)) { free(sc); free(pt); free(inf); return 0; }        /* the reference's rule, ecmult_impl.h:747-750 */
            secp256k1_scalar_get_b32(sc + 32 * i, &s);
            inf[i] = (unsigned char)secp256k1_ge_is_infinity(&p);
            if (inf[i]) memset(pt + 64 * i, 0, 64);
            else {
                secp256k1_fe_normalize_var(&p.x); secp256k1_fe_normalize_var(&p.y);
                secp256k1_fe_get_b32(pt + 64 * i, &p.x); secp256k1_fe_get_b32(pt + 64 * i + 32, &p.y);
            }
        }
        if (ok) {
            if (inp_g_sc != NULL) secp256k1_scalar_get_b32(g32, inp_g_sc);
            ok = be->ecmult_multi(be->engine, out, &rinf, inp_g_sc != NULL ? g32 : NULL, sc, pt, inf, n);
        }
        free(sc); free(pt); free(inf);
        if (ok) {
            SECP256K1_AMD_COUNT(secp256k1_amd_served);
            if (rinf) secp256k1_gej_set_infinity(r);
            else {
                secp256k1_ge a; secp256k1_fe x, y;
                secp256k1_fe_set_b32_mod(&x, out); secp256k1_fe_set_b32_mod(&y, out + 32);
                secp256k1_ge_set_xy(&a, &x, &y);
                secp256k1_gej_set_ge(r, &a);
            }
            return 1;
        }
        SECP256K1_AMD_COUNT(secp256k1_amd_fell_back);
    }
    return secp256k1_ecmult_multi_var(error_callback, scratch, r, inp_g_sc, cb, cbdata, n);
}

/* ---------------------------------------------------------------------------------------------------------------
 * Batch form of the single double multiplication: r[i] = na[i]*a[i] + ng[i]*G, the static secp256k1_ecmult (reference src/ecmult.h:47,
 * ecmult_impl.h:365-375) n times.  ng may be NULL (every ng[i] = 0, the reference's ng == NULL).  Returns 1.
 * --------------------------------------------------------------------------------------------------------------- */
static int secp256k1_ecmult_batch_amd(const secp256k1_callback *error_callback, secp256k1_gej *r, const secp256k1_gej *a, const secp256k1_scalar *na,
        const secp256k1_scalar *ng, size_t n) {
    const secp256k1_amd_backend *be = SECP256K1_AMD_LOAD_BE();
    size_t i;
    if (n == 0) return 1;
    if (be->ecmult_batch != NULL) {
        unsigned char *axy = (unsigned char*)checked_malloc(error_callback, 64 * n);
        unsigned char *ainf = (unsigned char*)checked_malloc(error_callback, n);
        unsigned char *sna = (unsigned char*)checked_malloc(error_callback, 32 * n);
        unsigned char *sng = (unsigned char*)checked_malloc(error_callback, 32 * n);
        unsigned char *rxy = (unsigned char*)checked_malloc(error_callback, 64 * n);
        int32_t *rinf = (int32_t*)checked_malloc(error_callback, sizeof(int32_t) * n);
        int ok = axy != NULL && ainf != NULL && sna != NULL && sng != NULL && rxy != NULL && rinf != NULL;
        if (ok) {
            for (i = 0; i < n; i++) {
                secp256k1_ge p;
                ainf[i] = (unsigned char)secp256k1_gej_is_infinity(&a[i]);
                if (ainf[i]) memset(axy + 64 * i, 0, 64);
                else {
                    secp256k1_gej t = a[i];
                    secp256k1_ge_set_gej_var(&p, &t);
                    secp256k1_fe_normalize_var(&p.x); secp256k1_fe_normalize_var(&p.y);
                    secp256k1_fe_get_b32(axy + 64 * i, &p.x); secp256k1_fe_get_b32(axy + 64 * i + 32, &p.y);
                }
                secp256k1_scalar_get_b32(sna + 32 * i, &na[i]);
                if (ng != NULL) secp256k1_scalar_get_b32(sng + 32 * i, &ng[i]);
            }
            ok = be->ecmult_batch(be->engine, rxy, rinf, axy, ainf, sna, ng != NULL ? sng : NULL, n);
            if (ok) {
                for (i = 0; i < n; i++) {
                    if (rinf[i]) secp256k1_gej_set_infinity(&r[i]);
                    else {
                        secp256k1_ge q; secp256k1_fe x, y;
                        secp256k1_fe_set_b32_mod(&x, rxy + 64 * i); secp256k1_fe_set_b32_mod(&y, rxy + 64 * i + 32);
                        secp256k1_ge_set_xy(&q, &x, &y);
                        secp256k1_gej_set_ge(&r[i], &q);
                    }
                }
            }
        }
        free(axy); free(ainf); free(sna); free(sng); free(rxy); free(rinf);
        if (ok) { SECP256K1_AMD_COUNT(secp256k1_amd_served); return 1; }
        SECP256K1_AMD_COUNT(secp256k1_amd_fell_back);
    }
    for (i = 0; i < n; i++) secp256k1_ecmult(&r[i], &a[i], &na[i], ng != NULL ? &ng[i] : NULL);
    return 1;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Batch form of the static secp256k1_bppp_rangeproof_norm_product_verify (reference src/modules/bppp/bppp_norm_product_impl.h:425-552):
 * n proofs of one length over ONE generator set; item i has its own transcript, rho, c_vec and commitment.  The transcripts are read,
 * not advanced (the reference's verify advances its own copy).  results[i] = the single call's return value.
 * --------------------------------------------------------------------------------------------------------------- */
#ifdef ENABLE_MODULE_BPPP
static int secp256k1_amd_bppp_norm_product_verify_batch(const secp256k1_context *ctx, secp256k1_scratch_space *scratch, int *results,
        const unsigned char *const *proofs, size_t proof_len, const secp256k1_sha256 *transcripts, const secp256k1_scalar *rhos,
        const secp256k1_bppp_generators *g_vec, size_t g_len, const secp256k1_scalar *const *c_vecs, size_t c_vec_len,
        const secp256k1_ge *commits, size_t n) {
    const secp256k1_amd_backend *be = SECP256K1_AMD_LOAD_BE();
    size_t i, k;
    if (n == 0) return 1;
    if (be->bppp_norm_product_verify_batch != NULL && sizeof(secp256k1_sha256) == 104) {
        unsigned char *pr = (unsigned char*)checked_malloc(&ctx->error_callback, proof_len * n + 1);
        unsigned char *tr = (unsigned char*)checked_malloc(&ctx->error_callback, 104 * n);
        unsigned char *rh = (unsigned char*)checked_malloc(&ctx->error_callback, 32 * n);
        unsigned char *gs = (unsigned char*)checked_malloc(&ctx->error_callback, 33 * g_vec->n + 1);
        unsigned char *cv = (unsigned char*)checked_malloc(&ctx->error_callback, 32 * c_vec_len * n + 1);
        unsigned char *cm = (unsigned char*)checked_malloc(&ctx->error_callback, 33 * n);
        int32_t *res32 = (int32_t*)checked_malloc(&ctx->error_callback, sizeof(int32_t) * n);
        int ok = pr != NULL && tr != NULL && rh != NULL && gs != NULL && cv != NULL && cm != NULL && res32 != NULL;
        if (ok) {
            for (k = 0; k < g_vec->n; k++) { secp256k1_ge t = g_vec->gens[k]; secp256k1_ge_serialize_ext(gs + 33 * k, &t); }
            for (i = 0; i < n; i++) {
                secp256k1_ge t = commits[i];
                memcpy(pr + proof_len * i, proofs[i], proof_len);
                memcpy(tr + 104 * i, &transcripts[i], 104);
                secp256k1_scalar_get_b32(rh + 32 * i, &rhos[i]);
                for (k = 0; k < c_vec_len; k++) secp256k1_scalar_get_b32(cv + 32 * (c_vec_len * i + k), &c_vecs[i][k]);
                secp256k1_ge_serialize_ext(cm + 33 * i, &t);
                res32[i] = 0;
            }
            ok = be->bppp_norm_product_verify_batch(be->engine, res32, pr, proof_len, tr, rh, gs, g_vec->n, g_len, cv, c_vec_len, cm, n);
            if (ok) for (i = 0; i < n; i++) results[i] = res32[i] != 0;
        }
        free(pr); free(tr); free(rh); free(gs); free(cv); free(cm); free(res32);
        if (ok) { SECP256K1_AMD_COUNT(secp256k1_amd_served); return 1; }
        SECP256K1_AMD_COUNT(secp256k1_amd_fell_back);
    }
    for (i = 0; i < n; i++) {
        /* the reference's verify modifies its generator set, c_vec and transcript: give it copies */
        secp256k1_sha256 t = transcripts[i];
        secp256k1_bppp_generators gcopy; secp256k1_scalar *cc; secp256k1_scalar rho = rhos[i];
        gcopy.n = g_vec->n;
        gcopy.gens = (secp256k1_ge*)checked_malloc(&ctx->error_callback, sizeof(secp256k1_ge) * (g_vec->n ? g_vec->n : 1));
        cc = (secp256k1_scalar*)checked_malloc(&ctx->error_callback, sizeof(secp256k1_scalar) * (c_vec_len ? c_vec_len : 1));
        results[i] = 0;
        if (gcopy.gens != NULL && cc != NULL) {
            memcpy(gcopy.gens, g_vec->gens, sizeof(secp256k1_ge) * g_vec->n);
            memcpy(cc, c_vecs[i], sizeof(secp256k1_scalar) * c_vec_len);
            results[i] = secp256k1_bppp_rangeproof_norm_product_verify(ctx, scratch, proofs[i], proof_len, &t, &rho, &gcopy, g_len, cc, c_vec_len, &commits[i]);
        }
        free(gcopy.gens); free(cc);
    }
    return 1;
}
#endif

/* ---------------------------------------------------------------------------------------------------------------
 * Batch form of secp256k1_schnorrsig_verify (reference include/secp256k1_schnorrsig.h:178); all messages msglen long.
 * --------------------------------------------------------------------------------------------------------------- */
#ifdef ENABLE_MODULE_SCHNORRSIG
SECP256K1_AMD_API int secp256k1_amd_schnorrsig_verify_batch(const secp256k1_context *ctx, int *results, const unsigned char *const *sigs64,
        const unsigned char *const *msgs, size_t msglen, const secp256k1_xonly_pubkey *const *pubkeys, size_t n) {
    const secp256k1_amd_backend *be = SECP256K1_AMD_LOAD_BE();
    size_t i;
    VERIFY_CHECK(ctx != NULL);
    ARG_CHECK(results != NULL);
    ARG_CHECK(sigs64 != NULL);
    ARG_CHECK(msgs != NULL || msglen == 0);
    ARG_CHECK(pubkeys != NULL);
    for (i = 0; i < n; i++) { ARG_CHECK(sigs64[i] != NULL); ARG_CHECK(msglen == 0 || msgs[i] != NULL); ARG_CHECK(pubkeys[i] != NULL); }
    if (n == 0) return 1;
    if (be->schnorrsig_verify_batch != NULL) {
        unsigned char *s = (unsigned char*)checked_malloc(&ctx->error_callback, 64 * n);
        unsigned char *m = (unsigned char*)checked_malloc(&ctx->error_callback, msglen * n + 1);
        unsigned char *pk = (unsigned char*)checked_malloc(&ctx->error_callback, 64 * n);
        int32_t *res32 = (int32_t*)checked_malloc(&ctx->error_callback, sizeof(int32_t) * n);
        int ok = s != NULL && m != NULL && pk != NULL && res32 != NULL;
        if (ok) {
            for (i = 0; i < n; i++) {
                memcpy(s + 64 * i, sigs64[i], 64);
                if (msglen != 0) memcpy(m + msglen * i, msgs[i], msglen);
                memcpy(pk + 64 * i, pubkeys[i]->data, 64);              /* pk_format 1: the opaque object as it lies in memory */
                res32[i] = 0;
            }
            ok = be->schnorrsig_verify_batch(be->engine, res32, s, m, msglen, pk, 1, n);
            if (ok) for (i = 0; i < n; i++) results[i] = res32[i] != 0;
        }
        free(s); free(m); free(pk); free(res32);
        if (ok) { SECP256K1_AMD_COUNT(secp256k1_amd_served); return 1; }
        SECP256K1_AMD_COUNT(secp256k1_amd_fell_back);
    }
    for (i = 0; i < n; i++) results[i] = secp256k1_schnorrsig_verify(ctx, sigs64[i], msglen != 0 ? msgs[i] : NULL, msglen, pubkeys[i]);
    return 1;
}
#endif

/* ---------------------------------------------------------------------------------------------------------------
 * secp256k1_schnorrsig_aggverify (reference include/secp256k1_schnorrsig_halfagg.h:94-101) with the reference's own argument
 * list: the engine checks the half-aggregate as ONE (2n+1)-term multi-scalar multiplication (the reference: 2n single
 * multiplications); the array of secp256k1_xonly_pubkey objects is handed over as it lies in memory (pk_format 1).
 * Returns the verdict; an engine that cannot give one leaves the call to the CPU.
 * --------------------------------------------------------------------------------------------------------------- */
#ifdef ENABLE_MODULE_SCHNORRSIG_HALFAGG
SECP256K1_AMD_API int secp256k1_amd_schnorrsig_aggverify(const secp256k1_context *ctx, const secp256k1_xonly_pubkey *pubkeys, const unsigned char *msgs32, size_t n,
        const unsigned char *aggsig, size_t aggsig_len) {
    const secp256k1_amd_backend *be = SECP256K1_AMD_LOAD_BE();
    VERIFY_CHECK(ctx != NULL);
    ARG_CHECK(pubkeys != NULL || n == 0);
    ARG_CHECK(msgs32 != NULL || n == 0);
    ARG_CHECK(aggsig != NULL);
    if (be->schnorrsig_aggverify != NULL && n != 0) {
        int32_t verdict = 0;
        if (be->schnorrsig_aggverify(be->engine, &verdict, (const unsigned char*)pubkeys, 1, msgs32, n, aggsig, aggsig_len)) {
            SECP256K1_AMD_COUNT(secp256k1_amd_served);
            return verdict != 0;
        }
        SECP256K1_AMD_COUNT(secp256k1_amd_fell_back);
    }
    return secp256k1_schnorrsig_aggverify(ctx, pubkeys, msgs32, n, aggsig, aggsig_len);
}
#endif

/* ---------------------------------------------------------------------------------------------------------------
 * Batch form of secp256k1_surjectionproof_verify (reference include/secp256k1_surjectionproof.h:256).
 * Item i: proofs[i], input_tags[i][0 .. n_input_tags[i]), output_tags[i].
 * --------------------------------------------------------------------------------------------------------------- */
#ifdef ENABLE_MODULE_SURJECTIONPROOF
SECP256K1_AMD_API int secp256k1_amd_surjectionproof_verify_batch(const secp256k1_context *ctx, int *results, const secp256k1_surjectionproof *const *proofs,
        const secp256k1_generator *const *input_tags, const size_t *n_input_tags, const secp256k1_generator *const *output_tags, size_t n) {
    const secp256k1_amd_backend *be = SECP256K1_AMD_LOAD_BE();
    size_t i;
    VERIFY_CHECK(ctx != NULL);
    ARG_CHECK(results != NULL);
    ARG_CHECK(proofs != NULL);
    ARG_CHECK(input_tags != NULL);
    ARG_CHECK(n_input_tags != NULL);
    ARG_CHECK(output_tags != NULL);
    for (i = 0; i < n; i++) { ARG_CHECK(proofs[i] != NULL); ARG_CHECK(input_tags[i] != NULL); ARG_CHECK(output_tags[i] != NULL); }
    if (n == 0) return 1;
    if (be->surjectionproof_verify_batch != NULL) {
        size_t ntags = 0, po = 0, to = 0;
        unsigned char *pbuf, *tags, *outs;
        uint64_t *poff, *toff;
        int32_t *res32;
        int ok;
        for (i = 0; i < n; i++) ntags += n_input_tags[i];
        pbuf = (unsigned char*)checked_malloc(&ctx->error_callback, SECP256K1_SURJECTIONPROOF_SERIALIZATION_BYTES_MAX * n);
        tags = (unsigned char*)checked_malloc(&ctx->error_callback, 64 * ntags + 1);
        outs = (unsigned char*)checked_malloc(&ctx->error_callback, 64 * n);
        poff = (uint64_t*)checked_malloc(&ctx->error_callback, sizeof(uint64_t) * (n + 1));
        toff = (uint64_t*)checked_malloc(&ctx->error_callback, sizeof(uint64_t) * (n + 1));
        res32 = (int32_t*)checked_malloc(&ctx->error_callback, sizeof(int32_t) * n);
        ok = pbuf != NULL && tags != NULL && outs != NULL && poff != NULL && toff != NULL && res32 != NULL;
        for (i = 0; ok && i < n; i++) {
            size_t len = SECP256K1_SURJECTIONPROOF_SERIALIZATION_BYTES_MAX, k;
            poff[i] = po; toff[i] = to;
            ok = secp256k1_surjectionproof_serialize(ctx, pbuf + po, &len, proofs[i]);
            po += len;
            for (k = 0; k < n_input_tags[i]; k++) memcpy(tags + 64 * (to + k), input_tags[i][k].data, 64);
            to += n_input_tags[i];
            memcpy(outs + 64 * i, output_tags[i]->data, 64);
            res32[i] = 0;
        }
        if (ok) {
            poff[n] = po; toff[n] = to;
            ok = be->surjectionproof_verify_batch(be->engine, res32, pbuf, poff, tags, toff, outs, n);
            if (ok) for (i = 0; i < n; i++) results[i] = res32[i] != 0;
        }
        free(pbuf); free(tags); free(outs); free(poff); free(toff); free(res32);
        if (ok) { SECP256K1_AMD_COUNT(secp256k1_amd_served); return 1; }
        SECP256K1_AMD_COUNT(secp256k1_amd_fell_back);
    }
    for (i = 0; i < n; i++) results[i] = secp256k1_surjectionproof_verify(ctx, proofs[i], input_tags[i], n_input_tags[i], output_tags[i]);
    return 1;
}
#endif

/* ---------------------------------------------------------------------------------------------------------------
 * Batch form of secp256k1_pedersen_verify_tally (reference include/secp256k1_generator.h:190): tally t checks
 * sum(pos[t][0..pcnt[t])) - sum(neg[t][0..ncnt[t])) == 0.
 * --------------------------------------------------------------------------------------------------------------- */
#ifdef ENABLE_MODULE_GENERATOR
SECP256K1_AMD_API int secp256k1_amd_pedersen_verify_tally_batch(const secp256k1_context *ctx, int *results,
        const secp256k1_pedersen_commitment *const *const *pos, const size_t *pcnt,
        const secp256k1_pedersen_commitment *const *const *neg, const size_t *ncnt, size_t n_tallies) {
    const secp256k1_amd_backend *be = SECP256K1_AMD_LOAD_BE();
    size_t t, k;
    VERIFY_CHECK(ctx != NULL);
    ARG_CHECK(results != NULL);
    ARG_CHECK(pos != NULL);
    ARG_CHECK(pcnt != NULL);
    ARG_CHECK(neg != NULL);
    ARG_CHECK(ncnt != NULL);
    for (t = 0; t < n_tallies; t++) { ARG_CHECK(pcnt[t] == 0 || pos[t] != NULL); ARG_CHECK(ncnt[t] == 0 || neg[t] != NULL); }
    if (n_tallies == 0) return 1;
    if (be->pedersen_verify_tally_batch != NULL) {
        size_t total = 0, o = 0;
        unsigned char *c33;
        uint64_t *off, *npos;
        int32_t *res32;
        int ok;
        for (t = 0; t < n_tallies; t++) total += pcnt[t] + ncnt[t];
        c33 = (unsigned char*)checked_malloc(&ctx->error_callback, 33 * total + 1);
        off = (uint64_t*)checked_malloc(&ctx->error_callback, sizeof(uint64_t) * (n_tallies + 1));
        npos = (uint64_t*)checked_malloc(&ctx->error_callback, sizeof(uint64_t) * n_tallies);
        res32 = (int32_t*)checked_malloc(&ctx->error_callback, sizeof(int32_t) * n_tallies);
        ok = c33 != NULL && off != NULL && npos != NULL && res32 != NULL;
        if (ok) {
            for (t = 0; t < n_tallies; t++) {
                off[t] = o; npos[t] = pcnt[t]; res32[t] = 0;
                for (k = 0; k < pcnt[t]; k++) { memcpy(c33 + 33 * o, pos[t][k]->data, 33); o++; }
                for (k = 0; k < ncnt[t]; k++) { memcpy(c33 + 33 * o, neg[t][k]->data, 33); o++; }
            }
            off[n_tallies] = o;
            ok = be->pedersen_verify_tally_batch(be->engine, res32, c33, off, npos, n_tallies);
            if (ok) for (t = 0; t < n_tallies; t++) results[t] = res32[t] != 0;
        }
        free(c33); free(off); free(npos); free(res32);
        if (ok) { SECP256K1_AMD_COUNT(secp256k1_amd_served); return 1; }
        SECP256K1_AMD_COUNT(secp256k1_amd_fell_back);
    }
    for (t = 0; t < n_tallies; t++) results[t] = secp256k1_pedersen_verify_tally(ctx, pos[t], pcnt[t], neg[t], ncnt[t]);
    return 1;
}
#endif
