/* oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Builds the *unmodified* reference (BlockstreamResearch/secp256k1-zkp) into
 * oracle/_ref/libsecp256k1_ref.so by #including its single translation unit from
 * where it lies (-I$(REF)/src, see oracle/Makefile); no reference source is copied
 * into this repository.  The reference keeps its hot-path internals `static`
 * (src/ecmult.h:47,62; src/secp256k1.c:263), so -- exactly like the reference's own
 * src/bench_ecmult.c:9 and src/tests.c do -- this file includes secp256k1.c and adds
 * thin byte-level wrappers (`ref_*`) around them.  Everything crosses the boundary in
 * serialised form (32-byte big-endian field elements / scalars, 64-byte x||y affine
 * points + an infinity flag), which is the level at which parity is defined
 * (limb values are representation dependent, bytes are not).
 *
 * The public API of the reference (secp256k1_context_create, secp256k1_rangeproof_sign,
 * secp256k1_rangeproof_verify, secp256k1_schnorrsig_verify, ...) is exported by the same
 * .so unchanged and is what tests use as the per-item oracle.
 */
#include "secp256k1.c"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define REF_EXPORT __attribute__((visibility("default")))

/* ---------- helpers: bytes <-> internal types ---------- */
static void ref_fe_from_b32(secp256k1_fe *r, const unsigned char *b) { secp256k1_fe_set_b32_mod(r, b); }
static void ref_fe_to_b32(unsigned char *b, const secp256k1_fe *a) {
    secp256k1_fe t = *a; secp256k1_fe_normalize_var(&t); secp256k1_fe_get_b32(b, &t);
}
static void ref_ge_from_b64(secp256k1_ge *r, const unsigned char *b, int inf) {
    if (inf) { secp256k1_ge_set_infinity(r); return; }
    {
        secp256k1_fe x, y;
        ref_fe_from_b32(&x, b); ref_fe_from_b32(&y, b + 32);
        secp256k1_ge_set_xy(r, &x, &y);
    }
}
static int ref_gej_to_b64(unsigned char *b, secp256k1_gej *j) {
    secp256k1_ge a;
    if (secp256k1_gej_is_infinity(j)) { memset(b, 0, 64); return 1; }
    secp256k1_ge_set_gej_var(&a, j);
    ref_fe_to_b32(b, &a.x); ref_fe_to_b32(b + 32, &a.y);
    return 0;
}
static void ref_scalar_from_b32(secp256k1_scalar *s, const unsigned char *b) { secp256k1_scalar_set_b32(s, b, NULL); }

/* ---------- field (src/field_5x52_impl.h, src/field_impl.h) ---------- */
REF_EXPORT void ref_fe_mul(unsigned char *r, const unsigned char *a, const unsigned char *b) {
    secp256k1_fe x, y; ref_fe_from_b32(&x, a); ref_fe_from_b32(&y, b); secp256k1_fe_mul(&x, &x, &y); ref_fe_to_b32(r, &x);
}
REF_EXPORT void ref_fe_sqr(unsigned char *r, const unsigned char *a) {
    secp256k1_fe x; ref_fe_from_b32(&x, a); secp256k1_fe_sqr(&x, &x); ref_fe_to_b32(r, &x);
}
REF_EXPORT void ref_fe_add(unsigned char *r, const unsigned char *a, const unsigned char *b) {
    secp256k1_fe x, y; ref_fe_from_b32(&x, a); ref_fe_from_b32(&y, b); secp256k1_fe_add(&x, &y); ref_fe_to_b32(r, &x);
}
REF_EXPORT void ref_fe_negate(unsigned char *r, const unsigned char *a) {
    secp256k1_fe x, y; ref_fe_from_b32(&x, a); secp256k1_fe_negate(&y, &x, 1); ref_fe_to_b32(r, &y);
}
REF_EXPORT void ref_fe_inv(unsigned char *r, const unsigned char *a) {
    secp256k1_fe x; ref_fe_from_b32(&x, a); secp256k1_fe_normalize_var(&x); secp256k1_fe_inv_var(&x, &x); ref_fe_to_b32(r, &x);
}
/* returns 1 iff a is a square; r = a^((p+1)/4) either way (src/field_impl.h:37-146) */
REF_EXPORT int ref_fe_sqrt(unsigned char *r, const unsigned char *a) {
    secp256k1_fe x, y; int ret; ref_fe_from_b32(&x, a); ret = secp256k1_fe_sqrt(&y, &x); ref_fe_to_b32(r, &y); return ret;
}
REF_EXPORT int ref_fe_is_square(const unsigned char *a) {
    secp256k1_fe x; ref_fe_from_b32(&x, a); secp256k1_fe_normalize_var(&x); return secp256k1_fe_is_square_var(&x);
}

/* ---------- scalar (src/scalar_4x64_impl.h, src/scalar_impl.h) ---------- */
REF_EXPORT int ref_scalar_set_b32(unsigned char *r, const unsigned char *a) {
    secp256k1_scalar s; int o; secp256k1_scalar_set_b32(&s, a, &o); secp256k1_scalar_get_b32(r, &s); return o;
}
REF_EXPORT void ref_scalar_mul(unsigned char *r, const unsigned char *a, const unsigned char *b) {
    secp256k1_scalar x, y; ref_scalar_from_b32(&x, a); ref_scalar_from_b32(&y, b); secp256k1_scalar_mul(&x, &x, &y); secp256k1_scalar_get_b32(r, &x);
}
REF_EXPORT void ref_scalar_add(unsigned char *r, const unsigned char *a, const unsigned char *b) {
    secp256k1_scalar x, y; ref_scalar_from_b32(&x, a); ref_scalar_from_b32(&y, b); secp256k1_scalar_add(&x, &x, &y); secp256k1_scalar_get_b32(r, &x);
}
REF_EXPORT void ref_scalar_negate(unsigned char *r, const unsigned char *a) {
    secp256k1_scalar x; ref_scalar_from_b32(&x, a); secp256k1_scalar_negate(&x, &x); secp256k1_scalar_get_b32(r, &x);
}
REF_EXPORT void ref_scalar_inverse(unsigned char *r, const unsigned char *a) {
    secp256k1_scalar x; ref_scalar_from_b32(&x, a); secp256k1_scalar_inverse_var(&x, &x); secp256k1_scalar_get_b32(r, &x);
}
/* GLV split k = r1 + lambda*r2 (src/scalar_impl.h:142-180) */
REF_EXPORT void ref_scalar_split_lambda(unsigned char *r1, unsigned char *r2, const unsigned char *k) {
    secp256k1_scalar a, b, x; ref_scalar_from_b32(&x, k); secp256k1_scalar_split_lambda(&a, &b, &x);
    secp256k1_scalar_get_b32(r1, &a); secp256k1_scalar_get_b32(r2, &b);
}

/* ---------- group (src/group_impl.h) : affine in, affine out ---------- */
/* r = a + b via gej_add_ge_var (group_impl.h:598-659); returns infinity flag */
REF_EXPORT int ref_ge_add(unsigned char *r64, const unsigned char *a64, int ainf, const unsigned char *b64, int binf) {
    secp256k1_ge a, b; secp256k1_gej j;
    ref_ge_from_b64(&a, a64, ainf); ref_ge_from_b64(&b, b64, binf);
    secp256k1_gej_set_ge(&j, &a);
    secp256k1_gej_add_ge_var(&j, &j, &b, NULL);
    return ref_gej_to_b64(r64, &j);
}
REF_EXPORT int ref_ge_double(unsigned char *r64, const unsigned char *a64, int ainf) {
    secp256k1_ge a; secp256k1_gej j;
    ref_ge_from_b64(&a, a64, ainf); secp256k1_gej_set_ge(&j, &a);
    secp256k1_gej_double_var(&j, &j, NULL);
    return ref_gej_to_b64(r64, &j);
}
/* lift x with the square y (group_impl.h:347-355); returns validity, y always written */
REF_EXPORT int ref_ge_set_xquad(unsigned char *r64, const unsigned char *x32) {
    secp256k1_fe x; secp256k1_ge g; int ret;
    ref_fe_from_b32(&x, x32); ret = secp256k1_ge_set_xquad(&g, &x);
    ref_fe_to_b32(r64, &g.x); ref_fe_to_b32(r64 + 32, &g.y);
    return ret;
}
REF_EXPORT int ref_pubkey_parse33(unsigned char *r64, const unsigned char *in33) {
    secp256k1_ge g; if (!secp256k1_eckey_pubkey_parse(&g, in33, 33)) return 0;
    ref_fe_to_b32(r64, &g.x); ref_fe_to_b32(r64 + 32, &g.y); return 1;
}

/* ---------- ecmult: r = na*A + ng*G (src/ecmult_impl.h:365-375) ---------- */
REF_EXPORT int ref_ecmult(unsigned char *r64, const unsigned char *a64, int ainf, const unsigned char *na32, const unsigned char *ng32) {
    secp256k1_ge a; secp256k1_gej aj, rj; secp256k1_scalar na, ng;
    ref_ge_from_b64(&a, a64, ainf); secp256k1_gej_set_ge(&aj, &a);
    ref_scalar_from_b32(&na, na32);
    if (ng32) ref_scalar_from_b32(&ng, ng32);
    secp256k1_ecmult(&rj, &aj, &na, ng32 ? &ng : NULL);
    return ref_gej_to_b64(r64, &rj);
}
/* n independent double-mults; infs may be NULL. Returns nothing; out_inf[i] set. */
REF_EXPORT void ref_ecmult_batch(unsigned char *r64, int *out_inf, const unsigned char *a64, const unsigned char *ainf,
                                 const unsigned char *na32, const unsigned char *ng32, size_t n) {
    size_t i;
    for (i = 0; i < n; i++) {
        out_inf[i] = ref_ecmult(r64 + 64 * i, a64 + 64 * i, ainf ? ainf[i] : 0, na32 + 32 * i, ng32 ? ng32 + 32 * i : NULL);
    }
}

/* ---------- ecmult_multi_var (src/ecmult_impl.h:823-867) ---------- */
typedef struct { const unsigned char *sc; const unsigned char *pt; const unsigned char *inf; } ref_msm_cbdata;
static int ref_msm_cb(secp256k1_scalar *sc, secp256k1_ge *pt, size_t idx, void *data) {
    ref_msm_cbdata *d = (ref_msm_cbdata *)data;
    ref_scalar_from_b32(sc, d->sc + 32 * idx);
    ref_ge_from_b64(pt, d->pt + 64 * idx, d->inf ? d->inf[idx] : 0);
    return 1;
}
/* algo: 0 = ecmult_multi_var dispatcher, 1 = strauss_batch_single, 2 = pippenger_batch_single, 3 = simple (scratch NULL).
 * returns -1 on failure, else infinity flag of the result. */
REF_EXPORT int ref_ecmult_multi(unsigned char *r64, const unsigned char *g_sc32, const unsigned char *sc32, const unsigned char *pt64,
                                const unsigned char *inf, size_t n, int algo) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_scratch *scratch = NULL;
    secp256k1_scalar g; secp256k1_gej rj; ref_msm_cbdata d; int ok;
    size_t bucket_window = secp256k1_pippenger_bucket_window(n ? n : 1);
    size_t sz = secp256k1_pippenger_scratch_size(n ? n : 1, bucket_window) + secp256k1_strauss_scratch_size(n ? n : 1) + 4096;
    d.sc = sc32; d.pt = pt64; d.inf = inf;
    if (g_sc32) ref_scalar_from_b32(&g, g_sc32);
    if (algo != 3) scratch = secp256k1_scratch_space_create(ctx, sz);
    if (algo == 1) ok = secp256k1_ecmult_strauss_batch_single(&ctx->error_callback, scratch, &rj, g_sc32 ? &g : NULL, ref_msm_cb, &d, n);
    else if (algo == 2) ok = secp256k1_ecmult_pippenger_batch_single(&ctx->error_callback, scratch, &rj, g_sc32 ? &g : NULL, ref_msm_cb, &d, n);
    else ok = secp256k1_ecmult_multi_var(&ctx->error_callback, scratch, &rj, g_sc32 ? &g : NULL, ref_msm_cb, &d, n);
    if (scratch) secp256k1_scratch_space_destroy(ctx, scratch);
    secp256k1_context_destroy(ctx);
    if (!ok) return -1;
    return ref_gej_to_b64(r64, &rj);
}
REF_EXPORT size_t ref_pippenger_bucket_window(size_t n) { return (size_t)secp256k1_pippenger_bucket_window(n); }

/* ---------- BP++ norm argument (src/modules/bppp/bppp_norm_product_impl.h) ---------- */
/* transcript state is passed as the reference's own secp256k1_sha256 object bytes (src/hash.h) */
REF_EXPORT size_t ref_sha256_state_size(void) { return sizeof(secp256k1_sha256); }
REF_EXPORT void ref_sha256_state_from_prefix(unsigned char *state, const unsigned char *prefix, size_t len) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_sha256 sha; secp256k1_sha256_initialize(&sha);
    if (len) secp256k1_sha256_write(secp256k1_get_hash_context(ctx), &sha, prefix, len);
    memcpy(state, &sha, sizeof(sha));
    secp256k1_context_destroy(ctx);
}
/* the test-side "commit to initial data" transcript (modules/bppp/tests_impl.h:273-306), restated with public pieces */
REF_EXPORT void ref_bppp_transcript_init(unsigned char *state, const unsigned char *rho32, const unsigned char *gens33, size_t n_gens,
                                         size_t g_len, const unsigned char *c_vec32, size_t c_len, const unsigned char *commit33) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    const secp256k1_hash_ctx *hc = secp256k1_get_hash_context(ctx);
    secp256k1_sha256 t; unsigned char le[8]; size_t i;
    secp256k1_bppp_sha256_tagged_commitment_init(&t);
    secp256k1_sha256_write(hc, &t, commit33, 33);
    secp256k1_sha256_write(hc, &t, rho32, 32);
    secp256k1_bppp_le64(le, g_len); secp256k1_sha256_write(hc, &t, le, 8);
    secp256k1_bppp_le64(le, n_gens); secp256k1_sha256_write(hc, &t, le, 8);
    for (i = 0; i < n_gens; i++) secp256k1_sha256_write(hc, &t, gens33 + 33 * i, 33);
    secp256k1_bppp_le64(le, c_len); secp256k1_sha256_write(hc, &t, le, 8);
    for (i = 0; i < c_len; i++) secp256k1_sha256_write(hc, &t, c_vec32 + 32 * i, 32);
    memcpy(state, &t, sizeof(t));
    secp256k1_context_destroy(ctx);
}
static secp256k1_bppp_generators *ref_gens_parse(const unsigned char *gens33, size_t n) {
    secp256k1_bppp_generators *g = (secp256k1_bppp_generators *)malloc(sizeof(*g)); size_t i;
    g->n = n; g->gens = (secp256k1_ge *)malloc((n ? n : 1) * sizeof(secp256k1_ge));
    for (i = 0; i < n; i++) {
        if (!secp256k1_eckey_pubkey_parse(&g->gens[i], gens33 + 33 * i, 33)) { free(g->gens); free(g); return NULL; }
    }
    return g;
}
/* deterministic generators (modules/bppp/main_impl.h:18-48), serialised 33 bytes each */
REF_EXPORT int ref_bppp_generators(unsigned char *out33, size_t n) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_bppp_generators *g = secp256k1_bppp_generators_create(ctx, n); size_t i;
    if (!g) { secp256k1_context_destroy(ctx); return 0; }
    for (i = 0; i < n; i++) secp256k1_eckey_pubkey_serialize33(&g->gens[i], out33 + 33 * i);
    secp256k1_bppp_generators_destroy(ctx, g); secp256k1_context_destroy(ctx);
    return 1;
}
REF_EXPORT int ref_bppp_norm_verify(const unsigned char *proof, size_t plen, const unsigned char *transcript_state, const unsigned char *rho32,
                                    const unsigned char *gens33, size_t n_gens, size_t g_len, const unsigned char *c_vec32, size_t c_len,
                                    const unsigned char *commit33) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_scratch *scratch = secp256k1_scratch_space_create(ctx, 4 * 1000 * 1000);
    secp256k1_bppp_generators *g = ref_gens_parse(gens33, n_gens);
    secp256k1_sha256 t; secp256k1_scalar rho; secp256k1_scalar *c; secp256k1_ge commit; size_t i; int ret = 0, overflow;
    if (g) {
        memcpy(&t, transcript_state, sizeof(t));
        secp256k1_scalar_set_b32(&rho, rho32, &overflow);
        c = (secp256k1_scalar *)malloc((c_len ? c_len : 1) * sizeof(*c));
        for (i = 0; i < c_len; i++) secp256k1_scalar_set_b32(&c[i], c_vec32 + 32 * i, &overflow);
        if (secp256k1_ge_parse_ext(&commit, commit33)) {
            ret = secp256k1_bppp_rangeproof_norm_product_verify(ctx, scratch, proof, plen, &t, &rho, g, g_len, c, c_len, &commit);
        }
        free(c); free(g->gens); free(g);
    }
    secp256k1_scratch_space_destroy(ctx, scratch); secp256k1_context_destroy(ctx);
    return ret;
}
/* commit (bppp_norm_product_impl.h:105-151) then prove (:223-367); used only to synthesise test/bench inputs */
REF_EXPORT int ref_bppp_commit(unsigned char *commit33, const unsigned char *gens33, size_t n_gens, const unsigned char *n_vec32, size_t g_len,
                               const unsigned char *l_vec32, const unsigned char *c_vec32, size_t h_len, const unsigned char *mu32) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_scratch *scratch = secp256k1_scratch_space_create(ctx, 4 * 1000 * 1000);
    secp256k1_bppp_generators *g = ref_gens_parse(gens33, n_gens);
    secp256k1_scalar *nv = malloc(g_len * 32), *lv = malloc(h_len * 32), *cv = malloc(h_len * 32), mu; secp256k1_ge commit; size_t i; int ret = 0;
    if (g) {
        for (i = 0; i < g_len; i++) ref_scalar_from_b32(&nv[i], n_vec32 + 32 * i);
        for (i = 0; i < h_len; i++) { ref_scalar_from_b32(&lv[i], l_vec32 + 32 * i); ref_scalar_from_b32(&cv[i], c_vec32 + 32 * i); }
        ref_scalar_from_b32(&mu, mu32);
        ret = secp256k1_bppp_commit(ctx, scratch, &commit, g, nv, g_len, lv, h_len, cv, h_len, &mu);
        if (ret) secp256k1_ge_serialize_ext(commit33, &commit);
        free(g->gens); free(g);
    }
    free(nv); free(lv); free(cv);
    secp256k1_scratch_space_destroy(ctx, scratch); secp256k1_context_destroy(ctx);
    return ret;
}
REF_EXPORT int ref_bppp_norm_prove(unsigned char *proof, size_t *plen, const unsigned char *transcript_state, const unsigned char *rho32,
                                   const unsigned char *gens33, size_t n_gens, const unsigned char *n_vec32, size_t g_len,
                                   const unsigned char *l_vec32, const unsigned char *c_vec32, size_t h_len) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_scratch *scratch = secp256k1_scratch_space_create(ctx, 4 * 1000 * 1000);
    secp256k1_bppp_generators *g = ref_gens_parse(gens33, n_gens);
    secp256k1_scalar *nv = malloc(g_len * 32), *lv = malloc(h_len * 32), *cv = malloc(h_len * 32), rho; secp256k1_sha256 t; size_t i; int ret = 0;
    if (g) {
        memcpy(&t, transcript_state, sizeof(t));
        for (i = 0; i < g_len; i++) ref_scalar_from_b32(&nv[i], n_vec32 + 32 * i);
        for (i = 0; i < h_len; i++) { ref_scalar_from_b32(&lv[i], l_vec32 + 32 * i); ref_scalar_from_b32(&cv[i], c_vec32 + 32 * i); }
        ref_scalar_from_b32(&rho, rho32);
        ret = secp256k1_bppp_rangeproof_norm_product_prove(ctx, scratch, proof, plen, &t, &rho, g->gens, n_gens, nv, g_len, lv, h_len, cv, h_len);
        free(g->gens); free(g);
    }
    free(nv); free(lv); free(cv);
    secp256k1_scratch_space_destroy(ctx, scratch); secp256k1_context_destroy(ctx);
    return ret;
}

/* ---------- batch loops in C (CPU baseline timing; OpenMP when built with -fopenmp) ---------- */
/* proofs are n records of `stride` bytes, length plens[i]; commits n x 33 (parsed internally), one generator (64B opaque) per item */
REF_EXPORT void ref_rangeproof_verify_many(int *results, uint64_t *min_v, uint64_t *max_v, const unsigned char *commits33, const unsigned char *proofs,
                                           size_t stride, const size_t *plens, const unsigned char *gens64, size_t n, int threads) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    long i;
    (void)threads;
#ifdef _OPENMP
    #pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
#endif
    for (i = 0; i < (long)n; i++) {
        secp256k1_pedersen_commitment c; secp256k1_generator g;
        results[i] = 0;
        memcpy(g.data, gens64 + 64 * i, 64);
        if (secp256k1_pedersen_commitment_parse(ctx, &c, commits33 + 33 * i)) {
            results[i] = secp256k1_rangeproof_verify(ctx, &min_v[i], &max_v[i], &c, proofs + stride * i, plens[i], NULL, 0, &g);
        }
    }
    secp256k1_context_destroy(ctx);
}
/* synthesise n proofs as src/bench_rangeproof.c:26-36 does (nonce = commit bytes, no message), with caller-chosen min_bits/exp/min_value */
REF_EXPORT int ref_rangeproof_make_many(unsigned char *commits33, unsigned char *proofs, size_t stride, size_t *plens, const unsigned char *blinds32,
                                        const uint64_t *values, const unsigned char *gens64, uint64_t min_value, int exp, int min_bits, size_t n, int threads) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    long i; int ok = 1;
    (void)threads;
#ifdef _OPENMP
    #pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
#endif
    for (i = 0; i < (long)n; i++) {
        secp256k1_pedersen_commitment c; secp256k1_generator g; size_t len = stride; unsigned char ser[33];
        memcpy(g.data, gens64 + 64 * i, 64);
        if (!secp256k1_pedersen_commit(ctx, &c, blinds32 + 32 * i, values[i], &g)) { ok = 0; continue; }
        secp256k1_pedersen_commitment_serialize(ctx, ser, &c);
        memcpy(commits33 + 33 * i, ser, 33);
        if (!secp256k1_rangeproof_sign(ctx, proofs + stride * i, &len, min_value, &c, blinds32 + 32 * i, ser, exp, min_bits, values[i], NULL, 0, NULL, 0, &g)) { ok = 0; len = 0; }
        plens[i] = len;
    }
    secp256k1_context_destroy(ctx);
    return ok;
}
/* the same two loops with an extra_commit per item (extra: n records of `estride` bytes, length elens[i]; Elements commits to the output's script there) */
REF_EXPORT void ref_rangeproof_verify_many_extra(int *results, uint64_t *min_v, uint64_t *max_v, const unsigned char *commits33, const unsigned char *proofs,
                                                 size_t stride, const size_t *plens, const unsigned char *extra, size_t estride, const size_t *elens,
                                                 const unsigned char *gens64, size_t n, int threads) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    long i;
    (void)threads;
#ifdef _OPENMP
    #pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
#endif
    for (i = 0; i < (long)n; i++) {
        secp256k1_pedersen_commitment c; secp256k1_generator g;
        results[i] = 0;
        memcpy(g.data, gens64 + 64 * i, 64);
        if (secp256k1_pedersen_commitment_parse(ctx, &c, commits33 + 33 * i)) {
            results[i] = secp256k1_rangeproof_verify(ctx, &min_v[i], &max_v[i], &c, proofs + stride * i, plens[i], elens[i] ? extra + estride * i : NULL, elens[i], &g);
        }
    }
    secp256k1_context_destroy(ctx);
}
REF_EXPORT int ref_rangeproof_make_many_extra(unsigned char *commits33, unsigned char *proofs, size_t stride, size_t *plens, const unsigned char *blinds32,
                                              const uint64_t *values, const unsigned char *gens64, const unsigned char *extra, size_t estride, const size_t *elens,
                                              uint64_t min_value, int exp, int min_bits, size_t n, int threads) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    long i; int ok = 1;
    (void)threads;
#ifdef _OPENMP
    #pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
#endif
    for (i = 0; i < (long)n; i++) {
        secp256k1_pedersen_commitment c; secp256k1_generator g; size_t len = stride; unsigned char ser[33];
        memcpy(g.data, gens64 + 64 * i, 64);
        if (!secp256k1_pedersen_commit(ctx, &c, blinds32 + 32 * i, values[i], &g)) { ok = 0; continue; }
        secp256k1_pedersen_commitment_serialize(ctx, ser, &c);
        memcpy(commits33 + 33 * i, ser, 33);
        if (!secp256k1_rangeproof_sign(ctx, proofs + stride * i, &len, min_value, &c, blinds32 + 32 * i, ser, exp, min_bits, values[i], NULL, 0,
                                       elens[i] ? extra + estride * i : NULL, elens[i], &g)) { ok = 0; len = 0; }
        plens[i] = len;
    }
    secp256k1_context_destroy(ctx);
    return ok;
}
REF_EXPORT void ref_schnorrsig_verify_many(int *results, const unsigned char *sigs64, const unsigned char *msgs, size_t msglen, const unsigned char *pks32, size_t n, int threads) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    long i;
    (void)threads;
#ifdef _OPENMP
    #pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 16)
#endif
    for (i = 0; i < (long)n; i++) {
        secp256k1_xonly_pubkey pk;
        results[i] = 0;
        if (secp256k1_xonly_pubkey_parse(ctx, &pk, pks32 + 32 * i)) {
            results[i] = secp256k1_schnorrsig_verify(ctx, sigs64 + 64 * i, msgs + msglen * i, msglen, &pk);
        }
    }
    secp256k1_context_destroy(ctx);
}
REF_EXPORT int ref_schnorrsig_make_many(unsigned char *sigs64, unsigned char *pks32, const unsigned char *seckeys32, const unsigned char *msgs32, size_t n, int threads) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    long i; int ok = 1;
    (void)threads;
#ifdef _OPENMP
    #pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 16)
#endif
    for (i = 0; i < (long)n; i++) {
        secp256k1_keypair kp; secp256k1_xonly_pubkey pk;
        if (!secp256k1_keypair_create(ctx, &kp, seckeys32 + 32 * i)) { ok = 0; continue; }
        secp256k1_keypair_xonly_pub(ctx, &pk, NULL, &kp);
        secp256k1_xonly_pubkey_serialize(ctx, pks32 + 32 * i, &pk);
        if (!secp256k1_schnorrsig_sign32(ctx, sigs64 + 64 * i, msgs32 + 32 * i, &kp, NULL)) ok = 0;
    }
    secp256k1_context_destroy(ctx);
    return ok;
}

/* ---------- surjection proofs (src/modules/surjection/main_impl.h) ---------- */
#include "../include/secp256k1_surjectionproof.h"
/* parse + verify on the wire format; tags are 64-byte secp256k1_generator objects */
REF_EXPORT int ref_surjectionproof_verify_ser(const unsigned char *proof_ser, size_t len, const unsigned char *in_tags64, size_t n_tags, const unsigned char *out_tag64) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_surjectionproof proof; secp256k1_generator *tags = (secp256k1_generator *)malloc((n_tags ? n_tags : 1) * sizeof(secp256k1_generator)), out; size_t i; int ret = 0;
    for (i = 0; i < n_tags; i++) memcpy(tags[i].data, in_tags64 + 64 * i, 64);
    memcpy(out.data, out_tag64, 64);
    if (secp256k1_surjectionproof_parse(ctx, &proof, proof_ser, len)) ret = secp256k1_surjectionproof_verify(ctx, &proof, tags, n_tags, &out);
    free(tags); secp256k1_context_destroy(ctx);
    return ret;
}
/* synthesise one proof the way the reference's tests do (tests_impl.h:300-340): n_inputs blinded asset tags, the output
 * re-blinds input `which`; returns serialised proof length (0 on failure) and the ephemeral tags (64 bytes each) */
REF_EXPORT size_t ref_surjection_make(unsigned char *proof_ser, size_t max_len, unsigned char *in_tags64, unsigned char *out_tag64,
                                      const unsigned char *seed32, size_t n_inputs, size_t n_used, size_t which) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_fixed_asset_tag *fixed = (secp256k1_fixed_asset_tag *)malloc(n_inputs * sizeof(*fixed)), fixed_out;
    secp256k1_generator *eph = (secp256k1_generator *)malloc(n_inputs * sizeof(*eph)), eph_out;
    unsigned char *blinds = (unsigned char *)malloc(32 * n_inputs), out_blind[32], h[32];
    secp256k1_surjectionproof proof; size_t i, input_index, len = max_len; int ok = 1;
    for (i = 0; i < n_inputs; i++) {
        secp256k1_sha256 s; unsigned char t[40]; memcpy(t, seed32, 32); t[32] = (unsigned char)i; t[33] = (unsigned char)(i >> 8); t[34] = 1;
        secp256k1_sha256_initialize(&s); secp256k1_sha256_write(secp256k1_get_hash_context(ctx), &s, t, 35); secp256k1_sha256_finalize(secp256k1_get_hash_context(ctx), &s, fixed[i].data);
        t[34] = 2;
        secp256k1_sha256_initialize(&s); secp256k1_sha256_write(secp256k1_get_hash_context(ctx), &s, t, 35); secp256k1_sha256_finalize(secp256k1_get_hash_context(ctx), &s, blinds + 32 * i);
        blinds[32 * i] &= 0x7F;
        ok &= secp256k1_generator_generate_blinded(ctx, &eph[i], fixed[i].data, blinds + 32 * i);
    }
    { secp256k1_sha256 s; unsigned char t[40]; memcpy(t, seed32, 32); t[32] = 0xFF; t[33] = 0xFF; t[34] = 3;
      secp256k1_sha256_initialize(&s); secp256k1_sha256_write(secp256k1_get_hash_context(ctx), &s, t, 35); secp256k1_sha256_finalize(secp256k1_get_hash_context(ctx), &s, out_blind); out_blind[0] &= 0x7F; }
    fixed_out = fixed[which];
    ok &= secp256k1_generator_generate_blinded(ctx, &eph_out, fixed_out.data, out_blind);
    memcpy(h, seed32, 32);
    ok = ok && secp256k1_surjectionproof_initialize(ctx, &proof, &input_index, fixed, n_inputs, n_used, &fixed_out, 1000000, h);
    ok = ok && secp256k1_surjectionproof_generate(ctx, &proof, eph, n_inputs, &eph_out, input_index, blinds + 32 * input_index, out_blind);
    ok = ok && secp256k1_surjectionproof_serialize(ctx, proof_ser, &len, &proof);
    for (i = 0; i < n_inputs; i++) memcpy(in_tags64 + 64 * i, eph[i].data, 64);
    memcpy(out_tag64, eph_out.data, 64);
    free(fixed); free(eph); free(blinds); secp256k1_context_destroy(ctx);
    return ok ? len : 0;
}

/* ---- half-aggregated Schnorr signatures (src/modules/schnorrsig_halfagg/main_impl.h) --------------------------------
 * keys travel as 32-byte x-only serialisations; a key that does not parse makes both calls return -1. */
REF_EXPORT int ref_halfagg_aggregate(unsigned char *aggsig, size_t *aggsig_len, const unsigned char *pks32, const unsigned char *msgs32,
                                     const unsigned char *sigs64, size_t n) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_xonly_pubkey *pk = (secp256k1_xonly_pubkey*)malloc(sizeof(secp256k1_xonly_pubkey) * (n ? n : 1));
    size_t i; int r = 1;
    for (i = 0; i < n; i++) if (!secp256k1_xonly_pubkey_parse(ctx, &pk[i], pks32 + 32 * i)) r = -1;
    if (r == 1) r = secp256k1_schnorrsig_aggregate(ctx, aggsig, aggsig_len, pk, msgs32, sigs64, n);
    free(pk);
    secp256k1_context_destroy(ctx);
    return r;
}
REF_EXPORT int ref_halfagg_verify(const unsigned char *pks32, const unsigned char *msgs32, size_t n, const unsigned char *aggsig, size_t aggsig_len) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_xonly_pubkey *pk = (secp256k1_xonly_pubkey*)malloc(sizeof(secp256k1_xonly_pubkey) * (n ? n : 1));
    size_t i; int r = 1;
    for (i = 0; i < n; i++) if (!secp256k1_xonly_pubkey_parse(ctx, &pk[i], pks32 + 32 * i)) r = -1;
    if (r == 1) r = secp256k1_schnorrsig_aggverify(ctx, n ? pk : NULL, n ? msgs32 : NULL, n, aggsig, aggsig_len);
    free(pk);
    secp256k1_context_destroy(ctx);
    return r;
}
/* the 64-byte in-memory secp256k1_xonly_pubkey objects of serialised keys (0 if one does not parse) */
REF_EXPORT int ref_xonly_objects(unsigned char *out64, const unsigned char *pks32, size_t n) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    size_t i; int ok = 1;
    for (i = 0; i < n; i++) {
        secp256k1_xonly_pubkey pk;
        if (!secp256k1_xonly_pubkey_parse(ctx, &pk, pks32 + 32 * i)) { ok = 0; memset(&pk, 0, sizeof(pk)); }
        memcpy(out64 + 64 * i, &pk, 64);
    }
    secp256k1_context_destroy(ctx);
    return ok;
}

/* ---- Pedersen commitments and tallies (src/modules/generator/main_impl.h:275-396) ------------------------------------ */
REF_EXPORT int ref_pedersen_commit_many(unsigned char *out33, const unsigned char *blinds32, const uint64_t *values, const unsigned char *gen64, size_t n) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_generator gen; size_t i; int ok = 1;
    memcpy(&gen, gen64, 64);
    for (i = 0; i < n; i++) {
        secp256k1_pedersen_commitment c;
        if (!secp256k1_pedersen_commit(ctx, &c, blinds32 + 32 * i, values[i], &gen)) { ok = 0; memset(out33 + 33 * i, 0, 33); continue; }
        secp256k1_pedersen_commitment_serialize(ctx, out33 + 33 * i, &c);
    }
    secp256k1_context_destroy(ctx);
    return ok;
}
/* blind_out = sum of the first npositive blinds - sum of the others */
REF_EXPORT int ref_pedersen_blind_sum(unsigned char *out32, const unsigned char *blinds32, size_t n, size_t npositive) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    const unsigned char **ptr = (const unsigned char**)malloc(sizeof(*ptr) * (n ? n : 1));
    size_t i; int r;
    for (i = 0; i < n; i++) ptr[i] = blinds32 + 32 * i;
    r = secp256k1_pedersen_blind_sum(ctx, out32, ptr, n, npositive);
    free(ptr);
    secp256k1_context_destroy(ctx);
    return r;
}
/* tally t = commitments [tally_off[t], tally_off[t+1]) of commits33, the first n_pos[t] on the positive side; -1 if one does not parse */
REF_EXPORT void ref_pedersen_verify_tally_many(int *results, const unsigned char *commits33, const uint64_t *tally_off, const uint64_t *n_pos, size_t n_tallies) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    size_t t;
    for (t = 0; t < n_tallies; t++) {
        const size_t a = (size_t)tally_off[t], cnt = (size_t)(tally_off[t + 1] - tally_off[t]), np = (size_t)n_pos[t];
        secp256k1_pedersen_commitment *c = (secp256k1_pedersen_commitment*)malloc(sizeof(*c) * (cnt ? cnt : 1));
        const secp256k1_pedersen_commitment **p = (const secp256k1_pedersen_commitment**)malloc(sizeof(*p) * (cnt ? cnt : 1));
        size_t i; int ok = 1;
        for (i = 0; i < cnt; i++) { p[i] = &c[i]; if (!secp256k1_pedersen_commitment_parse(ctx, &c[i], commits33 + 33 * (a + i))) ok = 0; }
        results[t] = ok ? secp256k1_pedersen_verify_tally(ctx, np ? p : NULL, np, cnt - np ? p + np : NULL, cnt - np) : -1;
        free(c); free(p);
    }
    secp256k1_context_destroy(ctx);
}

/* ---- rangeproof rewind (src/modules/rangeproof/main_impl.h:31-52, rangeproof_impl.h:364-485) -------------------------- */
/* proofs with a caller-chosen nonce and embedded message (message i = msgs + msg_len*i, msg_len <= 4096; may be 0) */
REF_EXPORT int ref_rangeproof_make_many_msg(unsigned char *commits33, unsigned char *proofs, size_t stride, size_t *plens, const unsigned char *blinds32,
                                            const uint64_t *values, const unsigned char *gens64, const unsigned char *nonces32, const unsigned char *msgs,
                                            size_t msg_len, uint64_t min_value, int exp, int min_bits, size_t n, int threads) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    long i; int ok = 1;
    (void)threads;
#ifdef _OPENMP
    #pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
#endif
    for (i = 0; i < (long)n; i++) {
        secp256k1_pedersen_commitment c; secp256k1_generator g; size_t len = stride;
        memcpy(g.data, gens64 + 64 * i, 64);
        if (!secp256k1_pedersen_commit(ctx, &c, blinds32 + 32 * i, values[i], &g)) { ok = 0; continue; }
        secp256k1_pedersen_commitment_serialize(ctx, commits33 + 33 * i, &c);
        if (!secp256k1_rangeproof_sign(ctx, proofs + stride * i, &len, min_value, &c, blinds32 + 32 * i, nonces32 + 32 * i, exp, min_bits, values[i],
                                       msg_len ? msgs + msg_len * i : NULL, msg_len, NULL, 0, &g)) { ok = 0; len = 0; }
        plens[i] = len;
    }
    secp256k1_context_destroy(ctx);
    return ok;
}
/* outlens: in = capacity of each message buffer (<= msg_stride), out = recovered length; msg_out may be NULL (then outlens is ignored) */
REF_EXPORT void ref_rangeproof_rewind_many(int *results, unsigned char *blind_out, uint64_t *value_out, unsigned char *msg_out, uint64_t *outlens, size_t msg_stride,
                                           const unsigned char *nonces32, uint64_t *min_v, uint64_t *max_v, const unsigned char *commits33, const unsigned char *proofs,
                                           size_t stride, const size_t *plens, const unsigned char *gens64, size_t n, int threads) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    long i;
    (void)threads;
#ifdef _OPENMP
    #pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
#endif
    for (i = 0; i < (long)n; i++) {
        secp256k1_pedersen_commitment c; secp256k1_generator g; size_t ol = msg_out ? (size_t)outlens[i] : 0;
        results[i] = 0;
        memcpy(g.data, gens64 + 64 * i, 64);
        if (secp256k1_pedersen_commitment_parse(ctx, &c, commits33 + 33 * i)) {
            results[i] = secp256k1_rangeproof_rewind(ctx, blind_out + 32 * i, &value_out[i], msg_out ? msg_out + msg_stride * i : NULL, msg_out ? &ol : NULL,
                                                     nonces32 + 32 * i, &min_v[i], &max_v[i], &c, proofs + stride * i, plens[i], NULL, 0, &g);
            if (msg_out) outlens[i] = ol;
        }
    }
    secp256k1_context_destroy(ctx);
}
