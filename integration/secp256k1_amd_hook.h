/* secp256k1_amd_hook.h -- the reference-side half of the MI355X drop-in boundary.
 *
 * What a secp256k1-zkp maintainer adds to the library (it is C89-compatible C and sees the library's internal types):
 * a context-independent function-pointer table modelled on the library's one existing pluggable seam, the SHA-256
 * compression hook (reference include/secp256k1.h:420-446), NULL => the existing CPU code runs.  Nothing hangs off
 * secp256k1_context (its layout is compared by secp256k1_context_eq, reference src/secp256k1.c:80-81).
 *
 * The table's members have exactly the C ABI of include/secp256k1_zkp_amd.h, so an application registers the engine with
 *
 *     s2k_engine *e = s2k_engine_create(0);
 *     s2k_engine_reserve(e, 16384);        (warm-up: the device's fixed-base tables are allocated and built NOW, so that a memory
 *                                           shortage shows here -- as a narrower table, s2k_engine_gtable_bits(e), or as a failure --
 *                                           and not as a latency spike or a CPU fallback inside the first verification)
 *     secp256k1_amd_backend b = {0};
 *     b.engine = e;
 *     b.rangeproof_verify_batch = (secp256k1_amd_rangeproof_verify_batch_fn)secp256k1_rangeproof_verify_batch;
 *     b.ecmult_multi            = (secp256k1_amd_ecmult_multi_fn)s2k_ecmult_multi;
 *     ...
 *     secp256k1_amd_set_backend(&b);
 *
 * Rules every adapter in secp256k1_amd_hook.c follows:
 *   - a backend call that returns 0 (engine-level failure) makes the adapter run the library's own CPU path for the
 *     whole batch: results are the reference's, a failed device never turns into a verdict;
 *   - argument checks and their illegal-callback behaviour are the library's (ARG_CHECK) and happen before any packing;
 *   - the caller owns every buffer; packing buffers are malloc'ed per call through checked_malloc (src/util.h:162-168)
 *     and freed before returning.
 *
 * This file is built and tested in the test tier only (oracle/Makefile target `hooked`, against /root/reference);
 * it is never linked into the product library.
 */
#ifndef SECP256K1_AMD_HOOK_H
#define SECP256K1_AMD_HOOK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* exported like the library's own API (the library is built with hidden visibility: include/secp256k1.h defines SECP256K1_API) */
#ifndef SECP256K1_AMD_API
# ifdef SECP256K1_API
#  define SECP256K1_AMD_API SECP256K1_API
# else
#  define SECP256K1_AMD_API
# endif
#endif

/* ABI of the engine entry points (include/secp256k1_zkp_amd.h); `engine` is the s2k_engine*. */
typedef int (*secp256k1_amd_rangeproof_verify_batch_fn)(void *engine, int32_t *results, uint64_t *min_value, uint64_t *max_value,
        const unsigned char *commits33, const unsigned char *proofs, const uint64_t *proof_off,
        const unsigned char *extra, const uint64_t *extra_off, const unsigned char *gens64, size_t n);
/* secp256k1_rangeproof_verify_batch_ptrs: the items as arrays of pointers to the library's own objects -- nothing is packed on this side */
typedef int (*secp256k1_amd_rangeproof_verify_batch_ptrs_fn)(void *engine, int32_t *results, uint64_t *min_value, uint64_t *max_value,
        const void *const *commit_objs, const unsigned char *const *proofs, const size_t *plens,
        const unsigned char *const *extra, const size_t *elens, const void *const *gen_objs, size_t n);
/* secp256k1_rangeproof_verify_batch_ptrs_submit / secp256k1_rangeproof_verify_batch_wait: the same, asynchronously (two batches in flight) */
typedef int (*secp256k1_amd_rangeproof_verify_batch_ptrs_submit_fn)(void *engine, uint64_t *ticket, int32_t *results, uint64_t *min_value, uint64_t *max_value,
        const void *const *commit_objs, const unsigned char *const *proofs, const size_t *plens,
        const unsigned char *const *extra, const size_t *elens, const void *const *gen_objs, size_t n);
typedef int (*secp256k1_amd_rangeproof_verify_batch_wait_fn)(void *engine, uint64_t ticket);
typedef int (*secp256k1_amd_ecmult_multi_fn)(void *engine, unsigned char *r_xy, int32_t *r_inf, const unsigned char *g_sc,
        const unsigned char *sc, const unsigned char *pt_xy, const unsigned char *pt_inf, size_t n);
typedef int (*secp256k1_amd_schnorrsig_verify_batch_fn)(void *engine, int32_t *results, const unsigned char *sigs, const unsigned char *msgs,
        size_t msglen, const unsigned char *pubkeys, int pk_format, size_t n);
typedef int (*secp256k1_amd_surjectionproof_verify_batch_fn)(void *engine, int32_t *results, const unsigned char *proofs, const uint64_t *proof_off,
        const unsigned char *input_tags64, const uint64_t *tag_off, const unsigned char *output_tags64, size_t n);
typedef int (*secp256k1_amd_pedersen_verify_tally_batch_fn)(void *engine, int32_t *results, const unsigned char *commits33,
        const uint64_t *tally_off, const uint64_t *n_pos, size_t n_tallies);

typedef int (*secp256k1_amd_schnorrsig_aggverify_fn)(void *engine, int32_t *result, const unsigned char *pubkeys, int pk_format,
                                                    const unsigned char *msgs32, size_t n, const unsigned char *aggsig, size_t aggsig_len);
typedef int (*secp256k1_amd_rangeproof_rewind_batch_fn)(void *engine, int32_t *results, unsigned char *blind_out, uint64_t *value_out, unsigned char *message_out,
                                                       uint64_t *outlen, size_t msg_stride, const unsigned char *nonces, uint64_t *min_value, uint64_t *max_value,
                                                       const unsigned char *commits33, const unsigned char *proofs, const uint64_t *proof_off,
                                                       const unsigned char *extra, const uint64_t *extra_off, const unsigned char *gens64, size_t n);
/* s2k_ecmult_batch and secp256k1_bppp_norm_product_verify_batch */
typedef int (*secp256k1_amd_ecmult_batch_fn)(void *engine, unsigned char *r_xy, int32_t *r_inf, const unsigned char *a_xy, const unsigned char *a_inf,
        const unsigned char *na, const unsigned char *ng, size_t n);
typedef int (*secp256k1_amd_bppp_norm_product_verify_batch_fn)(void *engine, int32_t *results, const unsigned char *proofs, size_t proof_len,
        const unsigned char *transcripts, const unsigned char *rho, const unsigned char *gens33, size_t n_gens, size_t g_len,
        const unsigned char *c_vec, size_t c_vec_len, const unsigned char *commits33, size_t n);
typedef struct secp256k1_amd_backend {
    void *engine;
    secp256k1_amd_rangeproof_verify_batch_fn rangeproof_verify_batch;            /* may be NULL: that call stays on the CPU */
    secp256k1_amd_ecmult_multi_fn ecmult_multi;
    secp256k1_amd_schnorrsig_verify_batch_fn schnorrsig_verify_batch;
    secp256k1_amd_surjectionproof_verify_batch_fn surjectionproof_verify_batch;
    secp256k1_amd_pedersen_verify_tally_batch_fn pedersen_verify_tally_batch;
    secp256k1_amd_schnorrsig_aggverify_fn schnorrsig_aggverify;                   /* secp256k1_schnorrsig_aggverify_amd */
    secp256k1_amd_rangeproof_rewind_batch_fn rangeproof_rewind_batch;             /* secp256k1_rangeproof_rewind_batch */
    secp256k1_amd_rangeproof_verify_batch_ptrs_fn rangeproof_verify_batch_ptrs;   /* preferred over rangeproof_verify_batch when set: no packing here */
    secp256k1_amd_ecmult_batch_fn ecmult_batch;                                   /* s2k_ecmult_batch */
    secp256k1_amd_bppp_norm_product_verify_batch_fn bppp_norm_product_verify_batch;   /* secp256k1_bppp_norm_product_verify_batch */
    secp256k1_amd_rangeproof_verify_batch_ptrs_submit_fn rangeproof_verify_batch_ptrs_submit;   /* both or neither: the asynchronous pair */
    secp256k1_amd_rangeproof_verify_batch_wait_fn rangeproof_verify_batch_wait;
} secp256k1_amd_backend;

/* Install (copy) a backend table; NULL restores the pure CPU library.  The table is published with ONE pointer store (release order) and
 * every adapter works on the table it read at its entry, so a verification that runs concurrently sees either the old or the new table,
 * never a mixture (include/secp256k1.h:42-52: API calls on const contexts may run concurrently).  Meant to be called at start-up, like
 * secp256k1_context_set_sha256_compression; two installs must be a verification's duration apart (two table slots).
 * Thread safety of the engine behind the table: one s2k_engine serialises its callers (a mutex per engine; their launches share its stream
 * and scratch), so concurrent verifier threads are safe and take turns; give every thread its own engine for parallel submission. */
SECP256K1_AMD_API void secp256k1_amd_set_backend(const secp256k1_amd_backend *backend);
/* secp256k1_ecmult_multi_var calls with fewer terms than this stay on the CPU (default 256: an engine round trip costs ~0.45-0.6 ms whatever
 * the size, the CPU ~3-6 us per term); 0 sends everything to the engine. */
SECP256K1_AMD_API void secp256k1_amd_set_msm_min_terms(size_t n);
/* Counters for tests / monitoring: batches served by the backend, batches that fell back to the CPU after a backend failure. */
SECP256K1_AMD_API void secp256k1_amd_stats(size_t *served, size_t *fell_back);

#ifdef __cplusplus
}
#endif
#endif
