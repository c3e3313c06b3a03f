/* route_api.h -- TEST INFRASTRUCTURE, second half of route_modules.h: inserted into the throw-away copy of the reference's src/tests.c
 * right after its `#include "secp256k1.c"`.  Every call the reference's tests make to a public verifier on the hot path
 *     secp256k1_rangeproof_verify / secp256k1_rangeproof_rewind      (include/secp256k1_rangeproof.h:70-130)
 *     secp256k1_schnorrsig_verify                                    (include/secp256k1_schnorrsig.h:178)
 *     secp256k1_schnorrsig_aggverify                                 (include/secp256k1_schnorrsig_halfagg.h:94)
 *     secp256k1_pedersen_verify_tally                                (include/secp256k1_generator.h:190)
 *     secp256k1_surjectionproof_verify                               (include/secp256k1_surjectionproof.h:256)
 * first runs the reference (its return value and outputs are what the test sees, its ARG_CHECKs fire as the test expects), then the
 * engine's form of the same call on the same arguments, and aborts the program on any difference in verdict or outputs.  Calls that
 * probe API misuse (a NULL the reference answers with its illegal-argument callback) are not repeated on the engine. */
#ifndef S2K_ROUTE_API_H
#define S2K_ROUTE_API_H

#ifdef ENABLE_MODULE_RANGEPROOF
static int s2k_rt_rangeproof_verify(const secp256k1_context *ctx, uint64_t *min_value, uint64_t *max_value, const secp256k1_pedersen_commitment *commit,
                                    const unsigned char *proof, size_t plen, const unsigned char *extra_commit, size_t extra_commit_len, const secp256k1_generator *gen) {
    const int legal = ctx != NULL && min_value != NULL && max_value != NULL && commit != NULL && proof != NULL && gen != NULL && (extra_commit != NULL || extra_commit_len == 0);
    const int ret = secp256k1_rangeproof_verify(ctx, min_value, max_value, commit, proof, plen, extra_commit, extra_commit_len, gen);
    s2k_rt.calls[S2K_RT_RANGEPROOF]++;
    if (legal && s2k_rt_on()) {
        /* what the reference writes into zeroed outputs (it leaves them alone when the header does not parse); the engine reports 0 there */
        uint64_t rmin = 0, rmax = 0, gmin = 0, gmax = 0;
        const int again = secp256k1_rangeproof_verify(ctx, &rmin, &rmax, commit, proof, plen, extra_commit, extra_commit_len, gen);
        const int got = secp256k1_rangeproof_verify_amd(ctx, &gmin, &gmax, commit, proof, plen, extra_commit, extra_commit_len, gen);
        if (again != ret) s2k_rt_die("secp256k1_rangeproof_verify", "the reference is not deterministic?");
        if (s2k_last_status() != S2K_STATUS_OK) s2k_rt_die("secp256k1_rangeproof_verify", "the engine call failed");
        if (got != ret) s2k_rt_die("secp256k1_rangeproof_verify", ret ? "the reference accepts, the engine rejects" : "the reference rejects, the engine ACCEPTS");
        if (gmin != rmin || gmax != rmax) s2k_rt_die("secp256k1_rangeproof_verify", "different min_value / max_value");
        s2k_rt.checked[S2K_RT_RANGEPROOF]++;
        if (ret) s2k_rt.accepted[S2K_RT_RANGEPROOF]++;
    }
    return ret;
}
static int s2k_rt_rangeproof_rewind(const secp256k1_context *ctx, unsigned char *blind_out, uint64_t *value_out, unsigned char *message_out, size_t *outlen,
                                    const unsigned char *nonce, uint64_t *min_value, uint64_t *max_value, const secp256k1_pedersen_commitment *commit,
                                    const unsigned char *proof, size_t plen, const unsigned char *extra_commit, size_t extra_commit_len, const secp256k1_generator *gen) {
    /* rewinding re-signs: a context without the generator multiplication tables (secp256k1_context_static) is API misuse, main_impl.h:40 */
    const int legal = ctx != NULL && secp256k1_ecmult_gen_context_is_built(&ctx->ecmult_gen_ctx) && nonce != NULL && min_value != NULL && max_value != NULL && commit != NULL && proof != NULL && gen != NULL &&
                      (message_out == NULL) == (outlen == NULL) && (extra_commit != NULL || extra_commit_len == 0);       /* ARG_CHECK(message_out != NULL || outlen == NULL) */
    const size_t cap = (legal && message_out != NULL) ? *outlen : 0;
    const int ret = secp256k1_rangeproof_rewind(ctx, blind_out, value_out, message_out, outlen, nonce, min_value, max_value, commit, proof, plen, extra_commit, extra_commit_len, gen);
    s2k_rt.calls[S2K_RT_REWIND]++;
    if (legal && s2k_rt_on()) {
        unsigned char gblind[32], *gmsg = (unsigned char*)malloc(cap + 1);
        uint64_t gvalue = 0, gmin = 0, gmax = 0, golen = cap, poff[2], eoff[2];
        int32_t got = 0;
        if (gmsg == NULL) s2k_rt_die("secp256k1_rangeproof_rewind", "out of host memory");
        poff[0] = 0; poff[1] = plen; eoff[0] = 0; eoff[1] = extra_commit_len;
        if (!secp256k1_rangeproof_rewind_batch(s2k_rt.e, &got, gblind, &gvalue, message_out != NULL ? gmsg : NULL, message_out != NULL ? &golen : NULL, cap, nonce, &gmin, &gmax,
                                               commit->data, proof, poff, extra_commit, extra_commit != NULL ? eoff : NULL, gen->data, 1))
            s2k_rt_die("secp256k1_rangeproof_rewind", "the engine call failed");
        if ((got != 0) != (ret != 0)) s2k_rt_die("secp256k1_rangeproof_rewind", ret ? "the reference rewinds, the engine does not" : "the reference refuses, the engine REWINDS");
        if (ret) {            /* the reference leaves its outputs unspecified when it returns 0 */
            if (gmin != *min_value || gmax != *max_value) s2k_rt_die("secp256k1_rangeproof_rewind", "different min_value / max_value");
            if (value_out != NULL && gvalue != *value_out) s2k_rt_die("secp256k1_rangeproof_rewind", "different value");
            if (blind_out != NULL && memcmp(gblind, blind_out, 32) != 0) s2k_rt_die("secp256k1_rangeproof_rewind", "different blinding factor");
            if (message_out != NULL && (golen != (uint64_t)*outlen || memcmp(gmsg, message_out, *outlen) != 0)) s2k_rt_die("secp256k1_rangeproof_rewind", "different message");
            s2k_rt.accepted[S2K_RT_REWIND]++;
        }
        free(gmsg);
        s2k_rt.checked[S2K_RT_REWIND]++;
    }
    return ret;
}
# define secp256k1_rangeproof_verify s2k_rt_rangeproof_verify
# define secp256k1_rangeproof_rewind s2k_rt_rangeproof_rewind
#endif

#ifdef ENABLE_MODULE_SCHNORRSIG
static int s2k_rt_schnorrsig_verify(const secp256k1_context *ctx, const unsigned char *sig64, const unsigned char *msg, size_t msglen, const secp256k1_xonly_pubkey *pubkey) {
    const int legal = ctx != NULL && sig64 != NULL && pubkey != NULL && (msg != NULL || msglen == 0);
    const int ret = secp256k1_schnorrsig_verify(ctx, sig64, msg, msglen, pubkey);
    s2k_rt.calls[S2K_RT_SCHNORR]++;
    if (legal && s2k_rt_on()) {
        static const unsigned char zero32[32] = {0};
        if (memcmp(pubkey->data, zero32, 32) != 0) {           /* an object with x = 0 is API misuse (secp256k1_pubkey_load's ARG_CHECK, src/secp256k1.c:240-254), not a verdict */
            const unsigned char nothing = 0;
            const int got = secp256k1_schnorrsig_verify_amd(ctx, sig64, msg != NULL ? msg : &nothing, msglen, pubkey);
            if (s2k_last_status() != S2K_STATUS_OK) s2k_rt_die("secp256k1_schnorrsig_verify", "the engine call failed");
            if (got != ret) s2k_rt_die("secp256k1_schnorrsig_verify", ret ? "the reference accepts, the engine rejects" : "the reference rejects, the engine ACCEPTS");
            s2k_rt.checked[S2K_RT_SCHNORR]++;
            if (ret) s2k_rt.accepted[S2K_RT_SCHNORR]++;
        }
    }
    return ret;
}
# define secp256k1_schnorrsig_verify s2k_rt_schnorrsig_verify
#endif

#ifdef ENABLE_MODULE_SCHNORRSIG_HALFAGG
static int s2k_rt_schnorrsig_aggverify(const secp256k1_context *ctx, const secp256k1_xonly_pubkey *pubkeys, const unsigned char *msgs32, size_t n,
                                       const unsigned char *aggsig, size_t aggsig_len) {
    const int legal = ctx != NULL && aggsig != NULL && (n == 0 || (pubkeys != NULL && msgs32 != NULL));
    const int ret = secp256k1_schnorrsig_aggverify(ctx, pubkeys, msgs32, n, aggsig, aggsig_len);
    s2k_rt.calls[S2K_RT_HALFAGG]++;
    if (legal && s2k_rt_on()) {
        int32_t got = 0;
        const unsigned char nothing[64] = {0};
        if (!secp256k1_schnorrsig_aggverify_amd(s2k_rt.e, &got, n != 0 ? (const unsigned char*)pubkeys : nothing, 1, n != 0 ? msgs32 : nothing, n, aggsig, aggsig_len))
            s2k_rt_die("secp256k1_schnorrsig_aggverify", "the engine call failed");
        if ((got != 0) != (ret != 0)) s2k_rt_die("secp256k1_schnorrsig_aggverify", ret ? "the reference accepts, the engine rejects" : "the reference rejects, the engine ACCEPTS");
        s2k_rt.checked[S2K_RT_HALFAGG]++;
        if (ret) s2k_rt.accepted[S2K_RT_HALFAGG]++;
    }
    return ret;
}
# define secp256k1_schnorrsig_aggverify s2k_rt_schnorrsig_aggverify
#endif

#ifdef ENABLE_MODULE_GENERATOR
static int s2k_rt_pedersen_verify_tally(const secp256k1_context *ctx, const secp256k1_pedersen_commitment *const *commits, size_t pcnt,
                                        const secp256k1_pedersen_commitment *const *ncommits, size_t ncnt) {
    int legal = ctx != NULL && (pcnt == 0 || commits != NULL) && (ncnt == 0 || ncommits != NULL);
    const int ret = secp256k1_pedersen_verify_tally(ctx, commits, pcnt, ncommits, ncnt);
    size_t i;
    s2k_rt.calls[S2K_RT_TALLY]++;
    for (i = 0; legal && i < pcnt; i++) legal = commits[i] != NULL;
    for (i = 0; legal && i < ncnt; i++) legal = ncommits[i] != NULL;
    if (legal && s2k_rt_on()) {
        const int got = secp256k1_pedersen_verify_tally_amd(ctx, (const void *const *)commits, pcnt, (const void *const *)ncommits, ncnt);
        if (s2k_last_status() != S2K_STATUS_OK) s2k_rt_die("secp256k1_pedersen_verify_tally", "the engine call failed");
        if (got != ret) s2k_rt_die("secp256k1_pedersen_verify_tally", ret ? "the reference accepts, the engine rejects" : "the reference rejects, the engine ACCEPTS");
        s2k_rt.checked[S2K_RT_TALLY]++;
        if (ret) s2k_rt.accepted[S2K_RT_TALLY]++;
    }
    return ret;
}
# define secp256k1_pedersen_verify_tally s2k_rt_pedersen_verify_tally
#endif

#ifdef ENABLE_MODULE_SURJECTIONPROOF
static int s2k_rt_surjectionproof_verify(const secp256k1_context *ctx, const secp256k1_surjectionproof *proof, const secp256k1_generator *ephemeral_input_tags,
                                         size_t n_ephemeral_input_tags, const secp256k1_generator *ephemeral_output_tag) {
    const int legal = ctx != NULL && proof != NULL && ephemeral_input_tags != NULL && ephemeral_output_tag != NULL;
    const int ret = secp256k1_surjectionproof_verify(ctx, proof, ephemeral_input_tags, n_ephemeral_input_tags, ephemeral_output_tag);
    s2k_rt.calls[S2K_RT_SURJECTION]++;
    if (legal && s2k_rt_on()) {
        const int got = secp256k1_surjectionproof_verify_amd(ctx, proof, ephemeral_input_tags, n_ephemeral_input_tags, ephemeral_output_tag);
        if (s2k_last_status() != S2K_STATUS_OK) s2k_rt_die("secp256k1_surjectionproof_verify", "the engine call failed");
        if (got != ret) s2k_rt_die("secp256k1_surjectionproof_verify", ret ? "the reference accepts, the engine rejects" : "the reference rejects, the engine ACCEPTS");
        s2k_rt.checked[S2K_RT_SURJECTION]++;
        if (ret) s2k_rt.accepted[S2K_RT_SURJECTION]++;
    }
    return ret;
}
# define secp256k1_surjectionproof_verify s2k_rt_surjectionproof_verify
#endif

#ifdef ENABLE_MODULE_BPPP
/* static in the module (src/modules/bppp/bppp_norm_product_impl.h:425-552); the reference's tests call it directly.  The reference's
 * verify overwrites its generators, c_vec and transcript, so the engine's operands are taken before it runs. */
static int s2k_rt_bppp_norm_product_verify(const secp256k1_context *ctx, secp256k1_scratch_space *scratch, const unsigned char *proof, size_t proof_len,
                                           secp256k1_sha256 *transcript, const secp256k1_scalar *rho, const secp256k1_bppp_generators *g_vec, size_t g_len,
                                           const secp256k1_scalar *c_vec, size_t c_vec_len, const secp256k1_ge *commit) {
    unsigned char *gs = NULL, *cv = NULL, tr[104], rh[32], cm[33];
    const int check = s2k_rt_on() && sizeof(secp256k1_sha256) == 104;
    size_t k, n_gens = g_vec->n;
    int ret;
    s2k_rt.calls[S2K_RT_BPPP]++;
    if (check) {
        gs = (unsigned char*)malloc(33 * n_gens + 1); cv = (unsigned char*)malloc(32 * c_vec_len + 1);
        if (gs == NULL || cv == NULL) s2k_rt_die("secp256k1_bppp_rangeproof_norm_product_verify", "out of host memory");
        for (k = 0; k < n_gens; k++) { secp256k1_ge t = g_vec->gens[k]; secp256k1_ge_serialize_ext(gs + 33 * k, &t); }
        for (k = 0; k < c_vec_len; k++) secp256k1_scalar_get_b32(cv + 32 * k, &c_vec[k]);
        memcpy(tr, transcript, 104);
        secp256k1_scalar_get_b32(rh, rho);
        { secp256k1_ge t = *commit; secp256k1_ge_serialize_ext(cm, &t); }
    }
    ret = secp256k1_bppp_rangeproof_norm_product_verify(ctx, scratch, proof, proof_len, transcript, rho, g_vec, g_len, c_vec, c_vec_len, commit);
    if (check) {
        int32_t got = 0;
        if (!secp256k1_bppp_norm_product_verify_batch(s2k_rt.e, &got, proof, proof_len, tr, rh, gs, n_gens, g_len, cv, c_vec_len, cm, 1))
            s2k_rt_die("secp256k1_bppp_rangeproof_norm_product_verify", "the engine call failed");
        if ((got != 0) != (ret != 0)) s2k_rt_die("secp256k1_bppp_rangeproof_norm_product_verify", ret ? "the reference accepts, the engine rejects" : "the reference rejects, the engine ACCEPTS");
        free(gs); free(cv);
        s2k_rt.checked[S2K_RT_BPPP]++;
        if (ret) s2k_rt.accepted[S2K_RT_BPPP]++;
    }
    return ret;
}
# define secp256k1_bppp_rangeproof_norm_product_verify s2k_rt_bppp_norm_product_verify
#endif
#endif
