/* oracle/zkp_oracle.h -- TEST INFRASTRUCTURE ONLY: byte-level entry points of the plain-C restatement (zkp_oracle.c).
 * Same conventions as the reference: 32-byte big-endian scalars / field elements, 64-byte x||y affine points with an
 * infinity flag, return 1 = valid / success. */
#ifndef ZKP_ORACLE_H
#define ZKP_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#define ZO_API __attribute__((visibility("default")))
ZO_API void zo_fe_mul(unsigned char *r, const unsigned char *a, const unsigned char *b);
ZO_API void zo_fe_inv(unsigned char *r, const unsigned char *a);
ZO_API int zo_fe_sqrt(unsigned char *r, const unsigned char *a);
ZO_API void zo_scalar_mul(unsigned char *r, const unsigned char *a, const unsigned char *b);
ZO_API void zo_scalar_split_lambda(unsigned char *r1, unsigned char *r2, const unsigned char *k);
ZO_API int zo_ge_add(unsigned char *r64, const unsigned char *a64, int ainf, const unsigned char *b64, int binf);
ZO_API int zo_ecmult(unsigned char *r64, const unsigned char *a64, int ainf, const unsigned char *na32, const unsigned char *ng32);
ZO_API int zo_ecmult_multi(unsigned char *r64, const unsigned char *g_sc32, const unsigned char *sc32, const unsigned char *pt64, const unsigned char *inf, size_t n);
ZO_API void zo_sha256(unsigned char *out32, const unsigned char *msg, size_t len);
ZO_API int zo_rangeproof_verify(uint64_t *min_value, uint64_t *max_value, const unsigned char *commit33, const unsigned char *proof, size_t plen,
                                const unsigned char *extra, size_t extra_len, const unsigned char *gen64);
ZO_API void zo_rangeproof_verify_many(int *results, uint64_t *min_v, uint64_t *max_v, const unsigned char *commits33, const unsigned char *proofs, size_t stride,
                                      const size_t *plens, const unsigned char *gens64, size_t n, int threads);
ZO_API int zo_schnorrsig_verify(const unsigned char *sig64, const unsigned char *msg, size_t msglen, const unsigned char *pk32);
/* secp256k1_schnorrsig_aggverify (modules/schnorrsig_halfagg/main_impl.h:108-198); keys as 32-byte x-only serialisations */
/* secp256k1_rangeproof_rewind (modules/rangeproof/main_impl.h:31-52, rangeproof_impl.h:61-108,339-485,652-680) */
ZO_API int zo_rangeproof_rewind(unsigned char *blind_out, uint64_t *value_out, unsigned char *message_out, size_t *outlen, const unsigned char *nonce32,
                                uint64_t *min_value, uint64_t *max_value, const unsigned char *commit33, const unsigned char *proof, size_t plen,
                                const unsigned char *extra, size_t extra_len, const unsigned char *gen64);
/* secp256k1_pedersen_verify_tally (modules/generator/main_impl.h:371-396) on 33-byte serialised commitments; -1 = unparseable */
ZO_API int zo_pedersen_verify_tally(const unsigned char *pos33, size_t pcnt, const unsigned char *neg33, size_t ncnt);
ZO_API int zo_schnorrsig_aggverify(const unsigned char *pks32, const unsigned char *msgs32, size_t n, const unsigned char *aggsig, size_t aggsig_len);
ZO_API int zo_bppp_norm_verify(const unsigned char *proof, size_t proof_len, const unsigned char *transcript104, const unsigned char *rho32,
                               const unsigned char *gens33, size_t n_gens, size_t g_len, const unsigned char *c_vec32, size_t c_len, const unsigned char *commit33);
ZO_API int zo_surjectionproof_verify(const unsigned char *proof, size_t plen, const unsigned char *in_tags64, size_t n_tags, const unsigned char *out_tag64);
#endif
