/* secp256k1_zkp_amd.h -- C ABI of the MI355X (gfx950) batch-verification engine.
 *
 * This is the drop-in boundary for the multi-scalar / double-scalar multiplication hot path of
 * BlockstreamResearch/secp256k1-zkp.  Every entry point names the reference interface it replaces
 * (paths relative to the reference tree).  Plain C: pointers and sizes only, no C++/torch types.
 *
 * Conventions (identical to the reference, include/secp256k1.h):
 *   - return 1 = success / valid, 0 = failure / invalid; verification never "errors" on malformed input.
 *   - scalars and field elements are 32-byte big-endian; affine points are 64 bytes x||y big-endian
 *     (the byte layout of `secp256k1_generator`, include/secp256k1_generator.h) with a separate infinity flag.
 *   - the caller owns every buffer; the engine keeps no pointer past a call.
 * Engine-level failures (no device, HIP error) make the *call* return 0 and set s2k_last_error(); a batch that
 * did not complete never reports an item as valid.
 *
 * `_dev` variants take pointers to device (HBM) memory of the engine's GPU and a hipStream_t passed as void*
 * (0 = the engine's own stream); they are asynchronous with respect to the host and are what bench.py times.
 * All device buffers must be 8-byte aligned.
 */
#ifndef SECP256K1_ZKP_AMD_H
#define SECP256K1_ZKP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S2K_API __attribute__((visibility("default")))

typedef struct s2k_engine s2k_engine;

/* Create an engine on HIP device `device` (builds the generator table on the GPU).  NULL on failure. */
S2K_API s2k_engine* s2k_engine_create(int device);
S2K_API void s2k_engine_destroy(s2k_engine* e);
/* Last engine-level error of the calling thread ("" if none). */
S2K_API const char* s2k_last_error(void);
/* Why the most recent call on this thread returned 0.  Batch entry points return 0 only for S2K_STATUS_* != OK (verdicts
 * go to results[]); the single-item `_amd` forms return the verdict itself, so for them 0 + S2K_STATUS_OK = "invalid" and
 * 0 + S2K_STATUS_ENGINE_FAILURE = "no verdict: take the CPU path" (a failed engine never yields 1).
 * The `_amd` forms reset the status on entry; batch callers may call s2k_clear_status() first. */
#define S2K_STATUS_OK 0
#define S2K_STATUS_ENGINE_FAILURE 1   /* no device, HIP error, out of memory */
#define S2K_STATUS_ILLEGAL_ARGUMENT 2 /* what the reference's ARG_CHECK would have rejected */
#define S2K_STATUS_BUSY 3             /* both staging sets of the host-buffer rangeproof calls are held by asynchronous tickets */
S2K_API int s2k_last_status(void);
S2K_API void s2k_clear_status(void);
/* Engine options.
 * S2K_OPT_RP_INPUTS_READY (default 0): the rangeproof `_dev` entry points run their first stage (header parse, message hash, ring
 *   bases, commitment lifts: latency bound) on side streams, two calls deep, underneath the previous call's ring kernel.  By default
 *   that stage still waits for everything queued on the caller's stream before the call, because the inputs may be produced there.
 *   Setting the option to 1 is the caller's promise that the input arrays of a `_dev` call are complete when the call is made (and
 *   stay untouched until its results are consumed); min_value/max_value may then be written before the stream reaches the call.
 *   Results are unaffected; host-buffer entry points ignore the option.  An option only: the environment cannot make this promise.
 * S2K_OPT_MSM_PIPELINE (default 0): s2k_ecmult_multi_dev and s2k_ecmult_multi_partial_dev keep TWO calls in flight for small sums (up
 *   to 2^13 terms, where one call is a chain of latency-bound launches): calls alternate between two internal stream / workspace sets
 *   and the caller's stream only waits for each call's result (give calls that may overlap different OUTPUT buffers: results of
 *   consecutive calls are not ordered against each other's writes).  Inputs stay ordered behind the caller's stream unless
 *   S2K_OPT_RP_INPUTS_READY is also set.  Larger sums fill the machine by themselves and run one after the other.
 * S2K_OPT_RP_SPLIT (default 1): two-piece double multiplication in the ring kernel (0: the one-piece form; same results).
 * S2K_OPT_GEN_CACHE_SLOTS (default 2, 0..8; $S2K_GEN_CACHE): how many rangeproof generators may have a fixed-base table at a time
 *   (21.5 GB of HBM each, see s2k_engine_cache_generator).  0 turns the shared-generator form of the ring kernel off.
 * S2K_OPT_GEN_CACHE_MIN (default 65536; $S2K_GEN_CACHE_MIN): an uncached generator gets a table automatically once this many proofs
 *   carrying it have VERIFIED (the last kernel of a call reports them through a device mailbox that the next call reads: junk proofs
 *   that merely name a generator never cost a table).  At most one automatic table is built per call, and it only takes a free slot
 *   or the slot of another automatic table -- never the table of secp256k1_generator_h or one requested through
 *   s2k_engine_cache_generator.
 * S2K_OPT_MAX_LANES (default 2^20, multiples of 256, >= 256): lanes per launch; larger batches run as consecutive sub-range launches.
 * S2K_OPT_STAGE_THREADS (default min(8, cores / 2); $S2K_STAGE_THREADS): host threads that gather a host-buffer rangeproof batch.
 * S2K_OPT_HALFAGG_HOST_CHAIN (default 1): the host-buffer half-aggregate verifier walks the randomizer hash chain on the host
 *   underneath the point-lifting kernel (0: on the device; same verdicts).
 * S2K_OPT_SYNC_SPLIT (default 1): a lone synchronous host-buffer rangeproof call that finds both staging sets free goes as two halves.
 * S2K_OPT_MSM_MAX_TERMS (default 0 = 238 609 293, what 32-bit bucket references index at nine digit windows): s2k_ecmult_multi[_dev] and
 *   s2k_ecmult_multi_partial_dev run a sum with more terms than this as consecutive launches over slices of the term arrays and add the
 *   partial sums -- the reference's own treatment of a sum that exceeds its scratch space (src/ecmult_impl.h:804-820, :856-865).  The
 *   result does not depend on the value; tests lower it to walk the loop at small sizes.
 * S2K_OPT_GTAB_BITS (20, 22, 24 or 26; default 26, $S2K_GTAB_BITS): digit width of the device's fixed-base tables (G and the rangeproof generator
 *   slots: 0.44 / 1.6 / 5.9 / 21.5 GB each, one more addition per fixed-base multiplication for every step down).  Changing it gives the device's
 *   tables back (after a device-wide wait: an administrative call) and the next call that needs one builds it at the new width; a width that
 *   does not fit still falls back to the next narrower one.  Results do not depend on it.
 * Environment: only start-up defaults are read from it, once, when an engine (or the device's table pool) is created: S2K_DEVICE,
 * S2K_GEN_CACHE, S2K_GEN_CACHE_MIN, S2K_STAGE_THREADS, S2K_GTAB_BITS.  Nothing on a call path reads the environment. */
#define S2K_OPT_RP_INPUTS_READY 1
#define S2K_OPT_RP_SPLIT 2
#define S2K_OPT_GEN_CACHE_SLOTS 3
#define S2K_OPT_GEN_CACHE_MIN 4
#define S2K_OPT_MSM_PIPELINE 5
#define S2K_OPT_MAX_LANES 6
#define S2K_OPT_STAGE_THREADS 7
#define S2K_OPT_HALFAGG_HOST_CHAIN 8
#define S2K_OPT_SYNC_SPLIT 9
#define S2K_OPT_MSM_MAX_TERMS 10
#define S2K_OPT_GTAB_BITS 11
S2K_API int s2k_engine_set_option(s2k_engine* e, int option, long value);
/* Rangeproof generator tables.  The four public keys of a Borromean ring differ by multiples of the proof's generator
 * (secp256k1_rangeproof_pub_expand, src/modules/rangeproof/rangeproof_impl.h:19-51), so when the engine holds a fixed-base table of that
 * generator a ring builds its odd-multiples tables once instead of four times (~20 % less work per proof).  The table of
 * secp256k1_generator_h (include/secp256k1_generator.h:36) is built at the first rangeproof call; other generators get one through
 * this call (gen64: the 64 bytes of a secp256k1_generator object, host memory) or automatically (S2K_OPT_GEN_CACHE_MIN); the least
 * recently used table makes room ("used" = served proofs that verified, whichever entry point they came through).  Proofs whose generator has no table take the general form of the kernel: results never depend on
 * the cache.  s2k_engine_generator_cached: 1 when gen64 has a table now. */
S2K_API int s2k_engine_cache_generator(s2k_engine* e, const unsigned char* gen64);
S2K_API int s2k_engine_generator_cached(s2k_engine* e, const unsigned char* gen64);
/* Make sure the per-batch HBM workspace can hold `n_items` rangeproofs (optional; calls grow it on demand). */
S2K_API int s2k_engine_reserve(s2k_engine* e, size_t n_items);
/* Block until everything queued on the engine's stream has finished. */
S2K_API int s2k_engine_sync(s2k_engine* e);
/* Device pointer + size (bytes) of the generator table, for tests. */
S2K_API const void* s2k_engine_gtable(s2k_engine* e, size_t* bytes);
/* Digit width D of the device's fixed-base tables (entry (w, v) = v * 2^(D w) * G for the magnitudes v of a signed D-bit digit): 26 =
 * 21.5 GB per table and 10 additions per fixed-base multiplication (the default, $S2K_GTAB_BITS at start-up), 24 / 22 / 20 = 5.9 / 1.6 /
 * 0.44 GB with 11 / 12 / 13 additions.  The first call that needs the table of G allocates the widest one that fits, from the wanted
 * width down, so an engine also exists on a partitioned or shared GPU; rangeproof generator tables take the same width.  Results do not
 * depend on the width.  The reference's knob of this kind is ECMULT_WINDOW_SIZE (src/ecmult.h:14-38). */
S2K_API int s2k_engine_gtable_bits(s2k_engine* e);
/* Device time (ms, HIP events) the construction of the table of G took on this device; negative before the table exists. */
S2K_API float s2k_engine_gtable_build_ms(s2k_engine* e);
/* Wall-clock of the most recent launch group on this engine as measured with hipEvents on its stream (ms);
 * `which`: 0 = whole call, 1 = dominant kernel only; 16 + k = dominant kernel of the k-th most recent rangeproof call (k < 32:
 * several calls may be in flight, see S2K_OPT_RP_INPUTS_READY).  Valid after s2k_engine_sync(). */
S2K_API float s2k_engine_last_ms(s2k_engine* e, int which);
/* 1 if the most recent MSM launch on this engine overflowed a bucket region and took the exact bucket-free path (diagnostics /
 * tests; synchronises the device). */
S2K_API int s2k_engine_last_msm_fallback(s2k_engine* e);
/* Work-list tallies of the most recent rangeproof call on this engine (diagnostics / tests; synchronises the device):
 * out[0] = ring groups the shared-generator form of the rings kernel was given, out[1] = rings the general form processed (its own
 * plus the ones handed back), out[2] = rings handed back because a ring key may be the point at infinity (the reference rejects such a
 * key, src/modules/rangeproof/borromean_impl.h:78), out[3] = rings handed back after an exceptional addition inside a step.
 * (A call of more than two launch groups reports its last two.) */
S2K_API int s2k_engine_rp_handback(s2k_engine* e, uint32_t out[4]);

/* ---- batch double multiplication ---------------------------------------------------------------------------
 * r[i] = na[i]*A[i] + ng[i]*G          replaces: static void secp256k1_ecmult(secp256k1_gej *r,
 *                                       const secp256k1_gej *a, const secp256k1_scalar *na,
 *                                       const secp256k1_scalar *ng)            (src/ecmult.h:47)
 * a_xy: n*64, a_inf: n bytes or NULL (all finite), na: n*32, ng: n*32 or NULL (treated as 0, as the reference's
 * ng == NULL).  Outputs: r_xy n*64 (zeroed when infinite), r_inf n*int32. */
S2K_API int s2k_ecmult_batch(s2k_engine* e, unsigned char* r_xy, int32_t* r_inf, const unsigned char* a_xy,
                             const unsigned char* a_inf, const unsigned char* na, const unsigned char* ng, size_t n);
S2K_API int s2k_ecmult_batch_dev(s2k_engine* e, void* stream, unsigned char* r_xy, int32_t* r_inf, const unsigned char* a_xy,
                                 const unsigned char* a_inf, const unsigned char* na, const unsigned char* ng, size_t n);

/* ---- multi-scalar multiplication ------------------------------------------------------------------------------
 * r = g_sc*G + sum_i sc[i]*pt[i]        replaces: static int secp256k1_ecmult_multi_var(const secp256k1_callback*,
 *                                       secp256k1_scratch*, secp256k1_gej *r, const secp256k1_scalar *inp_g_sc,
 *                                       secp256k1_ecmult_multi_callback cb, void *cbdata, size_t n)
 *                                                                               (src/ecmult.h:62, ecmult_impl.h:823-867)
 * The callback-pull model becomes arrays: sc n*32, pt_xy n*64, pt_inf n bytes or NULL; g_sc 32 bytes or NULL.
 * Entries with sc == 0 or an infinite point are skipped (ecmult_impl.h:523).  Output: r_xy 64 bytes, *r_inf. */
S2K_API int s2k_ecmult_multi(s2k_engine* e, unsigned char* r_xy, int32_t* r_inf, const unsigned char* g_sc,
                             const unsigned char* sc, const unsigned char* pt_xy, const unsigned char* pt_inf, size_t n);
S2K_API int s2k_ecmult_multi_dev(s2k_engine* e, void* stream, unsigned char* r_xy, int32_t* r_inf, const unsigned char* g_sc,
                                 const unsigned char* sc, const unsigned char* pt_xy, const unsigned char* pt_inf, size_t n);
/* K independent sums in one launch chain: r_k = g_sc_k * G + sum_{i in [offsets[k], offsets[k+1])} sc_i * P_i, k = 0 .. n_sums - 1 -- what a
 * caller of the reference gets from n_sums calls of secp256k1_ecmult_multi_var, e.g. bench_ecmult's 1 024-term sums
 * (src/bench_ecmult.c:262-276, :362-371), one per block or transaction.  ONE small sum is a chain of latency-bound launches (~0.4 ms
 * whatever its size); many of them side by side are binned, accumulated and recombined together (csrc/engine_msm_many.hip), the GPU-natural
 * counterpart of the reference's Strauss-batch regime (src/ecmult_impl.h:382-419).  sc / pt_xy / pt_inf: the sums' terms back to back;
 * offsets: n_sums + 1 increasing term indices starting at 0 -- a HOST array in both forms (read before the call returns); g_sc: n_sums * 32
 * bytes or NULL; r_xy n_sums * 64, r_inf n_sums.  Empty sums give infinity.  Sums of more than 8 192 terms take the single-sum path one
 * after the other. */
S2K_API int s2k_ecmult_multi_many(s2k_engine* e, unsigned char* r_xy, int32_t* r_inf, const unsigned char* g_sc, const unsigned char* sc,
                                  const unsigned char* pt_xy, const unsigned char* pt_inf, const uint64_t* offsets, size_t n_sums);
S2K_API int s2k_ecmult_multi_many_dev(s2k_engine* e, void* stream, unsigned char* r_xy, int32_t* r_inf, const unsigned char* g_sc, const unsigned char* sc,
                                      const unsigned char* pt_xy, const unsigned char* pt_inf, const uint64_t* offsets_host, size_t n_sums);
/* Partial sums for multi-GPU sharding: writes the Jacobian partial result as 3*9 limbs + flag (28 uint32) so that
 * ranks can all-gather raw limb buffers and finish with s2k_gej_sum (EC addition is not an RCCL op). */
S2K_API int s2k_ecmult_multi_partial_dev(s2k_engine* e, void* stream, uint32_t* r_gej28, const unsigned char* g_sc,
                                         const unsigned char* sc, const unsigned char* pt_xy, const unsigned char* pt_inf, size_t n);
S2K_API int s2k_gej_sum_dev(s2k_engine* e, void* stream, unsigned char* r_xy, int32_t* r_inf, const uint32_t* gej28, size_t count);
/* The other way to spread ONE sum over the GPUs of a node (BASELINE config 5: "Pippenger bucket windows sharded across 8 GPUs"):
 * every rank holds all n terms and owns share `part` of `parts` of the signed-digit windows of the bucket method
 * (ecmult_impl.h:516-591 generalised); it returns  sum_{w in share} 2^(c w) S_w  as a Jacobian partial.  The partials of all
 * shares add up to the full result exactly as the term-sharded partials do (all-gather of raw limbs + s2k_gej_sum_dev).
 * All ranks must pass the same n (the window width depends on it) and the same g_sc. */
S2K_API int s2k_ecmult_multi_window_partial_dev(s2k_engine* e, void* stream, uint32_t* r_gej28, const unsigned char* g_sc, const unsigned char* sc,
                                                const unsigned char* pt_xy, const unsigned char* pt_inf, size_t n, uint32_t part, uint32_t parts);

/* ---- BIP-340 batch verification -------------------------------------------------------------------------------
 * results[i] = secp256k1_schnorrsig_verify(ctx, sig64_i, msg_i, msglen, pubkey_i)
 *                                       (include/secp256k1_schnorrsig.h, src/modules/schnorrsig/main_impl.h:215-261)
 * sigs n*64, msgs n*msglen, pubkeys: pk_format 0 = n*32 x-only serialised keys (lifted on the GPU; an invalid key
 * gives 0, as secp256k1_xonly_pubkey_parse would have failed), 1 = n*64 `secp256k1_xonly_pubkey` opaque objects. */
S2K_API int secp256k1_schnorrsig_verify_batch(s2k_engine* e, int32_t* results, const unsigned char* sigs, const unsigned char* msgs,
                                              size_t msglen, const unsigned char* pubkeys, int pk_format, size_t n);
S2K_API int secp256k1_schnorrsig_verify_batch_dev(s2k_engine* e, void* stream, int32_t* results, const unsigned char* sigs,
                                                  const unsigned char* msgs, size_t msglen, const unsigned char* pubkeys, int pk_format, size_t n);

/* ---- half-aggregated Schnorr signature verification ------------------------------------------------------------------
 * *result = secp256k1_schnorrsig_aggverify(ctx, pubkeys, msgs32, n, aggsig, aggsig_len)
 *                                       (include/secp256k1_schnorrsig_halfagg.h, src/modules/schnorrsig_halfagg/main_impl.h:108-198)
 * evaluated as one (2n+1)-term multi-scalar multiplication  -s*G + sum z_i*R_i + sum (z_i e_i)*P_i == infinity  on the GPU
 * (the reference does two single multiplications per signature).  pubkeys / pk_format as in the BIP-340 batch call
 * (0 = n*32 serialised x-only keys, 1 = n*64 secp256k1_xonly_pubkey objects); msgs32 n*32; aggsig = r_0|...|r_{n-1}|s,
 * aggsig_len must be 32*(n+1) or the verdict is 0.  The return value is the call's success, the verdict goes to *result. */
S2K_API int secp256k1_schnorrsig_aggverify_amd(s2k_engine* e, int32_t* result, const unsigned char* pubkeys, int pk_format,
                                               const unsigned char* msgs32, size_t n, const unsigned char* aggsig, size_t aggsig_len);

/* ---- Borromean rangeproof batch verification ---------------------------------------------------------------------
 * results[i], min_value[i], max_value[i] = secp256k1_rangeproof_verify(ctx, &min, &max, commit_i, proof_i, plen_i,
 *                                          extra_commit_i, extra_commit_len_i, gen_i)
 *                                       (include/secp256k1_rangeproof.h:70-80, src/modules/rangeproof/main_impl.h:54-71,
 *                                        rangeproof_impl.h:541-683, borromean_impl.h:53-104)
 * Packed form: proofs = all proofs back to back, proof_off[n+1] byte offsets (proof i = [off[i], off[i+1]));
 * commits n*33 = serialised Pedersen commitments (equivalently the first 33 bytes of each 64-byte
 * secp256k1_pedersen_commitment object; an encoding that secp256k1_pedersen_commitment_parse refuses -- a prefix other than 8 / 9, x >= p,
 * x not on the curve -- can never be such an object and gives results[i] = 0); gens n*64 = secp256k1_generator objects; extra / extra_off
 * likewise or NULL.
 * min_value / max_value are written exactly when the reference writes them (header parsed), starting from 0. */
S2K_API int secp256k1_rangeproof_verify_batch(s2k_engine* e, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                                              const unsigned char* commits33, const unsigned char* proofs, const uint64_t* proof_off,
                                              const unsigned char* extra, const uint64_t* extra_off, const unsigned char* gens64, size_t n);
/* The same batch given the way the reference's own callers hold it -- arrays of pointers to the objects: commit_objs[i] -> a
 * secp256k1_pedersen_commitment (its first 33 bytes are read: src/modules/generator/main_impl.h:266-279), proofs[i] / plens[i],
 * extra[i] / elens[i] (extra may be NULL; extra[i] may be NULL when elens[i] == 0), gen_objs[i] -> a secp256k1_generator (64 bytes).
 * Both host-buffer forms gather their inputs once, with a few host threads ($S2K_STAGE_THREADS, default min(8, cores / 2)), straight into
 * pinned staging memory and copy it to HBM piece by piece underneath the packing; results come back through pinned memory. */
S2K_API int secp256k1_rangeproof_verify_batch_ptrs(s2k_engine* e, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                                                   const void* const* commit_objs, const unsigned char* const* proofs, const size_t* plens,
                                                   const unsigned char* const* extra, const size_t* elens, const void* const* gen_objs, size_t n);
/* Asynchronous pair over the two host-buffer forms, for callers with a stream of batches (the blocks of a chain sync, the transactions of a
 * mempool): `_submit` gathers the inputs, queues the copies and the kernels and returns a ticket; `_wait(ticket)` blocks until that batch is
 * done and only then fills results / min_value / max_value (which must stay valid until then; `results` holds zeros in between).  The
 * INPUT arrays may be reused as soon as `_submit` returns.  At most TWO submissions may be in flight -- a third `_submit` fails
 * (S2K_STATUS_BUSY) until the older ticket has been waited for -- and tickets may be waited for in any order, from any thread.  With
 *     submit(k+1); wait(k); submit(k+2); wait(k+1); ...
 * the gathering and the PCIe copies of batch k+1 run underneath the kernels of batch k and the GPU never idles: the throughput of the
 * host-buffer path becomes that of the device-resident one (bench.py: dropin.two_in_flight).  The synchronous forms above are
 * submit + wait on one of the same two staging sets: concurrent synchronous callers (verifier threads sharing an engine) queue for a
 * set -- two of them overlap, none is turned away -- unless both sets are held by asynchronous tickets, which only the application can
 * free: the synchronous call then returns 0 at once with S2K_STATUS_BUSY.
 * A ticket is never 0. */
S2K_API int secp256k1_rangeproof_verify_batch_submit(s2k_engine* e, uint64_t* ticket, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                                                     const unsigned char* commits33, const unsigned char* proofs, const uint64_t* proof_off,
                                                     const unsigned char* extra, const uint64_t* extra_off, const unsigned char* gens64, size_t n);
S2K_API int secp256k1_rangeproof_verify_batch_ptrs_submit(s2k_engine* e, uint64_t* ticket, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                                                          const void* const* commit_objs, const unsigned char* const* proofs, const size_t* plens,
                                                          const unsigned char* const* extra, const size_t* elens, const void* const* gen_objs, size_t n);
S2K_API int secp256k1_rangeproof_verify_batch_wait(s2k_engine* e, uint64_t ticket);
S2K_API int secp256k1_rangeproof_verify_batch_dev(s2k_engine* e, void* stream, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                                                  const unsigned char* commits33, const unsigned char* proofs, const uint64_t* proof_off,
                                                  const unsigned char* extra, const uint64_t* extra_off, const unsigned char* gens64, size_t n);
/* Rewind: verification as above and, for the proofs that verify, recovery of the committed value, the blinding factor and the
 * embedded message from the nonce -- per item what
 *   secp256k1_rangeproof_rewind(ctx, blind_out, &value_out, message_out, &outlen, nonce, &min, &max, commit, proof, plen,
 *                               extra_commit, extra_commit_len, gen)   (include/secp256k1_rangeproof.h:102-130,
 *                               src/modules/rangeproof/main_impl.h:31-52, rangeproof_impl.h:61-108,339-485,652-680)
 * returns and writes.  nonces n*32; blind_out n*32; value_out n; message_out n*msg_stride or NULL; outlen n (in: capacity of
 * item i's message buffer, at most msg_stride; out: bytes recovered) -- required iff message_out is given.  Items with
 * results[i] == 0 get zeroed blind/value and outlen 0.  Nonces are secrets: they are copied to the GPU. */
S2K_API int secp256k1_rangeproof_rewind_batch(s2k_engine* e, int32_t* results, unsigned char* blind_out, uint64_t* value_out, unsigned char* message_out,
                                              uint64_t* outlen, size_t msg_stride, const unsigned char* nonces, uint64_t* min_value, uint64_t* max_value,
                                              const unsigned char* commits33, const unsigned char* proofs, const uint64_t* proof_off,
                                              const unsigned char* extra, const uint64_t* extra_off, const unsigned char* gens64, size_t n);
/* Single-item form with the reference's argument list (ctx is accepted and ignored: the engine is process-global,
 * lazily created on device $S2K_DEVICE or 0).  commit / gen point at the reference's 64-byte opaque objects. */
S2K_API int secp256k1_rangeproof_verify_amd(const void* ctx, uint64_t* min_value, uint64_t* max_value, const void* commit,
                                            const unsigned char* proof, size_t plen, const unsigned char* extra_commit,
                                            size_t extra_commit_len, const void* gen);

/* Further single-item forms with the reference's argument lists (ctx ignored; same process-global engine; see
 * s2k_last_status() for telling "invalid" from "engine failed"):
 *   secp256k1_schnorrsig_verify(ctx, sig64, msg, msglen, pubkey)                       include/secp256k1_schnorrsig.h:178
 *   secp256k1_pedersen_verify_tally(ctx, commits, pcnt, ncommits, ncnt)                include/secp256k1_generator.h:190
 *   secp256k1_surjectionproof_verify(ctx, proof, input_tags, n_input_tags, output_tag) include/secp256k1_surjectionproof.h:256
 * Pointers to opaque objects are passed as const void* (64-byte xonly_pubkey / pedersen_commitment / generator objects,
 * the secp256k1_surjectionproof struct). */
S2K_API int secp256k1_schnorrsig_verify_amd(const void* ctx, const unsigned char* sig64, const unsigned char* msg, size_t msglen, const void* pubkey);
S2K_API int secp256k1_pedersen_verify_tally_amd(const void* ctx, const void* const* commits, size_t pcnt, const void* const* ncommits, size_t ncnt);
S2K_API int secp256k1_surjectionproof_verify_amd(const void* ctx, const void* proof, const void* ephemeral_input_tags, size_t n_ephemeral_input_tags,
                                                 const void* ephemeral_output_tag);

/* ---- Pedersen commitment tallies ------------------------------------------------------------------------------------
 * results[t] = secp256k1_pedersen_verify_tally(ctx, pos_t, pcnt_t, neg_t, ncnt_t)
 *                                       (include/secp256k1_generator.h:174-196, src/modules/generator/main_impl.h:371-396)
 * commits33: all commitments of all tallies back to back in serialised form (33 bytes; equivalently the first 33 bytes of each
 * secp256k1_pedersen_commitment object); tally t owns commitments [tally_off[t], tally_off[t+1]) and the first n_pos[t] of them
 * are its positive list, the rest its negative list.  A commitment that is not a valid encoding (cannot come out of
 * secp256k1_pedersen_commitment_parse) makes its tally 0. */
S2K_API int secp256k1_pedersen_verify_tally_batch(s2k_engine* e, int32_t* results, const unsigned char* commits33, const uint64_t* tally_off,
                                                  const uint64_t* n_pos, size_t n_tallies);

/* ---- surjection-proof batch verification -----------------------------------------------------------------------------
 * results[i] = secp256k1_surjectionproof_parse(ctx, &proof, ser_i, len_i) &&
 *              secp256k1_surjectionproof_verify(ctx, &proof, ephemeral_input_tags_i, n_i, &ephemeral_output_tag_i)
 *                                       (include/secp256k1_surjectionproof.h, src/modules/surjection/main_impl.h:45-82,360-402)
 * proofs: serialised proofs back to back with proof_off[n+1]; input_tags64: all items' input generators (64-byte
 * secp256k1_generator objects) back to back with tag_off[n+1] counted in tags; output_tags64: n*64. */
S2K_API int secp256k1_surjectionproof_verify_batch(s2k_engine* e, int32_t* results, const unsigned char* proofs, const uint64_t* proof_off,
                                                   const unsigned char* input_tags64, const uint64_t* tag_off, const unsigned char* output_tags64, size_t n);
S2K_API int secp256k1_surjectionproof_verify_batch_dev(s2k_engine* e, void* stream, int32_t* results, const unsigned char* proofs,
                                                       const uint64_t* proof_off, const unsigned char* input_tags64, const uint64_t* tag_off,
                                                       const unsigned char* output_tags64, size_t n);

/* ---- Bulletproofs++ norm-argument batch verification ---------------------------------------------------------------
 * results[i] = secp256k1_bppp_rangeproof_norm_product_verify(ctx, scratch, proof_i, proof_len, &transcript_i, &rho_i,
 *                                       g_vec, g_len, c_vec_i, c_vec_len, &commit_i)
 *                                       (src/modules/bppp/bppp_norm_product_impl.h:425-552)
 * All items share the generator set (gens33: n_gens compressed points, G_i first then H_i, as
 * secp256k1_bppp_generators_serialize writes them), g_len, c_vec_len and proof_len.  transcripts: n * 104 bytes, each the
 * SHA-256 state {uint32 s[8]; uint8 buf[64]; uint64 bytes} of the parent protocol (src/hash.h); rho n*32; c_vec
 * n*c_vec_len*32; commits n*33 (33 zero bytes = infinity, secp256k1.c:895).
 * The engine keeps a fixed-base table for the most recent generator set (75.5 MB per generator, built on the first call with that
 * set and reused while gens33 stays byte-identical); sets of more than 256 generators take the table-free path. */
S2K_API int secp256k1_bppp_norm_product_verify_batch(s2k_engine* e, int32_t* results, const unsigned char* proofs, size_t proof_len,
                                                     const unsigned char* transcripts, const unsigned char* rho, const unsigned char* gens33,
                                                     size_t n_gens, size_t g_len, const unsigned char* c_vec, size_t c_vec_len,
                                                     const unsigned char* commits33, size_t n);

/* ---- `_dev` forms of the calls above (every array in HBM of the engine's GPU, stream-ordered, nothing read back) ------------------
 * Same arguments and per-item results as the host forms.  Where a call needs a few bytes on the host to plan its launches, that is
 * an explicit extra argument: gens33_host (the BP++ generator set, cache key of its fixed-base table), tally_off_host / n_pos_host
 * (the ragged tally sizes; the call returns once they have been consumed).  The half-aggregate verdict lands in result_dev[0]. */
S2K_API int secp256k1_bppp_norm_product_verify_batch_dev(s2k_engine* e, void* stream, int32_t* results, const unsigned char* proofs, size_t proof_len,
                                                         const unsigned char* transcripts, const unsigned char* rho, const unsigned char* gens33_dev,
                                                         const unsigned char* gens33_host, size_t n_gens, size_t g_len, const unsigned char* c_vec,
                                                         size_t c_vec_len, const unsigned char* commits33, size_t n);
S2K_API int secp256k1_schnorrsig_aggverify_dev(s2k_engine* e, void* stream, int32_t* result_dev, const unsigned char* pubkeys, int pk_format,
                                               const unsigned char* msgs32, size_t n, const unsigned char* aggsig, size_t aggsig_len);
/* The same with the randomizer hash's chain walked by the caller.  z_i hashes the whole prefix r_0|x(P_0)|m_0|...|m_i (main_impl.h:153-163): a
 * serial chain of 1.5 SHA-256 blocks per signature -- 118 ms on the device for 2^15 signatures, 2.6 ms on one host core with SHA extensions.
 * chain_states (HBM): ((3 n) >> 1) x 8 words, the state after every full 64-byte block behind the tag midstate; s2k_halfagg_chain_states
 * computes them from host copies of the inputs.  NULL = the plain `_dev` form (device chain).  The states are input data of the caller: a
 * wrong array gives a wrong verdict for that aggregate only. */
S2K_API int secp256k1_schnorrsig_aggverify_dev_chain(s2k_engine* e, void* stream, int32_t* result_dev, const unsigned char* pubkeys, int pk_format,
                                                     const unsigned char* msgs32, size_t n, const unsigned char* aggsig, size_t aggsig_len,
                                                     const uint32_t* chain_states);
S2K_API int s2k_halfagg_chain_states(uint32_t* states_out, const unsigned char* pubkeys, int pk_format, const unsigned char* msgs32, size_t n,
                                     const unsigned char* aggsig);
S2K_API int secp256k1_pedersen_verify_tally_batch_dev(s2k_engine* e, void* stream, int32_t* results, const unsigned char* commits33,
                                                      const uint64_t* tally_off_host, const uint64_t* n_pos_host, size_t n_tallies);
S2K_API int secp256k1_rangeproof_rewind_batch_dev(s2k_engine* e, void* stream, int32_t* results, unsigned char* blind_out, uint64_t* value_out,
                                                  unsigned char* message_out, uint64_t* outlen, size_t msg_stride, const unsigned char* nonces,
                                                  uint64_t* min_value, uint64_t* max_value, const unsigned char* commits33, const unsigned char* proofs,
                                                  const uint64_t* proof_off, const unsigned char* extra, const uint64_t* extra_off,
                                                  const unsigned char* gens64, size_t n);

/* ---- Bulletproofs++ commitments on the fixed-base tables -----------------------------------------------
 * commits33[i] = serialize_ext( secp256k1_bppp_commit(ctx, scratch, &commit, gens, n_vec_i, g_len, l_vec_i, h_len, c_vec_i, h_len, &mu_i) )
 *              = v G + sum n_i G_i + sum l_j H_j ,  v = sum n_i^2 mu^(i+1) + <l, c>
 *                                       (static, src/modules/bppp/bppp_norm_product_impl.h:105-151; serialisation src/secp256k1.c:885-891:
 *                                        33 zero bytes = infinity)
 * gens33: n_gens = g_len + h_len compressed generators (G_i first, then H_j), shared by the batch and cached as a fixed-base table
 * like the verifier's; n_vec n*g_len*32, l_vec / c_vec n*h_len*32, mu n*32 (scalars, big-endian, reduced mod the group order).
 * results (may be NULL): 1 per item when the generator set parsed, else 0 (the reference cannot be handed an unparsed set).
 * The `_dev` form takes every array in HBM plus the generator set a second time on the host (gens33_host: the table's cache key). */
S2K_API int secp256k1_bppp_commit_batch(s2k_engine* e, unsigned char* commits33, int32_t* results, const unsigned char* gens33, size_t n_gens, size_t g_len,
                                        const unsigned char* n_vec, const unsigned char* l_vec, const unsigned char* c_vec, size_t h_len,
                                        const unsigned char* mu, size_t n);
S2K_API int secp256k1_bppp_commit_batch_dev(s2k_engine* e, void* stream, unsigned char* commits33, int32_t* results, const unsigned char* gens33_dev,
                                            const unsigned char* gens33_host, size_t n_gens, size_t g_len, const unsigned char* n_vec,
                                            const unsigned char* l_vec, const unsigned char* c_vec, size_t h_len, const unsigned char* mu, size_t n);

/* ---- engine groups: the GPUs of one node behind one handle ------------------------------------------------------------------------
 * A group holds one engine per entry of `devices` (HIP device ordinals; an ordinal may appear more than once -- engines on one device
 * share that device's tables) and one host thread per engine.  Independent items are replica work ("batches of independent
 * proofs: replicas only"): `_group` batch calls cut the batch into contiguous index ranges, one per engine, and run them concurrently
 * through the engines' host-buffer entry points; arguments, per-item results and error conventions are those of the single-engine
 * calls (a failing engine fails the call and zeroes `results`).  One LARGE multi-scalar multiplication is sharded by terms
 * (BASELINE config 5): every engine sums its slice to a 112-byte Jacobian partial, the partials are copied to the first engine's
 * device (hipMemcpyPeerAsync; devices that can reach each other over xGMI are made peers) and summed there.  s2k_ecmult_multi_group
 * takes host arrays (each engine uploads its slice); s2k_ecmult_multi_group_dev takes, per engine i, device pointers to a slice that
 * is already resident on engine i's GPU (n_per_engine[i] terms; pt_inf_dev may be NULL; g_sc_dev0: 32 bytes on engine 0's GPU or
 * NULL) -- the result comes back to the host in both.  One group call runs at a time; the engines of a group may also be used directly
 * (s2k_group_engine), e.g. to cache a generator on every device.
 * A group handle fits the reference-side hook's `engine` slot (integration/secp256k1_amd_hook.h): the `_group` functions have the
 * single-engine prototypes with the handle type changed. */
typedef struct s2k_group s2k_group;
S2K_API s2k_group* s2k_group_create(const int* devices, int n);
S2K_API void s2k_group_destroy(s2k_group* g);
S2K_API int s2k_group_size(const s2k_group* g);
S2K_API s2k_engine* s2k_group_engine(s2k_group* g, int i);
S2K_API int secp256k1_rangeproof_verify_batch_group(s2k_group* g, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                                                    const unsigned char* commits33, const unsigned char* proofs, const uint64_t* proof_off,
                                                    const unsigned char* extra, const uint64_t* extra_off, const unsigned char* gens64, size_t n);
S2K_API int secp256k1_rangeproof_verify_batch_ptrs_group(s2k_group* g, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                                                         const void* const* commit_objs, const unsigned char* const* proofs, const size_t* plens,
                                                         const unsigned char* const* extra, const size_t* elens, const void* const* gen_objs, size_t n);
S2K_API int secp256k1_schnorrsig_verify_batch_group(s2k_group* g, int32_t* results, const unsigned char* sigs, const unsigned char* msgs,
                                                    size_t msglen, const unsigned char* pubkeys, int pk_format, size_t n);
S2K_API int s2k_ecmult_multi_group(s2k_group* g, unsigned char* r_xy, int32_t* r_inf, const unsigned char* g_sc, const unsigned char* sc,
                                   const unsigned char* pt_xy, const unsigned char* pt_inf, size_t n);
S2K_API int s2k_ecmult_multi_group_dev(s2k_group* g, unsigned char* r_xy, int32_t* r_inf, const unsigned char* g_sc_dev0,
                                       const unsigned char* const* sc_dev, const unsigned char* const* pt_xy_dev,
                                       const unsigned char* const* pt_inf_dev, const size_t* n_per_engine);

#ifdef __cplusplus
}
#endif
#endif
