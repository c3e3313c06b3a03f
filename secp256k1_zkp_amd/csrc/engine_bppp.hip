#include "engine_internal.h"

// ------------------------------------------------------------------------------------------------------------
// Bulletproofs++ norm-argument batch verification (bppp.h)
// ------------------------------------------------------------------------------------------------------------
__global__ void k_bp_gens(u32* gens18, int* gens_ok, const unsigned char* gens33, u32 n_gens) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_gens) return;
    ge p; const int ok = bp_parse33(p, gens33 + 33 * i);
    fe_norm_weak(p.x); fe_norm_weak(p.y);
    for (int k = 0; k < 9; k++) { gens18[18 * i + k] = p.x.n[k]; gens18[18 * i + 9 + k] = p.y.n[k]; }
    if (!ok) atomicAnd(gens_ok, 0);
}
__global__ void __launch_bounds__(64)
k_bp_prologue(u32* term_sc, int* proof_ok, bp_shape sh, const unsigned char* proofs, size_t proof_len, const unsigned char* transcripts,
              const unsigned char* rho, const unsigned char* c_vec, u32* sg_factors, size_t n) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    proof_ok[p] = bp_prologue(term_sc + p * sh.n_terms * 8, sh, proofs + p * proof_len, transcripts + p * 104, rho + 32 * p, c_vec + p * sh.h_len * 32,
                              sg_factors + p * (8 * BP_MAX_LOG_G));
}
// the g_len - 1 scalars s_g[1..] of every proof, one lane each (bp_sg_entry)
__global__ void __launch_bounds__(256)
k_bp_sg(u32* term_sc, const u32* sg_factors, const int* proof_ok, bp_shape sh, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t p = t / (sh.g_len - 1); const u32 i = 1u + (u32)(t % (sh.g_len - 1));
    if (p >= n || !proof_ok[p]) return;
    bp_sg_entry(term_sc + p * sh.n_terms * 8, sg_factors + p * (8 * BP_MAX_LOG_G), sh, i);
}
// terms t0 .. t0 + tcount - 1 of every proof, one lane each (full double-and-add)
__global__ void __launch_bounds__(256, 2)
k_bp_terms(u32* out28, unsigned char* term_ok, bp_shape sh, const u32* term_sc, const int* proof_ok, const u32* gens18, const unsigned char* proofs,
           size_t proof_len, const unsigned char* commits33, const u32* gtab, u32* ptab, size_t n, u32 t0, u32 tcount) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t p = t / tcount; const u32 ti = t0 + (u32)(t % tcount);
    const int inrange = p < n;
    int live = inrange;
    if (!live) p = 0;
    live &= proof_ok[p];
    __shared__ u32 s_dig[S2K_DIG_WORDS * 256];
    const lane_mem lm{ptab + t * S2K_PTAB_WORDS, S2K_LANE_DIG(s_dig)};
    gej o; const int ok = bp_term(o, sh, ti, term_sc + p * sh.n_terms * 8, gens18, proofs + p * proof_len, commits33 + 33 * p, live, gtab, lm);
    if (inrange) { gej_store28(out28 + (p * sh.n_terms + ti) * 28, o); term_ok[p * sh.n_terms + ti] = (unsigned char)ok; }
}
// generator terms 0 .. n_gens - 1 of every proof through the generator set's fixed-base table
__global__ void __launch_bounds__(256, 2)
k_bp_terms_fixed(u32* out28, unsigned char* term_ok, bp_shape sh, const u32* term_sc, const int* proof_ok, const u32* tab, size_t tab_stride, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t p = t / sh.n_gens; const u32 ti = (u32)(t % sh.n_gens);
    if (p >= n) return;
    gej o; gej_set_infinity(o);
    if (proof_ok[p]) bp_term_fixed(o, tab + (size_t)ti * tab_stride, term_sc + (p * sh.n_terms + ti) * 8);
    gej_store28(out28 + (p * sh.n_terms + ti) * 28, o); term_ok[p * sh.n_terms + ti] = 1;
}
// the set's tables (bppp.h): lane (generator, window) writes the window's base (and window 0's lane the table's header); then the seeds and
// the fill of gtable.h for every generator at once -- the generator rides in the grid
__global__ void k_bp_tab_base(u32* tab, size_t stride, const u32* gens18, u32 n_gens, u32 D) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x, W = gtab_windows_for(D);
    if (t >= n_gens * W) return;
    const u32 gen = t / W, w = t % W;
    u32* T = tab + (size_t)gen * stride;
    if (w == 0) gtab_write_header(T, D);
    ge p; for (int i = 0; i < 9; i++) { p.x.n[i] = gens18[18 * gen + i]; p.y.n[i] = gens18[18 * gen + 9 + i]; }
    gtab_build_base(T, D, w, &p);
}
__global__ void __launch_bounds__(256)
k_bp_tab_seeds(u32* tab, size_t stride, gtab_fill_plan p) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x, per = gtab_seeds_per_window(p);
    const u32 w = t / per;
    if (w < p.W) gtab_build_seed(tab + (size_t)blockIdx.y * stride, p, w, t % per);
}
__global__ void __launch_bounds__(256, 2)
k_bp_tab_fill(u32* tab, size_t stride, gtab_fill_plan p, u32 runs) {
    const u32 b = 1u + blockIdx.x * 256u + threadIdx.x, run = blockIdx.y % runs, w = blockIdx.y / runs;
    if (b <= gtab_fill_cols(p)) gtab_fill_run(tab + (size_t)blockIdx.z * stride, p, w, b, 1u + run * GTAB_FILL_RUN);
}
__global__ void k_bp_final(int32_t* results, const u32* sums28, const int* proof_ok, const unsigned char* term_ok, const int* gens_ok, u32 n_terms, size_t n) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int ok = proof_ok[p] & *gens_ok;
    for (u32 t = 0; t < n_terms; t++) ok &= term_ok[p * n_terms + t];
    ok &= (int)sums28[p * 28 + 27];            // res2 - res1 must be the point at infinity (gej_eq_var, :551)
    results[p] = ok;
}
// fixed-base table of a generator set: built on first use, kept for the following calls (a deployment has one set).
// Sets too large for the table (> 256 generators = 19 GB) take the general path (*fixed = 0).  gens18: the set's affine points on
// the device (k_bp_gens already queued on st); gens33: the serialised set on the HOST (the cache key).
static int bp_table_cached(const s2k_engine* e, const unsigned char* gens33, size_t n_gens) {
    return n_gens <= 256 && e->bp_tab && e->bp_key.size() == 33 * n_gens && memcmp(e->bp_key.data(), gens33, 33 * n_gens) == 0;
}
static int bp_ensure_table(s2k_engine* e, hipStream_t st, const u32* gens18, const int* gens_ok_dev, const unsigned char* gens33, size_t n_gens, int* fixed) {
    *fixed = n_gens <= 256;
    if (*fixed && !bp_table_cached(e, gens33, n_gens)) {
        // (the old table may still be read by this engine's earlier calls: retired, not freed -- no wait for the device)
        if (e->bp_tab) { engine_retire_dev(e, e->bp_tab, e->bp_tab_bytes); e->bp_tab = nullptr; e->bp_tab_bytes = 0; }
        e->bp_key.clear();
        u32 D = bp_tab_bits_for(n_gens);
        for (;; D--) {                                        // (narrower when the memory is not there; 17 bits = 0.07 GB per generator)
            if (hipMalloc((void**)&e->bp_tab, bp_tab_words(n_gens, D) * sizeof(u32)) == hipSuccess) break;
            (void)hipGetLastError(); e->bp_tab = nullptr;
            if (D == 17u) break;
        }
        if (!e->bp_tab) *fixed = 0;
        else {
            e->bp_tab_bytes = bp_tab_words(n_gens, D) * sizeof(u32); e->bp_tab_stride = bp_tab_stride(D);
            const gtab_fill_plan p = gtab_make_fill_plan(D);
            const u32 seeds = p.W * gtab_seeds_per_window(p), runs = gtab_fill_runs(p);
            hipLaunchKernelGGL(k_bp_tab_base, dim3((unsigned)((n_gens * p.W + 63) / 64)), dim3(64), 0, st, e->bp_tab, e->bp_tab_stride, gens18, (u32)n_gens, D);
            hipLaunchKernelGGL(k_bp_tab_seeds, dim3((seeds + 255) / 256, (unsigned)n_gens), dim3(256), 0, st, e->bp_tab, e->bp_tab_stride, p);
            hipLaunchKernelGGL(k_bp_tab_fill, dim3((gtab_fill_cols(p) + 255) / 256, p.W * runs, (unsigned)n_gens), dim3(256), 0, st, e->bp_tab, e->bp_tab_stride, p, runs);
            HIPCHK(hipGetLastError());
            int ok_host = 0;
            HIPCHK(hipMemcpyAsync(&ok_host, gens_ok_dev, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));             // the key is only remembered for a table whose build is known to have completed
            e->bp_gens_ok = ok_host;
            e->bp_key.assign(gens33, gens33 + 33 * n_gens);
        }
    }
    return 1;
}
static size_t bpv_ws_bytes(size_t n, const bp_shape& sh) {
    const size_t T = sh.n_terms, nt = n * T;
    return ws_need({(size_t)sh.n_gens * 18 * 4, 64, nt * 8 * 4, 4 * n, nt * 28 * 4, nt + 64, (n * (T / 1024 + 1) + 64) * 28 * 4, (n * (T / 1024 + 1) + 64) * 28 * 4, n * 8 * BP_MAX_LOG_G * 4});
}
// device pointers in (gens33_host: the generator set once more on the host, the fixed-base table's cache key); one launch group
static int bpv_launch(s2k_engine* e, hipStream_t st, ws_carver& c, int32_t* d_res, const bp_shape& sh, const unsigned char* d_pr, size_t proof_len, const unsigned char* d_tr,
                      const unsigned char* d_rho, const unsigned char* d_g33, const unsigned char* gens33_host, const unsigned char* d_cv, const unsigned char* d_cm, size_t n) {
    const size_t T = sh.n_terms, nt = n * T, n_gens = sh.n_gens;
    u32* gens18 = c.take<u32>(n_gens * 18); int* gens_ok = c.take<int>(16); u32* term_sc = c.take<u32>(nt * 8); int* proof_ok = c.take<int>(n);
    u32* out28 = c.take<u32>(nt * 28); unsigned char* term_ok = c.take<unsigned char>(nt + 64);
    u32* bufA = c.take<u32>((n * (T / 1024 + 1) + 64) * 28); u32* bufB = c.take<u32>((n * (T / 1024 + 1) + 64) * 28);
    u32* sg_factors = c.take<u32>(n * 8 * BP_MAX_LOG_G);
    // (the reference accepts larger sets: this is "not supported here", an engine-level failure that sends a hooked caller to its CPU path,
    // not an illegal argument that would read as a rejected proof)
    if (sh.log_g > BP_MAX_LOG_G) return s2k_fail("secp256k1_bppp_norm_product_verify_batch", "g_len above 256 is not supported by this engine");
    if (!engine_ptab(e, ((nt + 255) / 256) * 256)) return 0;
    ENGINE_GTAB(e, st);
    HIPCHK(hipMemsetAsync(d_res, 0, sizeof(int32_t) * n, st));
    HIPCHK(hipEventRecord(e->ev[0], st));
    int fixed = 0;
    if (bp_table_cached(e, gens33_host, n_gens)) {             // the set's table is there: its generators need not be decompressed again
        fixed = 1;
        launch_set_word(st, (u32*)gens_ok, (u32)e->bp_gens_ok);
    } else {
        launch_set_word(st, (u32*)gens_ok, 1u);
        hipLaunchKernelGGL(k_bp_gens, dim3((unsigned)((n_gens + 63) / 64)), dim3(64), 0, st, gens18, gens_ok, d_g33, (u32)n_gens);
        if (!bp_ensure_table(e, st, gens18, gens_ok, gens33_host, n_gens, &fixed)) return 0;
    }
    hipLaunchKernelGGL(k_bp_prologue, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, term_sc, proof_ok, sh, d_pr, proof_len, d_tr, d_rho, d_cv, sg_factors, n);
    if (sh.g_len > 1) hipLaunchKernelGGL(k_bp_sg, dim3((unsigned)((n * (sh.g_len - 1) + 255) / 256)), dim3(256), 0, st, term_sc, sg_factors, proof_ok, sh, n);
    HIPCHK(hipEventRecord(e->ev[2], st));
    {
        const u32 t0 = fixed ? (u32)n_gens : 0u, tcount = (u32)T - t0;
        // the generator terms (fixed-base tables: throughput bound, fills the machine) run on the side stream next to the proof's own
        // points (one full double multiplication per lane, ~13 lanes per proof: latency bound at batch sizes like 2^12)
        if (fixed) {
            HIPCHK(hipEventRecord(e->ev_msm_fork, st));
            HIPCHK(hipStreamWaitEvent(e->stream2, e->ev_msm_fork, 0));
            hipLaunchKernelGGL(k_bp_terms_fixed, dim3((unsigned)((n * n_gens + 255) / 256)), dim3(256), 0, e->stream2, out28, term_ok, sh, term_sc, proof_ok, e->bp_tab, e->bp_tab_stride, n);
            HIPCHK(hipEventRecord(e->ev_msm_join, e->stream2));
        }
        hipLaunchKernelGGL(k_bp_terms, dim3((unsigned)((n * tcount + 255) / 256)), dim3(256), 0, st, out28, term_ok, sh, term_sc, proof_ok, gens18, d_pr, proof_len, d_cm,
                           e->gtab, e->ptab, n, t0, tcount);
        if (fixed) HIPCHK(hipStreamWaitEvent(st, e->ev_msm_join, 0));
    }
    HIPCHK(hipEventRecord(e->ev[3], st));
    const u32* sums = launch_gej_reduce(st, out28, bufA, bufB, (u32)n, (u32)T);
    hipLaunchKernelGGL(k_bp_final, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_res, sums, proof_ok, term_ok, gens_ok, (u32)T, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
}
extern "C" int secp256k1_bppp_norm_product_verify_batch_dev(s2k_engine* e, void* stream, int32_t* results, const unsigned char* proofs, size_t proof_len,
                                                            const unsigned char* transcripts, const unsigned char* rho, const unsigned char* gens33_dev,
                                                            const unsigned char* gens33_host, size_t n_gens, size_t g_len, const unsigned char* c_vec, size_t c_vec_len,
                                                            const unsigned char* commits33, size_t n) {
    if (!e) return s2k_fail("secp256k1_bppp_norm_product_verify_batch_dev", "null engine");
    if (n == 0) return 1;
    if (!results || !proofs || !transcripts || !rho || !gens33_dev || !gens33_host || !c_vec || !commits33)
        return s2k_fail_arg("secp256k1_bppp_norm_product_verify_batch_dev", "illegal argument (ARG_CHECK)");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    bp_shape sh;
    if (!bp_make_shape(sh, g_len, c_vec_len, n_gens, proof_len)) { HIPCHK(hipMemsetAsync(results, 0, sizeof(int32_t) * n, st)); return 1; }   // :446-461
    const size_t per = std::max<size_t>(1, e->max_lanes / sh.n_terms);        // proofs per launch group
    if (!engine_workspace(e, bpv_ws_bytes(std::min(n, per), sh))) return 0;
    for (size_t p0 = 0; p0 < n; p0 += per) {
        const size_t m = std::min(n - p0, per);
        ws_carver c{e->ws, 0};
        if (!bpv_launch(e, st, c, results + p0, sh, proofs + p0 * proof_len, proof_len, transcripts + 104 * p0, rho + 32 * p0, gens33_dev, gens33_host,
                        c_vec + 32 * c_vec_len * p0, commits33 + 33 * p0, m)) return 0;
    }
    return 1;
}
extern "C" int secp256k1_bppp_norm_product_verify_batch(s2k_engine* e, int32_t* results, const unsigned char* proofs, size_t proof_len,
                                                        const unsigned char* transcripts, const unsigned char* rho, const unsigned char* gens33,
                                                        size_t n_gens, size_t g_len, const unsigned char* c_vec, size_t c_vec_len,
                                                        const unsigned char* commits33, size_t n) {
    if (!e) return s2k_fail("secp256k1_bppp_norm_product_verify_batch", "null engine");
    if (n == 0) return 1;
    memset(results, 0, sizeof(int32_t) * n);
    bp_shape sh;
    if (!bp_make_shape(sh, g_len, c_vec_len, n_gens, proof_len)) return 1;   // :446-461: every item 0
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    const size_t per = std::max<size_t>(1, e->max_lanes / sh.n_terms);
    const size_t inner = bpv_ws_bytes(std::min(n, per), sh);
    if (!engine_workspace(e, inner + ws_need({4 * n, n * proof_len + 64, 104 * n, 32 * n, 33 * n_gens, 32 * c_vec_len * n, 33 * n}))) return 0;
    ws_carver c{e->ws, inner};
    int32_t* d_res = c.take<int32_t>(n); unsigned char* d_pr = c.take<unsigned char>(n * proof_len + 64); unsigned char* d_tr = c.take<unsigned char>(104 * n);
    unsigned char* d_rho = c.take<unsigned char>(32 * n); unsigned char* d_g33 = c.take<unsigned char>(33 * n_gens);
    unsigned char* d_cv = c.take<unsigned char>(32 * c_vec_len * n); unsigned char* d_cm = c.take<unsigned char>(33 * n);
    hipStream_t st = e->stream;
    stream_guard sg(e, st);
    HIPCHK(hipMemcpyAsync(d_pr, proofs, n * proof_len, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_tr, transcripts, 104 * n, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_rho, rho, 32 * n, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_g33, gens33, 33 * n_gens, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_cv, c_vec, 32 * c_vec_len * n, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_cm, commits33, 33 * n, hipMemcpyHostToDevice, st));
    if (!secp256k1_bppp_norm_product_verify_batch_dev(e, nullptr, d_res, d_pr, proof_len, d_tr, d_rho, d_g33, gens33, n_gens, g_len, d_cv, c_vec_len, d_cm, n)) return 0;
    HIPCHK(hipMemcpyAsync(results, d_res, 4 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 1;
}

// ---- secp256k1_bppp_commit, batched (bppp.h) ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_bpc_scalars(u32* v8, const unsigned char* n_vec, const unsigned char* l_vec, const unsigned char* c_vec, const unsigned char* mu, u32 g_len, u32 h_len, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bpc_v_scalar(v8 + 8 * i, n_vec + (size_t)32 * g_len * i, g_len, l_vec + (size_t)32 * h_len * i, c_vec + (size_t)32 * h_len * i, h_len, mu + 32 * i);
}
// lane (item, t): t < n_gens -> scalar_t * generator_t (fixed-base table, or the general double-and-add when there is none), t == n_gens -> v * G
__global__ void __launch_bounds__(256, 2)
k_bpc_terms(u32* out28, const u32* v8, const unsigned char* n_vec, const unsigned char* l_vec, u32 g_len, u32 h_len, const u32* tab, size_t tab_stride, const u32* gens18, const int* gens_ok,
            const u32* gtab, u32* ptab, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 T = g_len + h_len + 1;
    const size_t i = t / T; const u32 ti = (u32)(t % T);
    const int live = (i < n) && *gens_ok;
    const size_t ii = i < n ? i : 0;
    u32 k8[8];
    if (ti < g_len + h_len) {
        scalar k; sc_set_b32(k, ti < g_len ? n_vec + ((size_t)g_len * ii + ti) * 32 : l_vec + ((size_t)h_len * ii + (ti - g_len)) * 32, nullptr);
        for (int q = 0; q < 8; q++) k8[q] = live ? k.d[q] : 0u;
    } else for (int q = 0; q < 8; q++) k8[q] = live ? v8[8 * ii + q] : 0u;
    gej o;
    if (ti == g_len + h_len) bp_term_fixed(o, gtab, k8);            // (the table of G has the same format: one routine)
    else if (tab) bp_term_fixed(o, tab + (size_t)ti * tab_stride, k8);
    else {
        __shared__ u32 s_dig[S2K_DIG_WORDS * 256];
        const lane_mem lm{ptab + t * S2K_PTAB_WORDS, S2K_LANE_DIG(s_dig)};
        gej A; ge p; for (int q = 0; q < 9; q++) { p.x.n[q] = gens18[18 * ti + q]; p.y.n[q] = gens18[18 * ti + 9 + q]; }
        gej_set_ge(A, p);
        scalar k, g; for (int q = 0; q < 8; q++) k.d[q] = k8[q]; sc_set_zero(g);
        ecmult_lane(o, A, k, g, 0, gtab, lm);
    }
    if (i < n) gej_store28(out28 + t * 28, o);
}
// secp256k1_ge_serialize_ext (src/secp256k1.c:885-891): 33 zero bytes for infinity, else 0x02/0x03 || x; results[i] = the set parsed
__global__ void __launch_bounds__(64)
k_bpc_final(unsigned char* commits33, int32_t* results, const u32* sums28, const int* gens_ok, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    gej r; gej_load28(r, sums28 + i * 28);
    ge a; ge_set_gej(a, r);
    unsigned char* o = commits33 + 33 * i;
    if (r.inf || !*gens_ok) { for (int k = 0; k < 33; k++) o[k] = 0; }
    else { o[0] = (unsigned char)(2 | fe_is_odd(a.y)); fe_get_b32(o + 1, a.x); }
    if (results) results[i] = *gens_ok;
}
static int bpc_launch(s2k_engine* e, hipStream_t st, ws_carver& c, unsigned char* d_out33, int32_t* d_res, const unsigned char* d_g33, const unsigned char* gens33_host,
                      size_t n_gens, size_t g_len, size_t h_len, const unsigned char* d_nv, const unsigned char* d_lv, const unsigned char* d_cv, const unsigned char* d_mu, size_t n) {
    const size_t T = n_gens + 1;
    ENGINE_GTAB(e, st);
    u32* gens18 = c.take<u32>(n_gens * 18); int* gens_ok = c.take<int>(16); u32* v8 = c.take<u32>(8 * n);
    u32* out28 = c.take<u32>(n * T * 28); u32* bufA = c.take<u32>((n * (T / 1024 + 1) + 64) * 28); u32* bufB = c.take<u32>((n * (T / 1024 + 1) + 64) * 28);
    launch_set_word(st, (u32*)gens_ok, 1u);
    HIPCHK(hipEventRecord(e->ev[0], st));
    hipLaunchKernelGGL(k_bp_gens, dim3((unsigned)((n_gens + 63) / 64)), dim3(64), 0, st, gens18, gens_ok, d_g33, (u32)n_gens);
    int fixed = 0;
    if (!bp_ensure_table(e, st, gens18, gens_ok, gens33_host, n_gens, &fixed)) return 0;
    if (!fixed && !engine_ptab(e, ((n * T + 255) / 256) * 256)) return 0;
    hipLaunchKernelGGL(k_bpc_scalars, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, v8, d_nv, d_lv, d_cv, d_mu, (u32)g_len, (u32)h_len, n);
    HIPCHK(hipEventRecord(e->ev[2], st));
    hipLaunchKernelGGL(k_bpc_terms, dim3((unsigned)((n * T + 255) / 256)), dim3(256), 0, st, out28, v8, d_nv, d_lv, (u32)g_len, (u32)h_len, fixed ? e->bp_tab : (const u32*)nullptr, e->bp_tab_stride,
                       gens18, gens_ok, e->gtab, e->ptab, n);
    HIPCHK(hipEventRecord(e->ev[3], st));
    const u32* sums = launch_gej_reduce(st, out28, bufA, bufB, (u32)n, (u32)T);
    hipLaunchKernelGGL(k_bpc_final, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_out33, d_res, sums, gens_ok, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
}
static size_t bpc_ws_bytes(size_t n, size_t n_gens) {
    const size_t T = n_gens + 1;
    return ws_need({n_gens * 18 * 4, 64, 32 * n, n * T * 28 * 4, (n * (T / 1024 + 1) + 64) * 28 * 4, (n * (T / 1024 + 1) + 64) * 28 * 4});
}
extern "C" int secp256k1_bppp_commit_batch_dev(s2k_engine* e, void* stream, unsigned char* commits33, int32_t* results, const unsigned char* gens33_dev,
                                               const unsigned char* gens33_host, size_t n_gens, size_t g_len, const unsigned char* n_vec, const unsigned char* l_vec,
                                               const unsigned char* c_vec, size_t h_len, const unsigned char* mu, size_t n) {
    if (!e) return s2k_fail("secp256k1_bppp_commit_batch_dev", "null engine");
    if (!commits33 || !gens33_dev || !gens33_host || !n_vec || !l_vec || !c_vec || !mu || n_gens != g_len + h_len || n_gens == 0)
        return s2k_fail_arg("secp256k1_bppp_commit_batch_dev", "illegal argument (ARG_CHECK)");
    if (n == 0) return 1;
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    const size_t per = std::max<size_t>(1, e->max_lanes / (n_gens + 1));
    if (!engine_workspace(e, bpc_ws_bytes(std::min(n, per), n_gens))) return 0;
    for (size_t i0 = 0; i0 < n; i0 += per) {
        const size_t m = std::min(n - i0, per);
        ws_carver c{e->ws, 0};
        if (!bpc_launch(e, st, c, commits33 + 33 * i0, results ? results + i0 : nullptr, gens33_dev, gens33_host, n_gens, g_len, h_len, n_vec + 32 * g_len * i0,
                        l_vec + 32 * h_len * i0, c_vec + 32 * h_len * i0, mu + 32 * i0, m)) return 0;
    }
    return 1;
}
extern "C" int secp256k1_bppp_commit_batch(s2k_engine* e, unsigned char* commits33, int32_t* results, const unsigned char* gens33, size_t n_gens, size_t g_len,
                                           const unsigned char* n_vec, const unsigned char* l_vec, const unsigned char* c_vec, size_t h_len, const unsigned char* mu, size_t n) {
    if (!e) return s2k_fail("secp256k1_bppp_commit_batch", "null engine");
    if (!commits33 || !gens33 || !n_vec || !l_vec || !c_vec || !mu || n_gens != g_len + h_len || n_gens == 0)
        return s2k_fail_arg("secp256k1_bppp_commit_batch", "illegal argument (ARG_CHECK)");
    if (n == 0) return 1;
    if (results) memset(results, 0, sizeof(int32_t) * n);
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    const size_t per = std::max<size_t>(1, e->max_lanes / (n_gens + 1));
    const size_t io = ws_need({33 * n, 4 * n, 33 * n_gens, 32 * g_len * n, 32 * h_len * n, 32 * h_len * n, 32 * n});
    if (!engine_workspace(e, bpc_ws_bytes(std::min(n, per), n_gens) + io)) return 0;
    hipStream_t st = e->stream;
    stream_guard sg(e, st);
    ws_carver c0{e->ws, bpc_ws_bytes(std::min(n, per), n_gens)};
    unsigned char* d_out = c0.take<unsigned char>(33 * n); int32_t* d_res = c0.take<int32_t>(n); unsigned char* d_g33 = c0.take<unsigned char>(33 * n_gens);
    unsigned char* d_nv = c0.take<unsigned char>(32 * g_len * n); unsigned char* d_lv = c0.take<unsigned char>(32 * h_len * n);
    unsigned char* d_cv = c0.take<unsigned char>(32 * h_len * n); unsigned char* d_mu = c0.take<unsigned char>(32 * n);
    HIPCHK(hipMemcpyAsync(d_g33, gens33, 33 * n_gens, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_nv, n_vec, 32 * g_len * n, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_lv, l_vec, 32 * h_len * n, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_cv, c_vec, 32 * h_len * n, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_mu, mu, 32 * n, hipMemcpyHostToDevice, st));
    if (!secp256k1_bppp_commit_batch_dev(e, nullptr, d_out, d_res, d_g33, gens33, n_gens, g_len, d_nv, d_lv, d_cv, h_len, d_mu, n)) return 0;
    HIPCHK(hipMemcpyAsync(commits33, d_out, 33 * n, hipMemcpyDeviceToHost, st));
    if (results) HIPCHK(hipMemcpyAsync(results, d_res, 4 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 1;
}

