#include "engine_internal.h"

// ------------------------------------------------------------------------------------------------------------
// K independent multi-scalar multiplications in ONE launch chain (s2k_ecmult_multi_many[_dev])
// ------------------------------------------------------------------------------------------------------------
// The reference meets many small sums one at a time: below ECMULT_PIPPENGER_THRESHOLD = 88 points it runs Strauss in batches
// (src/ecmult_impl.h:382-419, :848-855), above it one bucket pass per call (:516-591) -- bench_ecmult's own shape is 1 024-term sums
// (src/bench_ecmult.c:262-276, :362-371).  One such sum on this machine is a chain of latency-bound launches (~0.4 ms whatever
// its size, msm.h); K of them side by side are a throughput problem again, so this path is organised for WORK, not for depth:
//   k_mm_prep     1 lane / term            byte decode + GLV split, as for one sum (msm_prep_term); the K optional G terms ride at the end
//   k_mm_bin      1 workgroup / (sum, window): digits counted in an LDS histogram, scanned, and the references scattered into the
//                                          (sum, window)'s own compact region -- an exact counting sort (no fixed-capacity regions,
//                                          hence nothing that can overflow and no exact path to queue)
//   k_mm_buckets  1 lane / bucket          the bucket's references summed (lean XYZZ accumulation, msm.h), the sum scaled by the
//                                          bucket's weight, a tree over the workgroup: per lane at full occupancy this is the cheapest
//                                          form of the window total (the single-sum path trades work for depth instead: k_msm_slices)
//   k_mm_combine  1 wavefront / sum        Horner over the windows in the wave-cooperative arithmetic (cofield.h) + the affine result
// The window width minimises the work of one sum of the largest size in the batch.  Sums above MM_MAX_TERMS go through the single-sum
// path one after the other (they fill the machine by themselves).
#define MM_MAX_TERMS 8192u
#define MM_BIN_THREADS 256
#define MM_RUN_MAX_C 6u            /* window totals by running sums (k_mm_window_run) up to this width, by weights + trees above */

// sub / top_vals: the top window of a 128-bit half only has 128 - c (windows - 1) live bits (plus the carry), i.e. a handful of digit values
// that would each collect a large share of the sum's terms in ONE bucket -- and a lane walks its bucket alone.  As in the single-sum path
// (k_msm_bin, msm_bucket_weight) every value v of the top window is spread over `sub` buckets, (v - 1) sub + (term index mod sub) + 1, all
// of weight v.  (First version without it: k_mm_sums 7-14 ms for 256 sums of 1 024 terms, the time of the one lane per sum that walked
// ~800 references; 1.2 ms at c = 8, whose top window only holds carries.  profiles/r06i_mm_break.txt)
struct mm_plan { u32 c, windows, nb, has_g, sub, top_vals; unsigned long long n_terms, n_sums; };
static void mm_plan_fill(mm_plan& P, u32 c, u32 has_g, size_t n_terms, size_t n_sums) {
    P.c = c; P.windows = (129 + c - 1) / c; P.nb = (1u << (c - 1)) + 1u; P.has_g = has_g; P.n_terms = n_terms; P.n_sums = n_sums;
    const u32 top_bits = 128u - c * (P.windows - 1);
    P.top_vals = top_bits >= c - 1 ? P.nb - 1 : (1u << top_bits);
    P.sub = 1; while (P.sub * 2 * P.top_vals <= P.nb - 1) P.sub *= 2;
}
__host__ __device__ __forceinline__ u32 mm_bucket_weight(const mm_plan& P, u32 w, u32 b) { return (w + 1 == P.windows && b) ? (b - 1u) / P.sub + 1u : b; }

// terms of sum s: [off[s], off[s + 1]) of the caller's arrays, plus its G term (record n_terms + s) when the call has them
__device__ __forceinline__ u32 mm_sum_terms(const unsigned long long* off, u32 s, u32 has_g) { return (u32)(off[s + 1] - off[s]) + has_g; }
// first reference slot of (sum s, window w): the regions are packed in (sum, window) order, 2 slots per term and window
__device__ __forceinline__ size_t mm_region(const unsigned long long* off, const mm_plan& P, u32 s, u32 w) {
    const size_t tbase = (size_t)off[s] + (size_t)s * P.has_g, tcount = (size_t)(off[s + 1] - off[s]) + P.has_g;
    return 2 * ((size_t)P.windows * tbase + (size_t)w * tcount);
}
__device__ __forceinline__ size_t mm_term_index(const unsigned long long* off, const mm_plan& P, u32 s, u32 j) {      // j-th term of sum s -> record index
    const u32 n = (u32)(off[s + 1] - off[s]);
    return j < n ? (size_t)off[s] + j : (size_t)P.n_terms + s;
}

__global__ void __launch_bounds__(256)
k_mm_prep(u32* term, u32* halves, const unsigned char* g_sc, const unsigned char* sc, const unsigned char* pt, const unsigned char* pt_inf, mm_plan P) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)P.n_terms + (P.has_g ? (size_t)P.n_sums : 0);
    if (i >= nt) return;
    const int isg = i >= P.n_terms;
    const unsigned char* ks = isg ? g_sc + 32 * (i - P.n_terms) : sc + 32 * i;
    const int pinf = isg ? 0 : (pt_inf ? pt_inf[i] != 0 : 0);
    if (((((size_t)sc) | ((size_t)pt) | ((size_t)g_sc)) & 15u) == 0u) msm_prep_term_aligned(term + i * MSM_TERM_WORDS, halves + i * MSM_HALF_WORDS, ks, isg ? sc : pt + 64 * i, pinf, isg);
    else msm_prep_term(term + i * MSM_TERM_WORDS, halves + i * MSM_HALF_WORDS, ks, isg ? sc : pt + 64 * i, pinf, isg);
}
// bucket of digit magnitude v (>= 1) of term j in window w.  (v <= top_vals in the top window: both halves are below 2^128,
// scalar_impl.h:183-285; the clamp only keeps an impossible value inside the bucket array)
__device__ __forceinline__ u32 mm_bucket_of(const mm_plan& P, u32 w, u32 v, u32 j) {
    if (w + 1 != P.windows) return v;
    v = v < P.top_vals ? v : P.top_vals;
    return (v - 1u) * P.sub + (j & (P.sub - 1u)) + 1u;
}
// exact counting sort of one (sum, window): bucket b's references end up at region + boff[b] .. + boff[b + 1]
__global__ void __launch_bounds__(MM_BIN_THREADS)
k_mm_bin(u32* __restrict__ refs, u32* __restrict__ boff, const u32* __restrict__ halves, const unsigned long long* __restrict__ off, mm_plan P) {
    __shared__ u32 s_cnt[520], s_pos[520], s_wave[8];
    const u32 s = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    const u32 tcount = mm_sum_terms(off, s, P.has_g);
    msm_plan pl = msm_plan_for(P.c);
    msm_wconst wc; msm_window_const(wc, w, P.c);
    for (u32 b = tid; b < 520; b += MM_BIN_THREADS) s_cnt[b] = 0;
    __syncthreads();
    for (u32 j = tid; j < tcount; j += MM_BIN_THREADS) {
        const u32* h = halves + mm_term_index(off, P, s, j) * MSM_HALF_WORDS;
#pragma unroll
        for (int half = 0; half < 2; half++) { const u32 key = msm_key_at(h, half, 0, wc, pl); if (key) atomicAdd(&s_cnt[mm_bucket_of(P, w, key >> 1, j)], 1u); }
    }
    __syncthreads();
    // exclusive scan of the nb <= 513 counters: three per lane, shuffle scan per wavefront, the wavefronts' totals by lane 0
    u32 c[3], sum = 0;
#pragma unroll
    for (int q = 0; q < 3; q++) { const u32 b = 3u * tid + (u32)q; c[q] = b < P.nb ? s_cnt[b] : 0u; sum += c[q]; }
    u32 inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 x = (u32)__shfl_up((int)inc, d, 64); if ((tid & 63u) >= (u32)d) inc += x; }
    if ((tid & 63u) == 63u) s_wave[tid >> 6] = inc;
    __syncthreads();
    if (tid == 0) { u32 run = 0; for (int q = 0; q < MM_BIN_THREADS / 64; q++) { const u32 x = s_wave[q]; s_wave[q] = run; run += x; } }
    __syncthreads();
    u32 run = s_wave[tid >> 6] + inc - sum;
    u32* const bo = boff + ((size_t)s * P.windows + w) * (P.nb + 1);
#pragma unroll
    for (int q = 0; q < 3; q++) { const u32 b = 3u * tid + (u32)q; if (b < P.nb) { s_pos[b] = run; bo[b] = run; } run += c[q]; if (b + 1 == P.nb) bo[P.nb] = run; }
    __syncthreads();
    u32* const dst = refs + mm_region(off, P, s, w);
    for (u32 j = tid; j < tcount; j += MM_BIN_THREADS) {
        const size_t ti = mm_term_index(off, P, s, j);
        const u32* h = halves + ti * MSM_HALF_WORDS;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const u32 key = msm_key_at(h, half, 0, wc, pl);
            if (key) { const u32 at = atomicAdd(&s_pos[mm_bucket_of(P, w, key >> 1, j)], 1u); dst[at] = (u32)(ti << 2) | ((u32)half << 1) | (key & 1u); }
        }
    }
}
// Bucket sums and bucket weights are separate launches, one lane per bucket of the whole batch, and the trees are the segmented tree sum
// the other families use (k_gej_reduce).  Fused into one kernel (sums, weight, tree per workgroup -- the first version) the three phases'
// code, ~80 KB with the exact accumulation and the complete addition inlined twice, did not fit the instruction cache two CUs share, and
// with workgroups of one dispatch in different phases the kernel ran 4-12x slower than its instruction count (profiles/r06h_mm_break.txt:
// 10.6 ms at c = 7, 2.4 ms at c = 8 where half the lanes idle, for 0.8 ms worth of additions).
__device__ __forceinline__ void mm_bucket_coords(u32& s, u32& w, u32& b, size_t g, const mm_plan& P) {
    const size_t sw = g / P.nb; b = (u32)(g - sw * P.nb); s = (u32)(sw / P.windows); w = (u32)(sw - (size_t)s * P.windows);
}
__global__ void __launch_bounds__(256, 2)
k_mm_sums(u32* out28, const u32* refs, const u32* boff, const u32* term, const unsigned long long* off, mm_plan P, size_t nbuckets) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nbuckets) return;
    u32 s, w, b; mm_bucket_coords(s, w, b, g, P);
    gej o; gej_set_infinity(o);
    if (b) {
        const u32* bo = boff + ((size_t)s * P.windows + w) * (P.nb + 1);
        const u32 lo = bo[b], hi = bo[b + 1];
        if (hi > lo) {
            const size_t first = mm_region(off, P, s, w);
            if (!msm_sum_refs_lean(o, refs, first + lo, first + hi, term)) msm_sum_refs(o, refs, first + lo, first + hi, term);
        }
    }
    gej_store28(out28 + g * 28, o);
}
__global__ void __launch_bounds__(256, 2)
k_mm_weights(u32* io28, mm_plan P, size_t nbuckets) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nbuckets) return;
    u32 s, w, b; mm_bucket_coords(s, w, b, g, P);
    const u32 weight = mm_bucket_weight(P, w, b);
    gej v, o; gej_load28(v, io28 + g * 28);
    if (weight < 2 || v.inf) return;                              // weight 1, unused slot 0, empty bucket: nothing to do
    msm_scale(o, v, weight);
    gej_store28(io28 + g * 28, o);
}
// Narrow windows (c <= 6, the widths of sums of a few dozen to a few hundred terms): the window total as the reference computes it, a running
// sum from the highest bucket down (src/ecmult_impl.h:581-588: running += bucket; total += running -- two additions per bucket and no
// multiplication by the weight), one lane per (sum, window).  With tens of thousands of (sum, window) pairs the lanes are there, and a
// 64-lane tree per 8- or 16-bucket window was 4.1 ms of a 7 ms call for 4 096 sums of 64 terms.  In the top window `sub` consecutive
// buckets share a weight: the total takes the running sum once per weight.
__global__ void __launch_bounds__(256, 2)
k_mm_window_run(u32* wsum28, const u32* buckets28, mm_plan P, size_t nsw) {
    const size_t sw = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sw >= nsw) return;
    const u32 w = (u32)(sw % P.windows);
    const u32 sub = (w + 1 == P.windows) ? P.sub : 1u;
    gej run, tot; gej_set_infinity(run); gej_set_infinity(tot);
    const u32* base = buckets28 + sw * P.nb * 28;
    for (u32 b = P.nb - 1; b >= 1; b--) {
        gej v, t; gej_load28(v, base + (size_t)b * 28);
        gej_add_var(t, run, v); run = t;
        if (((b - 1u) & (sub - 1u)) == 0u) { gej_add_var(t, tot, run); tot = t; }
    }
    gej_store28(wsum28 + sw * 28, tot);
}
__global__ void __launch_bounds__(64)
k_mm_combine(unsigned char* r_xy, int32_t* r_inf, const u32* wsum28, mm_plan P) {
    const u32 s = blockIdx.x;
    msm_plan pl = msm_plan_for(P.c);
    gej r; msm_combine(r, wsum28 + (size_t)s * P.windows * 28, pl);
    if (threadIdx.x) return;
    unsigned char* o = r_xy + 64 * (size_t)s;
    if (r.inf) { for (int k = 0; k < 64; k++) o[k] = 0; } else { ge a; ge_set_gej(a, r); ge_store_b64(o, a); }
    r_inf[s] = r.inf;
}

// window width: least work for one sum of n terms -- W windows x (2 n lean bucket additions of ~1 600 instructions + 2^(c-1) buckets x
// (c - 1 doublings + up to c - 1 additions for the weight, ~2 tree additions); narrow windows: two running-sum additions per bucket)
static u32 mm_pick_c(size_t n) {
#ifdef S2K_DIAG
    if (const char* v = getenv("S2K_MM_C")) { const int c = atoi(v); if (c >= 4 && c <= 10) return (u32)c; }
#endif
    double best = 0; u32 bc = 4;
    for (u32 c = 4; c <= 10; c++) {
        const double W = (double)((129 + c - 1) / c), nb = (double)(1u << (c - 1));
        const double cost = W * (2.0 * (double)n * 1600.0 + nb * (c <= MM_RUN_MAX_C ? 2.0 * 2400.0 : (double)(c - 1) * 3400.0 + 4800.0));
        if (c == 4 || cost < best) { best = cost; bc = c; }
    }
    return bc;
}

// workspace of one call beyond `front` bytes of staged inputs (sized once, up front: growing the workspace moves it)
static size_t mm_ws_bytes(const s2k_engine* e, const uint64_t* off_host, size_t K, u32 has_g) {
    const size_t N = (size_t)off_host[K];
    size_t nmax = 0;
    for (size_t s = 0; s < K; s++) nmax = std::max(nmax, (size_t)(off_host[s + 1] - off_host[s]));
    if (nmax > MM_MAX_TERMS) {
        size_t need = 0;
        for (size_t s = 0; s < K; s++) { const size_t nt = (size_t)(off_host[s + 1] - off_host[s]) + has_g; need = std::max(need, msm_ws_bytes(e, nt + 1, engine_msm_plan(e, nt ? nt : 1))); }
        return need;
    }
    const u32 c = mm_pick_c(nmax + has_g), W = (129 + c - 1) / c, nb = (1u << (c - 1)) + 1u;
    const size_t nt = N + (has_g ? K : 0), nsw = K * W;
    return ws_need({(K + 1) * 8, nt * MSM_TERM_WORDS * 4, nt * MSM_HALF_WORDS * 4, 2 * (size_t)W * nt * 4 + 64, nsw * (nb + 1) * 4, nsw * nb * 28 * 4, (nsw + 64) * 28 * 4, (nsw + 64) * 28 * 4});
}
// offsets (host, K + 1 entries) are read before the call returns; everything else is stream-ordered.  dev = 1: device pointers.
static int mm_impl(s2k_engine* e, hipStream_t st, size_t front, unsigned char* d_rxy, int32_t* d_rinf, const unsigned char* d_g, const unsigned char* d_sc,
                   const unsigned char* d_pt, const unsigned char* d_inf, const uint64_t* off_host, size_t K) {
    const size_t N = (size_t)off_host[K];
    size_t nmax = 0;
    for (size_t s = 0; s < K; s++) nmax = std::max(nmax, (size_t)(off_host[s + 1] - off_host[s]));
    const u32 has_g = d_g ? 1u : 0u;
    if (!engine_workspace(e, front + mm_ws_bytes(e, off_host, K, has_g))) return 0;
    if (nmax > MM_MAX_TERMS) {
        // large sums: the single-sum path, one after the other
        for (size_t s = 0; s < K; s++) {
            const size_t lo = (size_t)off_host[s], n = (size_t)(off_host[s + 1] - off_host[s]), nt = n + has_g;
            if (nt > msm_max_terms(e)) return s2k_fail("s2k_ecmult_multi_many", "a sum of this call exceeds what one launch indexes: use s2k_ecmult_multi for it");
            ws_carver c{e->ws, front}; u32* res = nullptr;
            const msm_out out{d_rxy + 64 * s, d_rinf + s, nullptr};
            if (!msm_launch(e, st, c, &res, d_g ? d_g + 32 * s : nullptr, d_sc + 32 * lo, d_pt + 64 * lo, d_inf ? d_inf + lo : nullptr, n, 0, 1, nullptr, &out)) return 0;
        }
        return 1;
    }
    mm_plan P; mm_plan_fill(P, mm_pick_c(nmax + has_g), has_g, N, K);
    const size_t nt = N + (has_g ? K : 0);
    if (nt >= (size_t(1) << 30)) return s2k_fail("s2k_ecmult_multi_many", "more than 2^30 terms in one call");
    const size_t nsw = K * P.windows, nbk = nsw * P.nb;
    if (nsw >= (size_t(1) << 31) || nbk >= (size_t(1) << 38)) {
        // (grid dimensions) -- split the batch
        const size_t half = K / 2;
        if (half == 0) return s2k_fail("s2k_ecmult_multi_many", "batch too large");
        std::vector<uint64_t> o2(off_host + half, off_host + K + 1);
        const uint64_t base = o2[0]; for (auto& v : o2) v -= base;
        if (!mm_impl(e, st, front, d_rxy, d_rinf, d_g, d_sc, d_pt, d_inf, off_host, half)) return 0;
        return mm_impl(e, st, front, d_rxy + 64 * half, d_rinf + half, d_g ? d_g + 32 * half : nullptr, d_sc + 32 * base, d_pt + 64 * base, d_inf ? d_inf + base : nullptr, o2.data(), K - half);
    }
    ws_carver c{e->ws, front};
    unsigned long long* d_off = c.take<unsigned long long>(K + 1);
    u32* term = c.take<u32>(nt * MSM_TERM_WORDS); u32* halves = c.take<u32>(nt * MSM_HALF_WORDS);
    u32* refs = c.take<u32>(2 * (size_t)P.windows * nt + 16); u32* boff = c.take<u32>(nsw * (P.nb + 1));
    u32* buckets = c.take<u32>(nbk * 28); u32* bufA = c.take<u32>((nsw + 64) * 28); u32* bufB = c.take<u32>((nsw + 64) * 28);
    HIPCHK(hipMemcpyAsync(d_off, off_host, (K + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(e->ev_fork, st));                     // (the host offset array must have been consumed before the call returns)
    if (nt) hipLaunchKernelGGL(k_mm_prep, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, term, halves, d_g, d_sc, d_pt, d_inf, P);
    hipLaunchKernelGGL(k_mm_bin, dim3((unsigned)K, P.windows), dim3(MM_BIN_THREADS), 0, st, refs, boff, (const u32*)halves, (const unsigned long long*)d_off, P);
    HIPCHK(hipEventRecord(e->ev[2], st));
    hipLaunchKernelGGL(k_mm_sums, dim3((unsigned)((nbk + 255) / 256)), dim3(256), 0, st, buckets, (const u32*)refs, (const u32*)boff, (const u32*)term, (const unsigned long long*)d_off, P, nbk);
    HIPCHK(hipEventRecord(e->ev[3], st));
    const u32* wsum = bufA;
    if (P.c <= MM_RUN_MAX_C) hipLaunchKernelGGL(k_mm_window_run, dim3((unsigned)((nsw + 255) / 256)), dim3(256), 0, st, bufA, (const u32*)buckets, P, nsw);
    else {
        hipLaunchKernelGGL(k_mm_weights, dim3((unsigned)((nbk + 255) / 256)), dim3(256), 0, st, buckets, P, nbk);
        wsum = launch_gej_reduce(st, buckets, bufA, bufB, (u32)nsw, P.nb);
    }
    hipLaunchKernelGGL(k_mm_combine, dim3((unsigned)K), dim3(64), 0, st, d_rxy, d_rinf, wsum, P);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventSynchronize(e->ev_fork));
    return 1;
}
static int mm_check_offsets(const char* who, const uint64_t* off, size_t K) {
    if (!off) return s2k_fail_arg(who, "illegal argument (ARG_CHECK)");
    if (off[0] != 0) return s2k_fail_arg(who, "offsets must start at 0");
    for (size_t s = 0; s < K; s++) if (off[s + 1] < off[s]) return s2k_fail_arg(who, "offsets must not decrease");
    return 1;
}
extern "C" int s2k_ecmult_multi_many_dev(s2k_engine* e, void* stream, unsigned char* r_xy, int32_t* r_inf, const unsigned char* g_sc, const unsigned char* sc,
                                         const unsigned char* pt_xy, const unsigned char* pt_inf, const uint64_t* offsets_host, size_t n_sums) {
    if (!e) return s2k_fail("s2k_ecmult_multi_many_dev", "null engine");
    if (n_sums == 0) return 1;
    if (!r_xy || !r_inf) return s2k_fail_arg("s2k_ecmult_multi_many_dev", "illegal argument (ARG_CHECK)");
    if (!mm_check_offsets("s2k_ecmult_multi_many_dev", offsets_host, n_sums)) return 0;
    if (offsets_host[n_sums] && (!sc || !pt_xy)) return s2k_fail_arg("s2k_ecmult_multi_many_dev", "illegal argument (ARG_CHECK)");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    HIPCHK(hipEventRecord(e->ev[0], st));
    if (!mm_impl(e, st, 0, r_xy, r_inf, g_sc, sc, pt_xy, pt_inf, offsets_host, n_sums)) return 0;
    HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
}
extern "C" int s2k_ecmult_multi_many(s2k_engine* e, unsigned char* r_xy, int32_t* r_inf, const unsigned char* g_sc, const unsigned char* sc,
                                     const unsigned char* pt_xy, const unsigned char* pt_inf, const uint64_t* offsets, size_t n_sums) {
    if (!e) return s2k_fail("s2k_ecmult_multi_many", "null engine");
    if (n_sums == 0) return 1;
    if (!r_xy || !r_inf) return s2k_fail_arg("s2k_ecmult_multi_many", "illegal argument (ARG_CHECK)");
    if (!mm_check_offsets("s2k_ecmult_multi_many", offsets, n_sums)) return 0;
    const size_t n = (size_t)offsets[n_sums], K = n_sums;
    if (n && (!sc || !pt_xy)) return s2k_fail_arg("s2k_ecmult_multi_many", "illegal argument (ARG_CHECK)");
    memset(r_inf, 0, sizeof(int32_t) * K);
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    const size_t front = ws_need({32 * n + 64, 64 * n + 64, n + 64, 32 * K + 64, 64 * K + 64, 4 * K + 64});
    hipStream_t st = e->stream;
    stream_guard sg(e, st);
    if (!engine_workspace(e, front + mm_ws_bytes(e, offsets, K, g_sc ? 1u : 0u))) return 0;      // (sized once: growing the workspace moves it)
    ws_carver c{e->ws, 0};
    unsigned char* d_sc = c.take<unsigned char>(32 * n + 64); unsigned char* d_pt = c.take<unsigned char>(64 * n + 64); unsigned char* d_inf = c.take<unsigned char>(n + 64);
    unsigned char* d_g = c.take<unsigned char>(32 * K + 64); unsigned char* d_r = c.take<unsigned char>(64 * K + 64); int32_t* d_ri = c.take<int32_t>(K + 16);
    if (n) {
        HIPCHK(hipMemcpyAsync(d_sc, sc, 32 * n, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_pt, pt_xy, 64 * n, hipMemcpyHostToDevice, st));
        if (pt_inf) HIPCHK(hipMemcpyAsync(d_inf, pt_inf, n, hipMemcpyHostToDevice, st));
    }
    if (g_sc) HIPCHK(hipMemcpyAsync(d_g, g_sc, 32 * K, hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(e->ev[0], st));
    if (!mm_impl(e, st, front, d_r, d_ri, g_sc ? d_g : nullptr, d_sc, d_pt, pt_inf ? d_inf : nullptr, offsets, K)) return 0;
    HIPCHK(hipEventRecord(e->ev[1], st));
    HIPCHK(hipMemcpyAsync(r_xy, d_r, 64 * K, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r_inf, d_ri, 4 * K, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 1;
}
