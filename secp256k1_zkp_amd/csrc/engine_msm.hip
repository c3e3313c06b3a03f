#include "engine_internal.h"

// ------------------------------------------------------------------------------------------------------------
// multi-scalar multiplication (msm.h)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_msm_prep(u32* term, u32* halves, const unsigned char* g_sc, const unsigned char* sc, const unsigned char* pt, const unsigned char* pt_inf, size_t n, size_t nt,
           u32* zero_a, u32 words_a, u32* zero_b, u32 words_b) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // the bucket counters and the flag words of this call are cleared here (they were two fill launches in front of every call)
    for (size_t z = i; z < words_a; z += (size_t)gridDim.x * blockDim.x) zero_a[z] = 0u;
    for (size_t z = i; z < words_b; z += (size_t)gridDim.x * blockDim.x) zero_b[z] = 0u;
    if (i >= nt) return;
    const int isg = (i == n);       // only when g_sc != NULL (nt == n + 1)
    const unsigned char* ks = isg ? g_sc : sc + 32 * i;
    const int pinf = isg ? 0 : (pt_inf ? pt_inf[i] != 0 : 0);
    // (the caller's arrays are plain byte arrays: the vector loads only when all three happen to be 16-byte aligned -- they are for every
    //  allocation of their own -- and then for every lane alike)
    if (((((size_t)sc) | ((size_t)pt) | ((size_t)g_sc)) & 15u) == 0u) msm_prep_term_aligned(term + i * MSM_TERM_WORDS, halves + i * MSM_HALF_WORDS, ks, isg ? sc : pt + 64 * i, pinf, isg);
    else msm_prep_term(term + i * MSM_TERM_WORDS, halves + i * MSM_HALF_WORDS, ks, isg ? sc : pt + 64 * i, pinf, isg);
}
// Binning: workgroup (chunk, window).  One sweep over the chunk's half-scalar records: the LDS histogram gives every digit its
// rank inside the workgroup, one global atomic per non-empty bucket reserves the workgroup's slots, then the references are
// written from the (bucket, rank) pairs kept in registers.
// refs layout: bucket k owns refs[k*cap .. k*cap + cap); gcnt[k] ends up as the bucket's full size even when it overflows.
// The top window only has 128 - c*(windows-1) live bits (both GLV halves are below 2^128, scalar_impl.h:183-285), so its
// few buckets are proportionally fuller: they get their own capacity.  Bucket k = w*nb + b starts at msm_region(k).
// The top window's few values would put n/2 points into each of a handful of buckets, and the depth of the partial-sum rounds follows
// the fullest region: so every value v of the top window is SPREAD over `sub` buckets, (v - 1) * sub + (term index mod sub) + 1 --
// all with the weight v (msm_bucket_weight) -- which brings the top window's regions down to the size of the others.
__host__ __device__ __forceinline__ u32 msm_bucket_weight(const msm_layout& L, const msm_plan& pl, u32 k) {      // k = w * nb + b, w local to the share
    const u32 b = k % pl.nb;
    const int top = (pl.w0 + k / pl.nb + 1 == pl.windows);
    return (top && b) ? (b - 1u) / L.sub + 1u : b;
}
// (w = window index inside the launch's share [pl.w0, pl.w0 + pl.wn); the top window, if the share has it, is its last one)
__device__ __forceinline__ size_t msm_region(const msm_layout& L, const msm_plan& pl, u32 w, u32 b) {
    return (pl.w0 + w + 1 < pl.windows) ? ((size_t)w * pl.nb + b) * L.cap : (size_t)(pl.wn - 1) * pl.nb * L.cap + (size_t)b * L.cap_top;
}
// WIDE (c = 14..16, the largest inputs): 2^(c-1) + 1 counters would not fit the LDS twice, so two 16-bit counters share a word (a chunk
// has at most 16 384 half-scalars, and a region of these plans fewer than 32 768 slots: msm_make_plan) and the (bucket, rank) pair in
// registers is 16 + 14 bits instead of 13 + 16.
#define MSM_BIN_THREADS 1024
template <int WIDE>
__global__ void __launch_bounds__(MSM_BIN_THREADS)
k_msm_bin(u32* __restrict__ refs, u32* __restrict__ gcnt, u32* __restrict__ flags, const u32* __restrict__ halves, size_t nt, msm_plan pl, msm_layout L, u32 chunk_dbg) {
    __shared__ u32 s_cnt[(WIDE ? 16385 : 4097) + 7];
    const u32 chunk = chunk_dbg & 0xFFFFFFu, dbg = chunk_dbg >> 24;      // dbg (S2K_MSM_BIN_DEBUG, diagnostic launches: results are meaningless): 1 no reference stores, 2 no global atomics, 4 no LDS atomics
    const u32 w = blockIdx.y, tid = threadIdx.x;
    const size_t t0 = (size_t)blockIdx.x * chunk;
    const size_t t1 = (t0 + chunk < nt) ? t0 + chunk : nt;
    for (u32 b = tid; b < (WIDE ? (pl.nb + 1u) / 2u : pl.nb); b += MSM_BIN_THREADS) s_cnt[b] = 0;
    msm_wconst wc; msm_window_const(wc, pl.w0 + w, pl.c);
    __syncthreads();
    // sweep: digit of every half-scalar of the chunk, rank inside the workgroup from the LDS histogram; the (bucket, rank, sign)
    // of the at most 8 terms x 2 halves a thread owns stay in registers (chunk <= 8 * MSM_BIN_THREADS)
    // (the kernel waits for memory three quarters of its time -- SQ_WAIT_ANY, profiles/r03f_msm_1048576_pmc.json -- so the records of four
    //  terms are requested before the first one is used: two round trips per lane instead of eight)
    u32 kv[8][2];
    int over = 0;
    const int top = (pl.w0 + w + 1 == pl.windows);
#pragma unroll
    for (int it0 = 0; it0 < 8; it0 += 4) {
        uint4 hv[4][3];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const size_t t = t0 + tid + (size_t)(it0 + j) * MSM_BIN_THREADS;
            const uint4* src = (const uint4*)(halves + (t < t1 ? t : t0) * MSM_HALF_WORDS);
#pragma unroll
            for (int q = 0; q < 3; q++) hv[j][q] = src[q];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int it = it0 + j;
            kv[it][0] = 0; kv[it][1] = 0;
            const size_t t = t0 + tid + (size_t)it * MSM_BIN_THREADS;
            if (t < t1) {
                u32 h[MSM_HALF_WORDS];
#pragma unroll
                for (int q = 0; q < 3; q++) { const uint4 v = hv[j][q]; h[4 * q] = v.x; h[4 * q + 1] = v.y; h[4 * q + 2] = v.z; h[4 * q + 3] = v.w; }
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const u32 key = msm_key_at(h, half, 0, wc, pl);          // window offset 0: local bucket index
                    if (key) {
                        u32 bkt = key >> 1;
                        if (top) {                                           // value v -> one of its `sub` buckets, by term index
                            if (bkt * L.sub > L.top_used - 1u) { over = 1; continue; }      // a value the top window cannot hold for a reduced half
                            bkt = (bkt - 1u) * L.sub + ((u32)t & (L.sub - 1u)) + 1u;
                        }
                        if (WIDE) {
                            const u32 sh = (bkt & 1u) * 16u;
                            const u32 rank = (atomicAdd(&s_cnt[bkt >> 1], 1u << sh) >> sh) & 0xFFFFu;
                            kv[it][half] = 0x40000000u | ((key & 1u) << 31) | (bkt << 14) | rank;
                        } else {
                            const u32 rank = (dbg & 4u) ? 0u : atomicAdd(&s_cnt[bkt], 1u); kv[it][half] = 0x40000000u | ((key & 1u) << 31) | (bkt << 16) | rank;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    if (WIDE) {
        for (u32 wd = tid; wd < (pl.nb + 1u) / 2u; wd += MSM_BIN_THREADS) {
            const u32 v = s_cnt[wd], c0 = v & 0xFFFFu, c1 = v >> 16;
            u32 b0 = c0 ? atomicAdd(&gcnt[w * pl.nb + 2u * wd], c0) : 0u;
            u32 b1 = c1 ? atomicAdd(&gcnt[w * pl.nb + 2u * wd + 1u], c1) : 0u;
            b0 = b0 < 49151u ? b0 : 49151u; b1 = b1 < 49151u ? b1 : 49151u;      // (beyond any region of these plans; base + rank stays inside 16 bits)
            s_cnt[wd] = b0 | (b1 << 16);
        }
    } else {
        // one global atomic per non-empty (workgroup, bucket); a lane's (up to five) atomics are all in flight before the first is awaited
        u32 cc[5], bb[5];
#pragma unroll
        for (int j = 0; j < 5; j++) { const u32 b = tid + (u32)j * MSM_BIN_THREADS; cc[j] = b < pl.nb ? s_cnt[b] : 0u; }
#pragma unroll
        for (int j = 0; j < 5; j++) { const u32 b = tid + (u32)j * MSM_BIN_THREADS; bb[j] = (cc[j] && !(dbg & 2u)) ? atomicAdd(&gcnt[w * pl.nb + b], cc[j]) : 0u; }
#pragma unroll
        for (int j = 0; j < 5; j++) { const u32 b = tid + (u32)j * MSM_BIN_THREADS; if (b < pl.nb) s_cnt[b] = bb[j]; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const size_t t = t0 + tid + (size_t)it * MSM_BIN_THREADS;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const u32 k = kv[it][half];
            if (k) {
                const u32 bkt = WIDE ? (k >> 14) & 0xFFFFu : (k >> 16) & 0x1FFFu;
                const u32 slot = WIDE ? ((s_cnt[bkt >> 1] >> ((bkt & 1u) * 16u)) & 0xFFFFu) + (k & 0x3FFFu) : s_cnt[bkt] + (k & 0xFFFFu);
                if (slot < (top ? L.cap_top : L.cap) && (!top || bkt < L.top_used)) { if (!(dbg & 1u)) refs[msm_region(L, pl, w, bkt) + slot] = (u32)(t << 2) | ((u32)half << 1) | (k >> 31); }
                else over = 1;
            }
        }
    }
    if (over) flags[0] = 1u;
}
// The single pass with its output ordered in LDS (c <= 13, from 2^15 terms).  Two thirds of k_msm_bin<0> are its lone 4-byte stores
// (profiles/r03g_msm_bin_parts.txt: one L2 request per reference).  Here a workgroup's references are first laid out by bucket in LDS --
// one 28-bit word each: bucket, term index inside the chunk, half, sign -- and then written with consecutive lanes on consecutive
// slots, so that the ~3 references a workgroup has for a bucket leave as one request.  6 144 terms per workgroup (48 KB of staging +
// counters: two workgroups per CU).
#define MSM_STAGED_PER_THREAD 6
#define MSM_STAGED_TERMS (MSM_STAGED_PER_THREAD * MSM_BIN_THREADS)
__global__ void __launch_bounds__(MSM_BIN_THREADS)
k_msm_bin_staged(u32* __restrict__ refs, u32* __restrict__ gcnt, u32* __restrict__ flags, const u32* __restrict__ halves, size_t nt, msm_plan pl, msm_layout L) {
    __shared__ u32 stage[2 * MSM_STAGED_TERMS];
    __shared__ u32 s_cnt[4104];                 // counts, then (in place) the buckets' offsets in `stage`; [4097]: the total
    __shared__ unsigned short s_gbase[4104];    // the workgroup's first slot in every bucket region
    __shared__ u32 s_wave[17];
    const u32 w = blockIdx.y, tid = threadIdx.x;
    const size_t t0 = (size_t)blockIdx.x * MSM_STAGED_TERMS;
    const size_t t1 = (t0 + MSM_STAGED_TERMS < nt) ? t0 + MSM_STAGED_TERMS : nt;
    for (u32 b = tid; b < 4104; b += MSM_BIN_THREADS) s_cnt[b] = 0;
    msm_wconst wc; msm_window_const(wc, pl.w0 + w, pl.c);
    __syncthreads();
    u32 kv[MSM_STAGED_PER_THREAD][2];           // valid << 30 | sign << 31 | bucket << 16 | rank
    int over = 0;
    const int top = (pl.w0 + w + 1 == pl.windows);
#pragma unroll
    for (int it0 = 0; it0 < MSM_STAGED_PER_THREAD; it0 += 3) {
        uint4 hv[3][3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const size_t t = t0 + tid + (size_t)(it0 + j) * MSM_BIN_THREADS;
            const uint4* src = (const uint4*)(halves + (t < t1 ? t : t0) * MSM_HALF_WORDS);
#pragma unroll
            for (int q = 0; q < 3; q++) hv[j][q] = src[q];
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int it = it0 + j;
            kv[it][0] = 0; kv[it][1] = 0;
            const size_t t = t0 + tid + (size_t)it * MSM_BIN_THREADS;
            if (t < t1) {
                u32 h[MSM_HALF_WORDS];
#pragma unroll
                for (int q = 0; q < 3; q++) { const uint4 v = hv[j][q]; h[4 * q] = v.x; h[4 * q + 1] = v.y; h[4 * q + 2] = v.z; h[4 * q + 3] = v.w; }
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const u32 key = msm_key_at(h, half, 0, wc, pl);
                    if (key) {
                        u32 bkt = key >> 1;
                        if (top) {
                            if (bkt * L.sub > L.top_used - 1u) { over = 1; continue; }
                            bkt = (bkt - 1u) * L.sub + ((u32)t & (L.sub - 1u)) + 1u;
                        }
                        const u32 rank = atomicAdd(&s_cnt[bkt], 1u);
                        kv[it][half] = 0x40000000u | ((key & 1u) << 31) | (bkt << 16) | rank;
                    }
                }
            }
        }
    }
    __syncthreads();
    // reservation + exclusive scan: lane t owns counters 4t .. 4t+3 (lane 1023 also the last one, 4096)
    u32 c[5], g[5], sum = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) { const u32 b = 4u * tid + (u32)j; c[j] = (j < 4 || tid == MSM_BIN_THREADS - 1) ? s_cnt[b] : 0u; sum += c[j]; }
#pragma unroll
    for (int j = 0; j < 5; j++) { const u32 b = 4u * tid + (u32)j; g[j] = (c[j] && b < pl.nb) ? atomicAdd(&gcnt[w * pl.nb + b], c[j]) : 0u; }
    u32 inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 x = (u32)__shfl_up((int)inc, d, 64); if ((tid & 63u) >= (u32)d) inc += x; }
    if ((tid & 63u) == 63u) s_wave[tid >> 6] = inc;
    __syncthreads();
    if (tid == 0) { u32 run = 0; for (int q = 0; q < 16; q++) { const u32 x = s_wave[q]; s_wave[q] = run; run += x; } s_wave[16] = run; }
    __syncthreads();
    u32 run = s_wave[tid >> 6] + inc - sum;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const u32 b = 4u * tid + (u32)j;
        if (j < 4 || tid == MSM_BIN_THREADS - 1) { s_cnt[b] = run; s_gbase[b] = (unsigned short)(g[j] < 65535u ? g[j] : 65535u); run += c[j]; }
    }
    if (tid == MSM_BIN_THREADS - 1) s_cnt[4097] = run;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < MSM_STAGED_PER_THREAD; it++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const u32 k = kv[it][half];
            if (k) {
                const u32 bkt = (k >> 16) & 0x1FFFu;
                stage[s_cnt[bkt] + (k & 0xFFFFu)] = (bkt << 15) | ((tid + (u32)it * MSM_BIN_THREADS) << 2) | ((u32)half << 1) | (k >> 31);
            }
        }
    }
    __syncthreads();
    const u32 total = s_cnt[4097], cap = top ? L.cap_top : L.cap;
    for (u32 i = tid; i < total; i += MSM_BIN_THREADS) {
        const u32 v = stage[i], bkt = v >> 15, slot = (u32)s_gbase[bkt] + (i - s_cnt[bkt]);
        if (slot < cap && (!top || bkt < L.top_used)) refs[msm_region(L, pl, w, bkt) + slot] = (u32)((t0 + ((v >> 2) & 8191u)) << 2) | (v & 3u);
        else over = 1;
    }
    if (over) flags[0] = 1u;
}
// Two-pass binning for the widest windows (c = 14..16: from 2^22 terms).  With 2^15 buckets per window a workgroup of the single-pass
// form has one or two references per bucket, every one a lone 4-byte store into its own region, and 2^15 counters to clear, reserve
// and scan per workgroup.  Here the references first go, as (reference, bucket) pairs, into COARSE bins of 2^shift adjacent buckets --
// the half-scalar records are read once for all windows -- and then one workgroup per coarse bin distributes its pairs over the bin's
// buckets, tile by tile through LDS, so that a bucket region is written in runs by exactly one workgroup; the bucket counts come out of
// that workgroup's running totals (no global atomics per bucket).  2^24 terms: 5.25 -> 3.3 ms; at 2^20 (c = 13) the single pass is as fast.
struct msm_coarse { u32 shift, nco, cap, cap_top; };          // nco bins per window; cap / cap_top pairs per bin (other windows / top window)
__host__ __device__ __forceinline__ size_t msm_coarse_region(const msm_coarse& C, const msm_plan& pl, u32 w, u32 co) {
    return (pl.w0 + w + 1 < pl.windows) ? ((size_t)w * C.nco + co) * C.cap : (size_t)(pl.wn - 1) * C.nco * C.cap + (size_t)co * C.cap_top;
}
// exclusive prefix over cnt[0..n) (n <= 320) by the first wavefront: five consecutive bins per lane, a shuffle scan across the lanes
__device__ __forceinline__ void msm_scan320(u32* off, const u32* cnt, u32 n) {
    if (threadIdx.x < 64) {
        const u32 l = threadIdx.x;
        u32 v[5], sum = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) { const u32 b = l * 5u + (u32)j; v[j] = b < n ? cnt[b] : 0u; sum += v[j]; }
        u32 inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 x = (u32)__shfl_up((int)inc, d, 64); if (l >= (u32)d) inc += x; }
        u32 run = inc - sum;
#pragma unroll
        for (int j = 0; j < 5; j++) { const u32 b = l * 5u + (u32)j; if (b <= n) off[b] = run; run += v[j]; }
    }
}
// pass 1: a workgroup takes MSM_COARSE_TERMS terms and ALL windows of the share -- the half-scalar records are read once (the single-pass
// form reads them once per window) and stay in registers for both sweeps: count, reserve (one global atomic per non-empty (window,
// coarse bin)), write.  (Measured variants, 2^24 terms: this one 2.06 ms; digits from one addition per half, msm_sum_full: 2.46;
// the pairs ordered in LDS and written in runs: 2.94 -- ten windows x five barriers of a 1 024-lane workgroup cost more than the
// runs save; one workgroup per window with the ranks kept in registers: 3.27.)
#define MSM_COARSE_PER_THREAD 2
#define MSM_COARSE_TERMS (MSM_COARSE_PER_THREAD * MSM_BIN_THREADS)
#define MSM_COARSE_LDS (9 * 257 + 7)          /* wn * nco: 9 x 257 (c = 16), 8 x 257 (c = 17), 9 x 129 (c = 15), 10 x 65 (c = 14) */
__global__ void __launch_bounds__(MSM_BIN_THREADS)
k_msm_bin_coarse(unsigned long long* __restrict__ pairs, u32* __restrict__ ccnt, u32* __restrict__ flags, const u32* __restrict__ halves, size_t nt,
                 msm_plan pl, msm_layout L, msm_coarse C) {
    __shared__ u32 s_cnt[MSM_COARSE_LDS];
    const u32 tid = threadIdx.x, nbins = pl.wn * C.nco;
    const size_t t0 = (size_t)blockIdx.x * MSM_COARSE_TERMS;
    for (u32 b = tid; b < nbins; b += MSM_BIN_THREADS) s_cnt[b] = 0;
    u32 h[MSM_COARSE_PER_THREAD][MSM_HALF_WORDS];
#pragma unroll
    for (int it = 0; it < MSM_COARSE_PER_THREAD; it++) {
        const size_t t = t0 + tid + (size_t)it * MSM_BIN_THREADS;
        if (t < nt) {
            const uint4* src = (const uint4*)(halves + t * MSM_HALF_WORDS);
#pragma unroll
            for (int q = 0; q < 3; q++) { const uint4 v = src[q]; h[it][4 * q] = v.x; h[it][4 * q + 1] = v.y; h[it][4 * q + 2] = v.z; h[it][4 * q + 3] = v.w; }
        } else {
#pragma unroll
            for (int q = 0; q < MSM_HALF_WORDS; q++) h[it][q] = 0;            // flags word 0: inactive
        }
    }
    __syncthreads();
    int over = 0;
    // sweep 1 counts, sweep 2 (after the reservation) takes its slots in the same histogram, which then holds the bins' bases
    for (int sweep = 0; sweep < 2; sweep++) {
#pragma unroll 1
        for (u32 w = 0; w < pl.wn; w++) {
            msm_wconst wc; msm_window_const(wc, pl.w0 + w, pl.c);
            const int top = (pl.w0 + w + 1 == pl.windows);
            const u32 cap = top ? C.cap_top : C.cap;
#pragma unroll
            for (int it = 0; it < MSM_COARSE_PER_THREAD; it++) {
                const size_t t = t0 + tid + (size_t)it * MSM_BIN_THREADS;
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const u32 key = msm_key_at(h[it], half, 0, wc, pl);
                    if (!key) continue;
                    u32 bkt = key >> 1;
                    if (top) {
                        if (bkt * L.sub > L.top_used - 1u) { over = 1; continue; }
                        bkt = (bkt - 1u) * L.sub + ((u32)t & (L.sub - 1u)) + 1u;
                    }
                    const u32 co = bkt >> C.shift;
                    const u32 slot = atomicAdd(&s_cnt[w * C.nco + co], 1u);
                    if (sweep) {
                        const u32 ref = (u32)(t << 2) | ((u32)half << 1) | (key & 1u);
                        if (slot < cap) pairs[msm_coarse_region(C, pl, w, co) + slot] = (unsigned long long)ref | ((unsigned long long)bkt << 32);
                        else over = 1;
                    }
                }
            }
        }
        if (sweep == 0) {
            __syncthreads();
            for (u32 b = tid; b < nbins; b += MSM_BIN_THREADS) {
                const u32 c = s_cnt[b];
                s_cnt[b] = c ? atomicAdd(&ccnt[b], c) : 0u;
            }
            __syncthreads();
        }
    }
    if (over) flags[0] = 1u;
}
// pass 2: one workgroup per (coarse bin, window), tiles of MSM_FINE_TILE pairs: counted per bucket, ordered by bucket in LDS, written out
// in runs behind what the earlier tiles put into the bucket's region.  The bucket counts come out of the running totals.
#define MSM_FINE_THREADS 256
#define MSM_FINE_TILE (8 * MSM_FINE_THREADS)
__global__ void __launch_bounds__(MSM_FINE_THREADS)
k_msm_bin_fine(u32* __restrict__ refs, u32* __restrict__ gcnt, u32* __restrict__ flags, const unsigned long long* __restrict__ pairs, const u32* __restrict__ ccnt,
               msm_plan pl, msm_layout L, msm_coarse C) {
    __shared__ u32 stage[MSM_FINE_TILE];
    __shared__ unsigned char stage_f[MSM_FINE_TILE];
    __shared__ u32 s_cnt[264], s_loff[264], s_tot[264];          // (up to 256 buckets per coarse bin: c = 17)
    const u32 co = blockIdx.x, w = blockIdx.y, tid = threadIdx.x, nfine = 1u << C.shift, first = co << C.shift;
    if (tid < nfine) s_tot[tid] = 0;
    const int top = (pl.w0 + w + 1 == pl.windows);
    const u32 ccap = top ? C.cap_top : C.cap, cap = top ? L.cap_top : L.cap;
    u32 n = ccnt[w * C.nco + co]; n = n < ccap ? n : ccap;
    const unsigned long long* src = pairs + msm_coarse_region(C, pl, w, co);
    int over = 0;
    for (u32 i0 = 0; i0 < n; i0 += MSM_FINE_TILE) {
        if (tid < nfine) s_cnt[tid] = 0;
        unsigned long long pr[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { const u32 i = i0 + (u32)j * MSM_FINE_THREADS + tid; pr[j] = i < n ? src[i] : 0ull; }
        __syncthreads();
        u32 rank[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { const u32 bkt = (u32)(pr[j] >> 32); rank[j] = bkt ? atomicAdd(&s_cnt[bkt - first], 1u) : 0u; }      // (bucket 0 never occurs)
        __syncthreads();
        msm_scan320(s_loff, s_cnt, nfine);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u32 bkt = (u32)(pr[j] >> 32);
            if (bkt) { const u32 f = bkt - first, at = s_loff[f] + rank[j]; stage[at] = (u32)pr[j]; stage_f[at] = (unsigned char)f; }
        }
        __syncthreads();
        const u32 total = s_loff[nfine];
        for (u32 i = tid; i < total; i += MSM_FINE_THREADS) {
            const u32 f = stage_f[i], bkt = first + f, slot = s_tot[f] + (i - s_loff[f]);
            if (slot < cap && (!top || bkt < L.top_used)) refs[msm_region(L, pl, w, bkt) + slot] = stage[i];
            else over = 1;
        }
        __syncthreads();
        if (tid < nfine) s_tot[tid] += s_cnt[tid];
    }
    __syncthreads();
    if (tid < nfine && first + tid < pl.nb) gcnt[w * pl.nb + first + tid] = s_tot[tid];
    if (over) flags[0] = 1u;
}
// exclusive scan of in[0..nk) into off[0..nk] (and a copy in cur if non-null): tiles of 1024, then the tile totals
__global__ void __launch_bounds__(256)
k_scan_tiles(u32* off, u32* tile_sum, const u32* in, u32 nk) {
    __shared__ u32 part[256];
    const u32 t = threadIdx.x, base = blockIdx.x * 1024 + t * 4;
    u32 v[4]; u32 s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = (base + k < nk) ? in[base + k] : 0u; s += v[k]; }
    part[t] = s;
    __syncthreads();
    for (u32 d = 1; d < 256; d <<= 1) {
        const u32 x = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    u32 run = part[t] - s;
#pragma unroll
    for (int k = 0; k < 4; k++) { if (base + k < nk) off[base + k] = run; run += v[k]; }
    if (t == 255) tile_sum[blockIdx.x] = part[255];
}
__global__ void __launch_bounds__(256)
k_scan_fix(u32* off, u32* cur, const u32* tile_sum, u32 nk) {
    u32 pre = 0;
    for (u32 b = 0; b < blockIdx.x; b++) pre += tile_sum[b];
    const u32 t = threadIdx.x, base = blockIdx.x * 1024 + t * 4;
#pragma unroll
    for (int k = 0; k < 4; k++) if (base + k < nk) { const u32 o = off[base + k] + pre; off[base + k] = o; if (cur) cur[base + k] = o; }
    if (blockIdx.x == gridDim.x - 1 && t == 0) off[nk] = pre + tile_sum[blockIdx.x];
}
// cap: a bucket's count can exceed its region (the binning pass then raised the overflow flag and the launch's result comes from
// the exact path); clamping keeps every later kernel inside the memory the references were written to
__global__ void k_msm_counts(u32* cnt_out, u32* cnt_clamped, const u32* cnt_in, u32 nk, u32 T, msm_layout L, msm_plan pl) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nk) {
        u32 c = cnt_in[k];
        if (cnt_clamped) {
            const int top = (pl.w0 + k / pl.nb + 1 == pl.windows);
            const u32 cap = top ? ((k % pl.nb) < L.top_used ? L.cap_top : 0u) : L.cap;
            c = c < cap ? c : cap; cnt_clamped[k] = c;
        }
        cnt_out[k] = (c + T - 1) / T;
    }
}
#ifndef MSM_R1_WAVES
#define MSM_R1_WAVES 2
#endif
// RUN_MAJOR (the widest windows, from 2^22 terms): lane m takes run j = m / nk of bucket k = m % nk instead of the m-th run in bucket order.  The
// binning passes fill a bucket's region roughly in term order (workgroups of consecutive terms reserve their slots as they arrive), so run j
// of EVERY bucket draws its operands from about the same 1/runs-th of the term records -- and with the lanes of one run index in flight
// together, the 64-byte operand gathers of the whole machine fall into a window of a few hundred MB that the Infinity Cache holds, where
// bucket order sprays them over all 2.1 GB of records (2^24 terms) at every moment.  Lanes beyond a bucket's last run leave at once; the
// partial sums land where bucket order puts them (off_out[k] + j), so the later rounds do not change.
template <int RUN_MAJOR>
__global__ void __launch_bounds__(256, MSM_R1_WAVES)
k_msm_round1(u32* out28, const u32* refs, const u32* off_in, const u32* cnt_in, msm_layout L, msm_plan pl, const u32* off_out, const u32* term, u32 nk, u32 T) {
    const size_t m0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 k, j, m;
    if (RUN_MAJOR) {
        j = (u32)(m0 / nk); k = (u32)(m0 - (size_t)j * nk);
        const u32 o = off_out[k];
        if (j >= off_out[k + 1] - o) return;
        m = o + j;
    } else {
        if (m0 >= off_out[nk]) return;
        m = (u32)m0; k = msm_find_key(off_out, nk, m); j = m - off_out[k];
    }
    // bucket k's references: its region of the fixed-capacity layout
    const size_t first = msm_region(L, pl, k / pl.nb, k % pl.nb); (void)off_in;
    const size_t start = first + (size_t)j * T, end = min(start + T, first + cnt_in[k]);
    gej o;
    if (!msm_sum_refs_lean(o, refs, start, end, term)) msm_sum_refs(o, refs, start, end, term);      // (exceptional additions: adversarial inputs only)
    gej_store28(out28 + (size_t)m * 28, o);
}
__global__ void __launch_bounds__(256, 2)
k_msm_roundN(u32* out28, const u32* in28, const u32* off_in, const u32* off_out, u32 nk, u32 T) {
    const u32 m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= off_out[nk]) return;
    const u32 k = msm_find_key(off_out, nk, m), j = m - off_out[k];
    const u32 start = off_in[k] + j * T, end = min(start + T, off_in[k + 1]);
    gej acc; gej_set_infinity(acc);
    for (u32 i = start; i < end; i++) { gej v, s; gej_load28(v, in28 + (size_t)i * 28); gej_add_var(s, acc, v); acc = s; }
    gej_store28(out28 + (size_t)m * 28, acc);
}
__global__ void __launch_bounds__(256, 2)
k_msm_finish(u32* bucket_out28, const u32* in28, const u32* off_last, u32 nk, msm_plan pl, msm_layout L) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nk) return;
    const u32 b = msm_bucket_weight(L, pl, k);                   // the bucket's digit value (the top window's values own several buckets each)
    gej v, o; gej_set_infinity(v);
    if (b != 0 && off_last[k + 1] > off_last[k]) gej_load28(v, in28 + (size_t)off_last[k] * 28);
    msm_scale(o, v, b);
    gej_store28(bucket_out28 + (size_t)k * 28, o);
}
// Small inputs (msm_make_plan keeps every bucket region within MSM_ONE_ROUND_CAP references): rounds, bucket weights and the first level
// of the window sums in ONE launch -- workgroup (chunk of 256 buckets, window), a lane per bucket: the bucket's references summed, the sum
// scaled by the bucket's weight, a tree over the workgroup.  Replaces counts + scan + round 1 + finish + the first tree level: these
// sizes are a chain of latency-bound stages, and what counts is how many there are.
__global__ void __launch_bounds__(256)
k_msm_small_windows(u32* out28, const u32* refs, const u32* gcnt, const u32* term, msm_plan pl, msm_layout L, u32 nchunks) {
    __shared__ u32 sh[256 * 28];
    const u32 w = blockIdx.y, t = threadIdx.x, b = 1u + blockIdx.x * 256u + t;
    gej o; gej_set_infinity(o);
    if (b < pl.nb) {
        const u32 k = w * pl.nb + b;
        const int top = (pl.w0 + w + 1 == pl.windows);
        const u32 cap = top ? (b < L.top_used ? L.cap_top : 0u) : L.cap;
        u32 cnt = gcnt[k]; cnt = cnt < cap ? cnt : cap;            // (an overflowing region raised the flag: the result comes from the exact path)
        if (cnt) {
            const size_t first = msm_region(L, pl, w, b);
            gej v;
            if (!msm_sum_refs_lean(v, refs, first, first + cnt, term)) msm_sum_refs(v, refs, first, first + cnt, term);
            msm_scale(o, v, msm_bucket_weight(L, pl, k));
        }
    }
    gej_store28(sh + t * 28, o);
    __syncthreads();
    for (u32 d = 128; d >= 1; d >>= 1) {
        if (t < d) {
            gej a, c, r; gej_load28(a, sh + t * 28); gej_load28(c, sh + (t + d) * 28);
            gej_add_var(r, a, c);
            gej_store28(sh + t * 28, r);
        }
        __syncthreads();
    }
    if (t < 28) out28[((size_t)w * nchunks + blockIdx.x) * 28 + t] = sh[t];
}
// segmented tree sum: block (seg, chunk) adds up items [chunk*per_block, ...) of segment `seg` (seg_len items each).  BS lanes per block:
// 256 for long segments; 64 (one wavefront, six tree levels instead of eight, a quarter of the LDS) when a segment has at most 256
// items -- the per-proof sums of the BP++ verifier (~80 terms), the per-window sums of a small MSM -- where most of a 256-lane
// block would run its tree levels on points at infinity.
template <int BS>
__global__ void __launch_bounds__(BS)
k_gej_reduce(u32* out28, const u32* in28, u32 seg_len, u32 per_block, u32 nchunks, const u32* gate) {
    __shared__ u32 sh[BS * 28];
    if (gate && *gate == 0) return;                  // exact-path launches do nothing unless the overflow flag is up
    const u32 seg = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks, t = threadIdx.x;
    const u32 lo = chunk * per_block, hi = min(lo + per_block, seg_len);
    gej acc; gej_set_infinity(acc);
    for (u32 k = lo + t; k < hi; k += BS) {
        gej v; gej_load28(v, in28 + ((size_t)seg * seg_len + k) * 28);
        gej s; gej_add_var(s, acc, v); acc = s;
    }
    gej_store28(sh + t * 28, acc);
    __syncthreads();
    for (u32 d = BS / 2; d >= 1; d >>= 1) {
        if (t < d) {
            gej a, b, s; gej_load28(a, sh + t * 28); gej_load28(b, sh + (t + d) * 28);
            gej_add_var(s, a, b);
            gej_store28(sh + t * 28, s);
        }
        __syncthreads();
    }
    if (t == 0) for (int i = 0; i < 28; i++) out28[(size_t)blockIdx.x * 28 + i] = sh[i];
}
// ---- the latency tail (round 6): bucket sums -> window sums without a per-bucket multiplication ------------------------------------------
// The reference's window total is a running sum (ecmult_impl.h:581-588: 2 * 2^(c-1) dependent additions); rounds 2-5 of this engine gave
// every bucket its own double-and-add by its weight (k_msm_finish) and then summed the window in trees (k_gej_reduce) -- ~20 + 12 + 6
// DEPENDENT per-lane point operations of ~7 us each on a machine that is empty by then (a lone wavefront issues one instruction per
// ~7.6 cycles), 0.34 ms of a 2 ms call at 2^20 terms.  Here the weights never multiply a point:
//     sum_b b B_b  =  sum_j 2^j T_j,   T_j = sum of the buckets whose weight has bit j
// k_msm_slices: one workgroup per (chunk of buckets, bit j, window) adds up its share of T_j -- the same masked tree for every bit, all bits
//   side by side (the machine has the lanes: 13 slices x 10 windows x 4 chunks x 256), depth R - 1 + 8 additions;
// k_msm_window_sums: one workgroup per window, one WAVEFRONT per bit, in the wave-cooperative arithmetic (cofield.h: a point addition
//   in ~2 us, a doubling in ~1 us): wave j adds the chunks of T_j, doubles the sum j times, and a four-level tree over the wavefronts
//   leaves the window sum.  ~25 us where the bucket weights + the second tree level took 0.25 ms.
// `off_last` non-null: bucket k's partial sum is record off_last[k] of in28 if off_last[k + 1] > off_last[k] (the partial-sum rounds'
// packed output); null: record k (k_msm_bucket_sums), infinity flagged in the record.
// record (28 words) of a cooperative point; every lane of the wavefront calls it
S2K_D void cgej_store28(u32* p28, const cgej& a, int inf) {
    const u32 l = co_abs();
    cfe y = a.y; cfe_norm_weak(y);
    if (l < 9) { p28[l] = inf ? 0u : a.x.v; p28[9 + l] = inf ? 0u : y.v; p28[18 + l] = inf ? 0u : a.z.v; }
    if (l == 0) p28[27] = (u32)inf;
}
// acc <- acc + v with the infinity flags carried beside the cooperative points (all wave-uniform)
S2K_D void cgej_acc(cgej& acc, int& acc_inf, const cgej& v, int v_inf) {
    if (v_inf) return;
    if (acc_inf) { acc = v; acc_inf = 0; return; }
    acc_inf = cgej_add(acc, v);
}
#define MSM_SLICE_MAX 16                 /* weights are below 2^16 (c <= 16) */
template <int R, int BS>
__global__ void __launch_bounds__(BS)
k_msm_slices(u32* q28, const u32* in28, const u32* off_last, msm_plan pl, msm_layout L, u32 nchunks, u32 nslices) {
    __shared__ u32 sh[BS * 28];
    const u32 chunk = blockIdx.x, j = blockIdx.y, w = blockIdx.z, t = threadIdx.x;
    gej acc; gej_set_infinity(acc);
#pragma unroll 1
    for (int r = 0; r < R; r++) {
        const u32 b = 1u + (chunk * (u32)R + (u32)r) * (u32)BS + t;
        int take = 0; size_t rec = 0;
        if (b < pl.nb) {
            const u32 k = w * pl.nb + b;
            take = (int)((msm_bucket_weight(L, pl, k) >> j) & 1u);
            if (off_last) { rec = off_last[k]; take &= (off_last[k + 1] > off_last[k]); } else rec = k;
        }
        gej v; gej_set_infinity(v);
        if (take) gej_load28(v, in28 + rec * 28);
        if (r == 0) acc = v;
        else if (__any(!v.inf)) { gej s; gej_add_var(s, acc, v); acc = s; }
    }
    gej_store28(sh + t * 28, acc);
    __syncthreads();
    // the tree: per lane while a level has at least a wavefront's worth... and the last sixteen partial sums in the wave-cooperative
    // arithmetic (a level of <= 8 additions costs the same ~8 us per lane as a full one, a cooperative addition ~3 us): wavefront v adds
    // up records v, v + 4, v + 8, v + 12, wavefront 0 the four results.
    u32* const dst = q28 + (((size_t)w * nslices + j) * nchunks + chunk) * 28;
    if (BS == 256) {
        for (u32 d = BS / 2; d >= 16; d >>= 1) {
            if (t < d) {
                gej a, c, r; gej_load28(a, sh + t * 28); gej_load28(c, sh + (t + d) * 28);
                gej_add_var(r, a, c);
                gej_store28(sh + t * 28, r);
            }
            __syncthreads();
        }
        const u32 wave = t >> 6;
        cgej cacc; int cinf = 1; cacc.x.v = cacc.y.v = cacc.z.v = 0;
#pragma unroll 1
        for (u32 i = 0; i < 4; i++) { cgej v; const int vi = cgej_load28(v, sh + (wave + 4u * i) * 28); cgej_acc(cacc, cinf, v, vi); }
        __syncthreads();
        cgej_store28(sh + wave * 28, cacc, cinf);
        __syncthreads();
        if (wave == 0) {
            cinf = 1;
#pragma unroll 1
            for (u32 i = 0; i < 4; i++) { cgej v; const int vi = cgej_load28(v, sh + i * 28); cgej_acc(cacc, cinf, v, vi); }
            cgej_store28(dst, cacc, cinf);
        }
    } else {
        for (u32 d = BS / 2; d >= 1; d >>= 1) {
            if (t < d) {
                gej a, c, r; gej_load28(a, sh + t * 28); gej_load28(c, sh + (t + d) * 28);
                gej_add_var(r, a, c);
                gej_store28(sh + t * 28, r);
            }
            __syncthreads();
        }
        if (t < 28) dst[t] = sh[t];
    }
}
__global__ void __launch_bounds__(1024)
k_msm_window_sums(u32* wsum28, const u32* q28, u32 nchunks, u32 nslices) {
    __shared__ u32 sh[MSM_SLICE_MAX * 28];
    const u32 w = blockIdx.x, wave = threadIdx.x >> 6;
    cgej acc; int acc_inf = 1;
    acc.x.v = acc.y.v = acc.z.v = 0;
    if (wave < nslices) {
        const u32* src = q28 + ((size_t)w * nslices + wave) * nchunks * 28;
        // (the record of chunk ch + 1 is requested before the addition of chunk ch)
        cgej nx; int nxi = cgej_load28(nx, src);
        for (u32 ch = 0; ch < nchunks; ch++) {
            const cgej v = nx; const int vi = nxi;
            if (ch + 1 < nchunks) nxi = cgej_load28(nx, src + (size_t)(ch + 1) * 28);
            cgej_acc(acc, acc_inf, v, vi);
        }
        if (!acc_inf) {
#pragma unroll 1
            for (u32 k = 0; k < wave; k++) cgej_double(acc);
        }
    }
    for (u32 d = MSM_SLICE_MAX / 2; d >= 1; d >>= 1) {
        if (wave >= d && wave < 2 * d) cgej_store28(sh + wave * 28, acc, acc_inf);
        __syncthreads();
        if (wave < d) { cgej v; const int vi = cgej_load28(v, sh + (wave + d) * 28); cgej_acc(acc, acc_inf, v, vi); }
    }
    if (wave == 0) cgej_store28(wsum28 + (size_t)w * 28, acc, acc_inf);
}
// small inputs: the bucket sums alone (one lane per bucket walks its whole region), records in bucket order
__global__ void __launch_bounds__(256)
k_msm_bucket_sums(u32* out28, const u32* refs, const u32* gcnt, const u32* term, msm_plan pl, msm_layout L) {
    const u32 w = blockIdx.y, b = blockIdx.x * 256u + threadIdx.x;
    if (b >= pl.nb) return;
    gej o; gej_set_infinity(o);
    const u32 k = w * pl.nb + b;
    if (b) {
        const int top = (pl.w0 + w + 1 == pl.windows);
        const u32 cap = top ? (b < L.top_used ? L.cap_top : 0u) : L.cap;
        u32 cnt = gcnt[k]; cnt = cnt < cap ? cnt : cap;            // (an overflowing region raised the flag: the result comes from the exact path)
        if (cnt) {
            const size_t first = msm_region(L, pl, w, b);
            if (!msm_sum_refs_lean(o, refs, first, first + cnt, term)) msm_sum_refs(o, refs, first, first + cnt, term);
        }
    }
    gej_store28(out28 + (size_t)k * 28, o);
}
// counts -> run counts -> their exclusive prefix, in ONE launch of one workgroup (it was three: k_msm_counts, k_scan_tiles, k_scan_fix; the
// arrays have 10^4 .. 3 10^5 entries and the three launches were ~5 us each plus their gaps, per partial-sum round).  Wavefront v owns the
// contiguous segment [v seg, (v + 1) seg) and walks it 256 entries a step (one 16-byte load per lane, consecutive lanes on consecutive
// entries); first sweep: segment totals, second: the running prefix.  Only the top window (the share's last, if it has it) has other
// capacities than L.cap, so no entry needs a division.
// (ceil(c / T) through the reciprocal M = floor(2^32 / T) + 1 and one correction step: a 32-bit division is ~40 instructions, and four
//  of them per step per lane were most of this kernel's 43 us)
__device__ __forceinline__ u32 msm_run_count(u32 c, u32 k, u32 T, u32 M, u32 top_first, const msm_layout& L, u32* clamped) {
    if (clamped) {
        const u32 cap = k >= top_first ? ((k - top_first) < L.top_used ? L.cap_top : 0u) : L.cap;
        c = c < cap ? c : cap; clamped[k] = c;
    }
    if (T <= 1u) return c;
    const u32 x = c + T - 1u;
    u32 q = __umulhi(x, M);
    q -= (q * T > x) ? 1u : 0u;
    q += ((q + 1u) * T <= x) ? 1u : 0u;
    return q;
}
// Full steps (all 256 entries inside the segment: every step but the array's last) move whole 16-byte vectors and carry no bounds tests:
// with a test per entry the kernel was ~1 500 instructions and 130 branches per 2 048 entries, 43 us for 41 000 entries on its one CU.
typedef unsigned int msm_u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(1024)
k_msm_counts_scan(u32* cnt_all, u32* cnt_clamped0, u32* off_all, const u32* cnt_in, u32 nk, u32 stride, u32 T, u32 T2, msm_layout L, msm_plan pl) {
    // workgroup r = partial-sum round r + 1: the run counts of EVERY round follow from the bucket counts alone (round r + 1 sums runs of T2 of
    // round r's partial sums: ceil(ceil(min(count, capacity) / T) / T2 / ...)), so all the rounds' prefix arrays are made here, side by side,
    // right behind the binning pass -- a launch per round between the rounds (round 6's first version) left ~10 us + a gap on the critical path each
    __shared__ u32 s_wave[17];
    const u32 rnd = blockIdx.x;
    u32* const cnt_out = cnt_all + (size_t)rnd * stride; u32* const off_out = off_all + (size_t)rnd * stride;
    u32* const cnt_clamped = rnd == 0 ? cnt_clamped0 : nullptr;
    const u32 M2 = 0xFFFFFFFFu / (T2 > 1u ? T2 : 2u) + 1u;
    const u32 t = threadIdx.x, wave = t >> 6, lane = t & 63u;
    const u32 seg = (((nk + 15u) / 16u) + 255u) & ~255u, lo = wave * seg, hi = min(lo + seg, nk);
    const u32 top_first = (pl.w0 + pl.wn == pl.windows) ? (pl.wn - 1u) * pl.nb : 0xFFFFFFFFu;
    const u32 M = 0xFFFFFFFFu / (T > 1u ? T : 2u) + 1u;        // >= 2^32 / T: the estimate is floor(x / T) or one more
    u32 sum = 0;
    // (the vector of step s + 1 is requested before step s is worked on: a step is otherwise one exposed L2 round trip)
    msm_u4 nxt; nxt.x = nxt.y = nxt.z = nxt.w = 0u;
    if (lo + 256u <= hi) nxt = *(const msm_u4*)(cnt_in + lo + 4u * lane);
    for (u32 k0 = lo; k0 < hi; k0 += 256u) {
        const u32 k = k0 + 4u * lane;
        u32 c[4], r[4];
        if (k0 + 256u <= hi) {
            const msm_u4 v = nxt;
            if (k0 + 512u <= hi) nxt = *(const msm_u4*)(cnt_in + k + 256u);
            c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                { const u32 kk = k + (u32)q; const u32 cap = kk >= top_first ? ((kk - top_first) < L.top_used ? L.cap_top : 0u) : L.cap; c[q] = c[q] < cap ? c[q] : cap; }
                r[q] = msm_run_count(c[q], 0u, T, M, 0u, L, nullptr);
                for (u32 i = 0; i < rnd; i++) r[q] = msm_run_count(r[q], 0u, T2, M2, 0u, L, nullptr);
                sum += r[q];
            }
            if (cnt_clamped) { msm_u4 o; o.x = c[0]; o.y = c[1]; o.z = c[2]; o.w = c[3]; *(msm_u4*)(cnt_clamped + k) = o; }
            msm_u4 o; o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3]; *(msm_u4*)(cnt_out + k) = o;
        } else {
#pragma unroll 1
            for (int q = 0; q < 4; q++) if (k + q < hi) {
                const u32 kk = k + (u32)q; const u32 cap = kk >= top_first ? ((kk - top_first) < L.top_used ? L.cap_top : 0u) : L.cap;
                u32 cc = cnt_in[kk]; cc = cc < cap ? cc : cap;
                if (cnt_clamped) cnt_clamped[kk] = cc;
                u32 rr = msm_run_count(cc, 0u, T, M, 0u, L, nullptr);
                for (u32 i = 0; i < rnd; i++) rr = msm_run_count(rr, 0u, T2, M2, 0u, L, nullptr);
                cnt_out[kk] = rr; sum += rr;
            }
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) sum += (u32)__shfl_xor((int)sum, d, 64);
    if (lane == 0) s_wave[wave] = sum;
    __syncthreads();
    if (t == 0) { u32 run = 0; for (int q = 0; q < 16; q++) { const u32 x = s_wave[q]; s_wave[q] = run; run += x; } s_wave[16] = run; }
    __syncthreads();
    u32 carry = s_wave[wave];
    if (lo + 256u <= hi) nxt = *(const msm_u4*)(cnt_out + lo + 4u * lane);
    for (u32 k0 = lo; k0 < hi; k0 += 256u) {
        const u32 k = k0 + 4u * lane;
        const int full = (k0 + 256u <= hi);
        u32 v[4];
        if (full) { const msm_u4 x = nxt; if (k0 + 512u <= hi) nxt = *(const msm_u4*)(cnt_out + k + 256u); v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
        else {
#pragma unroll 1
            for (int q = 0; q < 4; q++) v[q] = (k + q < hi) ? cnt_out[k + q] : 0u;
        }
        const u32 s4 = v[0] + v[1] + v[2] + v[3];
        u32 inc = s4;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 x = (u32)__shfl_up((int)inc, d, 64); if (lane >= (u32)d) inc += x; }
        const u32 run = carry + inc - s4;
        if (full) { msm_u4 o; o.x = run; o.y = run + v[0]; o.z = run + v[0] + v[1]; o.w = run + v[0] + v[1] + v[2]; *(msm_u4*)(off_out + k) = o; }
        else {
            u32 rr = run;
#pragma unroll 1
            for (int q = 0; q < 4; q++) { if (k + q < hi) off_out[k + q] = rr; rr += v[q]; }
        }
        carry += (u32)__shfl((int)inc, 63, 64);
    }
    if (t == 0) off_out[nk] = s_wave[16];
}
// Horner over the share's windows (~c*windows sequential doublings: the latency floor of one MSM).  One wavefront, all 64 lanes
// running the same point through msm_combine: the runs of doublings spread each field element over the lanes (cofield.h), the
// additions in between are the serial code executed redundantly.  Its own launch bounds so that the point state stays in registers.
// ... and the end of the call in the same launch: when the binning pass overflowed a bucket region the exact path's result (side stream,
// joined before this launch) is published instead; the flag goes to the engine's status word; `copy28` (optional) receives the Jacobian
// record, r_xy / r_inf (optional) the affine result -- the inversion used to be a launch of its own (k_gej_finish) behind a third one
// (k_msm_pick).
__global__ void __launch_bounds__(64)
k_msm_combine(u32* out28, u32* copy28, unsigned char* r_xy, int32_t* r_inf, const u32* wsum28, msm_plan pl, const u32* flags, const u32* exact28, u32* dev_flags) {
    if (blockIdx.x) return;
    const u32 over = flags[0];
    gej r;
    if (over) gej_load28(r, exact28); else msm_combine(r, wsum28, pl);
    if (threadIdx.x) return;
    dev_flags[0] = over;
    gej_store28(out28, r);
    if (copy28) gej_store28(copy28, r);
    if (r_xy) {
        if (r.inf) { for (int k = 0; k < 64; k++) r_xy[k] = 0; } else { ge a; ge_set_gej(a, r); ge_store_b64(r_xy, a); }
        *r_inf = r.inf;
    }
}
// Bucket-free form: lane l sums (share of k_i)*P_i over the terms i = l, l + lanes, ... with one full double-and-add each
// (ecmult.h); the term with index n carries g_sc*G.  Gated by the overflow flag: the exact path of a bucket launch whose regions
// overflowed (until round 6 also the whole call for sums below 32 terms: 0.8 ms where the bucket pipeline takes 0.37, msm.h).
__global__ void __launch_bounds__(256, 2)
k_msm_direct(u32* out28, const u32* gate, const unsigned char* g_sc, const unsigned char* sc, const unsigned char* pt, const unsigned char* pt_inf,
             const u32* gtab, u32* ptab, size_t n, size_t nt, msm_plan pl) {
    if (gate && *gate == 0) return;
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x, lanes = (size_t)gridDim.x * blockDim.x;
    __shared__ u32 s_dig[S2K_DIG_WORDS * 256];
    const lane_mem lm{ptab + lane * S2K_PTAB_WORDS, S2K_LANE_DIG(s_dig)};
    gej acc; gej_set_infinity(acc);
    for (size_t i0 = 0; i0 < nt; i0 += lanes) {
        const size_t i = i0 + lane;
        const int live = i < nt, isg = live && (i == n);
        gej A; scalar k, g; gej_set_infinity(A); sc_set_zero(k); sc_set_zero(g);
        if (live && !isg) {
            ge a; ge_load_b64(a, pt + 64 * i); fe_norm_weak(a.x); fe_norm_weak(a.y); gej_set_ge(A, a);
            A.inf = pt_inf ? (pt_inf[i] != 0) : 0;
            scalar kk; sc_set_b32(kk, sc + 32 * i, nullptr); msm_share_scalar(k, kk, pl);
        }
        if (isg) { scalar gg; sc_set_b32(gg, g_sc, nullptr); msm_share_scalar(g, gg, pl); }
        gej R; ecmult_lane(R, A, k, g, 1, gtab, lm);
        gej s; gej_add_var(s, acc, R); acc = s;
    }
    gej_store28(out28 + lane * 28, acc);
}
// a handful of Jacobian records (the exchanged partials of one sharded sum: 112 bytes per GPU) -> their affine sum, in ONE launch of one
// wavefront: cooperative additions (~3 us each; a 64-lane tree of complete per-lane additions + a separate inversion launch was ~115 us)
__global__ void __launch_bounds__(64)
k_gej_sum_small(unsigned char* r_xy, int32_t* r_inf, const u32* in28, u32 count) {
    if (blockIdx.x) return;
    cgej acc; int acc_inf = 1; acc.x.v = acc.y.v = acc.z.v = 0;
    cgej nx; int nxi = count ? cgej_load28(nx, in28) : 1;
    for (u32 i = 0; i < count; i++) {
        const cgej v = nx; const int vi = nxi;
        if (i + 1 < count) nxi = cgej_load28(nx, in28 + (size_t)(i + 1) * 28);
        cgej_acc(acc, acc_inf, v, vi);
    }
    gej r; if (acc_inf) gej_set_infinity(r); else cgej_to_gej(r, acc);
    if (threadIdx.x) return;
    if (r.inf) { for (int k = 0; k < 64; k++) r_xy[k] = 0; } else { ge a; ge_set_gej(a, r); ge_store_b64(r_xy, a); }
    *r_inf = r.inf;
}
__global__ void k_gej_finish(unsigned char* r_xy, int32_t* r_inf, const u32* in28) {
    if (threadIdx.x || blockIdx.x) return;
    gej r; gej_load28(r, in28);
    ge a; ge_set_gej(a, r);
    if (r.inf) { for (int k = 0; k < 64; k++) r_xy[k] = 0; } else ge_store_b64(r_xy, a);
    *r_inf = r.inf;
}

// reduce `count` gej28 (one segment) down to one, ping-ponging between two scratch buffers; returns pointer to the result
const u32* launch_gej_reduce(hipStream_t st, const u32* in, u32* bufA, u32* bufB, u32 nseg, u32 seg_len, const u32* gate) {
    const u32* cur = in; u32* dst = bufA;
    while (seg_len > 1) {
        const u32 per_block = 1024, nchunks = (seg_len + per_block - 1) / per_block;
        if (seg_len <= 256) hipLaunchKernelGGL(k_gej_reduce<64>, dim3(nseg * nchunks), dim3(64), 0, st, dst, cur, seg_len, per_block, nchunks, gate);
        else hipLaunchKernelGGL(k_gej_reduce<256>, dim3(nseg * nchunks), dim3(256), 0, st, dst, cur, seg_len, per_block, nchunks, gate);
        cur = dst; dst = (dst == bufA) ? bufB : bufA; seg_len = nchunks;
    }
    return cur;
}
static size_t msm_refs_words(const msm_plan& pl, const msm_layout& L) {
    if (pl.wn == 0) return 8;
    const int has_top = (pl.w0 + pl.wn == pl.windows);
    return has_top ? (size_t)(pl.wn - 1) * pl.nb * L.cap + (size_t)L.top_used * L.cap_top : (size_t)pl.wn * pl.nb * L.cap;
}
// run lengths of the partial-sum rounds: round 1 sums up to T references per lane (about 1.3e5 lanes' worth at the largest
// sizes), later rounds up to MSM_T2 partial sums -- short, because there are only a few per bucket left and lanes are scarce
static msm_coarse msm_make_coarse(size_t nt, const msm_plan& pl, const msm_layout& L) {
    msm_coarse C;
    C.shift = pl.c > 16 ? 8u : (pl.c > 13 ? 7u : (pl.c > 7 ? 6u : 0u));
    C.nco = ((pl.nb - 1u) >> C.shift) + 1u;                       // <= 2^(16 - 1 - 7) + 1 = 257
    const double mean = 2.0 * (double)nt / (double)(pl.nb - 1) * (double)(1u << C.shift);
    // the layout's own capacities are per bucket: mean + 10 standard deviations (+ the non-uniform values of the highest windows, see
    // msm_make_layout); the same rule for the sum over 2^shift buckets
    const u32 top_bits = 128u - pl.c * (pl.windows - 1);
    C.cap = msm_cap_for(top_bits == 0 ? 1.5 * mean : mean);
    const u32 top_vals = (top_bits >= pl.c - 1) ? (pl.nb - 1) : (1u << top_bits);
    double mean_top = 8.0 * (double)nt / (double)top_vals; if (mean_top > 2.0 * (double)nt) mean_top = 2.0 * (double)nt;
    double per_bin = mean_top / (double)L.sub * (double)(1u << C.shift); if (per_bin > 2.0 * (double)nt) per_bin = 2.0 * (double)nt;
    C.cap_top = msm_cap_for(per_bin);
    return C;
}
static size_t msm_pairs_words(const s2k_engine* e, size_t nt, const msm_plan& pl, const msm_layout& L) {
    if (!(pl.c > 13 || e->msm_diag.two_pass)) return 8;
    const msm_coarse C = msm_make_coarse(nt, pl, L);
    return (size_t)pl.windows * C.nco * (size_t)std::max(C.cap, C.cap_top) + 8;
}
static u32 msm_run_len(const s2k_engine* e, size_t E, const msm_plan& pl, const msm_layout& L) {
    if (e->msm_diag.T >= 2 && e->msm_diag.T <= 1024) return (u32)e->msm_diag.T;      // diagnostic override (-DS2K_DIAG builds)
    // small inputs (the plan keeps their bucket regions short): one lane per bucket takes the whole region, no second round
    const u32 maxcap = msm_max_cap(pl, L);
    if (maxcap <= MSM_ONE_ROUND_CAP) return maxcap;
    // ~6 lanes per resident lane slot (131 072) at the largest sizes, so that the last, partly filled round of workgroups is a small share
    // (measured at 2^20 terms: T = 24 2.17 ms, T = 128 2.33 ms; at 2^22: T = 48 7.26 ms, T = 128 7.38 ms)
    // (round 6, profiles/r06f_msm_tsweep.txt: from 2^17 terms a floor of 16 instead of 8 -- 2^18 terms 0.92 -> 0.85 ms, 2^17 0.68 -> 0.66: half
    //  as many partial sums for the later rounds, which are pure latency there)
    const u32 lo = E >= (size_t(1) << 17) * 2 * 11 ? 16u : 8u;
    u32 T = (u32)(E / 786432); if (T < lo) T = lo; if (T > 64) T = 64; return T;
}
#define MSM_T2 4u                        /* run length of the later rounds: the smallest that the buffer sizes below assume */
#define MSM_T2_MAX 12u
#define MSM_MAX_ROUNDS 6u                 /* T >= 8 and T2 = 12 reach any region (< 2^15 references) in 5 */
// Run length of the later partial-sum rounds.  The NUMBER of rounds follows from the bucket-region capacity (nothing is read back): T, T T2,
// T T2^2, ... until the fullest region is covered, and a round is two launches (counts + scan, sums) of latency: the fewest rounds that
// T2 <= 12 allows are taken (2^20 terms: the top window's regions, capacity 2 516, need T2 = 11 for two later rounds).  WITHIN that number
// of rounds the smallest sufficient T2 is taken (round 6; it used to be at least 8): a lane of a later round adds up to T2 Jacobian
// partials one after the other at ~8 us each on a machine that is mostly empty by then, so 2^17 terms (capacity 256, T = 8: two later
// rounds need T2^2 >= 32) run 6 + 6 dependent additions instead of 8 + 8.
static u32 msm_later_run_len(const s2k_engine* e, u32 T, u32 maxcap) {
    if (e->msm_diag.T2 >= (int)MSM_T2 && e->msm_diag.T2 <= 64) return (u32)e->msm_diag.T2;      // diagnostic override (-DS2K_DIAG builds)
    auto rounds_for = [&](u32 t2) { int r = 1; size_t reach = T; while (reach < maxcap) { reach *= t2; r++; } return r; };
    const int fewest = rounds_for(MSM_T2_MAX);
    for (u32 t2 = MSM_T2; t2 < MSM_T2_MAX; t2++) if (rounds_for(t2) == fewest) return t2;
    return MSM_T2_MAX;
}
#define MSM_DIRECT_LANES 16384u          /* lanes of the bucket-free exact path (each walks its terms with a stride) */
msm_plan engine_msm_plan(const s2k_engine* e, size_t nt) { return msm_make_plan(nt, e->msm_diag.c); }
// the slice tail (k_msm_slices / k_msm_window_sums) serves the plans up to this width; the widest windows (from 2^22 terms, where the tail
// is a few percent of the call) keep the per-bucket multiplication: 2^15 buckets x 16 slices per window is more tree work than it saves
#define MSM_SLICE_TAIL_MAX_C 13u
static size_t msm_slice_words(const msm_plan& pl) { return (size_t)pl.windows * MSM_SLICE_MAX * ((pl.nb + 63) / 64 + 1) * 28; }
// one launch indexes its bucket references with 32 bits (term << 2 | half << 1 | sign): the largest sum that fits
size_t msm_max_terms(const s2k_engine* e) {
    if (e->msm_max_terms_opt) return e->msm_max_terms_opt;
    return ((size_t(1) << 32) - 1) / (2 * 9) - 1;             // 9 windows (c = 16) from 2^22 terms on: 238 609 293
}
size_t msm_ws_bytes(const s2k_engine* e, size_t nt, const msm_plan& pl) {
    const size_t nk = (size_t)pl.windows * pl.nb;
    const size_t E = nt * 2 * pl.windows;
    const msm_layout L = msm_make_layout(nt, pl); const size_t T = msm_run_len(e, E, pl, L);
    return ws_need({28 * 4, 64, (size_t)MSM_DIRECT_LANES * 28 * 4, 64 * 28 * 4 * 2, nt * MSM_TERM_WORDS * 4, nt * MSM_HALF_WORDS * 4, (nk + 1) * 4 * 7, 1024 * 4, (size_t)MSM_MAX_ROUNDS * (nk + 64) * 4, (size_t)MSM_MAX_ROUNDS * (nk + 64) * 4,
                    msm_refs_words(pl, L) * 4, nk * 28 * 4, msm_pairs_words(e, nt, pl, L) * 8, (size_t)pl.windows * 520 * 4, (nk + E / T + 2) * 28 * 4, (nk * 2 + E / T / MSM_T2 + 64) * 28 * 4,
                    (nk / 1024 + nt / 1024 + pl.windows + 64) * 28 * 4 * 2, msm_slice_words(pl) * 4, (size_t)pl.windows * 28 * 4}) + 32 * 256;
}
static void launch_scan(hipStream_t st, u32* off, u32* cur, u32* tile_sum, const u32* in, u32 nk) {
    const u32 tiles = (nk + 1023) / 1024;
    hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(256), 0, st, off, tile_sum, in, nk);
    hipLaunchKernelGGL(k_scan_fix, dim3(tiles), dim3(256), 0, st, off, cur, tile_sum, nk);
}
// core: leaves the Jacobian result (28 words) at *result28 (device).  Workspace must already be large enough.
// Fully stream-ordered: nothing is read back.  The number of partial-sum rounds follows from the bucket-region capacity (a
// bucket never holds more than its region), and an input that overflows a region (adversarially equal scalars) raises a
// device flag that un-gates the exact bucket-free path queued behind the bucket pipeline; k_msm_combine then publishes its
// result instead.  (part, parts): the share of the digit windows this launch owns (msm_plan_share) -- (0, 1) = all of them.
__global__ void k_set_word(u32* p, u32 v) { *p = v; }
void launch_set_word(hipStream_t st, u32* p, u32 v) { hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, st, p, v); }
__global__ void k_msm_flag_copy(u32* persist, const u32* flags) { persist[0] = flags[0]; }
// what a launch does with its result besides leaving the Jacobian record in the workspace: the affine result (r_xy 64 bytes + r_inf) and /
// or a copy of the record, written by the last kernel of the chain
static int msm_emit(hipStream_t st, const msm_out* out, const u32* res28) {
    if (!out) return 1;
    if (out->r_xy) hipLaunchKernelGGL(k_gej_finish, dim3(1), dim3(64), 0, st, out->r_xy, out->r_inf, res28);
    if (out->out28) HIPCHK(hipMemcpyAsync(out->out28, res28, 28 * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipGetLastError());
    return 1;
}
// k_msm_slices: BS lanes per workgroup, R buckets per lane (a serial addition there costs a tree level, ~7 us; a chunk more ~3 us in
// k_msm_window_sums)
struct msm_slice_cfg { u32 R, BS; };
static msm_slice_cfg msm_slice_config(const s2k_engine* e, u32 nb) {
    msm_slice_cfg g{4u, 256u};           // (measured, profiles/r06b_msm_variants.txt: the kernel is bound by its tree levels' issue slots -- fewer, fuller workgroups win at every size)
    const int v = e->msm_diag.slice_r;          // diagnostic override (-DS2K_DIAG builds): R + 16 * (64-lane workgroups)
    if (v > 0) { g.R = (u32)(v & 15); g.BS = (v & 16) ? 64u : 256u; if (g.R != 1 && g.R != 2 && g.R != 4 && g.R != 8) g.R = 4; }
    return g;
}
static u32 msm_slice_chunks(u32 nb, const msm_slice_cfg& g) { return (nb - 1u + g.BS * g.R - 1u) / (g.BS * g.R); }
static int msm_new_tail(const s2k_engine* e, const msm_plan& pl) { return !e->msm_diag.old_tail && pl.c <= (e->msm_diag.slice_maxc ? (u32)e->msm_diag.slice_maxc : MSM_SLICE_TAIL_MAX_C); }
// bucket sums (one record per bucket: in28[k], or in28[off_last[k]] where the partial-sum rounds packed them) -> window sums
static const u32* launch_window_sums(s2k_engine* e, hipStream_t st, u32* q28, u32* wsum28, const u32* in28, const u32* off_last, const msm_plan& pl, const msm_layout& L) {
    const msm_slice_cfg g = msm_slice_config(e, pl.nb);
    const u32 nchunks = msm_slice_chunks(pl.nb, g), nslices = pl.c;      // weights are at most 2^(c-1)
    const dim3 grid(nchunks, nslices, pl.wn);
    const u32 lds = (u32)e->msm_diag.slice_lds;          // diagnostic: unused dynamic LDS, to limit the workgroups per CU
#define S2K_SLICES(R_, BS_) hipLaunchKernelGGL((k_msm_slices<R_, BS_>), grid, dim3(BS_), lds, st, q28, in28, off_last, pl, L, nchunks, nslices)
    if (g.BS == 256u) { if (g.R == 1) S2K_SLICES(1, 256); else if (g.R == 2) S2K_SLICES(2, 256); else if (g.R == 4) S2K_SLICES(4, 256); else S2K_SLICES(8, 256); }
    else { if (g.R == 1) S2K_SLICES(1, 64); else if (g.R == 2) S2K_SLICES(2, 64); else if (g.R == 4) S2K_SLICES(4, 64); else S2K_SLICES(8, 64); }
#undef S2K_SLICES
    hipLaunchKernelGGL(k_msm_window_sums, dim3(pl.wn), dim3(1024), 0, st, wsum28, (const u32*)q28, nchunks, nslices);
    return wsum28;
}
int msm_launch(s2k_engine* e, hipStream_t st, ws_carver& c, u32** result28, const unsigned char* g_sc, const unsigned char* sc,
               const unsigned char* pt, const unsigned char* pt_inf, size_t n, u32 part, u32 parts, const msm_ctx* ctx, const msm_out* out) {
    const msm_ctx dflt{e->stream2, e->ev_msm_fork, e->ev_msm_join, 0u};
    const msm_ctx& X = ctx ? *ctx : dflt;
    const size_t nt = n + (g_sc ? 1 : 0);
    if (parts == 0 || part >= parts) return s2k_fail_arg("s2k_ecmult_multi", "window share out of range");
    ENGINE_GTAB(e, st);                                        // (the bucket-free exact path multiplies by G through the table)
    msm_plan pl = engine_msm_plan(e, nt ? nt : 1);
    // term references are packed as (u32)(term << 2 | half << 1 | sign): the entry points split larger sums (msm_max_terms)
    if (nt > msm_max_terms(e) || nt >= (size_t(1) << 30) || nt * 2 * pl.windows >= (size_t(1) << 32))
        return s2k_fail("s2k_ecmult_multi", "too many terms for 32-bit bucket references (2 * windows * n >= 2^32) in one launch");
    msm_plan_share(pl, part, parts);
    u32* final28 = c.take<u32>(28);
    *result28 = final28;
    u32* flags = c.take<u32>(16);                              // flags[0]: a bucket region overflowed
    u32* lanes = c.take<u32>((size_t)MSM_DIRECT_LANES * 28); u32* dbufA = c.take<u32>(64 * 28); u32* dbufB = c.take<u32>(64 * 28);
    if (nt == 0 || pl.wn == 0) {                               // empty sum / empty share: infinity
        HIPCHK(hipMemsetAsync(flags, 0, 64, st));
        HIPCHK(hipMemsetAsync(final28, 0, 27 * 4, st));
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, st, final28 + 27, 1u);
        hipLaunchKernelGGL(k_msm_flag_copy, dim3(1), dim3(1), 0, st, e->dev_flags, flags);
        HIPCHK(hipGetLastError());
        return msm_emit(st, out, final28);
    }
    if (!engine_ptab(e, 3 * (size_t)MSM_DIRECT_LANES)) return 0;
    u32* const direct_ptab = e->ptab + (size_t)X.arena * MSM_DIRECT_LANES * S2K_PTAB_WORDS;
    const u32 nk = pl.wn * pl.nb;
    const size_t E = nt * 2 * pl.wn;                           // upper bound on this share's bucket references
    const msm_layout L = msm_make_layout(nt, pl);
    const msm_plan full = engine_msm_plan(e, nt);              // (run length as msm_ws_bytes sized the buffers for: from the whole plan, not the share)
    const u32 T = msm_run_len(e, nt * 2 * pl.windows, full, L);
    const size_t bound1 = (size_t)nk + E / T + 2;
    u32* term = c.take<u32>(nt * MSM_TERM_WORDS); u32* halves = c.take<u32>(nt * MSM_HALF_WORDS);
    u32* gcnt = c.take<u32>(nk + 1); u32* gclamp = c.take<u32>(nk + 1); u32* spare = c.take<u32>(nk + 1);
    u32* cntA = c.take<u32>(nk + 1); u32* cntB = c.take<u32>(nk + 1); u32* offA = c.take<u32>(nk + 1); u32* offB = c.take<u32>(nk + 1);
    u32* tile_sum = c.take<u32>(1024); (void)spare;
    u32* cnt_all = c.take<u32>((size_t)MSM_MAX_ROUNDS * (nk + 64)); u32* off_all = c.take<u32>((size_t)MSM_MAX_ROUNDS * (nk + 64));
    u32* refs_cap = c.take<u32>(msm_refs_words(pl, L)); u32* buckets = c.take<u32>((size_t)nk * 28);
    unsigned long long* pairs = c.take<unsigned long long>(msm_pairs_words(e, nt, pl, L)); u32* ccnt = c.take<u32>((size_t)pl.windows * 520);
    u32* partA = c.take<u32>(bound1 * 28); u32* partB = c.take<u32>(((size_t)nk * 2 + E / T / MSM_T2 + 64) * 28);
    u32* bufA = c.take<u32>(((size_t)nk / 1024 + pl.windows + 64) * 28); u32* bufB = c.take<u32>(((size_t)nk / 1024 + pl.windows + 64) * 28);
    u32* q28 = c.take<u32>(msm_slice_words(full)); u32* wsum_new = c.take<u32>((size_t)pl.windows * 28);
    const unsigned bt = (unsigned)((nt + 255) / 256), bk = (nk + 255) / 256;
    hipLaunchKernelGGL(k_msm_prep, dim3(bt), dim3(256), 0, st, term, halves, g_sc, sc, pt, pt_inf, n, nt, flags, 16u, gcnt, nk + 1u);
    u32 chunk = 8192; while (chunk > 1024 && (nt + chunk - 1) / chunk * pl.wn < 1024) chunk >>= 1;       // enough workgroups to fill 256 CUs
    { const int v = e->msm_diag.chunk; if (v == 1024 || v == 2048 || v == 4096 || v == 8192) chunk = (u32)v; }      // diagnostic override (-DS2K_DIAG builds)
    u32 bin_dbg = 0;
#ifdef S2K_DIAG          /* diagnostic builds only: launches of k_msm_bin with parts switched off (results are meaningless then) */
    if (const char* bd = getenv("S2K_MSM_BIN_DEBUG")) bin_dbg = ((u32)atoi(bd) & 7u) << 24;
#endif
    const int two_pass = (pl.c > 13 || e->msm_diag.two_pass) && !e->msm_diag.one_pass;
    if (two_pass) {
        const msm_coarse C = msm_make_coarse(nt, pl, L);
        HIPCHK(hipMemsetAsync(ccnt, 0, (size_t)pl.wn * C.nco * 4, st));
        hipLaunchKernelGGL(k_msm_bin_coarse, dim3((unsigned)((nt + MSM_COARSE_TERMS - 1) / MSM_COARSE_TERMS)), dim3(MSM_BIN_THREADS), 0, st, pairs, ccnt, flags, halves, nt, pl, L, C);
        hipLaunchKernelGGL(k_msm_bin_fine, dim3(C.nco, pl.wn), dim3(MSM_FINE_THREADS), 0, st, refs_cap, gcnt, flags, (const unsigned long long*)pairs, (const u32*)ccnt, pl, L, C);
    } else if (pl.c <= 13 && nt >= (size_t(1) << 15) && !e->msm_diag.bin_plain) {
        hipLaunchKernelGGL(k_msm_bin_staged, dim3((unsigned)((nt + MSM_STAGED_TERMS - 1) / MSM_STAGED_TERMS), pl.wn), dim3(MSM_BIN_THREADS), 0, st, refs_cap, gcnt, flags, halves, nt, pl, L);
    } else if (pl.c > 13) hipLaunchKernelGGL(k_msm_bin<1>, dim3((unsigned)((nt + chunk - 1) / chunk), pl.wn), dim3(MSM_BIN_THREADS), 0, st, refs_cap, gcnt, flags, halves, nt, pl, L, chunk | bin_dbg);
    else hipLaunchKernelGGL(k_msm_bin<0>, dim3((unsigned)((nt + chunk - 1) / chunk), pl.wn), dim3(MSM_BIN_THREADS), 0, st, refs_cap, gcnt, flags, halves, nt, pl, L, chunk | bin_dbg);
    // exact path, un-gated only by the overflow flag the binning pass may have raised: on the side stream, so that its (normally
    // empty) launches do not sit behind the Horner tail of every call
    HIPCHK(hipEventRecord(X.fork, st));
    HIPCHK(hipStreamWaitEvent(X.side, X.fork, 0));
    hipLaunchKernelGGL(k_msm_direct, dim3(MSM_DIRECT_LANES / 256), dim3(256), 0, X.side, lanes, (const u32*)flags, g_sc, sc, pt, pt_inf, e->gtab, direct_ptab, n, nt, pl);
    const u32* ex = launch_gej_reduce(X.side, lanes, dbufA, dbufB, 1, MSM_DIRECT_LANES, flags);
    HIPCHK(hipEventRecord(X.join, X.side));
    const int new_tail = msm_new_tail(e, pl);
    const u32* wsum = nullptr;
    if (msm_max_cap(full, L) <= MSM_ONE_ROUND_CAP && !e->msm_diag.no_small) {
        if (new_tail) {
            hipLaunchKernelGGL(k_msm_bucket_sums, dim3((pl.nb + 255) / 256, pl.wn), dim3(256), 0, st, buckets, refs_cap, gcnt, term, pl, L);
            wsum = launch_window_sums(e, st, q28, wsum_new, buckets, nullptr, pl, L);
        } else {
            const u32 nchunks = (pl.nb - 1 + 255) / 256;
            hipLaunchKernelGGL(k_msm_small_windows, dim3(nchunks, pl.wn), dim3(256), 0, st, partA, refs_cap, gcnt, term, pl, L, nchunks);
            wsum = launch_gej_reduce(st, partA, bufA, bufB, pl.wn, nchunks);
        }
    } else {
        // rounds: a bucket holds at most its region's capacity, so the capacity fixes how many rounds reach "one partial per bucket"
        const int has_top = (pl.w0 + pl.wn == pl.windows);
        const u32 maxcap = std::max(pl.wn > (has_top ? 1u : 0u) ? L.cap : 0u, has_top ? L.cap_top : 0u);
        const u32 T2 = msm_later_run_len(e, T, maxcap);
        int rounds = 1; { size_t reach = T; while (reach < maxcap) { reach *= T2; rounds++; } }
        if (rounds > (int)MSM_MAX_ROUNDS) return s2k_fail("s2k_ecmult_multi", "internal: more partial-sum rounds than planned for");
        // every round's run counts and prefix arrays, in one launch behind the binning pass
        const u32 stride = (nk + 64u) & ~63u;
        if (!e->msm_diag.old_tail) hipLaunchKernelGGL(k_msm_counts_scan, dim3((unsigned)rounds), dim3(1024), 0, st, cnt_all, gclamp, off_all, (const u32*)gcnt, nk, stride, T, T2, L, pl);
        else { hipLaunchKernelGGL(k_msm_counts, dim3(bk), dim3(256), 0, st, cntA, gclamp, gcnt, nk, T, L, pl); launch_scan(st, offA, nullptr, tile_sum, cntA, nk); }
        u32* const off1 = e->msm_diag.old_tail ? offA : off_all;
        // round 1: references -> partial sums (at most T references each)
        HIPCHK(hipEventRecord(e->ev[2], st));
        const int run_major = e->msm_diag.run_major ? e->msm_diag.run_major > 0 : (pl.c > 13 && nt >= (size_t(1) << 25));
        // (measured, profiles/r06o_msm_runmajor.txt: 2^22 .. 2^24 terms the same either way -- 20.7 ms at 2^24, so the operand gathers are NOT what
        //  holds round 1 back there; 2^25 terms, 4.3 GB of records: 42.2 -> 39.4 ms.  Run order from 2^25 terms.)
        if (run_major) {
            const size_t lanes = (size_t)nk * ((maxcap + T - 1) / T);
            hipLaunchKernelGGL(k_msm_round1<1>, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, partA, refs_cap, (const u32*)nullptr, gclamp, L, pl, off1, term, nk, T);
        } else hipLaunchKernelGGL(k_msm_round1<0>, dim3((unsigned)((bound1 + 255) / 256)), dim3(256), 0, st, partA, refs_cap, (const u32*)nullptr, gclamp, L, pl, off1, term, nk, T);
        HIPCHK(hipEventRecord(e->ev[3], st));
        // rounds 2..R: partial sums of partial sums until every bucket holds at most one
        u32 *cin = cntA, *cout = cntB, *oin = off1, *oout = offB, *pin = partA, *pout = partB;
        size_t bound = bound1;
        for (int r = 2; r <= rounds; r++) {
            bound = (size_t)nk + bound / T2 + 2;
            if (!e->msm_diag.old_tail) oout = off_all + (size_t)(r - 1) * stride;
            else { hipLaunchKernelGGL(k_msm_counts, dim3(bk), dim3(256), 0, st, cout, (u32*)nullptr, cin, nk, T2, L, pl); launch_scan(st, oout, nullptr, tile_sum, cout, nk); }
            hipLaunchKernelGGL(k_msm_roundN, dim3((unsigned)((bound + 255) / 256)), dim3(256), 0, st, pout, pin, oin, oout, nk, T2);
            u32* t;
            t = cin; cin = cout; cout = t; t = pin; pin = pout; pout = t;
            if (!e->msm_diag.old_tail) oin = oout; else { t = oin; oin = oout; oout = t; }
        }
        if (new_tail) wsum = launch_window_sums(e, st, q28, wsum_new, pin, oin, pl, L);
        else {
            hipLaunchKernelGGL(k_msm_finish, dim3(bk), dim3(256), 0, st, buckets, pin, oin, nk, pl, L);
            wsum = launch_gej_reduce(st, buckets, bufA, bufB, pl.wn, pl.nb);
        }
    }
    HIPCHK(hipStreamWaitEvent(st, X.join, 0));
    hipLaunchKernelGGL(k_msm_combine, dim3(1), dim3(64), 0, st, final28, out ? out->out28 : (u32*)nullptr, out ? out->r_xy : (unsigned char*)nullptr,
                       out ? out->r_inf : (int32_t*)nullptr, wsum, pl, (const u32*)flags, ex, e->dev_flags);
    HIPCHK(hipGetLastError());
    return 1;
}
// Two calls in flight (see s2k_engine::msm_slot): used when the caller has promised that its input arrays are complete at call time
// (S2K_OPT_RP_INPUTS_READY) and the sum is SMALL (<= 2^13 terms): there a call is a chain of latency-bound launches that leaves most of
// the machine idle, and two chains side by side finish in little more than the time of one (measured, 1 024 terms: 0.49 -> 0.35 ms per
// call; profiles/r04e_msm_bare*.txt).  From ~2^14 terms on the partial-sum rounds fill the SIMDs, the other call's tail kernels only take
// issue slots from them, and two calls in flight are SLOWER than one after the other (2^20 terms: 2.31 against 2.04 ms) -- those sizes keep
// the plain path.  `finish`: 0 = the Jacobian partial to
// out28, 1 = affine result to r_xy / r_inf.  Everything runs on the slot's streams; the caller's stream waits for the result.
#define MSM_PIPE_MAX_TERMS (size_t(1) << 13)
static int msm_pipelined(s2k_engine* e, hipStream_t st, int finish, uint32_t* out28, unsigned char* r_xy, int32_t* r_inf, const unsigned char* g_sc,
                         const unsigned char* sc, const unsigned char* pt_xy, const unsigned char* pt_inf, size_t n) {
    const size_t nt = n + (g_sc ? 1 : 0);
    const msm_plan pl = engine_msm_plan(e, nt ? nt : 1);
    const size_t need = msm_ws_bytes(e, nt + 1, pl);
    const unsigned si = e->msm_seq++ & 1u;
    auto& S = e->msm_slot[si];
    if (need > S.ws_bytes) {           // (the outgrown buffer may still be in use by the slot's previous call: retired, not freed)
        void* buf = S.ws; size_t have = S.ws_bytes;
        const int ok = engine_grow_dev(e, &buf, &have, need, size_t(1) << 20);
        S.ws = (unsigned char*)buf; S.ws_bytes = have;
        if (!ok) return 0;
    }
    // the slot's streams are not the caller's: work of an earlier call of another kind (it shares the engine's table arena with this
    // call's exact path) must be over first -- EVERY slot waits once for the latest such call (its epoch), not only the slot that happens to
    // run right behind it; consecutive pipelined MSM calls do not wait for each other -- that is the point.  The caller's inputs: ordered
    // behind the caller's stream unless the caller has promised that they are complete (S2K_OPT_RP_INPUTS_READY).
    if (e->np_valid && S.seen_epoch != e->np_epoch) { HIPCHK(hipStreamWaitEvent(S.s, e->ev_last_np, 0)); S.seen_epoch = e->np_epoch; }
    if (!e->rp_inputs_ready) { HIPCHK(hipEventRecord(S.in, st)); HIPCHK(hipStreamWaitEvent(S.s, S.in, 0)); }
    e->cur_pipe = 1;
    const msm_ctx ctx{S.s2, S.fork, S.join, 1u + si};
    ws_carver c{S.ws, 0}; u32* res = nullptr;
    HIPCHK(hipEventRecord(e->ev[0], S.s));
    const msm_out out{finish ? r_xy : nullptr, finish ? r_inf : nullptr, finish ? nullptr : out28};
    if (!msm_launch(e, S.s, c, &res, g_sc, sc, pt_xy, pt_inf, n, 0, 1, &ctx, &out)) return 0;
    HIPCHK(hipEventRecord(e->ev[1], S.s));
    HIPCHK(hipEventRecord(S.done, S.s));
    HIPCHK(hipStreamWaitEvent(st, S.done, 0));               // the result is stream-ordered for the caller
    return 1;
}
// One sum through the engine's own workspace, whatever its size.  A launch indexes its bucket references with 32 bits, so a sum of more
// than msm_max_terms() terms goes as consecutive launches over slices of the term arrays whose Jacobian partial sums are added at the end --
// the reference's own treatment of a sum that exceeds its scratch space (ecmult_impl.h:804-820 computes the batch size, :856-865 loops over
// the batches and adds).  `front`: bytes of the workspace the caller has already carved (staged inputs).
static size_t msm_run_ws_bytes(const s2k_engine* e, size_t n, int has_g) {
    const size_t nt = n + (has_g ? 1 : 0), cap = msm_max_terms(e);
    if (nt <= cap) return msm_ws_bytes(e, nt + 1, engine_msm_plan(e, nt ? nt : 1));
    const size_t per = has_g ? cap - 1 : cap, pieces = (n + per - 1) / per;
    const size_t last = n - (pieces - 1) * per + 1;            // (the last launch is shorter and may plan a different window width)
    return ws_need({pieces * 28 * 4, (pieces / 1024 + 64) * 28 * 4, (pieces / 1024 + 64) * 28 * 4}) +
           std::max(msm_ws_bytes(e, cap + 1, engine_msm_plan(e, cap)), msm_ws_bytes(e, last + 1, engine_msm_plan(e, last)));
}
static int msm_run(s2k_engine* e, hipStream_t st, size_t front, const msm_out& out, const unsigned char* g_sc, const unsigned char* sc,
                   const unsigned char* pt_xy, const unsigned char* pt_inf, size_t n) {
    const size_t nt = n + (g_sc ? 1 : 0), cap = msm_max_terms(e);
    if (!engine_workspace(e, front + msm_run_ws_bytes(e, n, g_sc != nullptr))) return 0;
    if (nt <= cap) {
        ws_carver c{e->ws, front}; u32* res = nullptr;
        HIPCHK(hipEventRecord(e->ev[0], st));
        if (!msm_launch(e, st, c, &res, g_sc, sc, pt_xy, pt_inf, n, 0, 1, nullptr, &out)) return 0;
        HIPCHK(hipEventRecord(e->ev[1], st));
        return 1;
    }
    const size_t per = g_sc ? cap - 1 : cap, pieces = (n + per - 1) / per;
    if (pieces > (size_t(1) << 20)) return s2k_fail("s2k_ecmult_multi", "sum too large");
    ws_carver c{e->ws, front};
    u32* parts28 = c.take<u32>(pieces * 28); u32* bufA = c.take<u32>((pieces / 1024 + 64) * 28); u32* bufB = c.take<u32>((pieces / 1024 + 64) * 28);
    const size_t base = c.off;
    HIPCHK(hipEventRecord(e->ev[0], st));
    for (size_t i = 0; i < pieces; i++) {
        const size_t lo = i * per, cnt = std::min(per, n - lo);
        ws_carver ci{e->ws, base}; u32* res = nullptr;
        const msm_out oi{nullptr, nullptr, parts28 + i * 28};
        if (!msm_launch(e, st, ci, &res, i == 0 ? g_sc : nullptr, sc + 32 * lo, pt_xy + 64 * lo, pt_inf ? pt_inf + lo : nullptr, cnt, 0, 1, nullptr, &oi)) return 0;
    }
    const u32* r = launch_gej_reduce(st, parts28, bufA, bufB, 1, (u32)pieces);
    if (out.r_xy) hipLaunchKernelGGL(k_gej_finish, dim3(1), dim3(64), 0, st, out.r_xy, out.r_inf, r);
    if (out.out28) HIPCHK(hipMemcpyAsync(out.out28, r, 28 * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
}
extern "C" int s2k_ecmult_multi_partial_dev(s2k_engine* e, void* stream, uint32_t* r_gej28, const unsigned char* g_sc,
                                            const unsigned char* sc, const unsigned char* pt_xy, const unsigned char* pt_inf, size_t n) {
    if (!e) return s2k_fail("s2k_ecmult_multi_partial_dev", "null engine");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    const size_t nt = n + (g_sc ? 1 : 0);
    if (e->msm_pipeline && nt >= 1 && nt <= MSM_PIPE_MAX_TERMS) return msm_pipelined(e, st, 0, r_gej28, nullptr, nullptr, g_sc, sc, pt_xy, pt_inf, n);
    return msm_run(e, st, 0, msm_out{nullptr, nullptr, r_gej28}, g_sc, sc, pt_xy, pt_inf, n);
}
extern "C" int s2k_ecmult_multi_window_partial_dev(s2k_engine* e, void* stream, uint32_t* r_gej28, const unsigned char* g_sc, const unsigned char* sc,
                                                   const unsigned char* pt_xy, const unsigned char* pt_inf, size_t n, uint32_t part, uint32_t parts) {
    if (!e) return s2k_fail("s2k_ecmult_multi_window_partial_dev", "null engine");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    const size_t nt = n + (g_sc ? 1 : 0);
    const msm_plan pl = engine_msm_plan(e, nt ? nt : 1);
    if (!engine_workspace(e, msm_ws_bytes(e, nt + 1, pl))) return 0;
    ws_carver c{e->ws, 0}; u32* res = nullptr;
    HIPCHK(hipEventRecord(e->ev[0], st));
    const msm_out out{nullptr, nullptr, r_gej28};
    if (!msm_launch(e, st, c, &res, g_sc, sc, pt_xy, pt_inf, n, part, parts, nullptr, &out)) return 0;
    HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
}
extern "C" int s2k_ecmult_multi_dev(s2k_engine* e, void* stream, unsigned char* r_xy, int32_t* r_inf, const unsigned char* g_sc,
                                    const unsigned char* sc, const unsigned char* pt_xy, const unsigned char* pt_inf, size_t n) {
    if (!e) return s2k_fail("s2k_ecmult_multi_dev", "null engine");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    const size_t nt = n + (g_sc ? 1 : 0);
    if (e->msm_pipeline && nt >= 1 && nt <= MSM_PIPE_MAX_TERMS) return msm_pipelined(e, st, 1, nullptr, r_xy, r_inf, g_sc, sc, pt_xy, pt_inf, n);
    return msm_run(e, st, 0, msm_out{r_xy, r_inf, nullptr}, g_sc, sc, pt_xy, pt_inf, n);
}
extern "C" int s2k_gej_sum_dev(s2k_engine* e, void* stream, unsigned char* r_xy, int32_t* r_inf, const uint32_t* gej28, size_t count) {
    if (!e) return s2k_fail("s2k_gej_sum_dev", "null engine");
    if (count == 0) return s2k_fail("s2k_gej_sum_dev", "count == 0");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    if (count <= 64) {             // the partials of a node's GPUs: one wavefront adds them up in the cooperative arithmetic and publishes the affine sum
        hipLaunchKernelGGL(k_gej_sum_small, dim3(1), dim3(64), 0, st, r_xy, r_inf, (const u32*)gej28, (u32)count);
        HIPCHK(hipGetLastError());
        return 1;
    }
    if (!engine_workspace(e, ws_need({(count / 1024 + 64) * 28 * 4, (count / 1024 + 64) * 28 * 4}))) return 0;
    ws_carver c{e->ws, 0};
    u32* bufA = c.take<u32>((count / 1024 + 64) * 28); u32* bufB = c.take<u32>((count / 1024 + 64) * 28);
    const u32* r = launch_gej_reduce(st, gej28, bufA, bufB, 1, (u32)count);
    hipLaunchKernelGGL(k_gej_finish, dim3(1), dim3(64), 0, st, r_xy, r_inf, r);
    HIPCHK(hipGetLastError());
    return 1;
}
extern "C" int s2k_ecmult_multi(s2k_engine* e, unsigned char* r_xy, int32_t* r_inf, const unsigned char* g_sc,
                                const unsigned char* sc, const unsigned char* pt_xy, const unsigned char* pt_inf, size_t n) {
    if (!e) return s2k_fail("s2k_ecmult_multi", "null engine");
    if (!r_xy || !r_inf || (n && (!sc || !pt_xy))) return s2k_fail_arg("s2k_ecmult_multi", "illegal argument (ARG_CHECK)");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    // the staged inputs come first in the workspace, the MSM passes carve what follows (sized here, once: growing the workspace moves it)
    const size_t front = ws_need({32 * n + 64, 64 * n + 64, n + 64, 64, 64, 16});
    if (!engine_workspace(e, front + msm_run_ws_bytes(e, n, g_sc != nullptr))) return 0;
    ws_carver c{e->ws, 0};
    unsigned char* d_sc = c.take<unsigned char>(32 * n + 64); unsigned char* d_pt = c.take<unsigned char>(64 * n + 64); unsigned char* d_inf = c.take<unsigned char>(n + 64);
    unsigned char* d_g = c.take<unsigned char>(64); unsigned char* d_r = c.take<unsigned char>(64); int32_t* d_ri = c.take<int32_t>(4);
    hipStream_t st = e->stream;
    stream_guard sg(e, st);
    if (n) {
        HIPCHK(hipMemcpyAsync(d_sc, sc, 32 * n, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_pt, pt_xy, 64 * n, hipMemcpyHostToDevice, st));
        if (pt_inf) HIPCHK(hipMemcpyAsync(d_inf, pt_inf, n, hipMemcpyHostToDevice, st));
    }
    if (g_sc) HIPCHK(hipMemcpyAsync(d_g, g_sc, 32, hipMemcpyHostToDevice, st));
    if (!msm_run(e, st, front, msm_out{d_r, d_ri, nullptr}, g_sc ? d_g : nullptr, d_sc, d_pt, pt_inf ? d_inf : nullptr, n)) return 0;
    HIPCHK(hipMemcpyAsync(r_xy, d_r, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r_inf, d_ri, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 1;
}

// ------------------------------------------------------------------------------------------------------------
// Pedersen tallies (pedersen.h): one lane per commitment, then bounded-run partial sums per tally
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pt_load(u32* out28, u32* bad, const unsigned char* commits33, const unsigned long long* tally_off, const unsigned long long* n_pos, size_t n_tallies, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t t = pedersen_find_tally(tally_off, n_tallies, i);
    ge c; const int ok = pedersen_load(c, commits33 + 33 * i);
    if (i - tally_off[t] >= n_pos[t]) { fe_neg(c.y, c.y, 1); }          // the negative list (:388)
    gej j; gej_set_ge(j, c); j.inf = 0;
    gej_store28(out28 + i * 28, j);
    if (!ok) bad[t] = 1u;
}
__global__ void k_pt_final(int32_t* results, const u32* sums28, const u32* off_last, const u32* bad, size_t n_tallies) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tallies) return;
    int inf = 1;
    if (off_last[t + 1] > off_last[t]) inf = (int)sums28[(size_t)off_last[t] * 28 + 27];
    results[t] = inf && !bad[t];
}
// dev = 1: `results` and `commits33` are device pointers (the two small offset arrays always come from the host: the run
// lengths of every partial-sum round are derived from them before anything is launched) and nothing is read back
static int tally_impl(s2k_engine* e, void* stream, int32_t* results, const unsigned char* commits33, const uint64_t* tally_off,
                      const uint64_t* n_pos, size_t n_tallies, int dev) {
    if (!e) return s2k_fail("secp256k1_pedersen_verify_tally_batch", "null engine");
    if (n_tallies == 0) return 1;
    if (!results || !tally_off || !n_pos) return s2k_fail_arg("secp256k1_pedersen_verify_tally_batch", "illegal argument (ARG_CHECK)");
    if (!dev) memset(results, 0, sizeof(int32_t) * n_tallies);
    const size_t total = (size_t)tally_off[n_tallies];
    if (total >= ((size_t)1 << 32)) return s2k_fail("secp256k1_pedersen_verify_tally_batch", "more than 2^32 commitments in one call");
    for (size_t t = 0; t < n_tallies; t++)
        if (tally_off[t + 1] < tally_off[t] || n_pos[t] > tally_off[t + 1] - tally_off[t]) return s2k_fail("secp256k1_pedersen_verify_tally_batch", "malformed tally offsets");
    if (total && !commits33) return s2k_fail_arg("secp256k1_pedersen_verify_tally_batch", "illegal argument (ARG_CHECK)");
    // the offset arrays of every partial-sum round are known from the sizes alone: built here, uploaded once
    const u32 T = 8;
    std::vector<std::vector<u32>> offs;
    { std::vector<u32> o(n_tallies + 1); for (size_t t = 0; t <= n_tallies; t++) o[t] = (u32)tally_off[t]; offs.push_back(o); }
    for (;;) {
        const std::vector<u32>& in = offs.back();
        u32 mx = 0; for (size_t t = 0; t < n_tallies; t++) mx = std::max(mx, in[t + 1] - in[t]);
        if (mx <= 1) break;
        std::vector<u32> o(n_tallies + 1); o[0] = 0;
        for (size_t t = 0; t < n_tallies; t++) o[t + 1] = o[t] + (in[t + 1] - in[t] + T - 1) / T;
        offs.push_back(o);
    }
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    const size_t half = (size_t)offs.size() > 1 ? (size_t)offs[1][n_tallies] : 1;
    if (!engine_workspace(e, ws_need({33 * total + 64, 8 * (n_tallies + 1), 8 * n_tallies + 8, 4 * n_tallies, 4 * n_tallies + 4, offs.size() * (n_tallies + 1) * 4 + 256 * offs.size(),
                                      (total + 1) * 28 * 4, (half + 1) * 28 * 4}))) return 0;
    ws_carver c{e->ws, 0};
    unsigned char* d_c = c.take<unsigned char>(33 * total + 64); unsigned long long* d_off = c.take<unsigned long long>(n_tallies + 1);
    unsigned long long* d_np = c.take<unsigned long long>(n_tallies + 1); int32_t* d_res = c.take<int32_t>(n_tallies); u32* d_bad = c.take<u32>(n_tallies + 1);
    std::vector<u32*> d_offs; for (size_t r = 0; r < offs.size(); r++) d_offs.push_back(c.take<u32>(n_tallies + 1));
    u32* bufA = c.take<u32>((total + 1) * 28); u32* bufB = c.take<u32>((half + 1) * 28);
    hipStream_t st = (dev && stream) ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    if (dev) { d_c = (unsigned char*)commits33; d_res = results; }
    else if (total) HIPCHK(hipMemcpyAsync(d_c, commits33, 33 * total, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_off, tally_off, 8 * (n_tallies + 1), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_np, n_pos, 8 * n_tallies, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(d_bad, 0, 4 * (n_tallies + 1), st));
    for (size_t r = 0; r < offs.size(); r++) HIPCHK(hipMemcpyAsync(d_offs[r], offs[r].data(), 4 * (n_tallies + 1), hipMemcpyHostToDevice, st));
    if (dev) HIPCHK(hipEventRecord(e->ev_fork, st));           // the host-side offset arrays must have been consumed before this call returns
    HIPCHK(hipEventRecord(e->ev[0], st)); HIPCHK(hipEventRecord(e->ev[2], st));
    if (total) hipLaunchKernelGGL(k_pt_load, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, bufA, d_bad, d_c, d_off, d_np, n_tallies, total);
    HIPCHK(hipEventRecord(e->ev[3], st));
    u32 *pin = bufA, *pout = bufB;
    for (size_t r = 1; r < offs.size(); r++) {
        const size_t lanes = offs[r][n_tallies];
        hipLaunchKernelGGL(k_msm_roundN, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, pout, pin, d_offs[r - 1], d_offs[r], (u32)n_tallies, T);
        u32* tmp = pin; pin = pout; pout = tmp;
    }
    hipLaunchKernelGGL(k_pt_final, dim3((unsigned)((n_tallies + 255) / 256)), dim3(256), 0, st, d_res, pin, d_offs.back(), d_bad, n_tallies);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[1], st));
    if (dev) { HIPCHK(hipEventSynchronize(e->ev_fork)); return 1; }
    HIPCHK(hipMemcpyAsync(results, d_res, 4 * n_tallies, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 1;
}
extern "C" int secp256k1_pedersen_verify_tally_batch(s2k_engine* e, int32_t* results, const unsigned char* commits33, const uint64_t* tally_off,
                                                     const uint64_t* n_pos, size_t n_tallies) {
    return tally_impl(e, nullptr, results, commits33, tally_off, n_pos, n_tallies, 0);
}
extern "C" int secp256k1_pedersen_verify_tally_batch_dev(s2k_engine* e, void* stream, int32_t* results, const unsigned char* commits33, const uint64_t* tally_off_host,
                                                         const uint64_t* n_pos_host, size_t n_tallies) {
    return tally_impl(e, stream, results, commits33, tally_off_host, n_pos_host, n_tallies, 1);
}

