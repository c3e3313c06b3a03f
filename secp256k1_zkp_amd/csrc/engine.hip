// engine.hip -- kernels + C ABI of the gfx950 batch-verification engine (see include/secp256k1_zkp_amd.h).
//
// One translation unit: the per-lane arithmetic lives in the headers next to this file; this file holds the
// __global__ kernels, the device workspace and the extern "C" entry points.  Compiled with
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC   (see __graft_entry__.build / Makefile)
// There is no CPU implementation behind these entry points: without a HIP device every call fails loudly.
#include "gtable.h"
#include "sha256.h"
#include "rangeproof.h"
#include "rangeproof_rewind.h"
#include "schnorr.h"
#include "msm.h"
#include "bppp.h"
#include "surjection.h"
#include "halfagg.h"
#include "pedersen.h"
#include "host_sha256.h"
#include "../../include/secp256k1_zkp_amd.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
#include <mutex>
#include <condition_variable>
#include <string>
#include <thread>
#include <atomic>
#include <chrono>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// ------------------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
static thread_local int g_last_status = 0;            // S2K_STATUS_* of the most recent failing call on this thread
static int s2k_fail(const char* what, const char* detail) {
    g_last_error = std::string(what) + ": " + (detail ? detail : "");
    g_last_status = S2K_STATUS_ENGINE_FAILURE;
    return 0;
}
static int s2k_fail_busy(const char* what, const char* detail) {
    g_last_error = std::string(what) + ": " + (detail ? detail : "");
    g_last_status = S2K_STATUS_BUSY;
    return 0;
}
static int s2k_fail_arg(const char* what, const char* detail) {
    g_last_error = std::string(what) + ": " + (detail ? detail : "");
    g_last_status = S2K_STATUS_ILLEGAL_ARGUMENT;
    return 0;
}
#define HIPCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) return s2k_fail(#call, hipGetErrorString(_e)); } while (0)
#define HIPCHK_NULL(call) do { hipError_t _e = (call); if (_e != hipSuccess) { s2k_fail(#call, hipGetErrorString(_e)); return nullptr; } } while (0)

extern "C" const char* s2k_last_error(void) { return g_last_error.c_str(); }
extern "C" int s2k_last_status(void) { return g_last_status; }
extern "C" void s2k_clear_status(void) { g_last_status = S2K_STATUS_OK; g_last_error.clear(); }

// ------------------------------------------------------------------------------------------------------------
// engine object
// ------------------------------------------------------------------------------------------------------------
struct s2k_dev_pool;
struct s2k_engine {
    int device;
    hipStream_t stream;
    u32* gtab;                 // the device pool's generator table (S2K_GTAB_WORDS words) once a call of this engine has needed it (engine_gtab)
    unsigned char* ws;         // growable HBM workspace
    size_t ws_bytes;
    u32* ptab;                 // per-lane odd-multiples tables (S2K_PTAB_WORDS words per lane), grown on demand
    size_t ptab_lanes;
    hipEvent_t ev[4];          // [0],[1] whole call; [2],[3] dominant kernel
    hipStream_t stream2;       // side stream for kernels that can run next to the main sequence
    hipEvent_t ev_fork, ev_join;
    schnorr_midstate bip340;   // tagged-hash midstate, computed once on the host
    size_t max_lanes;          // lanes per launch (multiple of 256)
    int rp_split;              // rangeproof rings use the two-piece double multiplication (ecmult_lane_split); $S2K_RP_SPLIT=0 turns it off
    // Rangeproof pipeline (rp_launch): two sets of per-proof scratch records, so that the header / prologue / lift / key-sum stage of
    // one chunk (side streams, latency bound) runs underneath the rings kernel of the chunk before it (caller's stream).
    unsigned char* rp_mem[2]; size_t rp_mem_bytes;
    hipStream_t stream_pre;
    hipEvent_t ev_rp_in, ev_rp_fork[2], ev_rp_join[2], ev_rp_pre[2], ev_rp_done[2], ev_rp_draws, ev_rp_rewound;
    int rp_rewound_valid;
    int rp_done_valid[2]; unsigned rp_seq;
    const u32* rp_last_plan[2];   // the work-list headers of the most recent call's last two launch groups (s2k_engine_rp_handback)
    hipEvent_t ev_ring[32][2]; unsigned ring_seq;   // the dominant kernel of the 32 most recent rangeproof calls (several calls may be in flight)
    hipStream_t last_stream; int last_stream_valid; hipEvent_t ev_last;   // see stream_guard
    hipEvent_t ev_msm_fork, ev_msm_join;   // the MSM's gated exact path runs on the side stream, next to the bucket pipeline
    int rp_debug;              // diagnostic launches ($S2K_RP_DEBUG: rp_rings_shared's dbg bits; results are meaningless then)
    int rp_inputs_ready;       // S2K_OPT_RP_INPUTS_READY: the side-stream stage need not wait for earlier work of the caller's stream
    u32* host_flags;           // pinned, 64 bytes (diagnostic read-backs)
    u32* dev_flags;            // device, 64 bytes: [0] the most recent MSM launch overflowed a bucket region (exact path taken)
    std::vector<unsigned char> bp_key;   // serialised generator set the BP++ fixed-base table was built for
    u32* bp_tab;               // [n_gens][16][65536] affine multiples (bppp.h), kept across calls
    int bp_gens_ok;            // every generator of the cached set parsed (what k_bp_gens found when the table was built)
    // The generator table and the cache of rangeproof generator tables live in the device's pool (below); per engine: the mailbox through
    // which k_rp_final reports which tables served verified proofs and which uncached generators keep coming.
    struct s2k_dev_pool* pool;
    rp_gen_mbox* gen_mbox;     // device
    rp_gen_mbox* gen_mbox_host;   // pinned copy taken at the end of the previous rangeproof call
    hipEvent_t ev_mbox; int mbox_pending;
    // pinned staging of the host-buffer rangeproof entry points (rp_host_submit): inputs are packed into it by a few host threads, chunk by
    // chunk, and every finished chunk goes to HBM at once (true DMA from pinned memory: the copies overlap the packing of the next chunks)
    // Two such sets (pinned in / pinned out / their device images), so that a second batch can be gathered and copied while the first one
    // computes (secp256k1_rangeproof_verify_batch_submit / _wait); the copies run on their own stream.
    struct stage_set {
        unsigned char* in; size_t in_bytes; unsigned char* out; size_t out_bytes; unsigned char* dev; size_t dev_bytes;
        hipEvent_t ev_h2d, ev_out; int used;
        uint64_t ticket;                                  // 0: free; otherwise the submission that owns the set until it is waited for
        int sync_owned;                                   // the owner is a synchronous call (it hands the set back by itself)
        int32_t* results; uint64_t* min_value; uint64_t* max_value; size_t n, o_res, o_min, o_max;
    } stage[2];
    uint64_t next_ticket;
    std::condition_variable_any stage_cv;                 // a staging set was handed back (synchronous callers queue for one)
    hipStream_t stream_copy;
    int stage_threads;
    // Two MSM calls in flight (s2k_ecmult_multi_dev / _partial_dev with S2K_OPT_RP_INPUTS_READY): each slot has its own streams, events and
    // workspace, calls alternate between the slots, and the caller's stream only waits for a call's result -- the latency-bound tail of
    // call k (Horner, tree sums, bucket weights: ~25 small launches during which most CUs idle) runs underneath the binning and
    // partial-sum rounds of call k+1.
    struct msm_slot { hipStream_t s, s2; hipEvent_t fork, join, done, in; unsigned char* ws; size_t ws_bytes; unsigned long long seen_epoch; } msm_slot[2];
    unsigned msm_seq;
    int cur_pipe;              // the current entry-point call is a pipelined MSM (stream_guard, msm_pipelined)
    // Work that did NOT go through an MSM slot shares the engine's table arena with the slots' gated exact path: `np_epoch` counts such calls
    // and `ev_last_np` is recorded at the end of each; a slot waits for it whenever it has not yet seen the current epoch (msm_pipelined).
    hipEvent_t ev_last_np; unsigned long long np_epoch; int np_valid;
    int msm_pipeline;          // S2K_OPT_MSM_PIPELINE: small multi-scalar multiplications keep two calls in flight (msm_pipelined)
    int halfagg_host_chain;    // S2K_OPT_HALFAGG_HOST_CHAIN
    int sync_split;            // S2K_OPT_SYNC_SPLIT: a lone synchronous host-buffer rangeproof call goes as two halves
    int stage_log;             // diagnostic builds: phase times of a host-buffer call on stderr
    // diagnostic overrides of the MSM launcher (-DS2K_DIAG builds read them from the environment ONCE, at engine creation; 0 = the plan's choice)
    struct { int c, T, chunk, two_pass, one_pass, bin_plain, no_small; } msm_diag;
    u32* ha_pin; size_t ha_pin_words;    // pinned: the chain states of the half-aggregate randomizer hash, walked on the host (host_sha256.h)
    std::recursive_mutex mu;
};

// The workspace and the table arena are shared by every call of an engine.  Calls on ONE stream are ordered by the stream; a call on
// a different stream than the call before it first waits for that call's last work (an event recorded when every entry point leaves).
struct stream_guard {
    s2k_engine* e; hipStream_t st;
    stream_guard(s2k_engine* e_, hipStream_t st_) : e(e_), st(st_) {
        e->cur_pipe = 0;
        if (e->last_stream_valid && e->last_stream != st) { if (hipStreamWaitEvent(st, e->ev_last, 0) != hipSuccess) (void)hipGetLastError(); }
    }
    ~stream_guard() {
        if (hipEventRecord(e->ev_last, st) == hipSuccess) { e->last_stream = st; e->last_stream_valid = 1; } else (void)hipGetLastError();
        if (!e->cur_pipe) {            // a call that used the engine's own scratch: the MSM slots must not run their exact path under it
            if (hipEventRecord(e->ev_last_np, st) == hipSuccess) { e->np_epoch++; e->np_valid = 1; } else (void)hipGetLastError();
        }
    }
};
// per-lane table scratch for `lanes` concurrent ecmult_lane callers (lane = global thread index of the launch)
static int engine_ptab(s2k_engine* e, size_t lanes) {
    lanes = (lanes + 255) & ~size_t(255);
    if (lanes <= e->ptab_lanes) return 1;
    HIPCHK(hipDeviceSynchronize());                 // earlier launches (possibly on a caller's stream) may still use the old arena
    if (e->ptab) HIPCHK(hipFree(e->ptab));
    e->ptab = nullptr; e->ptab_lanes = 0;
    HIPCHK(hipMalloc((void**)&e->ptab, lanes * S2K_PTAB_WORDS * sizeof(u32)));
    e->ptab_lanes = lanes;
    return 1;
}
// the arena of the rings kernels for `rings` rings: the general form wants S2K_PTAB_WORDS per ring; the shared form, with S2K_RP_K rings per
// lane, S2K_RTAB_WORDS per ring plus per wavefront of 64 lanes the construction's parking area and the lanes' point / challenge parking
static int engine_rtab(s2k_engine* e, size_t rings) {
    rings = (rings + 255) & ~size_t(255);
    const size_t lanes = ((rings + S2K_RP_K - 1) / S2K_RP_K + 255) & ~size_t(255);
    const size_t words = lanes * S2K_RP_K * S2K_RTAB_WORDS + (lanes / 64) * (S2K_RRAW_WAVE_WORDS + (size_t)S2K_RP_K * RP_PARK_WORDS * 64);
    return engine_ptab(e, std::max(rings, (words + S2K_PTAB_WORDS - 1) / S2K_PTAB_WORDS));
}
// Upper bound on lanes per launch: keeps the per-lane table arena at 1.2 GB however large the batch is; bigger
// batches run as several launches over sub-ranges (same stream, so the order of results is unaffected).
// (engine field max_lanes; default 2^20, $S2K_MAX_LANES overrides it -- the tests use a small value to exercise the split)
static int engine_workspace(s2k_engine* e, size_t bytes) {
    if (bytes <= e->ws_bytes) return 1;
    HIPCHK(hipDeviceSynchronize());                 // earlier launches (possibly on a caller's stream) may still use the old workspace
    if (e->ws) HIPCHK(hipFree(e->ws));
    e->ws = nullptr; e->ws_bytes = 0;
    bytes = (bytes + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
    HIPCHK(hipMalloc((void**)&e->ws, bytes));
    e->ws_bytes = bytes;
    return 1;
}

// ------------------------------------------------------------------------------------------------------------
// byte helpers (device)
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ge_load_b64(ge& g, const unsigned char* p) { fe_set_b32_mod(g.x, p); fe_set_b32_mod(g.y, p + 32); }
__device__ __forceinline__ void ge_store_b64(unsigned char* p, const ge& g) { fe_get_b32(p, g.x); fe_get_b32(p + 32, g.y); }
__device__ __forceinline__ void gej_store28(u32* p, const gej& a) {
    fe x = a.x, y = a.y, z = a.z;
    fe_norm_weak(x); fe_norm_weak(y); fe_norm_weak(z);
#pragma unroll
    for (int i = 0; i < 9; i++) { p[i] = x.n[i]; p[9 + i] = y.n[i]; p[18 + i] = z.n[i]; }
    p[27] = (u32)a.inf;
}
__device__ __forceinline__ void gej_load28(gej& a, const u32* p) {
#pragma unroll
    for (int i = 0; i < 9; i++) { a.x.n[i] = p[i]; a.y.n[i] = p[9 + i]; a.z.n[i] = p[18 + i]; }
    a.inf = (int)p[27];
}

// ------------------------------------------------------------------------------------------------------------
// generator table construction (engine creation)
// ------------------------------------------------------------------------------------------------------------
__global__ void k_gtab_base(u32* gtab) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < S2K_GTAB_WINDOWS) gtab_build_base(gtab, w);
}
__global__ void __launch_bounds__(256)
k_gtab_entries(u32* gtab) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 w = t >> (S2K_GTAB_BITS - 1), v = (t & (S2K_GTAB_HALF - 1u)) + 1u;          // v = 1 .. 2^(D-1): the magnitudes of a signed D-bit digit
    // the top window only ever sees the bits that are left of a 256-bit scalar, plus the carry of the recoding
    if (w < S2K_GTAB_WINDOWS && v >= 2 && (w + 1 < S2K_GTAB_WINDOWS || v <= (1u << S2K_GTAB_TOP_BITS) + 1u)) gtab_build_entry(gtab, w, v);
}

// fixed-base table of another point than G (a rangeproof generator): window bases from the 64 generator bytes, then k_gtab_entries
__global__ void k_gen_base(u32* tab, const unsigned char* gen64) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= S2K_GTAB_WINDOWS) return;
    ge g; rp_load_generator(g, gen64);
    gtab_build_base(tab, w, &g);
}
// x of j * 4^ring * 10^exp * H for j = 1..3 (rp_ring_suspect): one multiplication per lane
__global__ void __launch_bounds__(256, 2)
k_gen_xmul(u32* __restrict__ xmul, const unsigned char* __restrict__ gen64, const u32* __restrict__ gtab, u32* __restrict__ ptab) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 total = RP_XMUL_EXPS * RP_MAX_RINGS * 3;
    const int live = t < total;
    const u32 tt = live ? t : 0;
    const u32 j = tt % 3 + 1, ring = (tt / 3) % RP_MAX_RINGS; const int ex = (int)(tt / (3 * RP_MAX_RINGS));
    ge g; rp_load_generator(g, gen64);
    gej A; gej_set_ge(A, g); A.inf = !live;
    scalar c, k, z; rp_ring_const(c, ex, ring); k = c;
    for (u32 i = 1; i < j; i++) sc_add(k, k, c);
    sc_set_zero(z);
    if (!live) sc_set_zero(k);
    __shared__ u32 s_dig[S2K_DIG_WORDS * 256];
    const lane_mem lm{ptab + (size_t)t * S2K_PTAB_WORDS, S2K_LANE_DIG(s_dig)};
    gej R; ecmult_lane(R, A, k, z, 0, gtab, lm);
    ge a; ge_set_gej(a, R);
    if (live) { u32 w[8]; fe_to_words(w, a.x); for (int i = 0; i < 8; i++) xmul[8 * tt + i] = w[i]; }
}

// ------------------------------------------------------------------------------------------------------------
// batch double multiplication  r = na*A + ng*G    (secp256k1_ecmult, src/ecmult.h:47)
// one multiplication per lane; inputs are gathered with byte loads (160 B per lane against ~1.5 M cycles of
// arithmetic -- the loads are noise), the result is converted to affine and serialised in the same kernel.
// ------------------------------------------------------------------------------------------------------------
#ifndef S2K_EB_WAVES
#define S2K_EB_WAVES 2
#endif
__global__ void __launch_bounds__(256, S2K_EB_WAVES)
k_ecmult_batch(unsigned char* __restrict__ r_xy, int32_t* __restrict__ r_inf, const unsigned char* __restrict__ a_xy,
               const unsigned char* __restrict__ a_inf, const unsigned char* __restrict__ na, const unsigned char* __restrict__ ng,
               const u32* __restrict__ gtab, u32* __restrict__ ptab, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int live = i < n;
    const size_t ii = live ? i : 0;
    gej A; scalar sa, sg;
    {
        ge a; ge_load_b64(a, a_xy + 64 * ii);
        gej_set_ge(A, a);
        A.inf = (a_inf ? (a_inf[ii] != 0) : 0) | !live;
    }
    sc_set_b32(sa, na + 32 * ii, nullptr);
    if (ng) sc_set_b32(sg, ng + 32 * ii, nullptr); else sc_set_zero(sg);
    if (!live) { sc_set_zero(sa); sc_set_zero(sg); }
    gej R;
    __shared__ u32 s_dig[S2K_DIG_WORDS * 256];
    const lane_mem lm{ptab + i * S2K_PTAB_WORDS, S2K_LANE_DIG(s_dig)};
    ecmult_lane(R, A, sa, sg, ng != nullptr, gtab, lm);
    ge out;
    ge_set_gej(out, R);
    if (live) {
        if (R.inf) { for (int k = 0; k < 64; k++) r_xy[64 * i + k] = 0; }
        else ge_store_b64(r_xy + 64 * i, out);
        r_inf[i] = R.inf;
    }
}

// ------------------------------------------------------------------------------------------------------------
// per-device table pool + generator-table cache (rangeproof.h, shared-generator form): host side
// ------------------------------------------------------------------------------------------------------------
// The big tables belong to the DEVICE, not to an engine: the 21.5 GB fixed-base table of G and the cache of rangeproof generator tables
// (21.5 GB each) are held once per HIP device in a reference-counted pool that every engine on that device shares.  A second engine on
// a device -- the documented way to give every verifier thread its own stream, scratch and lock -- costs a few streams and events,
// no table memory and no table build.  Tables are built lazily, by the first call that needs one (s2k_engine_reserve warms them up).
// Ordering between engines: a build is stream-ordered on the building engine's stream and publishes an event; every other stream
// that is about to read the table waits for that event until it is known to have completed.  The pool's mutex is held while an engine
// takes its view of the cache AND enqueues the kernels that use it, and a slot's memory is only ever rewritten (eviction, fewer slots)
// after a device-wide synchronisation under that mutex, so no kernel in flight can read a table that is being replaced.
// Lock order: engine mutex, then pool mutex.
// secp256k1_generator_h (src/modules/generator/main_impl.h:30-35): the generator of bench_rangeproof and of every non-asset caller
static const unsigned char k_generator_h[64] = {
    0x50, 0x92, 0x9b, 0x74, 0xc1, 0xa0, 0x49, 0x54, 0xb7, 0x8b, 0x4b, 0x60, 0x35, 0xe9, 0x7a, 0x5e, 0x07, 0x8a, 0x5a, 0x0f, 0x28, 0xec, 0x96, 0xd5, 0x47, 0xbf, 0xee, 0x9a, 0xce, 0x80, 0x3a, 0xc0,
    0x31, 0xd3, 0xc6, 0x86, 0x39, 0x73, 0x92, 0x6e, 0x04, 0x9e, 0x63, 0x7c, 0xb1, 0xb5, 0xf4, 0x0a, 0x36, 0xda, 0xc2, 0x8a, 0xf1, 0x76, 0x69, 0x68, 0xc3, 0x0c, 0x23, 0x13, 0xf3, 0xa3, 0x89, 0x04};
struct s2k_dev_pool {
    int device; int refs;
    std::recursive_mutex mu;
    u32* gtab; hipEvent_t ev_gtab; int gtab_state;           // 0: not built, 1: build queued (ev_gtab behind it), 2: known to be complete
    // Fixed-base tables of rangeproof generators: a small cache keyed by the 64 generator bytes.  Slot tables have the layout of gtab
    // (allocated when a slot is first used and then reused by whatever generator takes the slot); xmul is the x-table of the ring-base
    // multiples (RP_XMUL_WORDS).  gen_keys (device) is what k_rp_header matches a proof's generator against; gen_seen counts the VERIFIED
    // proofs met per uncached generator (k_rp_final reports them through each engine's device mailbox, read at that engine's next call)
    // and a generator is built once it reaches gen_min.  pinned: secp256k1_generator_h and generators cached explicitly -- an automatic
    // build never evicts those.
    struct gen_slot { unsigned char key[64]; u32* tab; u32* xmul; unsigned long long stamp; int valid; int pinned; hipEvent_t ev_ready; int done; } gen[RP_GEN_SLOTS];
    int gen_slots; unsigned long long gen_clock; size_t gen_min; int gen_h;
    unsigned char* gen_keys;   // device, [RP_GEN_SLOTS][64]
    std::vector<std::pair<std::array<unsigned char, 64>, size_t>> gen_seen;
};
static std::mutex g_pools_mu;
static std::vector<s2k_dev_pool*> g_pools;
static void pool_free_tables(s2k_dev_pool* p) {
    if (p->gtab) hipFree(p->gtab);
    p->gtab = nullptr; p->gtab_state = 0;
    for (int i = 0; i < RP_GEN_SLOTS; i++) {
        if (p->gen[i].tab) hipFree(p->gen[i].tab);
        if (p->gen[i].xmul) hipFree(p->gen[i].xmul);
        p->gen[i].tab = nullptr; p->gen[i].xmul = nullptr; p->gen[i].valid = 0;
    }
}
// (the caller has made `device` current)
static s2k_dev_pool* pool_acquire(int device) {
    std::lock_guard<std::mutex> g(g_pools_mu);
    for (auto* p : g_pools) if (p->device == device) { p->refs++; return p; }
    s2k_dev_pool* p = new s2k_dev_pool();
    p->device = device; p->refs = 1; p->gtab = nullptr; p->ev_gtab = nullptr; p->gtab_state = 0; p->gen_keys = nullptr;
    for (int i = 0; i < RP_GEN_SLOTS; i++) { auto& g2 = p->gen[i]; g2.tab = nullptr; g2.xmul = nullptr; g2.valid = 0; g2.stamp = 0; g2.pinned = 0; g2.ev_ready = nullptr; g2.done = 0; }
    p->gen_slots = 2; p->gen_clock = 0; p->gen_min = size_t(1) << 16; p->gen_h = 1;
    if (const char* gs = getenv("S2K_GEN_CACHE")) { const int v = atoi(gs); p->gen_slots = v < 0 ? 0 : (v > RP_GEN_SLOTS ? RP_GEN_SLOTS : v); }
    if (const char* gm = getenv("S2K_GEN_CACHE_MIN")) p->gen_min = (size_t)strtoull(gm, nullptr, 10);
#ifdef S2K_DIAG
    if (const char* gh = getenv("S2K_GEN_CACHE_H")) p->gen_h = atoi(gh) != 0;
#endif
    int ok = hipEventCreateWithFlags(&p->ev_gtab, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < RP_GEN_SLOTS; i++) ok = hipEventCreateWithFlags(&p->gen[i].ev_ready, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc((void**)&p->gen_keys, 64 * RP_GEN_SLOTS) == hipSuccess && hipMemset(p->gen_keys, 0, 64 * RP_GEN_SLOTS) == hipSuccess;
    if (!ok) {
        s2k_fail("s2k_engine_create", "cannot create the device's table pool");
        (void)hipGetLastError();
        if (p->ev_gtab) hipEventDestroy(p->ev_gtab);
        for (int i = 0; i < RP_GEN_SLOTS; i++) if (p->gen[i].ev_ready) hipEventDestroy(p->gen[i].ev_ready);
        if (p->gen_keys) hipFree(p->gen_keys);
        delete p; return nullptr;
    }
    g_pools.push_back(p);
    return p;
}
// (the caller has made the device current and has synchronised it: nothing of the leaving engine is in flight)
static void pool_release(s2k_dev_pool* p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(g_pools_mu);
    if (--p->refs > 0) return;
    g_pools.erase(std::remove(g_pools.begin(), g_pools.end(), p), g_pools.end());
    pool_free_tables(p);
    if (p->gen_keys) hipFree(p->gen_keys);
    if (p->ev_gtab) hipEventDestroy(p->ev_gtab);
    for (int i = 0; i < RP_GEN_SLOTS; i++) if (p->gen[i].ev_ready) hipEventDestroy(p->gen[i].ev_ready);
    delete p;
}
// The table of G, built by the first call on this device that needs it (stream-ordered on that call's stream); every later user's
// stream waits for the build's event until it is known to be over.  Returns the table or nullptr (no memory: s2k_fail was called).
static const u32* engine_gtab(s2k_engine* e, hipStream_t st) {
    s2k_dev_pool* p = e->pool;
    std::lock_guard<std::recursive_mutex> lock(p->mu);
    if (p->gtab_state == 2) return p->gtab;
    if (p->gtab_state == 0) {
        if (hipMalloc((void**)&p->gtab, sizeof(u32) * S2K_GTAB_WORDS) != hipSuccess) { (void)hipGetLastError(); p->gtab = nullptr; s2k_fail("engine_gtab", "no memory for the generator table (21.5 GB of HBM)"); return nullptr; }
        int ok = hipMemsetAsync(p->gtab, 0, sizeof(u32) * S2K_GTAB_WORDS, st) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(k_gtab_base, dim3(1), dim3(64), 0, st, p->gtab);
            hipLaunchKernelGGL(k_gtab_entries, dim3((unsigned)(((size_t)S2K_GTAB_WINDOWS << (S2K_GTAB_BITS - 1)) / 256)), dim3(256), 0, st, p->gtab);
            ok = hipGetLastError() == hipSuccess && hipEventRecord(p->ev_gtab, st) == hipSuccess;
        }
        if (!ok) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); hipFree(p->gtab); p->gtab = nullptr; s2k_fail("engine_gtab", "generator table build failed"); return nullptr; }
        p->gtab_state = 1;
        return p->gtab;
    }
    if (hipEventQuery(p->ev_gtab) == hipSuccess) { p->gtab_state = 2; return p->gtab; }
    (void)hipGetLastError();
    if (hipStreamWaitEvent(st, p->ev_gtab, 0) != hipSuccess) { (void)hipGetLastError(); s2k_fail("engine_gtab", "hipStreamWaitEvent failed"); return nullptr; }
    return p->gtab;
}
#define ENGINE_GTAB(e, st) do { if (!((e)->gtab = const_cast<u32*>(engine_gtab((e), (st))))) return 0; } while (0)
// The cache as the kernels of one launch see it; `st` / `sp`: the streams that will read the tables (made to wait for builds still in flight)
static rp_gen_dev gen_dev_view(s2k_engine* e, hipStream_t st, hipStream_t sp) {
    s2k_dev_pool* p = e->pool;
    rp_gen_dev gc; gc.keys = p->gen_keys; gc.valid = 0; gc.any = 0;
    for (int i = 0; i < RP_GEN_SLOTS; i++) {
        auto& g = p->gen[i];
        int v = i < p->gen_slots && g.valid;
        if (v && !g.done) {
            if (hipEventQuery(g.ev_ready) == hipSuccess) g.done = 1;
            else {
                (void)hipGetLastError();
                if (hipStreamWaitEvent(st, g.ev_ready, 0) != hipSuccess || (sp && hipStreamWaitEvent(sp, g.ev_ready, 0) != hipSuccess)) { (void)hipGetLastError(); v = 0; }      // cannot order: do without this table
            }
        }
        gc.tab[i] = v ? g.tab : nullptr; gc.xmul[i] = v ? g.xmul : nullptr;
        if (v) { gc.valid |= 1u << i; gc.any = (u32)i; }
    }
    return gc;
}
static int gen_cache_find(s2k_dev_pool* p, const unsigned char* key) {
    for (int i = 0; i < p->gen_slots; i++) if (p->gen[i].valid && !memcmp(p->gen[i].key, key, 64)) { p->gen[i].stamp = ++p->gen_clock; return i; }
    return -1;
}
// Builds (stream-ordered on `st`) the tables of `key` into a free slot or the least recently used one; -1 when there is no memory for a
// table (the proofs then simply keep the general form).  pinned = 0 is an AUTOMATIC build (a generator that kept coming on valid
// proofs): it only takes a free slot or the slot of another automatically built table -- never the table of secp256k1_generator_h or
// one the application asked for -- and returns -1 when there is none.  (Pool mutex held by the caller.)
static int gen_cache_build(s2k_engine* e, hipStream_t st, const unsigned char* key, int pinned) {
    s2k_dev_pool* p = e->pool;
    int slot = gen_cache_find(p, key);
    if (slot >= 0) { if (pinned) p->gen[slot].pinned = 1; return slot; }
    if (p->gen_slots <= 0) return -1;
    slot = -1;
    for (int i = 0; i < p->gen_slots; i++) {
        if (!p->gen[i].valid) { slot = i; break; }
        if (!pinned && p->gen[i].pinned) continue;
        if (slot < 0 || p->gen[i].stamp < p->gen[slot].stamp) slot = i;
    }
    if (slot < 0) return -1;
    s2k_dev_pool::gen_slot& g = p->gen[slot];
    const u32* gtab = engine_gtab(e, st);
    if (!gtab) return -1;
    e->gtab = const_cast<u32*>(gtab);
    // a slot whose memory may still be read -- by this engine's side streams or by another engine's kernels -- is rewritten only once the
    // device is idle (an eviction is a 0.3 s table build anyway)
    if (g.tab && hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (!g.tab) {
        if (hipMalloc((void**)&g.tab, sizeof(u32) * S2K_GTAB_WORDS) != hipSuccess) { (void)hipGetLastError(); g.tab = nullptr; return -1; }
        if (hipMalloc((void**)&g.xmul, sizeof(u32) * RP_XMUL_WORDS) != hipSuccess) { (void)hipGetLastError(); hipFree(g.tab); g.tab = nullptr; g.xmul = nullptr; return -1; }
    }
    if (!engine_ptab(e, 2048)) return -1;
    g.valid = 0;
    memcpy(g.key, key, 64);
    if (hipMemcpyAsync(p->gen_keys + 64 * slot, g.key, 64, hipMemcpyHostToDevice, st) != hipSuccess) { (void)hipGetLastError(); return -1; }
    hipLaunchKernelGGL(k_gen_base, dim3(1), dim3(64), 0, st, g.tab, p->gen_keys + 64 * slot);
    hipLaunchKernelGGL(k_gtab_entries, dim3((unsigned)(((size_t)S2K_GTAB_WINDOWS << (S2K_GTAB_BITS - 1)) / 256)), dim3(256), 0, st, g.tab);
    hipLaunchKernelGGL(k_gen_xmul, dim3((RP_XMUL_EXPS * RP_MAX_RINGS * 3 + 255) / 256), dim3(256), 0, st, g.xmul, p->gen_keys + 64 * slot, gtab, e->ptab);
    if (hipGetLastError() != hipSuccess || hipEventRecord(g.ev_ready, st) != hipSuccess) { (void)hipGetLastError(); return -1; }
    g.valid = 1; g.done = 0; g.pinned = pinned; g.stamp = ++p->gen_clock;
    return slot;
}
// `count` more verified proofs were seen with this (uncached) generator; returns 1 when it has now been seen often enough to deserve a table
static int gen_note_seen(s2k_dev_pool* p, const unsigned char* key, size_t count) {
    for (auto& it : p->gen_seen) if (!memcmp(it.first.data(), key, 64)) { it.second += count; return it.second >= p->gen_min; }
    if (p->gen_seen.size() >= 64) {                         // bounded: forget the least seen
        size_t lo = 0; for (size_t i = 1; i < p->gen_seen.size(); i++) if (p->gen_seen[i].second < p->gen_seen[lo].second) lo = i;
        p->gen_seen.erase(p->gen_seen.begin() + lo);
    }
    std::array<unsigned char, 64> k; memcpy(k.data(), key, 64);
    p->gen_seen.emplace_back(k, count);
    return count >= p->gen_min;
}
static void gen_forget_seen(s2k_dev_pool* p, const unsigned char* key) {
    for (size_t i = 0; i < p->gen_seen.size(); i++) if (!memcmp(p->gen_seen[i].first.data(), key, 64)) { p->gen_seen.erase(p->gen_seen.begin() + i); return; }
}
// Start of a rangeproof call (pool mutex held): (1) secp256k1_generator_h gets its table once, (2) what the final kernels of this engine's
// call before reported through the mailbox (its pinned copy is only read once the copy has completed): tables that served valid proofs
// get a fresh least-recently-used stamp, uncached generators are counted by their VALID proofs and at most one that is due is built per call.
static void gen_cache_service(s2k_engine* e, hipStream_t st) {
    s2k_dev_pool* p = e->pool;
    if (p->gen_slots <= 0) return;
    if (p->gen_h == 1) { p->gen_h = 2; (void)gen_cache_build(e, st, k_generator_h, 1); }
    if (e->mbox_pending && hipEventQuery(e->ev_mbox) == hipSuccess) {
        e->mbox_pending = 0;
        // (slot indices in the report are those of the view the reporting call took; a slot replaced since then just gets a fresh stamp early)
        for (int i = 0; i < p->gen_slots; i++) if (p->gen[i].valid && e->gen_mbox_host->hits[i]) p->gen[i].stamp = ++p->gen_clock;
        int built = 0;
        for (int m = 0; m < RP_GEN_MBOX; m++) {
            if (!e->gen_mbox_host->tag[m] || !e->gen_mbox_host->count[m]) continue;
            const unsigned char* key = e->gen_mbox_host->key[m];
            if (rp_gen_tag(key) != e->gen_mbox_host->tag[m]) continue;          // (key bytes of a slot whose claimant never wrote them)
            int cached = 0;
            for (int i = 0; i < p->gen_slots; i++) if (p->gen[i].valid && !memcmp(p->gen[i].key, key, 64)) cached = 1;
            if (cached) continue;
            if (gen_note_seen(p, key, e->gen_mbox_host->count[m]) && !built && gen_cache_build(e, st, key, 0) >= 0) { gen_forget_seen(p, key); built = 1; }
        }
    } else if (e->mbox_pending) (void)hipGetLastError();
}
// End of a rangeproof call: copy the mailbox out and clear it (both on `st`, behind the call's kernels)
static void gen_cache_collect(s2k_engine* e, hipStream_t st) {
    if (e->pool->gen_slots <= 0 || e->mbox_pending) return;
    if (hipMemcpyAsync(e->gen_mbox_host, e->gen_mbox, sizeof(rp_gen_mbox), hipMemcpyDeviceToHost, st) != hipSuccess) { (void)hipGetLastError(); return; }
    if (hipMemsetAsync(e->gen_mbox, 0, sizeof(rp_gen_mbox), st) != hipSuccess) { (void)hipGetLastError(); return; }
    if (hipEventRecord(e->ev_mbox, st) == hipSuccess) e->mbox_pending = 1; else (void)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// C ABI: engine lifecycle
// ------------------------------------------------------------------------------------------------------------
extern "C" s2k_engine* s2k_engine_create(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { s2k_fail("s2k_engine_create", "no HIP device visible (this engine has no CPU path)"); return nullptr; }
    if (device < 0 || device >= count) { s2k_fail("s2k_engine_create", "device ordinal out of range"); return nullptr; }
    HIPCHK_NULL(hipSetDevice(device));
    s2k_engine* e = new s2k_engine();
    e->device = device; e->ws = nullptr; e->ws_bytes = 0; e->gtab = nullptr; e->ptab = nullptr; e->ptab_lanes = 0; e->host_flags = nullptr; e->dev_flags = nullptr; e->bp_tab = nullptr; e->bp_gens_ok = 0;
    e->stream = nullptr; e->stream2 = nullptr; e->ev_fork = nullptr; e->ev_join = nullptr; for (int i = 0; i < 4; i++) e->ev[i] = nullptr;
    for (int i = 0; i < 32; i++) e->ev_ring[i][0] = e->ev_ring[i][1] = nullptr;
    e->ring_seq = 0;
    e->last_stream = nullptr; e->last_stream_valid = 0; e->ev_last = nullptr; e->ev_msm_fork = nullptr; e->ev_msm_join = nullptr;
    e->ev_rp_draws = nullptr; e->ev_rp_rewound = nullptr; e->rp_rewound_valid = 0;
    e->stream_pre = nullptr; e->ev_rp_in = nullptr; e->rp_mem_bytes = 0; e->rp_seq = 0; e->rp_inputs_ready = 0;
    e->rp_last_plan[0] = e->rp_last_plan[1] = nullptr;
    e->ha_pin = nullptr; e->ha_pin_words = 0;
    for (int i = 0; i < 2; i++) { auto& m = e->msm_slot[i]; m.s = m.s2 = nullptr; m.fork = m.join = m.done = m.in = nullptr; m.ws = nullptr; m.ws_bytes = 0; m.seen_epoch = 0; }
    e->msm_seq = 0; e->cur_pipe = 0; e->ev_last_np = nullptr; e->np_epoch = 0; e->np_valid = 0;
    e->msm_pipeline = 0; e->halfagg_host_chain = 1; e->sync_split = 1; e->stage_log = 0;
    e->msm_diag = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 2; i++) { e->rp_mem[i] = nullptr; e->ev_rp_fork[i] = e->ev_rp_join[i] = e->ev_rp_pre[i] = e->ev_rp_done[i] = nullptr; e->rp_done_valid[i] = 0; }
    e->rp_debug = 0;
    for (int i = 0; i < 2; i++) { auto& S = e->stage[i]; S.in = S.out = S.dev = nullptr; S.in_bytes = S.out_bytes = S.dev_bytes = 0; S.ev_h2d = S.ev_out = nullptr; S.used = 0; S.ticket = 0; S.sync_owned = 0; }
    e->next_ticket = 1; e->stream_copy = nullptr;
    { unsigned hc = std::thread::hardware_concurrency(); e->stage_threads = (int)std::min(8u, std::max(1u, hc / 2)); }
    if (const char* th = getenv("S2K_STAGE_THREADS")) { const int v = atoi(th); if (v >= 1 && v <= 64) e->stage_threads = v; }
#ifdef S2K_DIAG          /* diagnostic builds only (tools/rings_parts.py builds its own library with -DS2K_DIAG): a verifier's verdicts never depend on the environment */
    if (const char* sg = getenv("S2K_RP_DEBUG")) e->rp_debug = atoi(sg);
    e->stage_log = getenv("S2K_STAGE_LOG") != nullptr;
    {   auto num = [](const char* k) { const char* v = getenv(k); return v ? atoi(v) : 0; };
        e->msm_diag.c = num("S2K_MSM_C"); e->msm_diag.T = num("S2K_MSM_T"); e->msm_diag.chunk = num("S2K_MSM_CHUNK");
        e->msm_diag.two_pass = getenv("S2K_MSM_TWO_PASS") != nullptr; e->msm_diag.one_pass = getenv("S2K_MSM_ONE_PASS") != nullptr;
        e->msm_diag.bin_plain = getenv("S2K_MSM_BIN_PLAIN") != nullptr; e->msm_diag.no_small = getenv("S2K_MSM_NO_SMALL") != nullptr; }
#endif
    e->pool = nullptr;
    e->gen_mbox = nullptr; e->gen_mbox_host = nullptr; e->ev_mbox = nullptr; e->mbox_pending = 0;
#define S2K_CREATE_CHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) { s2k_fail(#call, hipGetErrorString(_e)); s2k_engine_destroy(e); return nullptr; } } while (0)
    schnorr_tag_midstate(e->bip340);
    e->max_lanes = size_t(1) << 20;
    e->rp_split = 1;
    S2K_CREATE_CHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    {   // the side stream carries throughput-bound kernels that run NEXT TO short latency-bound ones on the main stream (k_rp_lift beside
        // k_rp_prologue): lowest priority, so that the main stream's few waves are placed first instead of queueing behind 8192 others
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = 0; }
        S2K_CREATE_CHK(hipStreamCreateWithPriority(&e->stream2, hipStreamNonBlocking, lo));
    }
    S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    for (int i = 0; i < 32; i++) { S2K_CREATE_CHK(hipEventCreate(&e->ev_ring[i][0])); S2K_CREATE_CHK(hipEventCreate(&e->ev_ring[i][1])); }
    S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_last, hipEventDisableTiming));
    S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_last_np, hipEventDisableTiming));
    S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_msm_fork, hipEventDisableTiming));
    S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_msm_join, hipEventDisableTiming));
    S2K_CREATE_CHK(hipStreamCreateWithFlags(&e->stream_pre, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
        auto& m = e->msm_slot[i];
        S2K_CREATE_CHK(hipStreamCreateWithFlags(&m.s, hipStreamNonBlocking)); S2K_CREATE_CHK(hipStreamCreateWithFlags(&m.s2, hipStreamNonBlocking));
        S2K_CREATE_CHK(hipEventCreateWithFlags(&m.fork, hipEventDisableTiming)); S2K_CREATE_CHK(hipEventCreateWithFlags(&m.join, hipEventDisableTiming));
        S2K_CREATE_CHK(hipEventCreateWithFlags(&m.done, hipEventDisableTiming)); S2K_CREATE_CHK(hipEventCreateWithFlags(&m.in, hipEventDisableTiming));
    }
    S2K_CREATE_CHK(hipStreamCreateWithFlags(&e->stream_copy, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) { S2K_CREATE_CHK(hipEventCreateWithFlags(&e->stage[i].ev_h2d, hipEventDisableTiming)); S2K_CREATE_CHK(hipEventCreateWithFlags(&e->stage[i].ev_out, hipEventDisableTiming)); }
    S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_rp_in, hipEventDisableTiming));
    S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_rp_draws, hipEventDisableTiming));
    S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_rp_rewound, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) {
        S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_rp_fork[i], hipEventDisableTiming));
        S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_rp_join[i], hipEventDisableTiming));
        S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_rp_pre[i], hipEventDisableTiming));
        S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_rp_done[i], hipEventDisableTiming));
    }
    for (int i = 0; i < 4; i++) S2K_CREATE_CHK(hipEventCreate(&e->ev[i]));
    S2K_CREATE_CHK(hipHostMalloc((void**)&e->host_flags, 64, hipHostMallocDefault));
    S2K_CREATE_CHK(hipMalloc((void**)&e->dev_flags, 64));
    S2K_CREATE_CHK(hipMemset(e->dev_flags, 0, 64));
    S2K_CREATE_CHK(hipMalloc((void**)&e->gen_mbox, sizeof(rp_gen_mbox)));
    S2K_CREATE_CHK(hipMemset(e->gen_mbox, 0, sizeof(rp_gen_mbox)));
    S2K_CREATE_CHK(hipHostMalloc((void**)&e->gen_mbox_host, sizeof(rp_gen_mbox), hipHostMallocDefault));
    S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_mbox, hipEventDisableTiming));
    e->pool = pool_acquire(device);            // the device's tables: shared with every other engine on it, built on first use
    if (!e->pool) { s2k_engine_destroy(e); return nullptr; }
#undef S2K_CREATE_CHK
    return e;
}
extern "C" void s2k_engine_destroy(s2k_engine* e) {
    if (!e) return;
    hipSetDevice(e->device);
    hipDeviceSynchronize();            // `_dev` calls may have been issued on caller streams: nothing of this engine may still be in flight
    if (e->ws) hipFree(e->ws);
    if (e->ptab) hipFree(e->ptab);
    if (e->bp_tab) hipFree(e->bp_tab);
    if (e->host_flags) hipHostFree(e->host_flags);
    if (e->ha_pin) hipHostFree(e->ha_pin);
    for (int i = 0; i < 2; i++) {
        auto& m = e->msm_slot[i];
        if (m.ws) hipFree(m.ws);
        if (m.fork) hipEventDestroy(m.fork); if (m.join) hipEventDestroy(m.join); if (m.done) hipEventDestroy(m.done); if (m.in) hipEventDestroy(m.in);
        if (m.s) hipStreamDestroy(m.s); if (m.s2) hipStreamDestroy(m.s2);
    }
    for (int i = 0; i < 2; i++) {
        auto& S = e->stage[i];
        if (S.in) hipHostFree(S.in);
        if (S.out) hipHostFree(S.out);
        if (S.dev) hipFree(S.dev);
        if (S.ev_h2d) hipEventDestroy(S.ev_h2d);
        if (S.ev_out) hipEventDestroy(S.ev_out);
    }
    if (e->stream_copy) hipStreamDestroy(e->stream_copy);
    if (e->dev_flags) hipFree(e->dev_flags);
    pool_release(e->pool);
    if (e->gen_mbox) hipFree(e->gen_mbox);
    if (e->gen_mbox_host) hipHostFree(e->gen_mbox_host);
    if (e->ev_mbox) hipEventDestroy(e->ev_mbox);
    for (int i = 0; i < 4; i++) if (e->ev[i]) hipEventDestroy(e->ev[i]);
    for (int i = 0; i < 2; i++) {
        if (e->rp_mem[i]) hipFree(e->rp_mem[i]);
        if (e->ev_rp_fork[i]) hipEventDestroy(e->ev_rp_fork[i]);
        if (e->ev_rp_join[i]) hipEventDestroy(e->ev_rp_join[i]);
        if (e->ev_rp_pre[i]) hipEventDestroy(e->ev_rp_pre[i]);
        if (e->ev_rp_done[i]) hipEventDestroy(e->ev_rp_done[i]);
    }
    for (int i = 0; i < 32; i++) for (int j = 0; j < 2; j++) if (e->ev_ring[i][j]) hipEventDestroy(e->ev_ring[i][j]);
    if (e->ev_last) hipEventDestroy(e->ev_last);
    if (e->ev_last_np) hipEventDestroy(e->ev_last_np);
    if (e->ev_msm_fork) hipEventDestroy(e->ev_msm_fork);
    if (e->ev_msm_join) hipEventDestroy(e->ev_msm_join);
    if (e->ev_rp_in) hipEventDestroy(e->ev_rp_in);
    if (e->ev_rp_draws) hipEventDestroy(e->ev_rp_draws);
    if (e->ev_rp_rewound) hipEventDestroy(e->ev_rp_rewound);
    if (e->stream_pre) { hipStreamSynchronize(e->stream_pre); hipStreamDestroy(e->stream_pre); }
    if (e->ev_fork) hipEventDestroy(e->ev_fork);
    if (e->ev_join) hipEventDestroy(e->ev_join);
    if (e->stream2) { hipStreamSynchronize(e->stream2); hipStreamDestroy(e->stream2); }
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}
extern "C" int s2k_engine_sync(s2k_engine* e) {
    if (!e) return s2k_fail("s2k_engine_sync", "null engine");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 1;
}
extern "C" const void* s2k_engine_gtable(s2k_engine* e, size_t* bytes) {
    if (bytes) *bytes = sizeof(u32) * S2K_GTAB_WORDS;
    if (!e) return nullptr;
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    if (hipSetDevice(e->device) != hipSuccess) return nullptr;
    const u32* t = engine_gtab(e, e->stream);                  // (built now if no call has needed it yet)
    if (!t || hipStreamSynchronize(e->stream) != hipSuccess) return nullptr;
    e->gtab = const_cast<u32*>(t);
    return t;
}
extern "C" int s2k_engine_last_msm_fallback(s2k_engine* e) {
    if (!e) return 0;
    u32 f = 0;
    hipSetDevice(e->device);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&f, e->dev_flags, 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return f != 0;
}
extern "C" int s2k_engine_rp_handback(s2k_engine* e, uint32_t out[4]) {
    if (!e || !out) return s2k_fail_arg("s2k_engine_rp_handback", "illegal argument");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    for (int k = 0; k < 4; k++) out[k] = 0;
    for (int i = 0; i < 2; i++) {
        if (!e->rp_last_plan[i]) continue;
        u32 v[4];
        HIPCHK(hipMemcpy(v, e->rp_last_plan[i], sizeof(v), hipMemcpyDeviceToHost));
        for (int k = 0; k < 4; k++) out[k] += v[k];
    }
    return 1;
}
#ifdef S2K_PROF
// diagnostic builds only: read (and clear) the per-region cycle table of s2k_common.h
extern "C" __attribute__((visibility("default"))) int s2k_prof_read(unsigned long long out[16]) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(s2k_prof_slots), 16 * sizeof(unsigned long long)) != hipSuccess) return 0;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(s2k_prof_slots), z, sizeof(z)) == hipSuccess;
}
#endif
extern "C" float s2k_engine_last_ms(s2k_engine* e, int which) {
    float ms = -1.0f;
    if (!e) return ms;
    hipSetDevice(e->device);
    if (which >= 16 && which < 48) {                   // dominant kernel of the (which-16)-th most recent rangeproof call
        const unsigned back = (unsigned)(which - 16);
        if (back >= e->ring_seq) return ms;
        const unsigned i = (e->ring_seq - 1u - back) & 31u;
        if (hipEventElapsedTime(&ms, e->ev_ring[i][0], e->ev_ring[i][1]) != hipSuccess) { (void)hipGetLastError(); ms = -1.0f; }
        return ms;
    }
    if (which == 0) { if (hipEventElapsedTime(&ms, e->ev[0], e->ev[1]) != hipSuccess) ms = -1.0f; }
    else            { if (hipEventElapsedTime(&ms, e->ev[2], e->ev[3]) != hipSuccess) ms = -1.0f; }
    return ms;
}

// stage host buffers through the workspace: small helper that carves 256-byte aligned pieces
struct ws_carver {
    unsigned char* base; size_t off;
    template <class T> T* take(size_t count) {
        off = (off + 255) & ~size_t(255);
        T* p = (T*)(base + off); off += count * sizeof(T); return p;
    }
};
static size_t ws_need(std::initializer_list<size_t> sizes) {
    size_t t = 0; for (size_t s : sizes) t = ((t + 255) & ~size_t(255)) + s; return t + 256;
}

// ------------------------------------------------------------------------------------------------------------
// C ABI: ecmult batch
// ------------------------------------------------------------------------------------------------------------
extern "C" int s2k_ecmult_batch_dev(s2k_engine* e, void* stream, unsigned char* r_xy, int32_t* r_inf, const unsigned char* a_xy,
                                    const unsigned char* a_inf, const unsigned char* na, const unsigned char* ng, size_t n) {
    if (!e) return s2k_fail("s2k_ecmult_batch_dev", "null engine");
    if (n == 0) return 1;
    HIPCHK(hipSetDevice(e->device));
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    if (!engine_ptab(e, ((std::min(n, e->max_lanes) + 255) / 256) * 256)) return 0;
    ENGINE_GTAB(e, st);
    HIPCHK(hipEventRecord(e->ev[0], st));
    HIPCHK(hipEventRecord(e->ev[2], st));
    for (size_t i0 = 0; i0 < n; i0 += e->max_lanes) {
        const size_t m = std::min(n - i0, e->max_lanes);
        hipLaunchKernelGGL(k_ecmult_batch, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, r_xy + 64 * i0, r_inf + i0, a_xy + 64 * i0,
                           a_inf ? a_inf + i0 : nullptr, na + 32 * i0, ng ? ng + 32 * i0 : nullptr, e->gtab, e->ptab, m);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[3], st));
    HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
}
extern "C" int s2k_ecmult_batch(s2k_engine* e, unsigned char* r_xy, int32_t* r_inf, const unsigned char* a_xy,
                                const unsigned char* a_inf, const unsigned char* na, const unsigned char* ng, size_t n) {
    if (!e) return s2k_fail("s2k_ecmult_batch", "null engine");
    if (n == 0) return 1;
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    if (!engine_workspace(e, ws_need({64 * n, 4 * n, 64 * n, n, 32 * n, 32 * n}))) return 0;
    ws_carver w{e->ws, 0};
    unsigned char* d_r = w.take<unsigned char>(64 * n); int32_t* d_inf = w.take<int32_t>(n);
    unsigned char* d_a = w.take<unsigned char>(64 * n); unsigned char* d_ai = w.take<unsigned char>(n);
    unsigned char* d_na = w.take<unsigned char>(32 * n); unsigned char* d_ng = w.take<unsigned char>(32 * n);
    HIPCHK(hipMemcpyAsync(d_a, a_xy, 64 * n, hipMemcpyHostToDevice, e->stream));
    if (a_inf) HIPCHK(hipMemcpyAsync(d_ai, a_inf, n, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d_na, na, 32 * n, hipMemcpyHostToDevice, e->stream));
    if (ng) HIPCHK(hipMemcpyAsync(d_ng, ng, 32 * n, hipMemcpyHostToDevice, e->stream));
    if (!s2k_ecmult_batch_dev(e, nullptr, d_r, d_inf, d_a, a_inf ? d_ai : nullptr, d_na, ng ? d_ng : nullptr, n)) return 0;
    HIPCHK(hipMemcpyAsync(r_xy, d_r, 64 * n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(r_inf, d_inf, 4 * n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 1;
}


// ------------------------------------------------------------------------------------------------------------
// Borromean rangeproof batch verification (rangeproof.h): five kernels on one stream
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_rp_header(rp_ws ws, uint64_t* min_value, uint64_t* max_value, const unsigned char* proofs, const uint64_t* proof_off, const unsigned char* gens64,
            rp_gen_dev gc, size_t n) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (p == 0) { ws.plan[0] = 0; ws.plan[1] = 0; ws.plan[2] = 0; ws.plan[3] = 0; }           // the work lists of K3 (k_rp_sum fills them) and the hand-back tallies
    uint64_t mn, mx;
    rp_header(ws.rec[p], &mn, &mx, proofs + proof_off[p], proof_off[p + 1] - proof_off[p]);
    min_value[p] = mn; max_value[p] = mx;
    // which cached generator table (if any) serves this proof (a generator without one is reported by k_rp_final, for proofs that verified)
    ws.rec[p].gslot = rp_gen_lookup(gc, gens64 + 64 * p);
}
// three waves per 64 proofs: wave 0 commitment + min_value*H, wave 1 generator flag + message hash, wave 2 ring bases
__global__ void __launch_bounds__(192)
k_rp_prologue(rp_ws ws, const uint64_t* min_value, const unsigned char* commits33, const unsigned char* proofs,
              const uint64_t* proof_off, const unsigned char* extra, const uint64_t* extra_off, const unsigned char* gens64, size_t n) {
    // few, latency-bound waves that share the SIMDs with the throughput-bound lift kernel: ask the arbiter to issue them first
    __builtin_amdgcn_s_setprio(3);
    const u32 role = threadIdx.x >> 6;
    const size_t p = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
    int commit_ok = 0;
    if (p < n) {
        rp_rec& rec = ws.rec[p];
        if (role == 0) commit_ok = rp_pp_commit(rec, min_value[p], commits33 + 33 * p, gens64 + 64 * p);
        else if (role == 1) {
            const unsigned char* ex = nullptr; uint64_t exlen = 0;
            if (extra && extra_off) { ex = extra + extra_off[p]; exlen = extra_off[p + 1] - extra_off[p]; if (exlen == 0) ex = nullptr; }
            rp_pp_hash(rec, commits33 + 33 * p, proofs + proof_off[p], ex, exlen, gens64 + 64 * p);
        } else rp_pp_bases(rec, ws.bases + p * RP_MAX_RINGS * RP_GEJ_WORDS, gens64 + 64 * p, ws.dbases + p * RP_MAX_RINGS * RP_GEJ_WORDS);
    }
    __syncthreads();
    if (p < n && role == 0 && (ws.rec[p].hdr & 1u) && commit_ok) ws.rec[p].ok = 1;      // (a commitment encoding that does not parse: never valid)
}
__global__ void __launch_bounds__(256)
k_rp_lift(rp_ws ws, const unsigned char* proofs, const uint64_t* proof_off, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t p = t >> 5; const u32 ring = (u32)(t & 31);
    if (p >= n) return;
    const rp_rec& rec = ws.rec[p];
    if (!(rec.hdr & 1u) || ring + 1 >= rec.rings) return;          // needs the header only: runs next to k_rp_prologue
    rp_lift(rec, ws.pub0 + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS, ws.lift_ok + p * RP_MAX_RINGS + ring, proofs + proof_off[p], ring);
}
// K2 also writes the work lists of stage K3 (the rings kernels never look at a proof that needs no ring work, and no lane idles on a ring
// beyond a proof's count -- a 52-bit proof has 26 rings, a 32-bit one 16):
//   mapF[i] = proof | group << 20   groups of S2K_RP_K consecutive rings of the proofs whose generator has a cached table (shared form)
//   mapG[i] = proof | ring << 20    the single rings of all other proofs that passed the earlier stages (general form)
//   plan[0], plan[1] = the two list lengths (zeroed by k_rp_header).  One atomic per wavefront reserves its proofs' entries; the order of
//   the lists is irrelevant.  The shared-form kernel appends to mapG what it hands back.
__global__ void __launch_bounds__(64)
k_rp_sum(rp_ws ws, u32 gen_valid, size_t n) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 cF = 0, cG = 0, rings = 0;
    if (p < n) {
        rp_sum(ws.rec[p], ws.pub0 + p * RP_MAX_RINGS * RP_GEJ_WORDS, ws.lift_ok + p * RP_MAX_RINGS);
        const rp_rec& rec = ws.rec[p];
        if (rec.ok) {
            rings = rec.rings;
            const int fast = gen_valid && rec.gslot < RP_GEN_SLOTS && ((gen_valid >> rec.gslot) & 1u);
            if (fast) cF = (rings + S2K_RP_K - 1) / S2K_RP_K; else cG = rings;
        }
    }
    u32 pf = cF, pg = cG;                                    // inclusive prefix sums over the wavefront (the workgroup is one wavefront)
    for (int d = 1; d < 64; d <<= 1) {
        const u32 a = __shfl_up(pf, d), b = __shfl_up(pg, d);
        if ((int)threadIdx.x >= d) { pf += a; pg += b; }
    }
    u32 baseF = 0, baseG = 0;
    if (threadIdx.x == 63) { baseF = pf ? atomicAdd(&ws.plan[0], pf) : 0u; baseG = pg ? atomicAdd(&ws.plan[1], pg) : 0u; }
    baseF = __shfl(baseF, 63); baseG = __shfl(baseG, 63);
    u32 oF = baseF + pf - cF, oG = baseG + pg - cG;
    for (u32 g = 0; g < cF; g++) ws.mapF[oF + g] = (u32)p | (g << 20);
    for (u32 r = 0; r < cG; r++) ws.mapG[oG + r] = (u32)p | (r << 20);
}
#ifndef S2K_RINGS_WAVES
#define S2K_RINGS_WAVES 2
#endif
// K3 comes as two kernels:
//   k_rp_rings_shared  the shared-generator form (rangeproof.h: rp_rings_shared): lane t takes group mapF[t] -- S2K_RP_K consecutive rings of a
//                      proof whose generator has a cached fixed-base table; a wavefront that meets a suspect ring or an exceptional
//                      addition (adversarial inputs only) appends its rings to mapG instead;
//   k_rp_rings         the general form: lane t takes ring mapG[t].
// Two kernels rather than one with both bodies: each gets its own register allocation (the combined kernel spilled 325 VGPRs) and the
// hot loops of one form do not share the instruction cache with the other's.
__global__ void __launch_bounds__(256, S2K_RINGS_WAVES)
k_rp_rings_shared(rp_ws ws, const unsigned char* __restrict__ proofs, const uint64_t* __restrict__ proof_off, const u32* __restrict__ gtab, u32* __restrict__ ptab, u32* ev,
                  rp_gen_dev gc, u32 dbg) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 nF = ws.plan[0];
    if ((t & ~size_t(63)) >= nF) return;
    const int live = t < nF;
    const u32 item = ws.mapF[live ? t : 0];
    const size_t p = item & 0xFFFFFu; const u32 g = item >> 20;
    const rp_rec& rec = ws.rec[p];
    __shared__ u32 s_dig[S2K_RING_DIG_WORDS * 256];
    const u32 sl = rec.gslot < RP_GEN_SLOTS ? rec.gslot : gc.any;
    const size_t lanes = (size_t)gridDim.x * 256, wave = t >> 6, lane = t & 63;
    u32* const raw0 = ptab + lanes * S2K_RP_K * S2K_RTAB_WORDS;
    const rp_shared_mem M{ptab + t * S2K_RP_K * S2K_RTAB_WORDS, raw0 + wave * S2K_RRAW_WAVE_WORDS + lane,
                          raw0 + (lanes >> 6) * S2K_RRAW_WAVE_WORDS + wave * (S2K_RP_K * RP_PARK_WORDS * 64) + lane, S2K_LANE_DIG(s_dig)};
    const int served = rp_rings_shared<S2K_RP_K>(rec, ws.pub0 + (p * RP_MAX_RINGS + g * S2K_RP_K) * RP_GEJ_WORDS, ws.ring_out + p * RP_RING_OUT_BYTES, ws.ring_ok + p * RP_MAX_RINGS,
                                                 proofs + proof_off[p], g * S2K_RP_K, live, gtab, gc.tab[sl], gc.xmul[sl], M, ev ? ev + p * (RP_MAX_RINGS * 32) : nullptr, dbg);
    if (served != RP_SHARED_SERVED && live) {       // (wavefront-uniform verdict) hand this lane's rings to the general form
        const u32 r0 = g * S2K_RP_K, cnt = rec.rings - r0 < S2K_RP_K ? rec.rings - r0 : S2K_RP_K;
        const u32 base = atomicAdd(&ws.plan[1], cnt);
        for (u32 i = 0; i < cnt; i++) ws.mapG[base + i] = (u32)p | ((r0 + i) << 20);
        atomicAdd(&ws.plan[served == RP_SHARED_SUSPECT ? 2 : 3], cnt);      // diagnostics: s2k_engine_rp_handback
    }
}
__global__ void __launch_bounds__(256, S2K_RINGS_WAVES)
k_rp_rings(rp_ws ws, const unsigned char* __restrict__ proofs, const uint64_t* __restrict__ proof_off, const u32* __restrict__ gtab, u32* __restrict__ ptab, u32* ev, int split) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 nG = ws.plan[1];
    if ((t & ~size_t(63)) >= nG) return;
    const int live = t < nG;
    const u32 item = ws.mapG[live ? t : 0];
    const size_t p = item & 0xFFFFFu; const u32 ring = item >> 20;
    const rp_rec& rec = ws.rec[p];
    __shared__ u32 s_dig[S2K_DIG_WORDS * 256];
    const lane_mem lm{ptab + t * S2K_PTAB_WORDS, S2K_LANE_DIG(s_dig)};
    rp_ring(rec, ws.bases + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS, ws.pub0 + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS,
            ws.ring_out + p * RP_RING_OUT_BYTES + ring * 33, ws.ring_ok + p * RP_MAX_RINGS + ring, proofs + proof_off[p], ring, live, gtab, lm, ev ? ev + (p * RP_MAX_RINGS + ring) * 32 : nullptr,
            split ? ws.dbases + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS : (const u32*)nullptr, split ? ws.tcur + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS : (u32*)nullptr);
}
__global__ void __launch_bounds__(64)
k_rp_final(rp_ws ws, int32_t* results, const unsigned char* proofs, const uint64_t* proof_off, const unsigned char* gens64, rp_gen_mbox* mbox, size_t n) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int ok = 0; u32 gslot = RP_GSLOT_NONE;
    if (p < n) {
        ok = rp_final(ws.rec[p], ws.ring_out + p * RP_RING_OUT_BYTES, ws.ring_ok + p * RP_MAX_RINGS, proofs + proof_off[p]);
        results[p] = ok;
        gslot = ws.rec[p].gslot;
    }
    // the generator-table cache's bookkeeping, from VERIFIED proofs only: which cached tables were of use (least-recently-used stamps),
    // which uncached generators keep coming (candidates for a table)
    if (mbox) {
        for (u32 sl = 0; sl < RP_GEN_SLOTS; sl++) {
            const unsigned long long m = __ballot(ok && gslot == sl);
            if (m && threadIdx.x == 0) atomicAdd(&mbox->hits[sl], (u32)__popcll(m));
        }
        rp_gen_report_miss(mbox, gens64 + 64 * (p < n ? p : 0), ok && gslot == RP_GSLOT_NONE);
    }
}

// rewinding (rangeproof_rewind.h): one lane per proof that verified
struct rp_rewind_args {
    u32* ev; u32* prep; u32* secs;                  // scratch, one chunk: [m][128][8], [m][128][8], [m][32][8] words
    unsigned char* blind_out; uint64_t* value_out; unsigned char* msg_out; uint64_t* outlen; size_t msg_stride; const unsigned char* nonces;
};
// the DRBG replay of every structurally valid proof (rp_rewind_draws): needs nothing from the ring verification, so rp_launch runs
// it on a side stream underneath the rings kernel
__global__ void __launch_bounds__(64)
k_rp_rewind_draws(rp_ws ws, rp_rewind_args ra, const unsigned char* proofs, const uint64_t* proof_off, const unsigned char* gens64, size_t n) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || !ws.rec[p].ok) return;
    rp_rewind_draws(ws.rec[p], proofs + proof_off[p], ra.nonces + 32 * p, gens64 + 64 * p, ra.prep + p * 1024, ra.secs + p * 256);
}
__global__ void __launch_bounds__(256, 2)
k_rp_rewind(rp_ws ws, rp_rewind_args ra, int32_t* results, const uint64_t* min_value, const unsigned char* proofs, const uint64_t* proof_off,
            const unsigned char* gens64, const u32* gtab, u32* ptab, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int inrange = t < n;
    const size_t p = inrange ? t : 0;
    int ok = inrange && results[p];
    scalar blind; u64 value = 0, mlen = 0; sc_set_zero(blind);
    const unsigned char* proof = proofs + proof_off[p];
    if (ok) {
        mlen = (ra.msg_out && ra.outlen) ? ra.outlen[p] : 0;
        if (mlen > ra.msg_stride) mlen = ra.msg_stride;
        ok = rp_rewind_recover(blind, value, ra.msg_out ? ra.msg_out + p * ra.msg_stride : nullptr, &mlen, ws.rec[p], proof,
                               ra.ev + p * 1024, ra.prep + p * 1024, ra.secs + p * 256);
    }
    // the commitment must be blind*G + (value*scale + min_value)*gen  (rangeproof_impl.h:662-672)
    u64 vv = 0;
    if (ok) {
        u32 off; int exp, mant; u64 scale, mn, mx;
        rp_getheader(off, exp, mant, scale, &mn, &mx, proof, proof_off[p + 1] - proof_off[p]);
        vv = value * scale + min_value[p];
    }
    gej A; scalar sv;
    { ge g; fe_set_b32_mod(g.x, gens64 + 64 * p); fe_set_b32_mod(g.y, gens64 + 64 * p + 32); fe_norm_weak(g.x); fe_norm_weak(g.y); gej_set_ge(A, g); }
    sc_set_u64(sv, vv);
    if (!ok) { sc_set_zero(sv); sc_set_zero(blind); }
    __shared__ u32 s_dig[S2K_DIG_WORDS * 256];
    const lane_mem lm{ptab + t * S2K_PTAB_WORDS, S2K_LANE_DIG(s_dig)};
    gej R; ecmult_lane(R, A, sv, blind, 1, gtab, lm);
    ge a; ge_set_gej(a, R);
    if (ok) {
        fe cx, cy, d;
        for (int i = 0; i < 9; i++) { cx.n[i] = ws.rec[p].commit[i]; cy.n[i] = ws.rec[p].commit[9 + i]; }
        ok &= !R.inf;
        fe_neg(d, a.x, 1); fe_add(d, cx); ok &= fe_normalizes_to_zero(d);
        fe_neg(d, a.y, 1); fe_add(d, cy); ok &= fe_normalizes_to_zero(d);
    }
    if (inrange) {
        results[p] = ok;
        if (!ok) { sc_set_zero(blind); vv = 0; mlen = 0; }
        sc_get_b32(ra.blind_out + 32 * p, blind);
        ra.value_out[p] = vv;
        if (ra.outlen) ra.outlen[p] = mlen;
    }
}

static size_t rp_ws_bytes(size_t n) {
    return ws_need({64, (n * RP_MAX_RINGS / S2K_RP_K + 64) * 4, (n * RP_MAX_RINGS + 64) * 4, n * sizeof(rp_rec), n * RP_MAX_RINGS * RP_GEJ_WORDS * 4, n * RP_MAX_RINGS * RP_GEJ_WORDS * 4, n * RP_MAX_RINGS * RP_GEJ_WORDS * 4,
                    n * RP_MAX_RINGS * RP_GEJ_WORDS * 4, n * RP_MAX_RINGS, n * RP_RING_OUT_BYTES, n * RP_MAX_RINGS});
}
static void rp_ws_carve(rp_ws& w, ws_carver& c, size_t n) {
    w.plan = c.take<u32>(16); w.mapF = c.take<u32>(n * RP_MAX_RINGS / S2K_RP_K + 64); w.mapG = c.take<u32>(n * RP_MAX_RINGS + 64);
    w.rec = c.take<rp_rec>(n);
    w.bases = c.take<u32>(n * RP_MAX_RINGS * RP_GEJ_WORDS);
    w.pub0 = c.take<u32>(n * RP_MAX_RINGS * RP_GEJ_WORDS);
    w.dbases = c.take<u32>(n * RP_MAX_RINGS * RP_GEJ_WORDS);
    w.tcur = c.take<u32>(n * RP_MAX_RINGS * RP_GEJ_WORDS);
    w.lift_ok = c.take<unsigned char>(n * RP_MAX_RINGS);
    w.ring_out = c.take<unsigned char>(n * RP_RING_OUT_BYTES);
    w.ring_ok = c.take<unsigned char>(n * RP_MAX_RINGS);
}
// the two scratch sets of the pipeline, each for `nw` proofs
static int engine_rp_slots(s2k_engine* e, size_t nw) {
    const size_t bytes = (rp_ws_bytes(nw) + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
    if (bytes <= e->rp_mem_bytes) return 1;
    HIPCHK(hipDeviceSynchronize());                 // earlier launches may still use the old records
    for (int i = 0; i < 2; i++) { if (e->rp_mem[i]) HIPCHK(hipFree(e->rp_mem[i])); e->rp_mem[i] = nullptr; e->rp_done_valid[i] = 0; e->rp_last_plan[i] = nullptr; }
    e->rp_mem_bytes = 0;
    for (int i = 0; i < 2; i++) HIPCHK(hipMalloc((void**)&e->rp_mem[i], bytes));
    e->rp_mem_bytes = bytes;
    return 1;
}
// Launches the five stages, chunk by chunk (RP_CHUNK proofs), as a two-deep pipeline:
//   side streams   : header -> { prologue (1 lane/proof, latency bound) || lift (1 lane/ring, lowest priority) } -> key sum
//   caller's stream: rings (the 98 %) -> final [-> rewind]
// Chunk i+1's side-stream stage runs while chunk i's rings kernel owns the machine; the scratch records alternate between two sets and a
// set is reused only after the rings/final that read it.  By default the side-stream stage of a call also waits for everything the
// caller had queued on `st` before the call (its inputs may still be in the making); with S2K_OPT_RP_INPUTS_READY the caller
// promises the input arrays are complete when the call is made, and the first stage of call k+1 then also runs under call k's rings.
#define RP_CHUNK (e->max_lanes / RP_MAX_RINGS)     /* proofs per launch group */
static int rp_launch(s2k_engine* e, hipStream_t st, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                     const unsigned char* commits33, const unsigned char* proofs, const uint64_t* proof_off, const unsigned char* extra,
                     const uint64_t* extra_off, const unsigned char* gens64, size_t n, const rp_rewind_args* rewind = nullptr, int inputs_on_stream = 0,
                     hipEvent_t inputs_ev = nullptr) {
    const size_t nw = std::min(n, RP_CHUNK);
    if (!engine_rp_slots(e, nw)) return 0;
    if (!engine_rtab(e, nw * RP_MAX_RINGS)) return 0;
    // the device's tables: held from the moment this call takes its view of the generator-table cache until its last kernel is queued
    std::lock_guard<std::recursive_mutex> pool_lock(e->pool->mu);
    ENGINE_GTAB(e, st);
    const hipStream_t sp = e->stream_pre;
    gen_cache_service(e, st);
    const rp_gen_dev gc = gen_dev_view(e, st, sp);           // (streams that will read a table still being built wait for its event)
    HIPCHK(hipMemsetAsync(results, 0, sizeof(int32_t) * n, st));          // a batch that does not complete never shows an item as valid
    HIPCHK(hipEventRecord(e->ev[0], st));
    // (inputs_ev: the inputs arrive on another stream, which recorded this event behind them -- the host-buffer entry points' copy stream)
    if (inputs_ev) HIPCHK(hipStreamWaitEvent(sp, inputs_ev, 0));
    if (inputs_on_stream || (!inputs_ev && !e->rp_inputs_ready)) { HIPCHK(hipEventRecord(e->ev_rp_in, st)); HIPCHK(hipStreamWaitEvent(sp, e->ev_rp_in, 0)); }
    for (size_t p0 = 0; p0 < n; p0 += RP_CHUNK) {
        const size_t m = std::min(n - p0, RP_CHUNK);
        const unsigned b64 = (unsigned)((m + 63) / 64), b256 = (unsigned)((m * 32 + 255) / 256);
        const int slot = (int)(e->rp_seq++ & 1u);
        ws_carver c{e->rp_mem[slot], 0}; rp_ws w; rp_ws_carve(w, c, nw);
        if (p0 == 0) e->rp_last_plan[slot ^ 1] = nullptr;
        e->rp_last_plan[slot] = w.plan;
        // ---- side streams
        if (e->rp_done_valid[slot]) HIPCHK(hipStreamWaitEvent(sp, e->ev_rp_done[slot], 0));
        hipLaunchKernelGGL(k_rp_header, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, sp, w, min_value + p0, max_value + p0, proofs, proof_off + p0, gens64 + 64 * p0,
                           gc, m);
        HIPCHK(hipEventRecord(e->ev_rp_fork[slot], sp));
        HIPCHK(hipStreamWaitEvent(e->stream2, e->ev_rp_fork[slot], 0));
        hipLaunchKernelGGL(k_rp_lift, dim3(b256), dim3(256), 0, e->stream2, w, proofs, proof_off + p0, m);
        HIPCHK(hipEventRecord(e->ev_rp_join[slot], e->stream2));
        hipLaunchKernelGGL(k_rp_prologue, dim3(b64), dim3(192), 0, sp, w, min_value + p0, commits33 + 33 * p0, proofs, proof_off + p0, extra,
                           extra_off ? extra_off + p0 : nullptr, gens64 + 64 * p0, m);
        HIPCHK(hipStreamWaitEvent(sp, e->ev_rp_join[slot], 0));
        hipLaunchKernelGGL(k_rp_sum, dim3(b64), dim3(64), 0, sp, w, gc.valid, m);
        HIPCHK(hipEventRecord(e->ev_rp_pre[slot], sp));
        if (rewind) {
            // rewinding: the replay of the prover's random stream (serial per proof, ~1 500 SHA-256 compressions) only needs the header
            // and the commitment, so it goes behind the first stage on the side stream and runs underneath the rings kernel; its
            // scratch (prep / secs) is one set per call, so it waits for the recovery pass of the chunk before
            rp_rewind_args ra = *rewind; ra.nonces += 32 * p0;
            if (e->rp_rewound_valid) HIPCHK(hipStreamWaitEvent(sp, e->ev_rp_rewound, 0));
            hipLaunchKernelGGL(k_rp_rewind_draws, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, sp, w, ra, proofs, proof_off + p0, gens64 + 64 * p0, m);
            HIPCHK(hipEventRecord(e->ev_rp_draws, sp));
        }
        // ---- caller's stream
        HIPCHK(hipStreamWaitEvent(st, e->ev_rp_pre[slot], 0));
        const unsigned rq = e->ring_seq & 31u;
        if (p0 == 0) { HIPCHK(hipEventRecord(e->ev[2], st)); HIPCHK(hipEventRecord(e->ev_ring[rq][0], st)); }
        if (gc.valid) hipLaunchKernelGGL(k_rp_rings_shared, dim3((unsigned)((m * (RP_MAX_RINGS / S2K_RP_K) + 255) / 256)), dim3(256), 0, st, w, proofs, proof_off + p0, e->gtab, e->ptab,
                                         rewind ? rewind->ev : (u32*)nullptr, gc, (u32)e->rp_debug);
        hipLaunchKernelGGL(k_rp_rings, dim3(b256), dim3(256), 0, st, w, proofs, proof_off + p0, e->gtab, e->ptab, rewind ? rewind->ev : (u32*)nullptr, e->rp_split);
        if (p0 == 0) { HIPCHK(hipEventRecord(e->ev[3], st)); HIPCHK(hipEventRecord(e->ev_ring[rq][1], st)); e->ring_seq++; }
        hipLaunchKernelGGL(k_rp_final, dim3(b64), dim3(64), 0, st, w, results + p0, proofs, proof_off + p0, gens64 + 64 * p0,
                           e->pool->gen_slots > 0 ? e->gen_mbox : (rp_gen_mbox*)nullptr, m);
        if (rewind) {
            rp_rewind_args ra = *rewind;                      // scratch is per chunk, the caller's arrays are per batch
            ra.blind_out += 32 * p0; ra.value_out += p0; ra.nonces += 32 * p0;
            if (ra.msg_out) ra.msg_out += ra.msg_stride * p0;
            if (ra.outlen) ra.outlen += p0;
            HIPCHK(hipStreamWaitEvent(st, e->ev_rp_draws, 0));
            hipLaunchKernelGGL(k_rp_rewind, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, w, ra, results + p0, min_value + p0, proofs, proof_off + p0,
                               gens64 + 64 * p0, e->gtab, e->ptab, m);
            HIPCHK(hipEventRecord(e->ev_rp_rewound, st)); e->rp_rewound_valid = 1;
        }
        HIPCHK(hipEventRecord(e->ev_rp_done[slot], st));
        e->rp_done_valid[slot] = 1;
    }
    HIPCHK(hipGetLastError());
    gen_cache_collect(e, st);
    HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
}
// Builds the tables of one generator now (host bytes: the 64-byte secp256k1_generator object, include/secp256k1_generator.h:22-24).
// Returns 1 when the generator has a table afterwards.
extern "C" int s2k_engine_cache_generator(s2k_engine* e, const unsigned char* gen64) {
    if (!e || !gen64) return s2k_fail_arg("s2k_engine_cache_generator", "illegal argument");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    stream_guard sg(e, e->stream);
    std::lock_guard<std::recursive_mutex> pool_lock(e->pool->mu);
    if (e->pool->gen_h == 1 && !memcmp(gen64, k_generator_h, 64)) e->pool->gen_h = 2;
    if (gen_cache_build(e, e->stream, gen64, 1) < 0) return s2k_fail("s2k_engine_cache_generator", "no slot or no memory for a generator table (S2K_GEN_CACHE)");
    return 1;
}
// 1 when `gen64` currently has a table
extern "C" int s2k_engine_generator_cached(s2k_engine* e, const unsigned char* gen64) {
    if (!e || !gen64) return 0;
    std::lock_guard<std::recursive_mutex> lock(e->pool->mu);
    for (int i = 0; i < e->pool->gen_slots; i++) if (e->pool->gen[i].valid && !memcmp(e->pool->gen[i].key, gen64, 64)) return 1;
    return 0;
}
extern "C" int secp256k1_rangeproof_verify_batch_dev(s2k_engine* e, void* stream, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                                                     const unsigned char* commits33, const unsigned char* proofs, const uint64_t* proof_off,
                                                     const unsigned char* extra, const uint64_t* extra_off, const unsigned char* gens64, size_t n) {
    if (!e) return s2k_fail("secp256k1_rangeproof_verify_batch_dev", "null engine");
    if (n == 0) return 1;
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    const size_t nw = std::min(n, RP_CHUNK);
    (void)nw;
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    return rp_launch(e, st, results, min_value, max_value, commits33, proofs, proof_off, extra, extra_off, gens64, n);
}
// ---- host-buffer form: the drop-in path ---------------------------------------------------------------------------------------------
// What an application hands over lives in pageable host memory: either packed arrays (secp256k1_rangeproof_verify_batch) or, the way the
// reference's own callers hold things, arrays of pointers to the objects (secp256k1_rangeproof_verify_batch_ptrs).  Both are gathered ONCE,
// straight into pinned staging memory laid out like the device buffers, by a few host threads; the proof bytes (98 % of the volume) go in
// RP_STAGE_CHUNKS pieces and every finished piece is queued for H2D at once, so the DMA runs underneath the packing of the next pieces.
// Then one launch of the stage pipeline over the whole batch, results back through pinned memory.
struct rp_host_src {
    // packed form
    const unsigned char* commits33; const unsigned char* proofs; const uint64_t* proof_off; const unsigned char* extra; const uint64_t* extra_off; const unsigned char* gens64;
    // pointer form (used when commit_objs != nullptr)
    const void* const* commit_objs; const unsigned char* const* proof_ptrs; const size_t* plens; const unsigned char* const* extra_ptrs; const size_t* elens; const void* const* gen_objs;
};
static int engine_stage(s2k_engine* e, s2k_engine::stage_set& S, size_t in_bytes, size_t out_bytes) {
    if (in_bytes > S.in_bytes || in_bytes + out_bytes + 512 > S.dev_bytes) {
        HIPCHK(hipDeviceSynchronize());
        if (S.in) HIPCHK(hipHostFree(S.in));
        if (S.dev) HIPCHK(hipFree(S.dev));
        S.in = nullptr; S.in_bytes = 0; S.dev = nullptr; S.dev_bytes = 0;
        in_bytes = (in_bytes + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
        HIPCHK(hipHostMalloc((void**)&S.in, in_bytes, hipHostMallocDefault));
        S.in_bytes = in_bytes;
        const size_t db = in_bytes + ((out_bytes + 65535) & ~size_t(65535)) + 65536;
        HIPCHK(hipMalloc((void**)&S.dev, db));
        S.dev_bytes = db;
    }
    if (out_bytes > S.out_bytes) {
        HIPCHK(hipDeviceSynchronize());
        if (S.out) HIPCHK(hipHostFree(S.out));
        S.out = nullptr; S.out_bytes = 0;
        out_bytes = (out_bytes + 65535) & ~size_t(65535);
        HIPCHK(hipHostMalloc((void**)&S.out, out_bytes, hipHostMallocDefault));
        S.out_bytes = out_bytes;
    }
    return 1;
}
#define RP_STAGE_CHUNKS 16
// Gather, copy and launch one batch; what comes back is the ticket of the staging set that now belongs to it (rp_host_wait hands it back).
// The copies go on the engine's copy stream and the first stage of the pipeline waits for them by event, so that a batch submitted while
// the one before it computes has its inputs in HBM -- and its header / prologue stage done -- by the time the rings kernel is free.
// `queue`: the caller's lock on the engine when it is a SYNCHRONOUS entry point -- such a call waits its turn for a staging set (several
// verifier threads on one engine take turns, two of them overlapping) where the asynchronous `_submit` reports "two in flight".
struct thread_joiner { std::vector<std::thread>& w; ~thread_joiner() { for (auto& t : w) if (t.joinable()) t.join(); } };
static int rp_host_submit_impl(s2k_engine* e, const char* who, uint64_t* ticket, int32_t* results, uint64_t* min_value, uint64_t* max_value, const rp_host_src& src, size_t n,
                               std::unique_lock<std::recursive_mutex>* queue);
// (these functions sit right behind extern "C" entry points: nothing may leave them as a C++ exception -- a failed allocation or thread
// start is an engine failure like any other, and the packing threads are joined on every path)
static int rp_host_submit(s2k_engine* e, const char* who, uint64_t* ticket, int32_t* results, uint64_t* min_value, uint64_t* max_value, const rp_host_src& src, size_t n,
                          std::unique_lock<std::recursive_mutex>* queue = nullptr) {
    try { return rp_host_submit_impl(e, who, ticket, results, min_value, max_value, src, n, queue); }
    catch (const std::exception& ex) { (void)hipStreamSynchronize(e->stream_copy); return s2k_fail(who, ex.what()); }
    catch (...) { (void)hipStreamSynchronize(e->stream_copy); return s2k_fail(who, "unexpected exception"); }
}
static int rp_host_submit_impl(s2k_engine* e, const char* who, uint64_t* ticket, int32_t* results, uint64_t* min_value, uint64_t* max_value, const rp_host_src& src, size_t n,
                               std::unique_lock<std::recursive_mutex>* queue) {
    const int ptrs = src.commit_objs != nullptr;
    const int has_extra = ptrs ? (src.extra_ptrs != nullptr) : (src.extra != nullptr && src.extra_off != nullptr);
    int si = -1;
    const auto t_queue = std::chrono::steady_clock::now();
    for (;;) {
        const int pref = (int)(e->next_ticket & 1u);
        si = !e->stage[pref].ticket ? pref : (!e->stage[pref ^ 1].ticket ? (pref ^ 1) : -1);
        if (si >= 0) break;
        if (!queue) return s2k_fail_busy(who, "two batches in flight already: wait for a ticket first");
        // a synchronous caller queues behind other synchronous callers (they hand their sets back by themselves); when both sets belong to
        // asynchronous tickets only the application can free one, so the call reports "busy" at once instead of stalling
        if (!e->stage[0].sync_owned && !e->stage[1].sync_owned) return s2k_fail_busy(who, "both staging sets are held by asynchronous tickets: wait for one first");
        // (bounded: an owner stuck in a hung device wait must not block every other synchronous caller for ever -- after two minutes the
        //  call reports "busy" and the hook's caller takes the CPU path)
        if (std::chrono::steady_clock::now() - t_queue > std::chrono::seconds(120)) return s2k_fail_busy(who, "waited 120 s for a staging set held by another synchronous call");
        e->stage_cv.wait_for(*queue, std::chrono::seconds(1));
    }
    s2k_engine::stage_set& S = e->stage[si];
    // sizes and offsets
    std::vector<uint64_t> poff_v, eoff_v;
    const uint64_t* poff = src.proof_off; const uint64_t* eoff = src.extra_off;
    if (ptrs) {
        poff_v.resize(n + 1); poff_v[0] = 0;
        for (size_t i = 0; i < n; i++) poff_v[i + 1] = poff_v[i] + src.plens[i];
        poff = poff_v.data();
        if (has_extra) { eoff_v.resize(n + 1); eoff_v[0] = 0; for (size_t i = 0; i < n; i++) eoff_v[i + 1] = eoff_v[i] + (src.extra_ptrs[i] ? src.elens[i] : 0); eoff = eoff_v.data(); }
    }
    const size_t pbytes = (size_t)poff[n], ebytes = has_extra ? (size_t)eoff[n] : 0;
    const bool tlog = e->stage_log != 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_begin = now();
    // one layout for the pinned staging area and for its device image
    size_t off = 0;
    auto take = [&](size_t bytes) { off = (off + 255) & ~size_t(255); const size_t o = off; off += bytes; return o; };
    const size_t o_com = take(33 * n), o_gen = take(64 * n), o_off = take(8 * (n + 1)), o_eoff = take(8 * (n + 1)), o_ex = take(ebytes + 64), o_pr = take(pbytes + 64);
    const size_t in_bytes = off + 256;
    const size_t o_res = 0, o_min = (4 * n + 255) & ~size_t(255), o_max = o_min + ((8 * n + 255) & ~size_t(255)), out_bytes = o_max + 8 * n + 256;
    if (!engine_stage(e, S, in_bytes, out_bytes)) return 0;
    unsigned char* const hs = S.in; unsigned char* const ds = S.dev; unsigned char* const dout = S.dev + ((in_bytes + 255) & ~size_t(255));
    hipStream_t st = e->stream, cp = e->stream_copy;
    stream_guard sg(e, st);
    // the set's device image is free once the batch that used it before has its results out (its pinned side: once that batch was waited for)
    if (S.used) HIPCHK(hipStreamWaitEvent(cp, S.ev_out, 0));
    // small arrays: packed by this thread, queued first
    if (ptrs) {
        for (size_t i = 0; i < n; i++) { memcpy(hs + o_com + 33 * i, src.commit_objs[i], 33); memcpy(hs + o_gen + 64 * i, src.gen_objs[i], 64); }
    } else {
        memcpy(hs + o_com, src.commits33, 33 * n); memcpy(hs + o_gen, src.gens64, 64 * n);
    }
    memcpy(hs + o_off, poff, 8 * (n + 1));
    if (has_extra) {
        memcpy(hs + o_eoff, eoff, 8 * (n + 1));
        if (ptrs) { for (size_t i = 0; i < n; i++) if (eoff[i + 1] > eoff[i]) memcpy(hs + o_ex + eoff[i], src.extra_ptrs[i], (size_t)(eoff[i + 1] - eoff[i])); }
        else if (ebytes) memcpy(hs + o_ex, src.extra, ebytes);
    }
    const auto t_small = now();
    HIPCHK(hipMemcpyAsync(ds + o_com, hs + o_com, o_pr - o_com, hipMemcpyHostToDevice, cp));          // everything in front of the proofs in one piece
    // proofs: RP_STAGE_CHUNKS pieces of whole proofs, packed by `nt` threads (piece c by thread c % nt), queued as they complete
    const int nchunk = (int)std::min<size_t>(RP_STAGE_CHUNKS, std::max<size_t>(1, n / 64));
    const int nt = (pbytes >= (size_t(4) << 20)) ? std::min(e->stage_threads, nchunk) : 1;
    std::vector<size_t> cb(nchunk + 1);
    for (int c = 0; c <= nchunk; c++) cb[c] = (size_t)((unsigned long long)n * (unsigned)c / (unsigned)nchunk);
    std::vector<std::atomic<int>> ready(nchunk);
    for (auto& r : ready) r.store(0);
    auto pack_piece = [&](int c) {
        if (ptrs) { for (size_t i = cb[c]; i < cb[c + 1]; i++) if (src.plens[i]) memcpy(hs + o_pr + poff[i], src.proof_ptrs[i], src.plens[i]); }
        else if (poff[cb[c + 1]] > poff[cb[c]]) memcpy(hs + o_pr + poff[cb[c]], src.proofs + poff[cb[c]], (size_t)(poff[cb[c + 1]] - poff[cb[c]]));
        ready[c].store(1, std::memory_order_release);
    };
    auto pack = [&](int t) { for (int c = t; c < nchunk; c += nt) pack_piece(c); };
    int ok = 1, next = 0;
    auto queue_ready = [&](bool all) {
        while (ok && next < nchunk && (all || ready[next].load(std::memory_order_acquire))) {
            const size_t b0 = (size_t)poff[cb[next]], b1 = (size_t)poff[cb[next + 1]];
            if (b1 > b0 && hipMemcpyAsync(ds + o_pr + b0, hs + o_pr + b0, b1 - b0, hipMemcpyHostToDevice, cp) != hipSuccess) ok = 0;
            next++;
        }
    };
    if (nt == 1) {
        pack(0);
        if (pbytes && hipMemcpyAsync(ds + o_pr, hs + o_pr, pbytes, hipMemcpyHostToDevice, cp) != hipSuccess) ok = 0;
    } else {
        std::vector<std::thread> workers;
        {
            thread_joiner joiner{workers};
            int started = 1;
            try { workers.reserve(nt); for (int t = 1; t < nt; t++) { workers.emplace_back(pack, t); started++; } }
            catch (...) { }                                                              // fewer workers than planned: this thread packs the rest
            for (int c = 0; c < nchunk; c += nt) { pack_piece(c); queue_ready(false); }      // this thread packs its share and queues whatever is ready, in order
            for (int t = started; t < nt; t++) pack(t);
        }
        queue_ready(true);
    }
    if (!ok) { (void)hipGetLastError(); (void)hipStreamSynchronize(cp); return s2k_fail(who, "host to device copy failed"); }
    HIPCHK(hipEventRecord(S.ev_h2d, cp));
    const auto t_packed = now();
    if (tlog) (void)hipStreamSynchronize(cp);
    const auto t_h2d = now();
    int32_t* d_res = (int32_t*)(dout + o_res); uint64_t* d_min = (uint64_t*)(dout + o_min); uint64_t* d_max = (uint64_t*)(dout + o_max);
    if (!rp_launch(e, st, d_res, d_min, d_max, ds + o_com, ds + o_pr, (const uint64_t*)(ds + o_off), has_extra ? ds + o_ex : nullptr, has_extra ? (const uint64_t*)(ds + o_eoff) : nullptr,
                   ds + o_gen, n, nullptr, 0, S.ev_h2d)) { (void)hipStreamSynchronize(cp); return 0; }
    HIPCHK(hipMemcpyAsync(S.out, dout, out_bytes - 256, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(S.ev_out, st));
    S.used = 1; S.ticket = e->next_ticket++; S.sync_owned = queue != nullptr; S.results = results; S.min_value = min_value; S.max_value = max_value; S.n = n; S.o_res = o_res; S.o_min = o_min; S.o_max = o_max;
    *ticket = S.ticket;
    if (tlog) fprintf(stderr, "[s2k stage] n=%zu threads=%d: small arrays %.2f ms, proofs packed+queued %.2f ms, H2D drained +%.2f ms, launch %.2f ms\n",
                      n, nt, ms(t_begin, t_small), ms(t_small, t_packed), ms(t_packed, t_h2d), ms(t_h2d, now()));
    return 1;
}
// blocks until the batch behind `ticket` is done, hands its results to the arrays given at submission and frees its staging set.
// Called WITHOUT the engine's mutex held across the wait (another thread may be submitting meanwhile).
static int rp_host_wait(s2k_engine* e, const char* who, uint64_t ticket) {
    s2k_engine::stage_set* S = nullptr; hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::recursive_mutex> lock(e->mu);
        for (int i = 0; i < 2; i++) if (ticket != 0 && e->stage[i].ticket == ticket) S = &e->stage[i];
        if (!S) return s2k_fail_arg(who, "unknown ticket (never issued, or waited for already)");
        ev = S->ev_out;
    }
    HIPCHK(hipSetDevice(e->device));
    const hipError_t err = hipEventSynchronize(ev);
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    if (S->ticket != ticket) return s2k_fail_arg(who, "unknown ticket (waited for by another thread meanwhile)");
    if (err != hipSuccess) { S->ticket = 0; S->sync_owned = 0; e->stage_cv.notify_all(); (void)hipGetLastError(); return s2k_fail(who, hipGetErrorString(err)); }      // (the arrays keep the zeros of submission time)
    memcpy(S->results, S->out + S->o_res, 4 * S->n); memcpy(S->min_value, S->out + S->o_min, 8 * S->n); memcpy(S->max_value, S->out + S->o_max, 8 * S->n);
    S->ticket = 0; S->sync_owned = 0;
    e->stage_cv.notify_all();
    return 1;
}
static int rp_ptrs_check(const char* who, int32_t* results, uint64_t* min_value, uint64_t* max_value, const void* const* commit_objs, const unsigned char* const* proofs,
                         const size_t* plens, const unsigned char* const* extra, const size_t* elens, const void* const* gen_objs, size_t n) {
    if (!results || !min_value || !max_value || !commit_objs || !proofs || !plens || !gen_objs || (extra && !elens)) return s2k_fail_arg(who, "illegal argument (ARG_CHECK)");
    for (size_t i = 0; i < n; i++) if (!commit_objs[i] || !gen_objs[i] || (!proofs[i] && plens[i]) || (extra && !extra[i] && elens[i]))
        return s2k_fail_arg(who, "illegal argument (ARG_CHECK): null item");
    return 1;
}
// ---- asynchronous pair: submit gathers + queues and returns; wait blocks for the results (at most two submissions in flight) ---------
extern "C" int secp256k1_rangeproof_verify_batch_submit(s2k_engine* e, uint64_t* ticket, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                                                        const unsigned char* commits33, const unsigned char* proofs, const uint64_t* proof_off,
                                                        const unsigned char* extra, const uint64_t* extra_off, const unsigned char* gens64, size_t n) {
    const char* who = "secp256k1_rangeproof_verify_batch_submit";
    if (!e) return s2k_fail(who, "null engine");
    if (!ticket || n == 0 || !results || !min_value || !max_value || !commits33 || !proofs || !proof_off || !gens64) return s2k_fail_arg(who, "illegal argument (ARG_CHECK)");
    memset(results, 0, sizeof(int32_t) * n);
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    rp_host_src src{}; src.commits33 = commits33; src.proofs = proofs; src.proof_off = proof_off; src.extra = extra; src.extra_off = extra_off; src.gens64 = gens64;
    return rp_host_submit(e, who, ticket, results, min_value, max_value, src, n);
}
extern "C" int secp256k1_rangeproof_verify_batch_ptrs_submit(s2k_engine* e, uint64_t* ticket, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                                                             const void* const* commit_objs, const unsigned char* const* proofs, const size_t* plens,
                                                             const unsigned char* const* extra, const size_t* elens, const void* const* gen_objs, size_t n) {
    const char* who = "secp256k1_rangeproof_verify_batch_ptrs_submit";
    if (!e) return s2k_fail(who, "null engine");
    if (!ticket || n == 0) return s2k_fail_arg(who, "illegal argument (ARG_CHECK)");
    if (!rp_ptrs_check(who, results, min_value, max_value, commit_objs, proofs, plens, extra, elens, gen_objs, n)) return 0;
    memset(results, 0, sizeof(int32_t) * n);
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    rp_host_src src{}; src.commit_objs = commit_objs; src.proof_ptrs = proofs; src.plens = plens; src.extra_ptrs = extra; src.elens = elens; src.gen_objs = gen_objs;
    return rp_host_submit(e, who, ticket, results, min_value, max_value, src, n);
}
extern "C" int secp256k1_rangeproof_verify_batch_wait(s2k_engine* e, uint64_t ticket) {
    if (!e) return s2k_fail("secp256k1_rangeproof_verify_batch_wait", "null engine");
    return rp_host_wait(e, "secp256k1_rangeproof_verify_batch_wait", ticket);
}
// A synchronous call that finds both staging sets free takes BOTH: the batch goes as two halves, and the second half is gathered and copied
// while the first one computes (a lone caller thread otherwise leaves the GPU idle for the ~3 ms of gathering and PCIe time of every
// batch).  With other callers about -- a second verifier thread, tickets in flight -- the sets are not both free and the call goes in one
// piece, as before: those callers overlap among themselves.  $S2K_SYNC_SPLIT=0 turns the halving off.
#define RP_SYNC_SPLIT_MIN 8192
static int rp_host_sync(s2k_engine* e, const char* who, int32_t* results, uint64_t* min_value, uint64_t* max_value, const rp_host_src& src, size_t n) {
    uint64_t ticket[2] = {0, 0};
    int parts = 1;
    {
        std::unique_lock<std::recursive_mutex> lock(e->mu);
        HIPCHK(hipSetDevice(e->device));
        if (e->sync_split && n >= RP_SYNC_SPLIT_MIN && !e->stage[0].ticket && !e->stage[1].ticket) parts = 2;
        const size_t h = parts == 2 ? ((n / 2 + 63) & ~size_t(63)) : n;
        rp_host_src a = src, b = src;
        std::vector<uint64_t> poff_b, eoff_b;
        if (parts == 2) {
            if (src.commit_objs) {
                b.commit_objs += h; b.proof_ptrs += h; b.plens += h; b.gen_objs += h;
                if (src.extra_ptrs) { b.extra_ptrs += h; b.elens += h; }
            } else {
                b.commits33 += 33 * h; b.gens64 += 64 * h;
                poff_b.resize(n - h + 1);
                for (size_t i = 0; i <= n - h; i++) poff_b[i] = src.proof_off[h + i] - src.proof_off[h];
                b.proofs += src.proof_off[h]; b.proof_off = poff_b.data();
                if (src.extra && src.extra_off) {
                    eoff_b.resize(n - h + 1);
                    for (size_t i = 0; i <= n - h; i++) eoff_b[i] = src.extra_off[h + i] - src.extra_off[h];
                    b.extra += src.extra_off[h]; b.extra_off = eoff_b.data();
                }
            }
        }
        if (!rp_host_submit(e, who, &ticket[0], results, min_value, max_value, a, h, &lock)) return 0;
        if (parts == 2 && !rp_host_submit(e, who, &ticket[1], results + h, min_value + h, max_value + h, b, n - h, &lock)) {
            lock.unlock();
            (void)rp_host_wait(e, who, ticket[0]);
            memset(results, 0, sizeof(int32_t) * n);                      // an engine failure never leaves part of a batch marked valid
            return 0;
        }
    }
    int ok = rp_host_wait(e, who, ticket[0]);
    if (parts == 2) ok &= rp_host_wait(e, who, ticket[1]);
    if (!ok) memset(results, 0, sizeof(int32_t) * n);
    return ok;
}
extern "C" int secp256k1_rangeproof_verify_batch(s2k_engine* e, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                                                 const unsigned char* commits33, const unsigned char* proofs, const uint64_t* proof_off,
                                                 const unsigned char* extra, const uint64_t* extra_off, const unsigned char* gens64, size_t n) {
    if (!e) return s2k_fail("secp256k1_rangeproof_verify_batch", "null engine");
    if (n == 0) return 1;
    memset(results, 0, sizeof(int32_t) * n);
    rp_host_src src{}; src.commits33 = commits33; src.proofs = proofs; src.proof_off = proof_off; src.extra = extra; src.extra_off = extra_off; src.gens64 = gens64;
    return rp_host_sync(e, "secp256k1_rangeproof_verify_batch", results, min_value, max_value, src, n);
}
extern "C" int secp256k1_rangeproof_verify_batch_ptrs(s2k_engine* e, int32_t* results, uint64_t* min_value, uint64_t* max_value, const void* const* commit_objs,
                                                      const unsigned char* const* proofs, const size_t* plens, const unsigned char* const* extra, const size_t* elens,
                                                      const void* const* gen_objs, size_t n) {
    const char* who = "secp256k1_rangeproof_verify_batch_ptrs";
    if (!e) return s2k_fail(who, "null engine");
    if (n == 0) return 1;
    if (!rp_ptrs_check(who, results, min_value, max_value, commit_objs, proofs, plens, extra, elens, gen_objs, n)) return 0;
    memset(results, 0, sizeof(int32_t) * n);
    rp_host_src src{}; src.commit_objs = commit_objs; src.proof_ptrs = proofs; src.plens = plens; src.extra_ptrs = extra; src.elens = elens; src.gen_objs = gen_objs;
    return rp_host_sync(e, who, results, min_value, max_value, src, n);
}
// rewind: verification + recovery (rangeproof_rewind.h); host buffers
extern "C" int secp256k1_rangeproof_rewind_batch(s2k_engine* e, int32_t* results, unsigned char* blind_out, uint64_t* value_out, unsigned char* message_out,
                                                 uint64_t* outlen, size_t msg_stride, const unsigned char* nonces, uint64_t* min_value, uint64_t* max_value,
                                                 const unsigned char* commits33, const unsigned char* proofs, const uint64_t* proof_off,
                                                 const unsigned char* extra, const uint64_t* extra_off, const unsigned char* gens64, size_t n) {
    if (!e) return s2k_fail("secp256k1_rangeproof_rewind_batch", "null engine");
    if (n == 0) return 1;
    if (!results || !blind_out || !value_out || !nonces || !min_value || !max_value || !commits33 || !proofs || !proof_off || !gens64 || (message_out && !outlen))
        return s2k_fail_arg("secp256k1_rangeproof_rewind_batch", "illegal argument (ARG_CHECK)");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    const size_t pbytes = (size_t)proof_off[n], ebytes = (extra && extra_off) ? (size_t)extra_off[n] : 0;
    const size_t nw = std::min(n, RP_CHUNK), mbytes = message_out ? msg_stride * n : 0;
    const size_t io = ws_need({4 * n, 8 * n, 8 * n, 33 * n, pbytes + 64, 8 * (n + 1), ebytes + 64, 8 * (n + 1), 64 * n, 32 * n, 32 * n, 8 * n, 8 * n, mbytes + 64,
                               nw * 4096, nw * 4096, nw * 1024});
    if (!engine_workspace(e, io)) return 0;
    ws_carver c{e->ws, 0}; (void)nw;
    int32_t* d_res = c.take<int32_t>(n); uint64_t* d_min = c.take<uint64_t>(n); uint64_t* d_max = c.take<uint64_t>(n);
    unsigned char* d_com = c.take<unsigned char>(33 * n); unsigned char* d_pr = c.take<unsigned char>(pbytes + 64);
    uint64_t* d_off = c.take<uint64_t>(n + 1); unsigned char* d_ex = c.take<unsigned char>(ebytes + 64); uint64_t* d_eoff = c.take<uint64_t>(n + 1);
    unsigned char* d_gen = c.take<unsigned char>(64 * n);
    rp_rewind_args ra;
    ra.nonces = c.take<unsigned char>(32 * n); ra.blind_out = c.take<unsigned char>(32 * n); ra.value_out = c.take<uint64_t>(n);
    ra.outlen = c.take<uint64_t>(n); ra.msg_out = message_out ? c.take<unsigned char>(mbytes + 64) : nullptr; ra.msg_stride = msg_stride;
    ra.ev = c.take<u32>(nw * 1024); ra.prep = c.take<u32>(nw * 1024); ra.secs = c.take<u32>(nw * 256);
    hipStream_t st = e->stream;
    stream_guard sg(e, st);
    HIPCHK(hipMemcpyAsync(d_com, commits33, 33 * n, hipMemcpyHostToDevice, st));
    if (pbytes) HIPCHK(hipMemcpyAsync(d_pr, proofs, pbytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_off, proof_off, 8 * (n + 1), hipMemcpyHostToDevice, st));
    if (ebytes) { HIPCHK(hipMemcpyAsync(d_ex, extra, ebytes, hipMemcpyHostToDevice, st)); }
    if (extra && extra_off) HIPCHK(hipMemcpyAsync(d_eoff, extra_off, 8 * (n + 1), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_gen, gens64, 64 * n, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync((void*)ra.nonces, nonces, 32 * n, hipMemcpyHostToDevice, st));
    if (message_out) { HIPCHK(hipMemcpyAsync(ra.outlen, outlen, 8 * n, hipMemcpyHostToDevice, st)); HIPCHK(hipMemsetAsync(ra.msg_out, 0, mbytes, st)); }
    else HIPCHK(hipMemsetAsync(ra.outlen, 0, 8 * n, st));
    if (!rp_launch(e, st, d_res, d_min, d_max, d_com, d_pr, d_off, (extra && extra_off) ? d_ex : nullptr, (extra && extra_off) ? d_eoff : nullptr, d_gen, n, &ra, 1)) return 0;
    HIPCHK(hipMemcpyAsync(results, d_res, 4 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(min_value, d_min, 8 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(max_value, d_max, 8 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(blind_out, ra.blind_out, 32 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(value_out, ra.value_out, 8 * n, hipMemcpyDeviceToHost, st));
    if (message_out) { HIPCHK(hipMemcpyAsync(message_out, ra.msg_out, mbytes, hipMemcpyDeviceToHost, st)); HIPCHK(hipMemcpyAsync(outlen, ra.outlen, 8 * n, hipMemcpyDeviceToHost, st)); }
    HIPCHK(hipStreamSynchronize(st));
    return 1;
}
// rewind with every array already in HBM (stream-ordered; scratch comes from the engine workspace)
extern "C" int secp256k1_rangeproof_rewind_batch_dev(s2k_engine* e, void* stream, int32_t* results, unsigned char* blind_out, uint64_t* value_out,
                                                     unsigned char* message_out, uint64_t* outlen, size_t msg_stride, const unsigned char* nonces,
                                                     uint64_t* min_value, uint64_t* max_value, const unsigned char* commits33, const unsigned char* proofs,
                                                     const uint64_t* proof_off, const unsigned char* extra, const uint64_t* extra_off, const unsigned char* gens64, size_t n) {
    if (!e) return s2k_fail("secp256k1_rangeproof_rewind_batch_dev", "null engine");
    if (n == 0) return 1;
    if (!results || !blind_out || !value_out || !nonces || !min_value || !max_value || !commits33 || !proofs || !proof_off || !gens64 || (message_out && !outlen))
        return s2k_fail_arg("secp256k1_rangeproof_rewind_batch_dev", "illegal argument (ARG_CHECK)");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    const size_t nw = std::min(n, RP_CHUNK);
    if (!engine_workspace(e, ws_need({8 * n, nw * 4096, nw * 4096, nw * 1024}))) return 0;
    ws_carver c{e->ws, 0};
    rp_rewind_args ra;
    ra.nonces = nonces; ra.blind_out = blind_out; ra.value_out = value_out; ra.msg_out = message_out; ra.msg_stride = msg_stride;
    ra.outlen = message_out ? outlen : c.take<uint64_t>(n);
    ra.ev = c.take<u32>(nw * 1024); ra.prep = c.take<u32>(nw * 1024); ra.secs = c.take<u32>(nw * 256);
    if (!message_out) HIPCHK(hipMemsetAsync(ra.outlen, 0, 8 * n, st));
    // inputs_on_stream = 1 whatever S2K_OPT_RP_INPUTS_READY says: the replay kernel writes ra.prep / ra.secs -- carved from the SHARED workspace
    // -- on the side stream, so that stream has to wait for whatever an earlier call queued on `st` may still be doing with the workspace
    return rp_launch(e, st, results, min_value, max_value, commits33, proofs, proof_off, extra, extra_off, gens64, n, &ra, 1);
}
// single-item forms with the reference's argument lists.  A 0 from these means "invalid" only while s2k_last_status() is
// S2K_STATUS_OK; an engine-level failure also returns 0 (never 1) and leaves S2K_STATUS_ENGINE_FAILURE for the caller's
// CPU fallback (integration/secp256k1_amd_hook.c does exactly that).
static s2k_engine* g_default_engine = nullptr;
static std::mutex g_default_mu;
static s2k_engine* default_engine() {
    std::lock_guard<std::mutex> lock(g_default_mu);
    if (!g_default_engine) {
        const char* d = getenv("S2K_DEVICE");
        g_default_engine = s2k_engine_create(d ? atoi(d) : 0);
    }
    return g_default_engine;
}
// include/secp256k1_rangeproof.h:70-80
extern "C" int secp256k1_rangeproof_verify_amd(const void* ctx, uint64_t* min_value, uint64_t* max_value, const void* commit,
                                               const unsigned char* proof, size_t plen, const unsigned char* extra_commit,
                                               size_t extra_commit_len, const void* gen) {
    (void)ctx;
    s2k_clear_status();
    if (!min_value || !max_value || !commit || !proof || !gen || (!extra_commit && extra_commit_len)) return s2k_fail_arg("secp256k1_rangeproof_verify_amd", "illegal argument (ARG_CHECK)");
    s2k_engine* e = default_engine();
    if (!e) return 0;
    int32_t res = 0; uint64_t off[2] = {0, plen}, eoff[2] = {0, extra_commit_len};
    if (!secp256k1_rangeproof_verify_batch(e, &res, min_value, max_value, (const unsigned char*)commit, proof, off,
                                           extra_commit_len ? extra_commit : nullptr, extra_commit_len ? eoff : nullptr, (const unsigned char*)gen, 1)) return 0;
    return res;
}
// include/secp256k1_schnorrsig.h:178 -- pubkey points at the 64-byte secp256k1_xonly_pubkey object
extern "C" int secp256k1_schnorrsig_verify_amd(const void* ctx, const unsigned char* sig64, const unsigned char* msg, size_t msglen, const void* pubkey) {
    (void)ctx;
    s2k_clear_status();
    if (!sig64 || (!msg && msglen) || !pubkey) return s2k_fail_arg("secp256k1_schnorrsig_verify_amd", "illegal argument (ARG_CHECK)");
    s2k_engine* e = default_engine();
    if (!e) return 0;
    int32_t res = 0; const unsigned char dummy = 0;
    if (!secp256k1_schnorrsig_verify_batch(e, &res, sig64, msglen ? msg : &dummy, msglen, (const unsigned char*)pubkey, 1, 1)) return 0;
    return res;
}
// include/secp256k1_generator.h:190 -- arrays of pointers to 64-byte secp256k1_pedersen_commitment objects
extern "C" int secp256k1_pedersen_verify_tally_amd(const void* ctx, const void* const* commits, size_t pcnt, const void* const* ncommits, size_t ncnt) {
    (void)ctx;
    s2k_clear_status();
    if ((!commits && pcnt) || (!ncommits && ncnt)) return s2k_fail_arg("secp256k1_pedersen_verify_tally_amd", "illegal argument (ARG_CHECK)");
    s2k_engine* e = default_engine();
    if (!e) return 0;
    std::vector<unsigned char> c33(33 * (pcnt + ncnt) + 1);
    for (size_t i = 0; i < pcnt; i++) memcpy(&c33[33 * i], commits[i], 33);
    for (size_t i = 0; i < ncnt; i++) memcpy(&c33[33 * (pcnt + i)], ncommits[i], 33);
    const uint64_t off[2] = {0, pcnt + ncnt}, npos[1] = {pcnt};
    int32_t res = 0;
    if (!secp256k1_pedersen_verify_tally_batch(e, &res, c33.data(), off, npos, 1)) return 0;
    return res;
}
// include/secp256k1_surjectionproof.h:256 -- proof points at a secp256k1_surjectionproof object (:50-62 of that header:
// size_t n_inputs; unsigned char used_inputs[256/8]; unsigned char data[32*(1+256)]), tags at arrays of 64-byte generators.
// The object is re-serialised (secp256k1_surjectionproof_serialize, main_impl.h:84-106) and takes the batch path, so an
// object that secp256k1_surjectionproof_parse could not have produced verifies as 0.
extern "C" int secp256k1_surjectionproof_verify_amd(const void* ctx, const void* proof, const void* ephemeral_input_tags, size_t n_ephemeral_input_tags,
                                                    const void* ephemeral_output_tag) {
    (void)ctx;
    s2k_clear_status();
    if (!proof || !ephemeral_input_tags || !ephemeral_output_tag) return s2k_fail_arg("secp256k1_surjectionproof_verify_amd", "illegal argument (ARG_CHECK)");
    s2k_engine* e = default_engine();
    if (!e) return 0;
    struct sj_obj { size_t n_inputs; unsigned char used[32]; unsigned char data[32 * 257]; };
    const sj_obj* o = (const sj_obj*)proof;
    if (o->n_inputs > 256) return 0;
    const size_t bm = (o->n_inputs + 7) / 8;
    size_t used = 0;
    for (size_t i = 0; i < bm; i++) used += (size_t)__builtin_popcount(o->used[i]);
    std::vector<unsigned char> ser(2 + bm + 32 * (1 + used));
    ser[0] = (unsigned char)(o->n_inputs & 0xFF); ser[1] = (unsigned char)(o->n_inputs >> 8);
    memcpy(&ser[2], o->used, bm);
    memcpy(&ser[2 + bm], o->data, 32 * (1 + used));
    const uint64_t poff[2] = {0, ser.size()}, toff[2] = {0, n_ephemeral_input_tags};
    int32_t res = 0;
    if (!secp256k1_surjectionproof_verify_batch(e, &res, ser.data(), poff, (const unsigned char*)ephemeral_input_tags, toff, (const unsigned char*)ephemeral_output_tag, 1)) return 0;
    return res;
}

// ------------------------------------------------------------------------------------------------------------
// BIP-340 batch verification (schnorr.h): one signature per lane
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
k_schnorr_verify(int32_t* __restrict__ results, schnorr_midstate mid, const unsigned char* __restrict__ sigs, const unsigned char* __restrict__ msgs,
                 size_t msglen, const unsigned char* __restrict__ pks, int pk_format, const u32* __restrict__ gtab, u32* __restrict__ ptab, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int live = i < n;
    const size_t ii = live ? i : 0;
    __shared__ u32 s_dig[S2K_DIG_WORDS * 256];
    const lane_mem lm{ptab + i * S2K_PTAB_WORDS, S2K_LANE_DIG(s_dig)};
    const int r = schnorr_verify_lane(mid, sigs + 64 * ii, msgs + msglen * ii, msglen, pks + (pk_format ? 64 : 32) * ii, pk_format, live, gtab, lm);
    if (live) results[i] = r;
}
extern "C" int secp256k1_schnorrsig_verify_batch_dev(s2k_engine* e, void* stream, int32_t* results, const unsigned char* sigs,
                                                     const unsigned char* msgs, size_t msglen, const unsigned char* pubkeys, int pk_format, size_t n) {
    if (!e) return s2k_fail("secp256k1_schnorrsig_verify_batch_dev", "null engine");
    if (n == 0) return 1;
    HIPCHK(hipSetDevice(e->device));
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    if (!engine_ptab(e, ((std::min(n, e->max_lanes) + 255) / 256) * 256)) return 0;
    ENGINE_GTAB(e, st);
    HIPCHK(hipMemsetAsync(results, 0, sizeof(int32_t) * n, st));          // a batch that does not complete never shows an item as valid
    HIPCHK(hipEventRecord(e->ev[0], st)); HIPCHK(hipEventRecord(e->ev[2], st));
    for (size_t i0 = 0; i0 < n; i0 += e->max_lanes) {
        const size_t m = std::min(n - i0, e->max_lanes);
        hipLaunchKernelGGL(k_schnorr_verify, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, results + i0, e->bip340, sigs + 64 * i0, msgs + msglen * i0, msglen,
                           pubkeys + (pk_format ? 64 : 32) * i0, pk_format, e->gtab, e->ptab, m);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[3], st)); HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
}
extern "C" int secp256k1_schnorrsig_verify_batch(s2k_engine* e, int32_t* results, const unsigned char* sigs, const unsigned char* msgs,
                                                 size_t msglen, const unsigned char* pubkeys, int pk_format, size_t n) {
    if (!e) return s2k_fail("secp256k1_schnorrsig_verify_batch", "null engine");
    if (results && n) memset(results, 0, sizeof(int32_t) * n);
    if (n == 0) return 1;
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    const size_t pkb = pk_format ? 64 : 32;
    if (!engine_workspace(e, ws_need({4 * n, 64 * n, msglen * n + 64, pkb * n}))) return 0;
    ws_carver w{e->ws, 0};
    int32_t* d_res = w.take<int32_t>(n); unsigned char* d_sig = w.take<unsigned char>(64 * n);
    unsigned char* d_msg = w.take<unsigned char>(msglen * n + 64); unsigned char* d_pk = w.take<unsigned char>(pkb * n);
    HIPCHK(hipMemcpyAsync(d_sig, sigs, 64 * n, hipMemcpyHostToDevice, e->stream));
    if (msglen) HIPCHK(hipMemcpyAsync(d_msg, msgs, msglen * n, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d_pk, pubkeys, pkb * n, hipMemcpyHostToDevice, e->stream));
    if (!secp256k1_schnorrsig_verify_batch_dev(e, nullptr, d_res, d_sig, d_msg, msglen, d_pk, pk_format, n)) return 0;
    HIPCHK(hipMemcpyAsync(results, d_res, 4 * n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 1;
}

// ------------------------------------------------------------------------------------------------------------
// multi-scalar multiplication (msm.h)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_msm_prep(u32* term, u32* halves, const unsigned char* g_sc, const unsigned char* sc, const unsigned char* pt, const unsigned char* pt_inf, size_t n, size_t nt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt) return;
    const int isg = (i == n);       // only when g_sc != NULL (nt == n + 1)
    msm_prep_term(term + i * MSM_TERM_WORDS, halves + i * MSM_HALF_WORDS, isg ? g_sc : sc + 32 * i, isg ? sc : pt + 64 * i,
                  isg ? 0 : (pt_inf ? pt_inf[i] != 0 : 0), isg);
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
#define MSM_COARSE_LDS (9 * 257 + 7)          /* wn * nco: 9 x 257 (c = 16), 9 x 129 (c = 15), 10 x 65 (c = 14) */
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
    __shared__ u32 s_cnt[136], s_loff[136], s_tot[136];
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
__global__ void __launch_bounds__(256, 2)
k_msm_round1(u32* out28, const u32* refs, const u32* off_in, const u32* cnt_in, msm_layout L, msm_plan pl, const u32* off_out, const u32* term, u32 nk, u32 T) {
    const u32 m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= off_out[nk]) return;
    const u32 k = msm_find_key(off_out, nk, m), j = m - off_out[k];
    // bucket k's references: its region of the fixed-capacity layout
    const size_t first = msm_region(L, pl, k / pl.nb, k % pl.nb); (void)off_in;
    const size_t start = first + (size_t)j * T, end = min(start + T, first + cnt_in[k]);
    gej o; msm_sum_refs(o, refs, start, end, term);
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
            gej v; msm_sum_refs(v, refs, first, first + cnt, term);
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
// Horner over the share's windows (~c*windows sequential doublings: the latency floor of one MSM).  One wavefront, all 64 lanes
// running the same point through msm_combine: the runs of doublings spread each field element over the lanes (cofield.h), the
// additions in between are the serial code executed redundantly.  Its own launch bounds so that the point state stays in registers.
__global__ void __launch_bounds__(64)
k_msm_combine(u32* out28, const u32* wsum28, msm_plan pl) {
    if (blockIdx.x) return;
    gej r; msm_combine(r, wsum28, pl);
    if (threadIdx.x == 0) gej_store28(out28, r);
}
// exact path: final <- exact result when the binning pass overflowed a bucket region; the flag goes to the engine's status word
__global__ void k_msm_pick(u32* final28, const u32* exact28, const u32* flags, u32* dev_flags) {
    if (threadIdx.x == 0) dev_flags[0] = flags[0];
    if (flags[0] == 0) return;
    for (int i = threadIdx.x; i < 28; i += blockDim.x) final28[i] = exact28[i];
}
// Bucket-free form: lane l sums (share of k_i)*P_i over the terms i = l, l + lanes, ... with one full double-and-add each
// (ecmult.h); the term with index n carries g_sc*G.  Two uses: small inputs (n < MSM_SMALL_N, the analogue of the reference
// switching to Strauss), and -- gated by the overflow flag -- the exact path of a bucket launch whose regions overflowed.
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
__global__ void k_gej_finish(unsigned char* r_xy, int32_t* r_inf, const u32* in28) {
    if (threadIdx.x || blockIdx.x) return;
    gej r; gej_load28(r, in28);
    ge a; ge_set_gej(a, r);
    if (r.inf) { for (int k = 0; k < 64; k++) r_xy[k] = 0; } else ge_store_b64(r_xy, a);
    *r_inf = r.inf;
}

// reduce `count` gej28 (one segment) down to one, ping-ponging between two scratch buffers; returns pointer to the result
static const u32* launch_gej_reduce(hipStream_t st, const u32* in, u32* bufA, u32* bufB, u32 nseg, u32 seg_len, const u32* gate = nullptr) {
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
    C.shift = pl.c > 13 ? 7u : (pl.c > 7 ? 6u : 0u);
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
    u32 T = (u32)(E / 786432); if (T < 8) T = 8; if (T > 64) T = 64; return T;
}
#define MSM_T2 8u
#define MSM_DIRECT_LANES 16384u          /* lanes of the bucket-free exact path (each walks its terms with a stride) */
static msm_plan engine_msm_plan(const s2k_engine* e, size_t nt) { return msm_make_plan(nt, e->msm_diag.c); }
static size_t msm_ws_bytes(const s2k_engine* e, size_t nt, const msm_plan& pl) {
    const size_t nk = (size_t)pl.windows * pl.nb;
    const size_t E = nt * 2 * pl.windows;
    const msm_layout L = msm_make_layout(nt, pl); const size_t T = msm_run_len(e, E, pl, L);
    return ws_need({28 * 4, 64, (size_t)MSM_DIRECT_LANES * 28 * 4, 64 * 28 * 4 * 2, nt * MSM_TERM_WORDS * 4, nt * MSM_HALF_WORDS * 4, (nk + 1) * 4 * 7, 1024 * 4,
                    msm_refs_words(pl, L) * 4, nk * 28 * 4, msm_pairs_words(e, nt, pl, L) * 8, (size_t)pl.windows * 520 * 4, (nk + E / T + 2) * 28 * 4, (nk * 2 + E / T / MSM_T2 + 64) * 28 * 4,
                    (nk / 1024 + nt / 1024 + pl.windows + 64) * 28 * 4 * 2}) + 32 * 256;
}
static void launch_scan(hipStream_t st, u32* off, u32* cur, u32* tile_sum, const u32* in, u32 nk) {
    const u32 tiles = (nk + 1023) / 1024;
    hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(256), 0, st, off, tile_sum, in, nk);
    hipLaunchKernelGGL(k_scan_fix, dim3(tiles), dim3(256), 0, st, off, cur, tile_sum, nk);
}
// core: leaves the Jacobian result (28 words) at *result28 (device).  Workspace must already be large enough.
// Fully stream-ordered: nothing is read back.  The number of partial-sum rounds follows from the bucket-region capacity (a
// bucket never holds more than its region), and an input that overflows a region (adversarially equal scalars) raises a
// device flag that un-gates the exact bucket-free path queued behind the bucket pipeline; k_msm_pick then publishes its
// result instead.  (part, parts): the share of the digit windows this launch owns (msm_plan_share) -- (0, 1) = all of them.
__global__ void k_set_word(u32* p, u32 v) { *p = v; }
__global__ void k_msm_flag_copy(u32* persist, const u32* flags) { persist[0] = flags[0]; }
// side: where the gated exact path runs (with its fork / join events); arena: which MSM_DIRECT_LANES-sized region of the engine's table
// arena its lanes use (0: the engine's own calls; 1, 2: the two pipelined slots)
struct msm_ctx { hipStream_t side; hipEvent_t fork, join; unsigned arena; };
static int msm_launch(s2k_engine* e, hipStream_t st, ws_carver& c, u32** result28, const unsigned char* g_sc, const unsigned char* sc,
                      const unsigned char* pt, const unsigned char* pt_inf, size_t n, u32 part = 0, u32 parts = 1, const msm_ctx* ctx = nullptr) {
    const msm_ctx dflt{e->stream2, e->ev_msm_fork, e->ev_msm_join, 0u};
    const msm_ctx& X = ctx ? *ctx : dflt;
    const size_t nt = n + (g_sc ? 1 : 0);
    if (parts == 0 || part >= parts) return s2k_fail_arg("s2k_ecmult_multi", "window share out of range");
    ENGINE_GTAB(e, st);                                        // (the bucket-free exact path multiplies by G through the table)
    msm_plan pl = engine_msm_plan(e, nt ? nt : 1);
    // term references are packed as (u32)(term << 2 | half << 1 | sign): refuse what those cannot index instead of wrapping silently
    if (nt >= (size_t(1) << 30) || nt * 2 * pl.windows >= (size_t(1) << 32))
        return s2k_fail("s2k_ecmult_multi", "too many terms for 32-bit bucket references (2 * windows * n >= 2^32): split the sum, partial sums add");
    msm_plan_share(pl, part, parts);
    u32* final28 = c.take<u32>(28);
    *result28 = final28;
    u32* flags = c.take<u32>(16);                              // flags[0]: a bucket region overflowed
    u32* lanes = c.take<u32>((size_t)MSM_DIRECT_LANES * 28); u32* dbufA = c.take<u32>(64 * 28); u32* dbufB = c.take<u32>(64 * 28);
    HIPCHK(hipMemsetAsync(flags, 0, 64, st));
    if (nt == 0 || pl.wn == 0) {                               // empty sum / empty share: infinity
        HIPCHK(hipMemsetAsync(final28, 0, 27 * 4, st));
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, st, final28 + 27, 1u);
        hipLaunchKernelGGL(k_msm_flag_copy, dim3(1), dim3(1), 0, st, e->dev_flags, flags);
        HIPCHK(hipGetLastError());
        return 1;
    }
    if (nt < MSM_SMALL_N) {
        const unsigned dl = (unsigned)(((nt + 255) / 256) * 256);
        if (!engine_ptab(e, 3 * (size_t)MSM_DIRECT_LANES)) return 0;
        HIPCHK(hipEventRecord(e->ev[2], st));
        hipLaunchKernelGGL(k_msm_direct, dim3(dl / 256), dim3(256), 0, st, lanes, (const u32*)nullptr, g_sc, sc, pt, pt_inf, e->gtab,
                           e->ptab + (size_t)X.arena * MSM_DIRECT_LANES * S2K_PTAB_WORDS, n, nt, pl);
        HIPCHK(hipEventRecord(e->ev[3], st));
        const u32* r = launch_gej_reduce(st, lanes, dbufA, dbufB, 1, dl);
        HIPCHK(hipMemcpyAsync(final28, r, 28 * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_msm_flag_copy, dim3(1), dim3(1), 0, st, e->dev_flags, flags);
        HIPCHK(hipGetLastError());
        return 1;
    }
    if (!engine_ptab(e, 3 * (size_t)MSM_DIRECT_LANES)) return 0;
    u32* const direct_ptab = e->ptab + (size_t)X.arena * MSM_DIRECT_LANES * S2K_PTAB_WORDS;
    const u32 nk = pl.wn * pl.nb;
    const size_t E = nt * 2 * pl.wn;                           // upper bound on this share's bucket references
    const msm_layout L = msm_make_layout(nt, pl);
    const msm_plan full = engine_msm_plan(e, nt);              // (run length as msm_ws_bytes sized the buffers for: from the whole plan, not the share)
    const u32 T = msm_run_len(e, nt * 2 * pl.windows, full, L), T2 = MSM_T2;
    const size_t bound1 = (size_t)nk + E / T + 2;
    u32* term = c.take<u32>(nt * MSM_TERM_WORDS); u32* halves = c.take<u32>(nt * MSM_HALF_WORDS);
    u32* gcnt = c.take<u32>(nk + 1); u32* gclamp = c.take<u32>(nk + 1); u32* spare = c.take<u32>(nk + 1);
    u32* cntA = c.take<u32>(nk + 1); u32* cntB = c.take<u32>(nk + 1); u32* offA = c.take<u32>(nk + 1); u32* offB = c.take<u32>(nk + 1);
    u32* tile_sum = c.take<u32>(1024); (void)spare;
    u32* refs_cap = c.take<u32>(msm_refs_words(pl, L)); u32* buckets = c.take<u32>((size_t)nk * 28);
    unsigned long long* pairs = c.take<unsigned long long>(msm_pairs_words(e, nt, pl, L)); u32* ccnt = c.take<u32>((size_t)pl.windows * 520);
    u32* partA = c.take<u32>(bound1 * 28); u32* partB = c.take<u32>(((size_t)nk * 2 + E / T / MSM_T2 + 64) * 28);
    u32* bufA = c.take<u32>(((size_t)nk / 1024 + pl.windows + 64) * 28); u32* bufB = c.take<u32>(((size_t)nk / 1024 + pl.windows + 64) * 28);
    HIPCHK(hipMemsetAsync(gcnt, 0, (nk + 1) * 4, st));
    const unsigned bt = (unsigned)((nt + 255) / 256), bk = (nk + 255) / 256;
    hipLaunchKernelGGL(k_msm_prep, dim3(bt), dim3(256), 0, st, term, halves, g_sc, sc, pt, pt_inf, n, nt);
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
    if (msm_max_cap(full, L) <= MSM_ONE_ROUND_CAP && !e->msm_diag.no_small) {
        const u32 nchunks = (pl.nb - 1 + 255) / 256;
        hipLaunchKernelGGL(k_msm_small_windows, dim3(nchunks, pl.wn), dim3(256), 0, st, partA, refs_cap, gcnt, term, pl, L, nchunks);
        const u32* wsum = launch_gej_reduce(st, partA, bufA, bufB, pl.wn, nchunks);
        hipLaunchKernelGGL(k_msm_combine, dim3(1), dim3(64), 0, st, final28, wsum, pl);
        HIPCHK(hipStreamWaitEvent(st, X.join, 0));
        hipLaunchKernelGGL(k_msm_pick, dim3(1), dim3(32), 0, st, final28, ex, (const u32*)flags, e->dev_flags);
        HIPCHK(hipGetLastError());
        return 1;
    }
    hipLaunchKernelGGL(k_msm_counts, dim3(bk), dim3(256), 0, st, cntA, gclamp, gcnt, nk, T, L, pl);
    // rounds: a bucket holds at most its region's capacity, so the capacity fixes how many rounds reach "one partial per bucket"
    const int has_top = (pl.w0 + pl.wn == pl.windows);
    const u32 maxcap = std::max(pl.wn > (has_top ? 1u : 0u) ? L.cap : 0u, has_top ? L.cap_top : 0u);
    int rounds = 1; { size_t reach = T; while (reach < maxcap) { reach *= T2; rounds++; } }
    // round 1: references -> partial sums (at most T references each)
    launch_scan(st, offA, nullptr, tile_sum, cntA, nk);
    HIPCHK(hipEventRecord(e->ev[2], st));
    hipLaunchKernelGGL(k_msm_round1, dim3((unsigned)((bound1 + 255) / 256)), dim3(256), 0, st, partA, refs_cap, (const u32*)nullptr, gclamp, L, pl, offA, term, nk, T);
    HIPCHK(hipEventRecord(e->ev[3], st));
    // rounds 2..R: partial sums of partial sums until every bucket holds at most one
    u32 *cin = cntA, *cout = cntB, *oin = offA, *oout = offB, *pin = partA, *pout = partB;
    size_t bound = bound1;
    for (int r = 2; r <= rounds; r++) {
        bound = (size_t)nk + bound / T2 + 2;
        hipLaunchKernelGGL(k_msm_counts, dim3(bk), dim3(256), 0, st, cout, (u32*)nullptr, cin, nk, T2, L, pl);
        launch_scan(st, oout, nullptr, tile_sum, cout, nk);
        hipLaunchKernelGGL(k_msm_roundN, dim3((unsigned)((bound + 255) / 256)), dim3(256), 0, st, pout, pin, oin, oout, nk, T2);
        u32* t;
        t = cin; cin = cout; cout = t; t = oin; oin = oout; oout = t; t = pin; pin = pout; pout = t;
    }
    hipLaunchKernelGGL(k_msm_finish, dim3(bk), dim3(256), 0, st, buckets, pin, oin, nk, pl, L);
    const u32* wsum = launch_gej_reduce(st, buckets, bufA, bufB, pl.wn, pl.nb);
    hipLaunchKernelGGL(k_msm_combine, dim3(1), dim3(64), 0, st, final28, wsum, pl);
    HIPCHK(hipStreamWaitEvent(st, X.join, 0));
    hipLaunchKernelGGL(k_msm_pick, dim3(1), dim3(32), 0, st, final28, ex, (const u32*)flags, e->dev_flags);
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
    if (need > S.ws_bytes) {
        HIPCHK(hipStreamSynchronize(S.s)); HIPCHK(hipStreamSynchronize(S.s2));
        if (S.ws) HIPCHK(hipFree(S.ws));
        S.ws = nullptr; S.ws_bytes = 0;
        const size_t bytes = (need + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
        HIPCHK(hipMalloc((void**)&S.ws, bytes));
        S.ws_bytes = bytes;
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
    if (!msm_launch(e, S.s, c, &res, g_sc, sc, pt_xy, pt_inf, n, 0, 1, &ctx)) return 0;
    if (finish) hipLaunchKernelGGL(k_gej_finish, dim3(1), dim3(64), 0, S.s, r_xy, r_inf, res);
    else HIPCHK(hipMemcpyAsync(out28, res, 28 * 4, hipMemcpyDeviceToDevice, S.s));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[1], S.s));
    HIPCHK(hipEventRecord(S.done, S.s));
    HIPCHK(hipStreamWaitEvent(st, S.done, 0));               // the result is stream-ordered for the caller
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
    if (e->msm_pipeline && nt >= MSM_SMALL_N && nt <= MSM_PIPE_MAX_TERMS) return msm_pipelined(e, st, 0, r_gej28, nullptr, nullptr, g_sc, sc, pt_xy, pt_inf, n);
    const msm_plan pl = engine_msm_plan(e, nt ? nt : 1);
    if (!engine_workspace(e, msm_ws_bytes(e, nt + 1, pl))) return 0;
    ws_carver c{e->ws, 0}; u32* res = nullptr;
    HIPCHK(hipEventRecord(e->ev[0], st));
    if (!msm_launch(e, st, c, &res, g_sc, sc, pt_xy, pt_inf, n)) return 0;
    HIPCHK(hipMemcpyAsync(r_gej28, res, 28 * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
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
    if (!msm_launch(e, st, c, &res, g_sc, sc, pt_xy, pt_inf, n, part, parts)) return 0;
    HIPCHK(hipMemcpyAsync(r_gej28, res, 28 * 4, hipMemcpyDeviceToDevice, st));
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
    if (e->msm_pipeline && nt >= MSM_SMALL_N && nt <= MSM_PIPE_MAX_TERMS) return msm_pipelined(e, st, 1, nullptr, r_xy, r_inf, g_sc, sc, pt_xy, pt_inf, n);
    const msm_plan pl = engine_msm_plan(e, nt ? nt : 1);
    if (!engine_workspace(e, msm_ws_bytes(e, nt + 1, pl))) return 0;
    ws_carver c{e->ws, 0}; u32* res = nullptr;
    HIPCHK(hipEventRecord(e->ev[0], st));
    if (!msm_launch(e, st, c, &res, g_sc, sc, pt_xy, pt_inf, n)) return 0;
    hipLaunchKernelGGL(k_gej_finish, dim3(1), dim3(64), 0, st, r_xy, r_inf, res);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
}
extern "C" int s2k_gej_sum_dev(s2k_engine* e, void* stream, unsigned char* r_xy, int32_t* r_inf, const uint32_t* gej28, size_t count) {
    if (!e) return s2k_fail("s2k_gej_sum_dev", "null engine");
    if (count == 0) return s2k_fail("s2k_gej_sum_dev", "count == 0");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
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
    const size_t nt = n + (g_sc ? 1 : 0);
    const msm_plan pl = engine_msm_plan(e, nt ? nt : 1);
    // the staged inputs come first in the workspace, the MSM passes carve what follows
    if (!engine_workspace(e, ws_need({32 * n + 64, 64 * n + 64, n + 64, 64, 64, 16}) + msm_ws_bytes(e, nt + 1, pl))) return 0;
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
    u32* res = nullptr;
    HIPCHK(hipEventRecord(e->ev[0], st));
    if (!msm_launch(e, st, c, &res, g_sc ? d_g : nullptr, d_sc, d_pt, pt_inf ? d_inf : nullptr, n)) return 0;
    hipLaunchKernelGGL(k_gej_finish, dim3(1), dim3(64), 0, st, d_r, d_ri, res);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[1], st));
    HIPCHK(hipMemcpyAsync(r_xy, d_r, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r_inf, d_ri, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 1;
}

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
k_bp_terms_fixed(u32* out28, unsigned char* term_ok, bp_shape sh, const u32* term_sc, const int* proof_ok, const u32* tab, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t p = t / sh.n_gens; const u32 ti = (u32)(t % sh.n_gens);
    if (p >= n) return;
    gej o; gej_set_infinity(o);
    if (proof_ok[p]) bp_term_fixed(o, tab, ti, term_sc + (p * sh.n_terms + ti) * 8);
    gej_store28(out28 + (p * sh.n_terms + ti) * 28, o); term_ok[p * sh.n_terms + ti] = 1;
}
__global__ void k_bp_tab_base(u32* tab, const u32* gens18, u32 n_gens) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_gens * BP_TAB_WINDOWS) bp_tab_build_base(tab, gens18, t / BP_TAB_WINDOWS, t % BP_TAB_WINDOWS);
}
__global__ void __launch_bounds__(256)
k_bp_tab_entries(u32* tab, size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const u32 v = (u32)(t & 0xFFFFu); const size_t gw = t >> BP_TAB_BITS;
    if (v >= 2) bp_tab_build_entry(tab, (u32)(gw / BP_TAB_WINDOWS), (u32)(gw % BP_TAB_WINDOWS), v);
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
        HIPCHK(hipStreamSynchronize(st));
        if (e->bp_tab) { hipFree(e->bp_tab); e->bp_tab = nullptr; }
        e->bp_key.clear();
        if (hipMalloc((void**)&e->bp_tab, bp_tab_words(n_gens) * sizeof(u32)) != hipSuccess) { (void)hipGetLastError(); e->bp_tab = nullptr; *fixed = 0; }
        else {
            const size_t total = (n_gens * BP_TAB_WINDOWS) << BP_TAB_BITS;
            hipLaunchKernelGGL(k_bp_tab_base, dim3((unsigned)((n_gens * BP_TAB_WINDOWS + 63) / 64)), dim3(64), 0, st, e->bp_tab, gens18, (u32)n_gens);
            hipLaunchKernelGGL(k_bp_tab_entries, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, e->bp_tab, total);
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
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, st, (u32*)gens_ok, (u32)e->bp_gens_ok);
    } else {
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, st, (u32*)gens_ok, 1u);
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
            hipLaunchKernelGGL(k_bp_terms_fixed, dim3((unsigned)((n * n_gens + 255) / 256)), dim3(256), 0, e->stream2, out28, term_ok, sh, term_sc, proof_ok, e->bp_tab, n);
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
k_bpc_terms(u32* out28, const u32* v8, const unsigned char* n_vec, const unsigned char* l_vec, u32 g_len, u32 h_len, const u32* tab, const u32* gens18, const int* gens_ok,
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
    if (ti == g_len + h_len) bpc_gmul(o, gtab, k8);
    else if (tab) bp_term_fixed(o, tab, ti, k8);
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
    hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, st, (u32*)gens_ok, 1u);
    HIPCHK(hipEventRecord(e->ev[0], st));
    hipLaunchKernelGGL(k_bp_gens, dim3((unsigned)((n_gens + 63) / 64)), dim3(64), 0, st, gens18, gens_ok, d_g33, (u32)n_gens);
    int fixed = 0;
    if (!bp_ensure_table(e, st, gens18, gens_ok, gens33_host, n_gens, &fixed)) return 0;
    if (!fixed && !engine_ptab(e, ((n * T + 255) / 256) * 256)) return 0;
    hipLaunchKernelGGL(k_bpc_scalars, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, v8, d_nv, d_lv, d_cv, d_mu, (u32)g_len, (u32)h_len, n);
    HIPCHK(hipEventRecord(e->ev[2], st));
    hipLaunchKernelGGL(k_bpc_terms, dim3((unsigned)((n * T + 255) / 256)), dim3(256), 0, st, out28, v8, d_nv, d_lv, (u32)g_len, (u32)h_len, fixed ? e->bp_tab : (const u32*)nullptr,
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

// ------------------------------------------------------------------------------------------------------------
// surjection-proof batch verification (surjection.h): one proof per lane
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
k_sj_verify(int32_t* __restrict__ results, const unsigned char* __restrict__ proofs, const uint64_t* __restrict__ proof_off,
            const unsigned char* __restrict__ in_tags, const uint64_t* __restrict__ tag_off, const unsigned char* __restrict__ out_tags,
            const u32* __restrict__ gtab, u32* __restrict__ ptab, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int live = i < n;
    const size_t ii = live ? i : 0;
    __shared__ u32 s_dig[S2K_DIG_WORDS * 256];
    const lane_mem lm{ptab + i * S2K_PTAB_WORDS, S2K_LANE_DIG(s_dig)};
    const int r = sj_verify_lane(proofs + proof_off[ii], proof_off[ii + 1] - proof_off[ii], in_tags + 64 * tag_off[ii], tag_off[ii + 1] - tag_off[ii],
                                 out_tags + 64 * ii, live, gtab, lm);
    if (live) results[i] = r;
}
extern "C" int secp256k1_surjectionproof_verify_batch_dev(s2k_engine* e, void* stream, int32_t* results, const unsigned char* proofs,
                                                          const uint64_t* proof_off, const unsigned char* input_tags64, const uint64_t* tag_off,
                                                          const unsigned char* output_tags64, size_t n) {
    if (!e) return s2k_fail("secp256k1_surjectionproof_verify_batch_dev", "null engine");
    if (n == 0) return 1;
    HIPCHK(hipSetDevice(e->device));
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    if (!engine_ptab(e, ((std::min(n, e->max_lanes) + 255) / 256) * 256)) return 0;
    ENGINE_GTAB(e, st);
    HIPCHK(hipMemsetAsync(results, 0, sizeof(int32_t) * n, st));          // a batch that does not complete never shows an item as valid
    HIPCHK(hipEventRecord(e->ev[0], st)); HIPCHK(hipEventRecord(e->ev[2], st));
    for (size_t i0 = 0; i0 < n; i0 += e->max_lanes) {     // offsets are absolute, so a sub-range only shifts the per-item arrays
        const size_t m = std::min(n - i0, e->max_lanes);
        hipLaunchKernelGGL(k_sj_verify, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, results + i0, proofs, proof_off + i0, input_tags64, tag_off + i0,
                           output_tags64 + 64 * i0, e->gtab, e->ptab, m);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[3], st)); HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
}
extern "C" int secp256k1_surjectionproof_verify_batch(s2k_engine* e, int32_t* results, const unsigned char* proofs, const uint64_t* proof_off,
                                                      const unsigned char* input_tags64, const uint64_t* tag_off, const unsigned char* output_tags64, size_t n) {
    if (!e) return s2k_fail("secp256k1_surjectionproof_verify_batch", "null engine");
    if (results && n) memset(results, 0, sizeof(int32_t) * n);
    if (n == 0) return 1;
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    const size_t pbytes = (size_t)proof_off[n], ntags = (size_t)tag_off[n];
    if (!engine_workspace(e, ws_need({4 * n, pbytes + 64, 8 * (n + 1), 64 * ntags + 64, 8 * (n + 1), 64 * n}))) return 0;
    ws_carver w{e->ws, 0};
    int32_t* d_res = w.take<int32_t>(n); unsigned char* d_pr = w.take<unsigned char>(pbytes + 64); uint64_t* d_po = w.take<uint64_t>(n + 1);
    unsigned char* d_in = w.take<unsigned char>(64 * ntags + 64); uint64_t* d_to = w.take<uint64_t>(n + 1); unsigned char* d_out = w.take<unsigned char>(64 * n);
    hipStream_t st = e->stream;
    stream_guard sg(e, st);
    if (pbytes) HIPCHK(hipMemcpyAsync(d_pr, proofs, pbytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_po, proof_off, 8 * (n + 1), hipMemcpyHostToDevice, st));
    if (ntags) HIPCHK(hipMemcpyAsync(d_in, input_tags64, 64 * ntags, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_to, tag_off, 8 * (n + 1), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_out, output_tags64, 64 * n, hipMemcpyHostToDevice, st));
    if (!secp256k1_surjectionproof_verify_batch_dev(e, nullptr, d_res, d_pr, d_po, d_in, d_to, d_out, n)) return 0;
    HIPCHK(hipMemcpyAsync(results, d_res, 4 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 1;
}
// ------------------------------------------------------------------------------------------------------------
// half-aggregated Schnorr signatures (halfagg.h): one (2n+1)-term MSM per aggregate
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_ha_points(unsigned char* pts, unsigned char* pkx32, u32* flags, const unsigned char* aggsig, const unsigned char* pks, int pk_format, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!ha_points(pts + 128 * i, pkx32 + 32 * i, aggsig + 32 * i, pks + (pk_format ? 64 : 32) * i, pk_format)) flags[0] = 1u;
}
__global__ void __launch_bounds__(256)
k_ha_schedule(u32* wk, const unsigned char* aggsig, const unsigned char* pkx32, const unsigned char* msgs32, size_t nblocks) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nblocks) ha_schedule(wk + 64 * j, aggsig, pkx32, msgs32, j);
}
// ha_rounds with the three-input logic spelled out (v_bitop3_b32: xor3 = 0x96, ch = 0xCA, maj = 0xE8): 14 vector
// instructions per round instead of the 18 the generic source compiles to -- this chain is pure single-wave issue latency
__device__ __forceinline__ void ha_rounds_dev(u32 s[8], const u32* __restrict__ wk) {
    u32 a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
#pragma unroll
    for (int t = 0; t < 64; t++) {
        const u32 S1 = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_alignbit(e, e, 6), __builtin_amdgcn_alignbit(e, e, 11), __builtin_amdgcn_alignbit(e, e, 25), 0x96);
        const u32 ch = __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA);
        const u32 t1 = h + S1 + ch + wk[t];
        const u32 S0 = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_alignbit(a, a, 2), __builtin_amdgcn_alignbit(a, a, 13), __builtin_amdgcn_alignbit(a, a, 22), 0x96);
        const u32 mj = __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + S0 + mj;
    }
    s[0] += a; s[1] += b; s[2] += c; s[3] += d; s[4] += e; s[5] += f; s[6] += g; s[7] += h;
}
__global__ void __launch_bounds__(64)
k_ha_chain(u32* states, const u32* __restrict__ wk, size_t nblocks) {
    u32 st[8]; ha_tag_midstate(st);
    // keep the state in vector registers: left to itself the compiler puts these wave-uniform values in SGPRs and then bounces
    // every rotate through v_alignbit_b32 + v_readfirstlane_b32 (the scalar unit has no rotate), which is ~2.5x slower
    for (int k = 0; k < 8; k++) S2K_OPAQUE(st[k]);
    for (size_t j = 0; j < nblocks; j++) {
        ha_rounds_dev(st, wk + 64 * j);
        if (threadIdx.x == 0) { for (int k = 0; k < 8; k++) states[8 * j + k] = st[k]; }
    }
}
__global__ void __launch_bounds__(256)
k_ha_scalars(unsigned char* sc, unsigned char* g32, u32* flags, const u32* states, schnorr_midstate bip340, const unsigned char* aggsig,
             const unsigned char* pkx32, const unsigned char* msgs32, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && !ha_gscalar(g32, aggsig + 32 * n)) flags[0] = 1u;
    if (i < n) ha_scalars(sc + 64 * i, states, bip340, aggsig, pkx32, msgs32, i);
}
__global__ void k_ha_final(int32_t* result, const u32* flags, const u32* res28) {
    if (threadIdx.x || blockIdx.x) return;
    *result = (flags[0] == 0u) && (res28[27] != 0u);
}
static size_t ha_ws_bytes(const s2k_engine* e, size_t n) {
    const size_t nblocks = (3 * n) >> 1, nt = 2 * n + 1;
    return ws_need({128 * n + 64, 32 * n + 64, nblocks * 256 + 64, nblocks * 32 + 64, 64 * n + 64, 64, 64, 16}) + msm_ws_bytes(e, nt + 1, engine_msm_plan(e, nt));
}
// The randomizer hash's chain on the host (host_sha256.h): state after every full 64-byte block of r_0|x(P_0)|m_0|r_1|..., into pinned
// memory.  pk_format 0: the serialised key IS x(P_i) (a key that does not parse makes the verdict 0 whatever is hashed); 1: the object's
// first 32 bytes are x, least significant byte first.
struct ha_host_src { const unsigned char* aggsig; const unsigned char* pks; int pk_format; const unsigned char* msgs32; };
static void ha_host_chain(u32* states, const ha_host_src& h, size_t nblocks) {
    uint32_t st[8]; ha_tag_midstate(st);
    const size_t pkb = h.pk_format ? 64 : 32;
    unsigned char blk[64];
    for (size_t j = 0; j < nblocks; j++) {
        for (int half = 0; half < 2; half++) {
            const size_t u = 2 * j + half, i = u / 3; const unsigned part = (unsigned)(u % 3);
            unsigned char* o = blk + 32 * half;
            if (part == 0) memcpy(o, h.aggsig + 32 * i, 32);
            else if (part == 2) memcpy(o, h.msgs32 + 32 * i, 32);
            else if (!h.pk_format) memcpy(o, h.pks + pkb * i, 32);
            else { for (int k = 0; k < 32; k++) o[k] = h.pks[pkb * i + 31 - k]; }
        }
        host_sha256_compress(st, blk);
        for (int k = 0; k < 8; k++) states[8 * j + k] = st[k];
    }
}
// device pointers in, verdict to d_res[0]; aggsig_len already checked to be 32 (n + 1).  host: the same inputs in host memory -- the
// hash chain is then walked on the host underneath the point-lifting kernel and its states uploaded (the device chain is 2.4 us per
// block on one wavefront: 118 ms for 2^15 signatures against ~2.6 ms here); nullptr: the device chain.
static int ha_launch(s2k_engine* e, hipStream_t st, ws_carver& c, int32_t* d_res, const unsigned char* d_pk, int pk_format, const unsigned char* d_msg, size_t n,
                     const unsigned char* d_agg, const ha_host_src* host = nullptr) {
    const size_t nblocks = (3 * n) >> 1;
    unsigned char* d_pts = c.take<unsigned char>(128 * n + 64);
    unsigned char* d_pkx = c.take<unsigned char>(32 * n + 64); u32* d_wk = c.take<u32>(nblocks * 64 + 16); u32* d_states = c.take<u32>(nblocks * 8 + 16);
    unsigned char* d_sc = c.take<unsigned char>(64 * n + 64); unsigned char* d_g = c.take<unsigned char>(64); u32* d_flags = c.take<u32>(16);
    if (host && nblocks * 8 > e->ha_pin_words) {
        HIPCHK(hipStreamSynchronize(st));                       // (an earlier call's upload may still read the old buffer)
        if (e->ha_pin) HIPCHK(hipHostFree(e->ha_pin));
        e->ha_pin = nullptr; e->ha_pin_words = 0;
        const size_t words = (nblocks * 8 + 4095) & ~size_t(4095);
        HIPCHK(hipHostMalloc((void**)&e->ha_pin, words * sizeof(u32), hipHostMallocDefault));
        e->ha_pin_words = words;
    }
    HIPCHK(hipMemsetAsync(d_res, 0, 4, st));
    HIPCHK(hipMemsetAsync(d_flags, 0, 64, st));
    HIPCHK(hipEventRecord(e->ev[0], st));
    const unsigned bn = (unsigned)((n + 255) / 256);
    if (n) hipLaunchKernelGGL(k_ha_points, dim3(bn), dim3(256), 0, st, d_pts, d_pkx, d_flags, d_agg, d_pk, pk_format, n);
    if (nblocks && host) {
        HIPCHK(hipGetLastError());
        ha_host_chain(e->ha_pin, *host, nblocks);              // the GPU lifts the points meanwhile
        HIPCHK(hipMemcpyAsync(d_states, e->ha_pin, nblocks * 8 * sizeof(u32), hipMemcpyHostToDevice, st));
    } else if (nblocks) {
        hipLaunchKernelGGL(k_ha_schedule, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, st, d_wk, d_agg, d_pkx, d_msg, nblocks);
        hipLaunchKernelGGL(k_ha_chain, dim3(1), dim3(64), 0, st, d_states, d_wk, nblocks);
    }
    hipLaunchKernelGGL(k_ha_scalars, dim3(bn ? bn : 1), dim3(256), 0, st, d_sc, d_g, d_flags, d_states, e->bip340, d_agg, d_pkx, d_msg, n);
    HIPCHK(hipGetLastError());
    u32* res28 = nullptr;
    // the MSM's points are R_0, P_0, R_1, P_1, ... with scalars z_0, z_0 e_0, z_1, z_1 e_1, ...; the generator term carries -s
    if (!msm_launch(e, st, c, &res28, d_g, d_sc, d_pts, nullptr, 2 * n)) return 0;
    hipLaunchKernelGGL(k_ha_final, dim3(1), dim3(64), 0, st, d_res, d_flags, res28);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
}
// every array in HBM, the verdict lands in result_dev[0] (stream-ordered)
extern "C" int secp256k1_schnorrsig_aggverify_dev(s2k_engine* e, void* stream, int32_t* result_dev, const unsigned char* pubkeys, int pk_format,
                                                  const unsigned char* msgs32, size_t n, const unsigned char* aggsig, size_t aggsig_len) {
    if (!e) return s2k_fail("secp256k1_schnorrsig_aggverify_dev", "null engine");
    if (!result_dev || !aggsig || ((!pubkeys || !msgs32) && n)) return s2k_fail_arg("secp256k1_schnorrsig_aggverify_dev", "illegal argument (ARG_CHECK)");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    if ((aggsig_len / 32) == 0 || (aggsig_len / 32) - 1 != n || (aggsig_len % 32) != 0) { HIPCHK(hipMemsetAsync(result_dev, 0, 4, st)); return 1; }     // main_impl.h:122-125
    if (!engine_workspace(e, ha_ws_bytes(e, n))) return 0;
    ws_carver c{e->ws, 0};
    return ha_launch(e, st, c, result_dev, pubkeys, pk_format, msgs32, n, aggsig);
}
extern "C" int secp256k1_schnorrsig_aggverify_amd(s2k_engine* e, int32_t* result, const unsigned char* pubkeys, int pk_format, const unsigned char* msgs32,
                                                  size_t n, const unsigned char* aggsig, size_t aggsig_len) {
    if (!e) return s2k_fail("secp256k1_schnorrsig_aggverify_amd", "null engine");
    if (!result || !aggsig || ((!pubkeys || !msgs32) && n)) return s2k_fail_arg("secp256k1_schnorrsig_aggverify_amd", "illegal argument (ARG_CHECK)");
    *result = 0;
    if ((aggsig_len / 32) == 0 || (aggsig_len / 32) - 1 != n || (aggsig_len % 32) != 0) return 1;          // main_impl.h:122-125
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    const size_t pkb = pk_format ? 64 : 32;
    const size_t io = ws_need({pkb * n + 64, 32 * n + 64, 32 * (n + 1), 16});
    if (!engine_workspace(e, ha_ws_bytes(e, n) + io)) return 0;
    ws_carver c0{e->ws, ha_ws_bytes(e, n)};
    unsigned char* d_pk = c0.take<unsigned char>(pkb * n + 64); unsigned char* d_msg = c0.take<unsigned char>(32 * n + 64);
    unsigned char* d_agg = c0.take<unsigned char>(32 * (n + 1)); int32_t* d_res = c0.take<int32_t>(4);
    hipStream_t st = e->stream;
    stream_guard sg(e, st);
    if (n) {
        HIPCHK(hipMemcpyAsync(d_pk, pubkeys, pkb * n, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_msg, msgs32, 32 * n, hipMemcpyHostToDevice, st));
    }
    HIPCHK(hipMemcpyAsync(d_agg, aggsig, 32 * (n + 1), hipMemcpyHostToDevice, st));
    ws_carver c{e->ws, 0};
    const int host_chain = e->halfagg_host_chain;      // S2K_OPT_HALFAGG_HOST_CHAIN 0: the device chain (same verdicts; tests)
    const ha_host_src hsrc{aggsig, pubkeys, pk_format, msgs32};
    if (!ha_launch(e, st, c, d_res, d_pk, pk_format, d_msg, n, d_agg, host_chain ? &hsrc : nullptr)) return 0;
    HIPCHK(hipMemcpyAsync(result, d_res, 4, hipMemcpyDeviceToHost, st));
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

// engine options (include/secp256k1_zkp_amd.h)
extern "C" int s2k_engine_set_option(s2k_engine* e, int option, long value) {
    if (!e) return s2k_fail("s2k_engine_set_option", "null engine");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    switch (option) {
    case S2K_OPT_RP_INPUTS_READY: e->rp_inputs_ready = value != 0; return 1;
    case S2K_OPT_RP_SPLIT: e->rp_split = value != 0; return 1;
    case S2K_OPT_MSM_PIPELINE: e->msm_pipeline = value != 0; return 1;
    case S2K_OPT_MAX_LANES: if (value < 256) return s2k_fail_arg("s2k_engine_set_option", "S2K_OPT_MAX_LANES needs at least 256 lanes"); e->max_lanes = (size_t)value & ~size_t(255); return 1;
    case S2K_OPT_STAGE_THREADS: if (value < 1 || value > 64) return s2k_fail_arg("s2k_engine_set_option", "S2K_OPT_STAGE_THREADS: 1..64"); e->stage_threads = (int)value; return 1;
    case S2K_OPT_HALFAGG_HOST_CHAIN: e->halfagg_host_chain = value != 0; return 1;
    case S2K_OPT_SYNC_SPLIT: e->sync_split = value != 0; return 1;
    case S2K_OPT_GEN_CACHE_SLOTS: {                             // (the cache belongs to the device: every engine on it sees the change)
        s2k_dev_pool* p = e->pool;
        std::lock_guard<std::recursive_mutex> pool_lock(p->mu);
        const int v = value < 0 ? 0 : (value > RP_GEN_SLOTS ? RP_GEN_SLOTS : (int)value);
        if (v < p->gen_slots) {                            // slots that go away give their tables back once nothing can still read them
            if (hipSetDevice(e->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return s2k_fail("s2k_engine_set_option", "device synchronisation failed");
            for (int i = v; i < p->gen_slots; i++) {
                if (p->gen[i].valid && !memcmp(p->gen[i].key, k_generator_h, 64) && p->gen_h == 2) p->gen_h = 1;
                if (p->gen[i].tab) hipFree(p->gen[i].tab);
                if (p->gen[i].xmul) hipFree(p->gen[i].xmul);
                p->gen[i].tab = nullptr; p->gen[i].xmul = nullptr; p->gen[i].valid = 0; p->gen[i].pinned = 0;
            }
        }
        p->gen_slots = v; return 1;
    }
    case S2K_OPT_GEN_CACHE_MIN: { std::lock_guard<std::recursive_mutex> pool_lock(e->pool->mu); e->pool->gen_min = value < 1 ? 1 : (size_t)value; return 1; }
    default: return s2k_fail_arg("s2k_engine_set_option", "unknown option");
    }
}
// pre-size the workspace for batches of n_items rangeproofs (optional: every call grows it on demand)
extern "C" int s2k_engine_reserve(s2k_engine* e, size_t n_items) {
    if (!e) return s2k_fail("s2k_engine_reserve", "null engine");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    const size_t nw = std::min(n_items, RP_CHUNK);
    if (!(engine_workspace(e, n_items * 5400) && engine_rp_slots(e, nw) && engine_ptab(e, nw * RP_MAX_RINGS))) return 0;
    // warm-up: the device's tables (the table of G; secp256k1_generator_h's when the generator-table cache is on) are built now rather than by
    // the first call that needs them
    stream_guard sg(e, e->stream);
    ENGINE_GTAB(e, e->stream);
    {
        std::lock_guard<std::recursive_mutex> pool_lock(e->pool->mu);
        if (e->pool->gen_slots > 0 && e->pool->gen_h == 1) { e->pool->gen_h = 2; (void)gen_cache_build(e, e->stream, k_generator_h, 1); }
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    return 1;
}


// ------------------------------------------------------------------------------------------------------------
// C ABI: engine groups -- the GPUs of one node behind one handle (include/secp256k1_zkp_amd.h, "engine groups")
// ------------------------------------------------------------------------------------------------------------
// One engine per entry of `devices` and one host thread per engine (its device stays current on that thread).  Independent items
// (rangeproofs, signatures) are REPLICA work: a batch is cut into contiguous index ranges, every engine runs its range through its own
// host-buffer entry point -- pinned staging, copies, kernels and results all proceed in parallel on the node's GPUs, no data crosses
// between them.  One large multi-scalar multiplication is sharded by TERMS: every engine sums its slice to a 112-byte Jacobian partial
// (s2k_ecmult_multi_partial_dev), the partials are copied to the first engine's device (hipMemcpyPeerAsync: xGMI when the devices are
// peers) and summed there (s2k_gej_sum_dev) -- EC addition is not a reduction operator of a collective library, and 112 bytes per
// device do not need one.
#include <functional>
struct s2k_group {
    std::vector<s2k_engine*> eng;
    struct worker { std::thread th; std::mutex mu; std::condition_variable cv; std::function<int()> job; int has_job = 0, done = 0, quit = 0, ok = 0, status = 0; std::string err; };
    std::vector<worker*> wk;
    std::mutex call_mu;                    // one group call at a time
    u32* gather = nullptr;                 // on eng[0]'s device: [n][28] Jacobian partials
    u32** partial = nullptr;               // partial[i] on eng[i]'s device: 28 words
    unsigned char* res_xy = nullptr; int32_t* res_inf = nullptr;      // on eng[0]'s device
    hipEvent_t* ev = nullptr;              // ev[i] on eng[i]'s device: partial i has arrived in `gather`
};
static void group_worker_main(s2k_group::worker* w, int device) {
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->quit) return;
        std::function<int()> job = std::move(w->job);
        w->has_job = 0;
        lk.unlock();
        int ok = 0;
        g_last_status = S2K_STATUS_OK; g_last_error.clear();
        try { ok = job(); } catch (const std::exception& ex) { ok = s2k_fail("s2k_group", ex.what()); } catch (...) { ok = s2k_fail("s2k_group", "unexpected exception"); }
        lk.lock();
        w->ok = ok; w->status = g_last_status; w->err = g_last_error; w->done = 1;
        w->cv.notify_all();
    }
}
// runs jobs[i] on worker i (all of them concurrently); 1 when every job returned 1, otherwise the first failure's status and message
static int group_run(s2k_group* g, std::vector<std::function<int()>>& jobs) {
    for (size_t i = 0; i < jobs.size(); i++) {
        auto* w = g->wk[i];
        std::lock_guard<std::mutex> lk(w->mu);
        w->job = std::move(jobs[i]); w->has_job = 1; w->done = 0;
        w->cv.notify_all();
    }
    int ok = 1;
    for (size_t i = 0; i < jobs.size(); i++) {
        auto* w = g->wk[i];
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return w->done != 0; });
        if (!w->ok && ok) { ok = 0; g_last_status = w->status ? w->status : S2K_STATUS_ENGINE_FAILURE; g_last_error = w->err; }
    }
    return ok;
}
extern "C" void s2k_group_destroy(s2k_group* g) {
    if (!g) return;
    for (auto* w : g->wk) {
        { std::lock_guard<std::mutex> lk(w->mu); w->quit = 1; w->cv.notify_all(); }
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    if (!g->eng.empty() && g->eng[0]) {
        (void)hipSetDevice(g->eng[0]->device);
        if (g->gather) hipFree(g->gather);
        if (g->res_xy) hipFree(g->res_xy);
        if (g->res_inf) hipFree(g->res_inf);
    }
    for (size_t i = 0; i < g->eng.size(); i++) {
        if (!g->eng[i]) continue;
        (void)hipSetDevice(g->eng[i]->device);
        if (g->partial && g->partial[i]) hipFree(g->partial[i]);
        if (g->ev && g->ev[i]) hipEventDestroy(g->ev[i]);
        s2k_engine_destroy(g->eng[i]);
    }
    delete[] g->partial; delete[] g->ev;
    delete g;
}
extern "C" s2k_group* s2k_group_create(const int* devices, int n) {
    if (!devices || n <= 0 || n > 64) { s2k_fail_arg("s2k_group_create", "illegal argument"); return nullptr; }
    s2k_group* g = new s2k_group();
    g->partial = new u32*[n](); g->ev = new hipEvent_t[n]();
    for (int i = 0; i < n; i++) {
        s2k_engine* e = s2k_engine_create(devices[i]);
        if (!e) { s2k_group_destroy(g); return nullptr; }
        g->eng.push_back(e);
        if (hipSetDevice(devices[i]) != hipSuccess || hipMalloc((void**)&g->partial[i], 28 * sizeof(u32)) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev[i], hipEventDisableTiming) != hipSuccess) { s2k_fail("s2k_group_create", "device allocation failed"); (void)hipGetLastError(); s2k_group_destroy(g); return nullptr; }
    }
    if (hipSetDevice(devices[0]) != hipSuccess || hipMalloc((void**)&g->gather, (size_t)n * 28 * sizeof(u32)) != hipSuccess ||
        hipMalloc((void**)&g->res_xy, 64) != hipSuccess || hipMalloc((void**)&g->res_inf, 16) != hipSuccess) { s2k_fail("s2k_group_create", "device allocation failed"); (void)hipGetLastError(); s2k_group_destroy(g); return nullptr; }
    // devices that can reach each other directly (xGMI) are made peers, so that the 112-byte partials do not bounce through the host
    for (int i = 1; i < n; i++) {
        int can = 0;
        if (devices[i] != devices[0] && hipDeviceCanAccessPeer(&can, devices[i], devices[0]) == hipSuccess && can) {
            if (hipSetDevice(devices[i]) == hipSuccess) { const hipError_t er = hipDeviceEnablePeerAccess(devices[0], 0); if (er != hipSuccess) (void)hipGetLastError(); }
        } else (void)hipGetLastError();
    }
    try {
        for (int i = 0; i < n; i++) { auto* w = new s2k_group::worker(); g->wk.push_back(w); w->th = std::thread(group_worker_main, w, devices[i]); }
    } catch (...) { s2k_fail("s2k_group_create", "cannot start worker threads"); s2k_group_destroy(g); return nullptr; }
    return g;
}
extern "C" int s2k_group_size(const s2k_group* g) { return g ? (int)g->eng.size() : 0; }
extern "C" s2k_engine* s2k_group_engine(s2k_group* g, int i) { return (g && i >= 0 && (size_t)i < g->eng.size()) ? g->eng[i] : nullptr; }
// share i of n items over k engines: [lo, hi)
static inline void group_share(size_t n, size_t k, size_t i, size_t& lo, size_t& hi) { lo = n * i / k; hi = n * (i + 1) / k; }

extern "C" int secp256k1_rangeproof_verify_batch_group(s2k_group* g, int32_t* results, uint64_t* min_value, uint64_t* max_value, const unsigned char* commits33,
                                                       const unsigned char* proofs, const uint64_t* proof_off, const unsigned char* extra, const uint64_t* extra_off,
                                                       const unsigned char* gens64, size_t n) {
    const char* who = "secp256k1_rangeproof_verify_batch_group";
    if (!g || g->eng.empty()) return s2k_fail(who, "null group");
    if (n == 0) return 1;
    if (!results || !min_value || !max_value || !commits33 || !proofs || !proof_off || !gens64) return s2k_fail_arg(who, "illegal argument (ARG_CHECK)");
    std::lock_guard<std::mutex> call(g->call_mu);
    memset(results, 0, sizeof(int32_t) * n);
    const size_t k = g->eng.size();
    std::vector<std::function<int()>> jobs(k);
    for (size_t i = 0; i < k; i++) {
        size_t lo, hi; group_share(n, k, i, lo, hi);
        s2k_engine* e = g->eng[i];
        jobs[i] = [=]() -> int {
            if (hi == lo) return 1;
            const size_t m = hi - lo;
            std::vector<uint64_t> po(m + 1), eo;
            for (size_t t = 0; t <= m; t++) po[t] = proof_off[lo + t] - proof_off[lo];
            const int has_extra = extra && extra_off;
            if (has_extra) { eo.resize(m + 1); for (size_t t = 0; t <= m; t++) eo[t] = extra_off[lo + t] - extra_off[lo]; }
            return secp256k1_rangeproof_verify_batch(e, results + lo, min_value + lo, max_value + lo, commits33 + 33 * lo, proofs + proof_off[lo], po.data(),
                                                     has_extra ? extra + extra_off[lo] : nullptr, has_extra ? eo.data() : nullptr, gens64 + 64 * lo, m);
        };
    }
    const int ok = group_run(g, jobs);
    if (!ok) memset(results, 0, sizeof(int32_t) * n);                  // an engine failure never leaves part of a batch marked valid
    return ok;
}
extern "C" int secp256k1_rangeproof_verify_batch_ptrs_group(s2k_group* g, int32_t* results, uint64_t* min_value, uint64_t* max_value, const void* const* commit_objs,
                                                            const unsigned char* const* proofs, const size_t* plens, const unsigned char* const* extra, const size_t* elens,
                                                            const void* const* gen_objs, size_t n) {
    const char* who = "secp256k1_rangeproof_verify_batch_ptrs_group";
    if (!g || g->eng.empty()) return s2k_fail(who, "null group");
    if (n == 0) return 1;
    if (!rp_ptrs_check(who, results, min_value, max_value, commit_objs, proofs, plens, extra, elens, gen_objs, n)) return 0;
    std::lock_guard<std::mutex> call(g->call_mu);
    memset(results, 0, sizeof(int32_t) * n);
    const size_t k = g->eng.size();
    std::vector<std::function<int()>> jobs(k);
    for (size_t i = 0; i < k; i++) {
        size_t lo, hi; group_share(n, k, i, lo, hi);
        s2k_engine* e = g->eng[i];
        jobs[i] = [=]() -> int {
            if (hi == lo) return 1;
            return secp256k1_rangeproof_verify_batch_ptrs(e, results + lo, min_value + lo, max_value + lo, commit_objs + lo, proofs + lo, plens + lo, extra ? extra + lo : nullptr,
                                                          extra ? elens + lo : nullptr, gen_objs + lo, hi - lo);
        };
    }
    const int ok = group_run(g, jobs);
    if (!ok) memset(results, 0, sizeof(int32_t) * n);
    return ok;
}
extern "C" int secp256k1_schnorrsig_verify_batch_group(s2k_group* g, int32_t* results, const unsigned char* sigs, const unsigned char* msgs, size_t msglen,
                                                       const unsigned char* pubkeys, int pk_format, size_t n) {
    const char* who = "secp256k1_schnorrsig_verify_batch_group";
    if (!g || g->eng.empty()) return s2k_fail(who, "null group");
    if (n == 0) return 1;
    if (!results || !sigs || (!msgs && msglen) || !pubkeys) return s2k_fail_arg(who, "illegal argument (ARG_CHECK)");
    std::lock_guard<std::mutex> call(g->call_mu);
    memset(results, 0, sizeof(int32_t) * n);
    const size_t k = g->eng.size(), pkb = pk_format ? 64 : 32;
    std::vector<std::function<int()>> jobs(k);
    for (size_t i = 0; i < k; i++) {
        size_t lo, hi; group_share(n, k, i, lo, hi);
        s2k_engine* e = g->eng[i];
        jobs[i] = [=]() -> int {
            if (hi == lo) return 1;
            return secp256k1_schnorrsig_verify_batch(e, results + lo, sigs + 64 * lo, msgs ? msgs + msglen * lo : nullptr, msglen, pubkeys + pkb * lo, pk_format, hi - lo);
        };
    }
    const int ok = group_run(g, jobs);
    if (!ok) memset(results, 0, sizeof(int32_t) * n);
    return ok;
}
// One sum over the group.  Per engine i: its slice's scalars / points (device memory of engine i's GPU when `resident`, host memory
// otherwise); the generator term goes with slice 0.  The result comes back to the host (r_xy 64 bytes, *r_inf).
static int group_msm(s2k_group* g, const char* who, unsigned char* r_xy, int32_t* r_inf, const unsigned char* g_sc, const unsigned char* const* sc,
                     const unsigned char* const* pt, const unsigned char* const* pt_inf, const size_t* cnt, int resident) {
    const size_t k = g->eng.size();
    std::lock_guard<std::mutex> call(g->call_mu);
    std::vector<std::function<int()>> jobs(k);
    const int dev0 = g->eng[0]->device;
    for (size_t i = 0; i < k; i++) {
        s2k_engine* e = g->eng[i];
        u32* part = g->partial[i]; u32* dst = g->gather + 28 * i; hipEvent_t ev = g->ev[i];
        const unsigned char* sci = sc[i]; const unsigned char* pti = pt[i]; const unsigned char* infi = pt_inf ? pt_inf[i] : nullptr; const size_t m = cnt[i];
        const unsigned char* gs = i == 0 ? g_sc : nullptr;
        jobs[i] = [=]() -> int {
            std::lock_guard<std::recursive_mutex> lock(e->mu);
            HIPCHK(hipSetDevice(e->device));
            hipStream_t st = e->stream;
            const unsigned char *d_sc = sci, *d_pt = pti, *d_inf = infi, *d_g = gs;
            if (!resident) {
                // slice to HBM: behind the MSM's own workspace need (s2k_ecmult_multi_partial_dev carves from the start)
                const size_t nt = m + (gs ? 1 : 0);
                const size_t base = ws_need({28 * 4}) + msm_ws_bytes(e, nt + 1, engine_msm_plan(e, nt ? nt : 1));
                if (!engine_workspace(e, base + ws_need({32 * m + 64, 64 * m + 64, m + 64, 64}))) return 0;
                ws_carver c{e->ws, base};
                unsigned char* a = c.take<unsigned char>(32 * m + 64); unsigned char* b = c.take<unsigned char>(64 * m + 64); unsigned char* ci = c.take<unsigned char>(m + 64);
                unsigned char* dg = c.take<unsigned char>(64);
                stream_guard sg(e, st);
                if (m) { HIPCHK(hipMemcpyAsync(a, sci, 32 * m, hipMemcpyHostToDevice, st)); HIPCHK(hipMemcpyAsync(b, pti, 64 * m, hipMemcpyHostToDevice, st)); }
                if (m && infi) HIPCHK(hipMemcpyAsync(ci, infi, m, hipMemcpyHostToDevice, st));
                if (gs) HIPCHK(hipMemcpyAsync(dg, gs, 32, hipMemcpyHostToDevice, st));
                d_sc = a; d_pt = b; d_inf = infi ? ci : nullptr; d_g = gs ? dg : nullptr;
            }
            if (!s2k_ecmult_multi_partial_dev(e, nullptr, part, d_g, d_sc, d_pt, d_inf, m)) return 0;
            HIPCHK(hipMemcpyPeerAsync(dst, dev0, part, e->device, 28 * sizeof(u32), st));
            HIPCHK(hipEventRecord(ev, st));
            return 1;
        };
    }
    if (!group_run(g, jobs)) return 0;
    (void)who;
    s2k_engine* e0 = g->eng[0];
    std::lock_guard<std::recursive_mutex> lock(e0->mu);
    HIPCHK(hipSetDevice(e0->device));
    for (size_t i = 0; i < k; i++) HIPCHK(hipStreamWaitEvent(e0->stream, g->ev[i], 0));
    if (!s2k_gej_sum_dev(e0, nullptr, g->res_xy, g->res_inf, g->gather, k)) return 0;
    HIPCHK(hipMemcpyAsync(r_xy, g->res_xy, 64, hipMemcpyDeviceToHost, e0->stream));
    HIPCHK(hipMemcpyAsync(r_inf, g->res_inf, 4, hipMemcpyDeviceToHost, e0->stream));
    HIPCHK(hipStreamSynchronize(e0->stream));
    return 1;
}
extern "C" int s2k_ecmult_multi_group(s2k_group* g, unsigned char* r_xy, int32_t* r_inf, const unsigned char* g_sc, const unsigned char* sc,
                                      const unsigned char* pt_xy, const unsigned char* pt_inf, size_t n) {
    const char* who = "s2k_ecmult_multi_group";
    if (!g || g->eng.empty()) return s2k_fail(who, "null group");
    if (!r_xy || !r_inf || (n && (!sc || !pt_xy))) return s2k_fail_arg(who, "illegal argument (ARG_CHECK)");
    const size_t k = g->eng.size();
    std::vector<const unsigned char*> a(k), b(k), c(k); std::vector<size_t> cnt(k);
    for (size_t i = 0; i < k; i++) { size_t lo, hi; group_share(n, k, i, lo, hi); a[i] = sc + 32 * lo; b[i] = pt_xy + 64 * lo; c[i] = pt_inf ? pt_inf + lo : nullptr; cnt[i] = hi - lo; }
    return group_msm(g, who, r_xy, r_inf, g_sc, a.data(), b.data(), pt_inf ? c.data() : nullptr, cnt.data(), 0);
}
extern "C" int s2k_ecmult_multi_group_dev(s2k_group* g, unsigned char* r_xy, int32_t* r_inf, const unsigned char* g_sc_dev0, const unsigned char* const* sc_dev,
                                          const unsigned char* const* pt_xy_dev, const unsigned char* const* pt_inf_dev, const size_t* n_per_engine) {
    const char* who = "s2k_ecmult_multi_group_dev";
    if (!g || g->eng.empty()) return s2k_fail(who, "null group");
    if (!r_xy || !r_inf || !sc_dev || !pt_xy_dev || !n_per_engine) return s2k_fail_arg(who, "illegal argument (ARG_CHECK)");
    return group_msm(g, who, r_xy, r_inf, g_sc_dev0, sc_dev, pt_xy_dev, pt_inf_dev, n_per_engine, 1);
}
