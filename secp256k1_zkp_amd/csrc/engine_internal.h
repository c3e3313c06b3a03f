// engine_internal.h -- what the translation units of the gfx950 batch-verification engine share (see include/secp256k1_zkp_amd.h):
// the engine object, the per-device table pool, error plumbing, workspace carving and the host functions one kernel family offers the others.
//
// The library is five translation units, one per kernel family, compiled separately (hipcc --offload-arch=gfx950 -O3 -fPIC -c) and linked
// into one shared object (see __graft_entry__.build):
//   engine_core.hip        table construction, the per-device pool and generator-table cache, engine lifecycle / options / groups,
//                          s2k_ecmult_batch, BIP-340, the single-item `_amd` forms
//   engine_rangeproof.hip  Borromean rangeproof verification + rewind (kernels, pipeline, host staging), surjection proofs
//   engine_msm.hip         multi-scalar multiplication (binning, partial-sum rounds, Horner, exact path), point sums, Pedersen tallies
//   engine_bppp.hip        Bulletproofs++ norm argument verification and bppp_commit
//   engine_halfagg.hip     half-aggregated Schnorr verification
// The per-lane arithmetic lives in the headers next to these files.  No device function is called across translation units (no -fgpu-rdc):
// a family that needs another family's kernels calls the HOST function that launches them.
// There is no CPU implementation behind the entry points: without a HIP device every call fails loudly.
#pragma once
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
// (engine_core.hip holds the per-thread message and status: S2K_STATUS_* of the most recent failing call on this thread)
int s2k_fail(const char* what, const char* detail);
int s2k_fail_busy(const char* what, const char* detail);
int s2k_fail_arg(const char* what, const char* detail);
#define HIPCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) return s2k_fail(#call, hipGetErrorString(_e)); } while (0)
#define HIPCHK_NULL(call) do { hipError_t _e = (call); if (_e != hipSuccess) { s2k_fail(#call, hipGetErrorString(_e)); return nullptr; } } while (0)


// ------------------------------------------------------------------------------------------------------------
// engine object
// ------------------------------------------------------------------------------------------------------------
struct s2k_dev_pool;
struct s2k_engine {
    int device;
    hipStream_t stream;
    u32* gtab;                 // the device pool's generator table once a call of this engine has needed it (engine_gtab)
    unsigned char* ws;         // growable HBM workspace
    size_t ws_bytes;
    u32* ptab;                 // per-lane odd-multiples tables (S2K_PTAB_WORDS words per lane), grown on demand
    size_t ptab_lanes;
    hipEvent_t ev[4];          // [0],[1] whole call; [2],[3] dominant kernel
    hipStream_t stream2;       // side stream for kernels that can run next to the main sequence
    hipEvent_t ev_fork, ev_join;
    schnorr_midstate bip340;   // tagged-hash midstate, computed once on the host
    size_t max_lanes;          // lanes per launch (multiple of 256)
    int rp_split;              // rangeproof rings use the two-piece double multiplication (ecmult_lane_split); S2K_OPT_RP_SPLIT = 0 turns it off (an option: product builds read no such variable from the environment)
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
    u32* bp_tab; size_t bp_tab_bytes, bp_tab_stride;   // the BP++ generator set's fixed-base tables (bppp.h: one table of G's format per generator, bp_tab_stride words apart), kept across calls
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
    struct { int c, T, chunk, two_pass, one_pass, bin_plain, no_small, T2, old_tail, slice_r, slice_lds, slice_maxc, run_major; } msm_diag;
    size_t msm_max_terms_opt;  // S2K_OPT_MSM_MAX_TERMS: sums with more terms go as several launches whose partial sums add (0: the 32-bit reference limit)
    u32* ha_pin; size_t ha_pin_words;    // pinned: the chain states of the half-aggregate randomizer hash, walked on the host (host_sha256.h)
    // Buffers this engine has outgrown.  Growing a buffer never waits for the device (round 6; it used to be hipDeviceSynchronize() + hipFree
    // under the engine's lock, i.e. a stall for the longest stream of EVERY engine on the GPU): the new buffer is allocated beside the old
    // one, which may still be read by launches in flight and is only handed back when the engine is known to be idle -- at destruction, at
    // s2k_engine_reserve, or when an allocation fails (engine_make_room: waits for THIS engine's streams only).
    std::vector<void*> retired_dev, retired_host; size_t retired_bytes;
    hipEvent_t ev_gen_read[RP_GEN_SLOTS];   // recorded behind the last kernel of a rangeproof call that read generator table i (see s2k_dev_pool::gen_slot::readers)
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
// per-lane table scratch / rings arena / workspace of an engine, grown on demand (engine_core.hip)
int engine_ptab(s2k_engine* e, size_t lanes);
// growth without a device-wide wait (see s2k_engine::retired_dev)
int engine_grow_dev(s2k_engine* e, void** buf, size_t* have, size_t need, size_t unit);
void engine_retire_dev(s2k_engine* e, void* p, size_t bytes);
void engine_retire_host(s2k_engine* e, void* p);
int engine_make_room(s2k_engine* e);
void gen_note_read(s2k_engine* e, hipStream_t st, u32 valid_mask);
int engine_rtab(s2k_engine* e, size_t rings);
int engine_workspace(s2k_engine* e, size_t bytes);

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

// stage host buffers through the workspace: small helper that carves 256-byte aligned pieces
struct ws_carver {
    unsigned char* base; size_t off;
    template <class T> T* take(size_t count) {
        off = (off + 255) & ~size_t(255);
        T* p = (T*)(base + off); off += count * sizeof(T); return p;
    }
};
static inline size_t ws_need(std::initializer_list<size_t> sizes) {
    size_t t = 0; for (size_t s : sizes) t = ((t + 255) & ~size_t(255)) + s; return t + 256;
}


static const unsigned char k_generator_h[64] = {
    0x50, 0x92, 0x9b, 0x74, 0xc1, 0xa0, 0x49, 0x54, 0xb7, 0x8b, 0x4b, 0x60, 0x35, 0xe9, 0x7a, 0x5e, 0x07, 0x8a, 0x5a, 0x0f, 0x28, 0xec, 0x96, 0xd5, 0x47, 0xbf, 0xee, 0x9a, 0xce, 0x80, 0x3a, 0xc0,
    0x31, 0xd3, 0xc6, 0x86, 0x39, 0x73, 0x92, 0x6e, 0x04, 0x9e, 0x63, 0x7c, 0xb1, 0xb5, 0xf4, 0x0a, 0x36, 0xda, 0xc2, 0x8a, 0xf1, 0x76, 0x69, 0x68, 0xc3, 0x0c, 0x23, 0x13, 0xf3, 0xa3, 0x89, 0x04};
struct s2k_dev_pool {
    int device; int refs;
    std::recursive_mutex mu;
    u32* gtab; hipEvent_t ev_gtab; int gtab_state;           // 0: not built, 1: build queued (ev_gtab behind it), 2: known to be complete
    hipEvent_t ev_build[2];                                   // around the construction kernels of the table of G (s2k_engine_gtable_build_ms)
    u32 gtab_bits, gtab_bits_wanted;                          // digit width of the device's tables: wanted ($S2K_GTAB_BITS, default 26) until the table of G exists, then what fitted (back to wanted when the tables are freed)
    // Fixed-base tables of rangeproof generators: a small cache keyed by the 64 generator bytes.  Slot tables have the layout of gtab
    // (allocated when a slot is first used and then reused by whatever generator takes the slot); xmul is the x-table of the ring-base
    // multiples (RP_XMUL_WORDS).  gen_keys (device) is what k_rp_header matches a proof's generator against; gen_seen counts the VERIFIED
    // proofs met per uncached generator (k_rp_final reports them through each engine's device mailbox, read at that engine's next call)
    // and a generator is built once it reaches gen_min.  pinned: secp256k1_generator_h and generators cached explicitly -- an automatic
    // build never evicts those.
    // readers: the engines whose kernels may still read this slot's tables (each records its ev_gen_read[slot] behind the last kernel of
    // every call that took the slot into its view, pool mutex held from the view to the record): a build that REUSES the slot's memory makes
    // its stream wait for exactly those events -- no host wait, no device-wide synchronisation (round 6).
    struct gen_slot { unsigned char key[64]; u32* tab; u32* xmul; unsigned long long stamp; int valid; int pinned; hipEvent_t ev_ready; int done; std::vector<s2k_engine*> readers; } gen[RP_GEN_SLOTS];
    int gen_slots; unsigned long long gen_clock; size_t gen_min; int gen_h;
    unsigned char* gen_keys;   // device, [RP_GEN_SLOTS][64]
    std::vector<std::pair<std::array<unsigned char, 64>, size_t>> gen_seen;
};

// ---- the device's tables and the generator-table cache (engine_core.hip) ---------------------------------------------------------------
const u32* engine_gtab(s2k_engine* e, hipStream_t st);
#define ENGINE_GTAB(e, st) do { if (!((e)->gtab = const_cast<u32*>(engine_gtab((e), (st))))) return 0; } while (0)
rp_gen_dev gen_dev_view(s2k_engine* e, hipStream_t st, hipStream_t sp);
int gen_cache_find(s2k_dev_pool* p, const unsigned char* key);
int gen_cache_build(s2k_engine* e, hipStream_t st, const unsigned char* key, int pinned);
void gen_cache_service(s2k_engine* e, hipStream_t st);
void gen_cache_collect(s2k_engine* e, hipStream_t st);
// ---- rangeproof family (engine_rangeproof.hip) --------------------------------------------------------------------------------------------
int engine_rp_slots(s2k_engine* e, size_t nw);
int rp_ptrs_check(const char* who, int32_t* results, uint64_t* min_value, uint64_t* max_value, const void* const* commit_objs, const unsigned char* const* proofs,
                  const size_t* plens, const unsigned char* const* extra, const size_t* elens, const void* const* gen_objs, size_t n);
// ---- multi-scalar multiplication family (engine_msm.hip) ----------------------------------------------------------------------------------
// side: where the gated exact path runs (with its fork / join events); arena: which MSM_DIRECT_LANES-sized region of the engine's table
// arena its lanes use (0: the engine's own calls; 1, 2: the two pipelined slots)
struct msm_ctx { hipStream_t side; hipEvent_t fork, join; unsigned arena; };
// where the last kernel of a launch also writes the result: the affine point (r_xy 64 bytes, r_inf) and / or a copy of the Jacobian record
struct msm_out { unsigned char* r_xy; int32_t* r_inf; u32* out28; };
size_t msm_max_terms(const s2k_engine* e);
msm_plan engine_msm_plan(const s2k_engine* e, size_t nt);
size_t msm_ws_bytes(const s2k_engine* e, size_t nt, const msm_plan& pl);
// core of every MSM entry point: leaves the Jacobian result (28 words) at *result28 (device), stream-ordered, nothing read back
int msm_launch(s2k_engine* e, hipStream_t st, ws_carver& c, u32** result28, const unsigned char* g_sc, const unsigned char* sc,
               const unsigned char* pt, const unsigned char* pt_inf, size_t n, u32 part = 0, u32 parts = 1, const msm_ctx* ctx = nullptr, const msm_out* out = nullptr);
// segmented tree sum of Jacobian records, ping-ponging between two scratch buffers; returns where the nseg results are
const u32* launch_gej_reduce(hipStream_t st, const u32* in, u32* bufA, u32* bufB, u32 nseg, u32 seg_len, const u32* gate = nullptr);
void launch_set_word(hipStream_t st, u32* p, u32 v);
