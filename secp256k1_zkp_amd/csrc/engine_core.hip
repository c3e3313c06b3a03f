#include "engine_internal.h"

// ------------------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
static thread_local int g_last_status = 0;            // S2K_STATUS_* of the most recent failing call on this thread
int s2k_fail(const char* what, const char* detail) {
    g_last_error = std::string(what) + ": " + (detail ? detail : "");
    g_last_status = S2K_STATUS_ENGINE_FAILURE;
    return 0;
}
int s2k_fail_busy(const char* what, const char* detail) {
    g_last_error = std::string(what) + ": " + (detail ? detail : "");
    g_last_status = S2K_STATUS_BUSY;
    return 0;
}
int s2k_fail_arg(const char* what, const char* detail) {
    g_last_error = std::string(what) + ": " + (detail ? detail : "");
    g_last_status = S2K_STATUS_ILLEGAL_ARGUMENT;
    return 0;
}
extern "C" const char* s2k_last_error(void) { return g_last_error.c_str(); }
extern "C" int s2k_last_status(void) { return g_last_status; }
extern "C" void s2k_clear_status(void) { g_last_status = S2K_STATUS_OK; g_last_error.clear(); }

// ------------------------------------------------------------------------------------------------------------
// engine scratch
// ------------------------------------------------------------------------------------------------------------
// ---- growth without a device-wide wait ------------------------------------------------------------------------------------------------
void engine_retire_dev(s2k_engine* e, void* p, size_t bytes) { if (p) { e->retired_dev.push_back(p); e->retired_bytes += bytes; } }
void engine_retire_host(s2k_engine* e, void* p) { if (p) e->retired_host.push_back(p); }
// waits for the work of THIS engine only -- its own streams and, through the event every entry point leaves behind, the caller streams its
// `_dev` calls were issued on -- and hands the outgrown buffers back
int engine_make_room(s2k_engine* e) {
    hipStream_t own[] = {e->stream, e->stream2, e->stream_pre, e->stream_copy, e->msm_slot[0].s, e->msm_slot[0].s2, e->msm_slot[1].s, e->msm_slot[1].s2};
    for (hipStream_t s : own) if (s) HIPCHK(hipStreamSynchronize(s));
    if (e->last_stream_valid) HIPCHK(hipEventSynchronize(e->ev_last));
    for (void* p : e->retired_dev) (void)hipFree(p);
    for (void* p : e->retired_host) (void)hipHostFree(p);
    e->retired_dev.clear(); e->retired_host.clear(); e->retired_bytes = 0;
    (void)hipGetLastError();
    return 1;
}
// *buf (of *have bytes) becomes at least `need` bytes: a new allocation of max(need, 1.5 x have) beside the old buffer, which is retired, not
// freed (launches in flight may still use it).  Only when there is no room beside it does the call wait -- for this engine's work alone --
// give everything outgrown back and take exactly what it needs.
#define S2K_RETIRED_MAX (size_t(16) << 30)
int engine_grow_dev(s2k_engine* e, void** buf, size_t* have, size_t need, size_t unit) {
    if (need <= *have) return 1;
    auto round = [&](size_t b) { return ((b + unit - 1) / unit) * unit; };          // (`unit` need not be a power of two: the per-lane table arena's is not)
    size_t want = round(std::max(need, *have + *have / 2));
    void* p = nullptr;
    if (e->retired_bytes + *have > S2K_RETIRED_MAX || hipMalloc(&p, want) != hipSuccess) {
        (void)hipGetLastError(); p = nullptr;
        if (!engine_make_room(e)) return 0;
        if (*buf) (void)hipFree(*buf);
        *buf = nullptr; *have = 0;
        want = round(need);
        HIPCHK(hipMalloc(&p, want));
    } else engine_retire_dev(e, *buf, *have);
    *buf = p; *have = want;
    return 1;
}
// per-lane table scratch for `lanes` concurrent ecmult_lane callers (lane = global thread index of the launch)
int engine_ptab(s2k_engine* e, size_t lanes) {
    lanes = (lanes + 255) & ~size_t(255);
    if (lanes <= e->ptab_lanes) return 1;
    size_t have = e->ptab_lanes * S2K_PTAB_WORDS * sizeof(u32); void* buf = e->ptab;
    const size_t per = 256 * S2K_PTAB_WORDS * sizeof(u32);
    if (!engine_grow_dev(e, &buf, &have, lanes * S2K_PTAB_WORDS * sizeof(u32), per)) { e->ptab = (u32*)buf; e->ptab_lanes = (have / per) * 256; return 0; }
    e->ptab = (u32*)buf; e->ptab_lanes = (have / per) * 256;
    return 1;
}
// the arena of the rings kernels for `rings` rings: the general form wants S2K_PTAB_WORDS per ring; the shared form, with S2K_RP_K rings per
// lane, S2K_RTAB_WORDS per ring plus per wavefront of 64 lanes the construction's parking area and the lanes' point / challenge parking
int engine_rtab(s2k_engine* e, size_t rings) {
    rings = (rings + 255) & ~size_t(255);
    const size_t lanes = ((rings + S2K_RP_K - 1) / S2K_RP_K + 255) & ~size_t(255);
    const size_t words = lanes * S2K_RP_K * S2K_RTAB_WORDS + (lanes / 64) * (S2K_RRAW_WAVE_WORDS + (size_t)S2K_RP_K * RP_PARK_WORDS * 64);
    return engine_ptab(e, std::max(rings, (words + S2K_PTAB_WORDS - 1) / S2K_PTAB_WORDS));
}
// Upper bound on lanes per launch: keeps the per-lane table arena at 1.2 GB however large the batch is; bigger
// batches run as several launches over sub-ranges (same stream, so the order of results is unaffected).
// (engine field max_lanes; default 2^20, $S2K_MAX_LANES overrides it -- the tests use a small value to exercise the split)
int engine_workspace(s2k_engine* e, size_t bytes) {
    if (bytes <= e->ws_bytes) return 1;
    void* buf = e->ws; size_t have = e->ws_bytes;
    const int ok = engine_grow_dev(e, &buf, &have, bytes, size_t(1) << 20);
    e->ws = (unsigned char*)buf; e->ws_bytes = have;
    return ok;
}

// ------------------------------------------------------------------------------------------------------------
// generator table construction (engine creation)
// ------------------------------------------------------------------------------------------------------------
// (gtable.h: window bases -> seeds -> one affine addition per remaining entry with a shared inversion per run of rows)
__global__ void k_gtab_base(u32* gtab, u32 D) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w == 0) gtab_write_header(gtab, D);
    if (w < gtab_windows_for(D)) gtab_build_base(gtab, D, w);
}
__global__ void __launch_bounds__(256)
k_gtab_seeds(u32* gtab, gtab_fill_plan p) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x, per = gtab_seeds_per_window(p);
    const u32 w = t / per;
    if (w < p.W) gtab_build_seed(gtab, p, w, t % per);
}
// block = 256 consecutive columns of one (window, run of rows)
__global__ void __launch_bounds__(256, 2)
k_gtab_fill(u32* gtab, gtab_fill_plan p, u32 runs) {
    const u32 b = 1u + blockIdx.x * 256u + threadIdx.x, run = blockIdx.y % runs, w = blockIdx.y / runs;
    if (b <= gtab_fill_cols(p)) gtab_fill_run(gtab, p, w, b, 1u + run * GTAB_FILL_RUN);
}
// the three launches; `base_done`: the window bases (and the header) are already there (a generator's table: k_gen_base wrote them)
static int launch_table_build(hipStream_t st, u32* tab, u32 D, int base_done) {
    const gtab_fill_plan p = gtab_make_fill_plan(D);
    // the top window only has entries in its first top_rows rows: the slots behind them (never addressed by a digit) are cleared, so that the
    // table's bytes are defined -- s2k_engine_gtable() shows the whole allocation, and a prefetch that strays there finds zeros, not stale HBM
    {   const size_t first = gtab_slot(D, p.W - 1, 0) + (size_t)p.top_rows * p.Kc, end = gtab_slots_for(D);
        if (end > first && hipMemsetAsync(tab + first * S2K_GTAB_ENTRY_WORDS, 0, (end - first) * S2K_GTAB_ENTRY_WORDS * sizeof(u32), st) != hipSuccess) return 0; }
    if (!base_done) hipLaunchKernelGGL(k_gtab_base, dim3(1), dim3(64), 0, st, tab, D);
    const u32 seeds = p.W * gtab_seeds_per_window(p);
    hipLaunchKernelGGL(k_gtab_seeds, dim3((seeds + 255) / 256), dim3(256), 0, st, tab, p);
    const u32 runs = gtab_fill_runs(p);
    hipLaunchKernelGGL(k_gtab_fill, dim3((gtab_fill_cols(p) + 255) / 256, p.W * runs), dim3(256), 0, st, tab, p, runs);
    return hipGetLastError() == hipSuccess;
}

// fixed-base table of another point than G (a rangeproof generator): window bases from the 64 generator bytes, then k_gtab_entries
__global__ void k_gen_base(u32* tab, const unsigned char* gen64, u32 D) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w == 0) gtab_write_header(tab, D);
    if (w >= gtab_windows_for(D)) return;
    ge g; rp_load_generator(g, gen64);
    gtab_build_base(tab, D, w, &g);
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
// takes its view of the cache AND enqueues the kernels that use it AND records, per slot in the view, an event behind the last of them
// (gen_note_read); a slot's memory is only ever rewritten (eviction) or freed (fewer slots) behind the events of the engines that read
// it, so no kernel in flight can read a table that is being replaced -- and nobody waits for engines that never touched the slot.
// Lock order: engine mutex, then pool mutex.
// secp256k1_generator_h (src/modules/generator/main_impl.h:30-35): the generator of bench_rangeproof and of every non-asset caller
static std::mutex g_pools_mu;
static std::vector<s2k_dev_pool*> g_pools;
static void pool_free_tables(s2k_dev_pool* p) {
    if (p->gtab) hipFree(p->gtab);
    p->gtab = nullptr; p->gtab_state = 0; p->gtab_bits = p->gtab_bits_wanted;
    for (int i = 0; i < RP_GEN_SLOTS; i++) {
        p->gen[i].readers.clear();
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
    p->gtab_bits = S2K_GTAB_BITS;
    if (const char* gb = getenv("S2K_GTAB_BITS")) { const int v = atoi(gb); if (v >= 20 && v <= S2K_GTAB_MAX_BITS && gtab_bits_ok((u32)v)) p->gtab_bits = (u32)v; }
    p->gtab_bits_wanted = p->gtab_bits;
    if (const char* gs = getenv("S2K_GEN_CACHE")) { const int v = atoi(gs); p->gen_slots = v < 0 ? 0 : (v > RP_GEN_SLOTS ? RP_GEN_SLOTS : v); }
    if (const char* gm = getenv("S2K_GEN_CACHE_MIN")) p->gen_min = (size_t)strtoull(gm, nullptr, 10);
#ifdef S2K_DIAG
    if (const char* gh = getenv("S2K_GEN_CACHE_H")) p->gen_h = atoi(gh) != 0;
#endif
    int ok = hipEventCreateWithFlags(&p->ev_gtab, hipEventDisableTiming) == hipSuccess;
    p->ev_build[0] = p->ev_build[1] = nullptr;
    ok = ok && hipEventCreate(&p->ev_build[0]) == hipSuccess && hipEventCreate(&p->ev_build[1]) == hipSuccess;
    for (int i = 0; ok && i < RP_GEN_SLOTS; i++) ok = hipEventCreateWithFlags(&p->gen[i].ev_ready, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc((void**)&p->gen_keys, 64 * RP_GEN_SLOTS) == hipSuccess && hipMemset(p->gen_keys, 0, 64 * RP_GEN_SLOTS) == hipSuccess;
    if (!ok) {
        s2k_fail("s2k_engine_create", "cannot create the device's table pool");
        (void)hipGetLastError();
        if (p->ev_gtab) hipEventDestroy(p->ev_gtab);
        for (int i = 0; i < 2; i++) if (p->ev_build[i]) hipEventDestroy(p->ev_build[i]);
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
    for (int i = 0; i < 2; i++) if (p->ev_build[i]) hipEventDestroy(p->ev_build[i]);
    for (int i = 0; i < RP_GEN_SLOTS; i++) if (p->gen[i].ev_ready) hipEventDestroy(p->gen[i].ev_ready);
    delete p;
}
// The table of G, built by the first call on this device that needs it (stream-ordered on that call's stream); every later user's
// stream waits for the build's event until it is known to be over.  Returns the table or nullptr (no memory: s2k_fail was called).
const u32* engine_gtab(s2k_engine* e, hipStream_t st) {
    s2k_dev_pool* p = e->pool;
    std::lock_guard<std::recursive_mutex> lock(p->mu);
    if (p->gtab_state == 2) return p->gtab;
    if (p->gtab_state == 0) {
        // The widest table the device has room for, from the wanted width down (26 bits = 21.5 GB, 24 = 5.9 GB, 22 = 1.6 GB, 20 = 0.44 GB):
        // a partitioned or shared GPU still gets an engine, with one more addition per fixed-base multiplication for every step down.
        // A width is only taken when the table leaves headroom: a device with 22-24 GB free would get the 21.5 GB table and then fail every
        // call for want of workspace, where the 5.9 GB one works.  Headroom: the workspace this engine has reserved or 4 GB, whichever is
        // larger, beside the table (generator-slot tables are optional: a slot that finds no memory leaves its proofs on the general form).
        p->gtab = nullptr;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = ~size_t(0); }
        const size_t headroom = std::max(size_t(4) << 30, e->ws_bytes / 2);
        for (u32 D = p->gtab_bits_wanted; D >= 20u; D -= 2u) {
            if (!gtab_bits_ok(D)) continue;
            const size_t bytes = sizeof(u32) * gtab_words_for(D);
            if (D > 20u && (free_b < bytes || free_b - bytes < headroom)) continue;
            if (hipMalloc((void**)&p->gtab, bytes) == hipSuccess) { p->gtab_bits = D; break; }
            (void)hipGetLastError(); p->gtab = nullptr;
        }
        if (!p->gtab) { s2k_fail("engine_gtab", "no memory for the generator table (0.44 GB of HBM at the narrowest width)"); return nullptr; }
        const int ok = hipEventRecord(p->ev_build[0], st) == hipSuccess && launch_table_build(st, p->gtab, p->gtab_bits, 0) &&
                       hipEventRecord(p->ev_build[1], st) == hipSuccess && hipEventRecord(p->ev_gtab, st) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); hipFree(p->gtab); p->gtab = nullptr; s2k_fail("engine_gtab", "generator table build failed"); return nullptr; }
        p->gtab_state = 1;
        return p->gtab;
    }
    if (hipEventQuery(p->ev_gtab) == hipSuccess) { p->gtab_state = 2; return p->gtab; }
    (void)hipGetLastError();
    if (hipStreamWaitEvent(st, p->ev_gtab, 0) != hipSuccess) { (void)hipGetLastError(); s2k_fail("engine_gtab", "hipStreamWaitEvent failed"); return nullptr; }
    return p->gtab;
}
// The cache as the kernels of one launch see it; `st` / `sp`: the streams that will read the tables (made to wait for builds still in flight)
rp_gen_dev gen_dev_view(s2k_engine* e, hipStream_t st, hipStream_t sp) {
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
// End of a call whose kernels read the tables of the slots in `valid_mask` (pool mutex still held since the view was taken): one event per
// slot behind the call's last kernel on `st`, which every side stream of the call has been joined into
void gen_note_read(s2k_engine* e, hipStream_t st, u32 valid_mask) {
    s2k_dev_pool* p = e->pool;
    for (int i = 0; i < RP_GEN_SLOTS; i++) {
        if (!((valid_mask >> i) & 1u)) continue;
        if (hipEventRecord(e->ev_gen_read[i], st) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(st); continue; }      // (cannot mark: make the reads over instead)
        auto& r = p->gen[i].readers;
        if (std::find(r.begin(), r.end(), e) == r.end()) r.push_back(e);
    }
}
int gen_cache_find(s2k_dev_pool* p, const unsigned char* key) {
    for (int i = 0; i < p->gen_slots; i++) if (p->gen[i].valid && !memcmp(p->gen[i].key, key, 64)) { p->gen[i].stamp = ++p->gen_clock; return i; }
    return -1;
}
// Builds (stream-ordered on `st`) the tables of `key` into a free slot or the least recently used one; -1 when there is no memory for a
// table (the proofs then simply keep the general form).  pinned = 0 is an AUTOMATIC build (a generator that kept coming on valid
// proofs): it only takes a free slot or the slot of another automatically built table -- never the table of secp256k1_generator_h or
// one the application asked for -- and returns -1 when there is none.  (Pool mutex held by the caller.)
int gen_cache_build(s2k_engine* e, hipStream_t st, const unsigned char* key, int pinned) {
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
    // a slot whose memory may still be read -- by this engine's side streams or by another engine's kernels -- is rewritten only behind its
    // readers: the build's stream waits for the event each reading engine recorded behind its last reading kernel (gen_note_read).  Nothing
    // waits on the host, and engines that never touched this slot are not involved at all (until round 5: hipDeviceSynchronize() here, with
    // the engine's and the pool's locks held -- every verifier thread on the GPU stalled for the longest stream in flight).
    if (g.tab) {
        if (!g.done && hipStreamWaitEvent(st, g.ev_ready, 0) != hipSuccess) { (void)hipGetLastError(); return -1; }      // (a build of this slot still in flight)
        for (s2k_engine* r : g.readers) if (hipStreamWaitEvent(st, r->ev_gen_read[slot], 0) != hipSuccess) { (void)hipGetLastError(); return -1; }
        g.readers.clear();
    }
    if (!g.tab) {
        if (hipMalloc((void**)&g.tab, sizeof(u32) * gtab_words_for(p->gtab_bits)) != hipSuccess) { (void)hipGetLastError(); g.tab = nullptr; return -1; }      // (the width of the table of G)
        if (hipMalloc((void**)&g.xmul, sizeof(u32) * RP_XMUL_WORDS) != hipSuccess) { (void)hipGetLastError(); hipFree(g.tab); g.tab = nullptr; g.xmul = nullptr; return -1; }
    }
    if (!engine_ptab(e, 2048)) return -1;
    g.valid = 0;
    memcpy(g.key, key, 64);
    if (hipMemcpyAsync(p->gen_keys + 64 * slot, g.key, 64, hipMemcpyHostToDevice, st) != hipSuccess) { (void)hipGetLastError(); return -1; }
    hipLaunchKernelGGL(k_gen_base, dim3(1), dim3(64), 0, st, g.tab, p->gen_keys + 64 * slot, p->gtab_bits);
    if (!launch_table_build(st, g.tab, p->gtab_bits, 1)) { (void)hipGetLastError(); return -1; }
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
void gen_cache_service(s2k_engine* e, hipStream_t st) {
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
void gen_cache_collect(s2k_engine* e, hipStream_t st) {
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
    e->device = device; e->ws = nullptr; e->ws_bytes = 0; e->gtab = nullptr; e->ptab = nullptr; e->ptab_lanes = 0; e->host_flags = nullptr; e->dev_flags = nullptr; e->bp_tab = nullptr; e->bp_tab_bytes = 0; e->bp_tab_stride = 0; e->bp_gens_ok = 0;
    e->stream = nullptr; e->stream2 = nullptr; e->ev_fork = nullptr; e->ev_join = nullptr; for (int i = 0; i < 4; i++) e->ev[i] = nullptr;
    for (int i = 0; i < 32; i++) e->ev_ring[i][0] = e->ev_ring[i][1] = nullptr;
    e->ring_seq = 0;
    e->last_stream = nullptr; e->last_stream_valid = 0; e->ev_last = nullptr; e->ev_msm_fork = nullptr; e->ev_msm_join = nullptr;
    e->ev_rp_draws = nullptr; e->ev_rp_rewound = nullptr; e->rp_rewound_valid = 0;
    e->stream_pre = nullptr; e->ev_rp_in = nullptr; e->rp_mem_bytes = 0; e->rp_seq = 0; e->rp_inputs_ready = 0;
    e->rp_last_plan[0] = e->rp_last_plan[1] = nullptr;
    e->ha_pin = nullptr; e->ha_pin_words = 0; e->retired_bytes = 0;
    for (int i = 0; i < RP_GEN_SLOTS; i++) e->ev_gen_read[i] = nullptr;
    for (int i = 0; i < 2; i++) { auto& m = e->msm_slot[i]; m.s = m.s2 = nullptr; m.fork = m.join = m.done = m.in = nullptr; m.ws = nullptr; m.ws_bytes = 0; m.seen_epoch = 0; }
    e->msm_seq = 0; e->cur_pipe = 0; e->ev_last_np = nullptr; e->np_epoch = 0; e->np_valid = 0;
    e->msm_pipeline = 0; e->halfagg_host_chain = 1; e->sync_split = 1; e->stage_log = 0;
    e->msm_diag = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; e->msm_max_terms_opt = 0;
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
        e->msm_diag.c = num("S2K_MSM_C"); e->msm_diag.T = num("S2K_MSM_T"); e->msm_diag.chunk = num("S2K_MSM_CHUNK"); e->msm_diag.T2 = num("S2K_MSM_T2");
        e->msm_diag.two_pass = getenv("S2K_MSM_TWO_PASS") != nullptr; e->msm_diag.one_pass = getenv("S2K_MSM_ONE_PASS") != nullptr;
        e->msm_diag.bin_plain = getenv("S2K_MSM_BIN_PLAIN") != nullptr; e->msm_diag.no_small = getenv("S2K_MSM_NO_SMALL") != nullptr;
        e->msm_diag.old_tail = getenv("S2K_MSM_OLD_TAIL") != nullptr; e->msm_diag.slice_r = num("S2K_MSM_SLICE_R"); e->msm_diag.slice_lds = num("S2K_MSM_SLICE_LDS"); e->msm_diag.slice_maxc = num("S2K_MSM_SLICE_MAXC"); e->msm_diag.run_major = num("S2K_MSM_RUN_MAJOR"); }
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
    for (int i = 0; i < RP_GEN_SLOTS; i++) S2K_CREATE_CHK(hipEventCreateWithFlags(&e->ev_gen_read[i], hipEventDisableTiming));
    e->pool = pool_acquire(device);            // the device's tables: shared with every other engine on it, built on first use
    if (!e->pool) { s2k_engine_destroy(e); return nullptr; }
#undef S2K_CREATE_CHK
    return e;
}
extern "C" void s2k_engine_destroy(s2k_engine* e) {
    if (!e) return;
    hipSetDevice(e->device);
    hipDeviceSynchronize();            // `_dev` calls may have been issued on caller streams: nothing of this engine may still be in flight
    for (void* p : e->retired_dev) hipFree(p);
    for (void* p : e->retired_host) hipHostFree(p);
    e->retired_dev.clear(); e->retired_host.clear();
    if (e->pool) { std::lock_guard<std::recursive_mutex> pool_lock(e->pool->mu); for (int i = 0; i < RP_GEN_SLOTS; i++) { auto& r = e->pool->gen[i].readers; r.erase(std::remove(r.begin(), r.end(), e), r.end()); } }
    for (int i = 0; i < RP_GEN_SLOTS; i++) if (e->ev_gen_read[i]) hipEventDestroy(e->ev_gen_read[i]);
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
    if (bytes) *bytes = 0;
    if (!e) return nullptr;
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    if (hipSetDevice(e->device) != hipSuccess) return nullptr;
    const u32* t = engine_gtab(e, e->stream);                  // (built now if no call has needed it yet)
    if (!t || hipStreamSynchronize(e->stream) != hipSuccess) return nullptr;
    e->gtab = const_cast<u32*>(t);
    if (bytes) *bytes = sizeof(u32) * gtab_words_for(e->pool->gtab_bits);
    return t;
}
// device time of the construction of the table of G (the three kernels of gtable.h, by HIP events on the building stream), in ms; < 0 when
// the table does not exist yet or the events are not available (another process's pool built nothing here)
extern "C" float s2k_engine_gtable_build_ms(s2k_engine* e) {
    if (!e) return -1.0f;
    std::lock_guard<std::recursive_mutex> lock(e->pool->mu);
    float ms = -1.0f;
    if (e->pool->gtab_state == 0 || hipSetDevice(e->device) != hipSuccess) return ms;
    if (hipEventSynchronize(e->pool->ev_build[1]) != hipSuccess || hipEventElapsedTime(&ms, e->pool->ev_build[0], e->pool->ev_build[1]) != hipSuccess) { (void)hipGetLastError(); ms = -1.0f; }
    return ms;
}
extern "C" int s2k_engine_gtable_bits(s2k_engine* e) {
    if (!e) return 0;
    std::lock_guard<std::recursive_mutex> lock(e->pool->mu);
    return (int)e->pool->gtab_bits;                            // the width wanted until the table exists, the width it has afterwards
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
    case S2K_OPT_MSM_MAX_TERMS:
        if (value != 0 && (value < 64 || (size_t)value > ((size_t(1) << 32) - 1) / 18 - 1)) return s2k_fail_arg("s2k_engine_set_option", "S2K_OPT_MSM_MAX_TERMS: 0 or 64 .. 238609293");
        e->msm_max_terms_opt = (size_t)value; return 1;
    case S2K_OPT_GEN_CACHE_SLOTS: {                             // (the cache belongs to the device: every engine on it sees the change)
        s2k_dev_pool* p = e->pool;
        std::lock_guard<std::recursive_mutex> pool_lock(p->mu);
        const int v = value < 0 ? 0 : (value > RP_GEN_SLOTS ? RP_GEN_SLOTS : (int)value);
        if (v < p->gen_slots) {                            // slots that go away give their tables back once nothing can still read them
            // (waits for the slots' own builds and readers -- their events -- not for the device)
            if (hipSetDevice(e->device) != hipSuccess) return s2k_fail("s2k_engine_set_option", "hipSetDevice failed");
            for (int i = v; i < p->gen_slots; i++) {
                if (p->gen[i].tab && p->gen[i].valid && hipEventSynchronize(p->gen[i].ev_ready) != hipSuccess) return s2k_fail("s2k_engine_set_option", "waiting for a table build failed");
                for (s2k_engine* r : p->gen[i].readers) if (hipEventSynchronize(r->ev_gen_read[i]) != hipSuccess) return s2k_fail("s2k_engine_set_option", "waiting for a table's readers failed");
                p->gen[i].readers.clear();
                if (p->gen[i].valid && !memcmp(p->gen[i].key, k_generator_h, 64) && p->gen_h == 2) p->gen_h = 1;
                if (p->gen[i].tab) hipFree(p->gen[i].tab);
                if (p->gen[i].xmul) hipFree(p->gen[i].xmul);
                p->gen[i].tab = nullptr; p->gen[i].xmul = nullptr; p->gen[i].valid = 0; p->gen[i].pinned = 0;
            }
        }
        p->gen_slots = v; return 1;
    }
    case S2K_OPT_GTAB_BITS: {                                   // (the tables belong to the device: every engine on it sees the change)
        if (!(value == 20 || value == 22 || value == 24 || value == 26) || !gtab_bits_ok((u32)value)) return s2k_fail_arg("s2k_engine_set_option", "S2K_OPT_GTAB_BITS: 20, 22, 24 or 26");
        s2k_dev_pool* p = e->pool;
        std::lock_guard<std::recursive_mutex> pool_lock(p->mu);
        if (p->gtab_bits_wanted == (u32)value && (p->gtab_state == 0 || p->gtab_bits == (u32)value)) return 1;
        // an administrative call: the device's tables are given back and rebuilt at the new width by the next call that needs them; nothing
        // of ANY engine on the device may still read them
        if (hipSetDevice(e->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return s2k_fail("s2k_engine_set_option", "device synchronisation failed");
        int had_h = 0;
        for (int i = 0; i < p->gen_slots; i++) if (p->gen[i].valid && !memcmp(p->gen[i].key, k_generator_h, 64)) had_h = 1;
        p->gtab_bits_wanted = (u32)value;
        pool_free_tables(p);
        if (had_h && p->gen_h == 2) p->gen_h = 1;            // (secp256k1_generator_h gets its table again at the next rangeproof call)
        e->gtab = nullptr;
        return 1;
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
    const size_t nw = std::min(n_items, e->max_lanes / RP_MAX_RINGS);      // (proofs per launch group, rp_launch)
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
    if (!e->retired_dev.empty() || !e->retired_host.empty()) return engine_make_room(e);      // an explicit sizing call: outgrown buffers go back now
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

