#include "engine_internal.h"

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
int engine_rp_slots(s2k_engine* e, size_t nw) {
    const size_t bytes = (rp_ws_bytes(nw) + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
    if (bytes <= e->rp_mem_bytes) return 1;
    // (no device-wide wait: the outgrown sets are retired -- chunks in flight keep reading them -- and new ones allocated beside them)
    const size_t want = std::max(bytes, e->rp_mem_bytes + e->rp_mem_bytes / 2);
    void* nw2[2] = {nullptr, nullptr};
    int ok = hipMalloc(&nw2[0], want) == hipSuccess && hipMalloc(&nw2[1], want) == hipSuccess;
    size_t got = want;
    if (!ok) {
        (void)hipGetLastError();
        for (int i = 0; i < 2; i++) if (nw2[i]) { (void)hipFree(nw2[i]); nw2[i] = nullptr; }
        if (!engine_make_room(e)) return 0;
        for (int i = 0; i < 2; i++) { if (e->rp_mem[i]) (void)hipFree(e->rp_mem[i]); e->rp_mem[i] = nullptr; }
        e->rp_mem_bytes = 0; got = bytes;
        for (int i = 0; i < 2; i++) HIPCHK(hipMalloc(&nw2[i], got));
    } else for (int i = 0; i < 2; i++) engine_retire_dev(e, e->rp_mem[i], e->rp_mem_bytes);
    for (int i = 0; i < 2; i++) { e->rp_mem[i] = (unsigned char*)nw2[i]; e->rp_done_valid[i] = 0; e->rp_last_plan[i] = nullptr; }
    e->rp_mem_bytes = got;
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
    gen_note_read(e, st, gc.valid);              // (an eviction of one of these slots waits for exactly this point of this stream)
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
    // (the staging set belongs to this call alone -- its previous owner handed it back after its copies had completed -- so the old buffers
    //  are not in use; they are retired rather than freed because hipFree / hipHostFree wait for the whole device)
    if (in_bytes > S.in_bytes || in_bytes + out_bytes + 512 > S.dev_bytes) {
        engine_retire_host(e, S.in); engine_retire_dev(e, S.dev, S.dev_bytes);
        S.in = nullptr; S.in_bytes = 0; S.dev = nullptr; S.dev_bytes = 0;
        in_bytes = (in_bytes + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
        const size_t db = in_bytes + ((out_bytes + 65535) & ~size_t(65535)) + 65536;
        if (hipHostMalloc((void**)&S.in, in_bytes, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&S.dev, db) != hipSuccess) {
            (void)hipGetLastError();
            if (S.in) { (void)hipHostFree(S.in); S.in = nullptr; }
            if (!engine_make_room(e)) return 0;
            HIPCHK(hipHostMalloc((void**)&S.in, in_bytes, hipHostMallocDefault));
            HIPCHK(hipMalloc((void**)&S.dev, db));
        }
        S.in_bytes = in_bytes; S.dev_bytes = db;
    }
    if (out_bytes > S.out_bytes) {
        engine_retire_host(e, S.out);
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
int rp_ptrs_check(const char* who, int32_t* results, uint64_t* min_value, uint64_t* max_value, const void* const* commit_objs, const unsigned char* const* proofs,
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
