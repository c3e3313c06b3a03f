// engine.hip -- kernels + C ABI of the gfx950 batch-verification engine (see include/secp256k1_zkp_amd.h).
//
// One translation unit: the per-lane arithmetic lives in the headers next to this file; this file holds the
// __global__ kernels, the device workspace and the extern "C" entry points.  Compiled with
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC   (see __graft_entry__.build / Makefile)
// There is no CPU implementation behind these entry points: without a HIP device every call fails loudly.
#include "gtable.h"
#include "sha256.h"
#include "rangeproof.h"
#include "schnorr.h"
#include "../../include/secp256k1_zkp_amd.h"

#include <hip/hip_runtime.h>
#include <mutex>
#include <string>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// ------------------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
static int s2k_fail(const char* what, const char* detail) {
    g_last_error = std::string(what) + ": " + (detail ? detail : "");
    return 0;
}
#define HIPCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) return s2k_fail(#call, hipGetErrorString(_e)); } while (0)
#define HIPCHK_NULL(call) do { hipError_t _e = (call); if (_e != hipSuccess) { s2k_fail(#call, hipGetErrorString(_e)); return nullptr; } } while (0)

extern "C" const char* s2k_last_error(void) { return g_last_error.c_str(); }

// ------------------------------------------------------------------------------------------------------------
// engine object
// ------------------------------------------------------------------------------------------------------------
struct s2k_engine {
    int device;
    hipStream_t stream;
    u32* gtab;                 // generator table (S2K_GTAB_WORDS words)
    unsigned char* ws;         // growable HBM workspace
    size_t ws_bytes;
    hipEvent_t ev[4];          // [0],[1] whole call; [2],[3] dominant kernel
    schnorr_midstate bip340;   // tagged-hash midstate, computed once on the host
    std::mutex mu;
};

static int engine_workspace(s2k_engine* e, size_t bytes) {
    if (bytes <= e->ws_bytes) return 1;
    HIPCHK(hipStreamSynchronize(e->stream));
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
__global__ void k_gtab_entries(u32* gtab) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 w = t >> 8, b = t & 255u;
    if (w < S2K_GTAB_WINDOWS && b >= 2) gtab_build_entry(gtab, w, b);
}

// ------------------------------------------------------------------------------------------------------------
// batch double multiplication  r = na*A + ng*G    (secp256k1_ecmult, src/ecmult.h:47)
// one multiplication per lane; inputs are gathered with byte loads (160 B per lane against ~1.5 M cycles of
// arithmetic -- the loads are noise), the result is converted to affine and serialised in the same kernel.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
k_ecmult_batch(unsigned char* __restrict__ r_xy, int32_t* __restrict__ r_inf, const unsigned char* __restrict__ a_xy,
               const unsigned char* __restrict__ a_inf, const unsigned char* __restrict__ na, const unsigned char* __restrict__ ng,
               const u32* __restrict__ gtab, size_t n) {
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
    ecmult_lane(R, A, sa, sg, ng != nullptr, gtab);
    ge out;
    ge_set_gej(out, R);
    if (live) {
        if (R.inf) { for (int k = 0; k < 64; k++) r_xy[64 * i + k] = 0; }
        else ge_store_b64(r_xy + 64 * i, out);
        r_inf[i] = R.inf;
    }
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
    e->device = device; e->ws = nullptr; e->ws_bytes = 0; e->gtab = nullptr;
    schnorr_tag_midstate(e->bip340);
    HIPCHK_NULL(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    for (int i = 0; i < 4; i++) HIPCHK_NULL(hipEventCreate(&e->ev[i]));
    HIPCHK_NULL(hipMalloc((void**)&e->gtab, sizeof(u32) * S2K_GTAB_WORDS));
    HIPCHK_NULL(hipMemsetAsync(e->gtab, 0, sizeof(u32) * S2K_GTAB_WORDS, e->stream));
    hipLaunchKernelGGL(k_gtab_base, dim3(1), dim3(64), 0, e->stream, e->gtab);
    hipLaunchKernelGGL(k_gtab_entries, dim3(S2K_GTAB_WINDOWS * 256 / 64), dim3(64), 0, e->stream, e->gtab);
    HIPCHK_NULL(hipGetLastError());
    HIPCHK_NULL(hipStreamSynchronize(e->stream));
    return e;
}
extern "C" void s2k_engine_destroy(s2k_engine* e) {
    if (!e) return;
    hipSetDevice(e->device);
    hipStreamSynchronize(e->stream);
    if (e->ws) hipFree(e->ws);
    if (e->gtab) hipFree(e->gtab);
    for (int i = 0; i < 4; i++) hipEventDestroy(e->ev[i]);
    hipStreamDestroy(e->stream);
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
    return e ? e->gtab : nullptr;
}
extern "C" float s2k_engine_last_ms(s2k_engine* e, int which) {
    float ms = -1.0f;
    if (!e) return ms;
    hipSetDevice(e->device);
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
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    HIPCHK(hipEventRecord(e->ev[0], st));
    HIPCHK(hipEventRecord(e->ev[2], st));
    hipLaunchKernelGGL(k_ecmult_batch, dim3(blocks), dim3(256), 0, st, r_xy, r_inf, a_xy, a_inf, na, ng, e->gtab, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[3], st));
    HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
}
extern "C" int s2k_ecmult_batch(s2k_engine* e, unsigned char* r_xy, int32_t* r_inf, const unsigned char* a_xy,
                                const unsigned char* a_inf, const unsigned char* na, const unsigned char* ng, size_t n) {
    if (!e) return s2k_fail("s2k_ecmult_batch", "null engine");
    if (n == 0) return 1;
    std::lock_guard<std::mutex> lock(e->mu);
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
__global__ void __launch_bounds__(64)
k_rp_prologue(rp_ws ws, uint64_t* min_value, uint64_t* max_value, const unsigned char* commits33, const unsigned char* proofs,
              const uint64_t* proof_off, const unsigned char* extra, const uint64_t* extra_off, const unsigned char* gens64, size_t n) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint64_t o0 = proof_off[p], o1 = proof_off[p + 1];
    const unsigned char* ex = nullptr; uint64_t exlen = 0;
    if (extra && extra_off) { ex = extra + extra_off[p]; exlen = extra_off[p + 1] - extra_off[p]; if (exlen == 0) ex = nullptr; }
    uint64_t mn, mx;
    rp_prologue(ws.rec[p], ws.bases + p * RP_MAX_RINGS * RP_GEJ_WORDS, &mn, &mx, commits33 + 33 * p, proofs + o0, o1 - o0, ex, exlen, gens64 + 64 * p);
    min_value[p] = mn; max_value[p] = mx;
}
__global__ void __launch_bounds__(256)
k_rp_lift(rp_ws ws, const unsigned char* proofs, const uint64_t* proof_off, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t p = t >> 5; const u32 ring = (u32)(t & 31);
    if (p >= n) return;
    const rp_rec& rec = ws.rec[p];
    if (!rec.ok || ring + 1 >= rec.rings) return;
    rp_lift(rec, ws.pub0 + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS, ws.lift_ok + p * RP_MAX_RINGS + ring, proofs + proof_off[p], ring);
}
__global__ void __launch_bounds__(64)
k_rp_sum(rp_ws ws, size_t n) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    rp_sum(ws.rec[p], ws.pub0 + p * RP_MAX_RINGS * RP_GEJ_WORDS, ws.lift_ok + p * RP_MAX_RINGS);
}
__global__ void __launch_bounds__(256, 2)
k_rp_rings(rp_ws ws, const unsigned char* __restrict__ proofs, const uint64_t* __restrict__ proof_off, const u32* __restrict__ gtab, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t p = t >> 5; const u32 ring = (u32)(t & 31);
    int live = p < n;
    if (!live) p = 0;
    const rp_rec& rec = ws.rec[p];
    live &= (ring < rec.rings);
    rp_ring(rec, ws.bases + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS, ws.pub0 + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS,
            ws.ring_out + (p * RP_MAX_RINGS + ring) * 36, ws.ring_ok + p * RP_MAX_RINGS + ring, proofs + proof_off[p], ring, live, gtab);
}
__global__ void __launch_bounds__(64)
k_rp_final(rp_ws ws, int32_t* results, const unsigned char* proofs, const uint64_t* proof_off, size_t n) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    results[p] = rp_final(ws.rec[p], ws.ring_out + p * RP_MAX_RINGS * 36, ws.ring_ok + p * RP_MAX_RINGS, proofs + proof_off[p]);
}

static size_t rp_ws_bytes(size_t n) {
    return ws_need({n * sizeof(rp_rec), n * RP_MAX_RINGS * RP_GEJ_WORDS * 4, n * RP_MAX_RINGS * RP_GEJ_WORDS * 4, n * RP_MAX_RINGS, n * RP_MAX_RINGS * 36, n * RP_MAX_RINGS});
}
static void rp_ws_carve(rp_ws& w, ws_carver& c, size_t n) {
    w.rec = c.take<rp_rec>(n);
    w.bases = c.take<u32>(n * RP_MAX_RINGS * RP_GEJ_WORDS);
    w.pub0 = c.take<u32>(n * RP_MAX_RINGS * RP_GEJ_WORDS);
    w.lift_ok = c.take<unsigned char>(n * RP_MAX_RINGS);
    w.ring_out = c.take<unsigned char>(n * RP_MAX_RINGS * 36);
    w.ring_ok = c.take<unsigned char>(n * RP_MAX_RINGS);
}
// launches the five stages; `w` must already point into device memory
static int rp_launch(s2k_engine* e, hipStream_t st, const rp_ws& w, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                     const unsigned char* commits33, const unsigned char* proofs, const uint64_t* proof_off, const unsigned char* extra,
                     const uint64_t* extra_off, const unsigned char* gens64, size_t n) {
    const unsigned b64 = (unsigned)((n + 63) / 64), b256 = (unsigned)((n * 32 + 255) / 256);
    HIPCHK(hipEventRecord(e->ev[0], st));
    hipLaunchKernelGGL(k_rp_prologue, dim3(b64), dim3(64), 0, st, w, min_value, max_value, commits33, proofs, proof_off, extra, extra_off, gens64, n);
    hipLaunchKernelGGL(k_rp_lift, dim3(b256), dim3(256), 0, st, w, proofs, proof_off, n);
    hipLaunchKernelGGL(k_rp_sum, dim3(b64), dim3(64), 0, st, w, n);
    HIPCHK(hipEventRecord(e->ev[2], st));
    hipLaunchKernelGGL(k_rp_rings, dim3(b256), dim3(256), 0, st, w, proofs, proof_off, e->gtab, n);
    HIPCHK(hipEventRecord(e->ev[3], st));
    hipLaunchKernelGGL(k_rp_final, dim3(b64), dim3(64), 0, st, w, results, proofs, proof_off, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
}
extern "C" int secp256k1_rangeproof_verify_batch_dev(s2k_engine* e, void* stream, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                                                     const unsigned char* commits33, const unsigned char* proofs, const uint64_t* proof_off,
                                                     const unsigned char* extra, const uint64_t* extra_off, const unsigned char* gens64, size_t n) {
    if (!e) return s2k_fail("secp256k1_rangeproof_verify_batch_dev", "null engine");
    if (n == 0) return 1;
    std::lock_guard<std::mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    if (!engine_workspace(e, rp_ws_bytes(n))) return 0;
    ws_carver c{e->ws, 0}; rp_ws w; rp_ws_carve(w, c, n);
    return rp_launch(e, stream ? (hipStream_t)stream : e->stream, w, results, min_value, max_value, commits33, proofs, proof_off, extra, extra_off, gens64, n);
}
extern "C" int secp256k1_rangeproof_verify_batch(s2k_engine* e, int32_t* results, uint64_t* min_value, uint64_t* max_value,
                                                 const unsigned char* commits33, const unsigned char* proofs, const uint64_t* proof_off,
                                                 const unsigned char* extra, const uint64_t* extra_off, const unsigned char* gens64, size_t n) {
    if (!e) return s2k_fail("secp256k1_rangeproof_verify_batch", "null engine");
    if (n == 0) return 1;
    std::lock_guard<std::mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    const size_t pbytes = (size_t)proof_off[n], ebytes = (extra && extra_off) ? (size_t)extra_off[n] : 0;
    const size_t io = ws_need({4 * n, 8 * n, 8 * n, 33 * n, pbytes + 64, 8 * (n + 1), ebytes + 64, 8 * (n + 1), 64 * n});
    if (!engine_workspace(e, rp_ws_bytes(n) + io)) return 0;
    ws_carver c{e->ws, 0}; rp_ws w; rp_ws_carve(w, c, n);
    int32_t* d_res = c.take<int32_t>(n); uint64_t* d_min = c.take<uint64_t>(n); uint64_t* d_max = c.take<uint64_t>(n);
    unsigned char* d_com = c.take<unsigned char>(33 * n); unsigned char* d_pr = c.take<unsigned char>(pbytes + 64);
    uint64_t* d_off = c.take<uint64_t>(n + 1); unsigned char* d_ex = c.take<unsigned char>(ebytes + 64); uint64_t* d_eoff = c.take<uint64_t>(n + 1);
    unsigned char* d_gen = c.take<unsigned char>(64 * n);
    hipStream_t st = e->stream;
    HIPCHK(hipMemcpyAsync(d_com, commits33, 33 * n, hipMemcpyHostToDevice, st));
    if (pbytes) HIPCHK(hipMemcpyAsync(d_pr, proofs, pbytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_off, proof_off, 8 * (n + 1), hipMemcpyHostToDevice, st));
    if (ebytes) { HIPCHK(hipMemcpyAsync(d_ex, extra, ebytes, hipMemcpyHostToDevice, st)); }
    if (extra && extra_off) HIPCHK(hipMemcpyAsync(d_eoff, extra_off, 8 * (n + 1), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_gen, gens64, 64 * n, hipMemcpyHostToDevice, st));
    if (!rp_launch(e, st, w, d_res, d_min, d_max, d_com, d_pr, d_off, (extra && extra_off) ? d_ex : nullptr, (extra && extra_off) ? d_eoff : nullptr, d_gen, n)) return 0;
    HIPCHK(hipMemcpyAsync(results, d_res, 4 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(min_value, d_min, 8 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(max_value, d_max, 8 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 1;
}
// single-item form with the reference's argument list (include/secp256k1_rangeproof.h:70-80)
static s2k_engine* g_default_engine = nullptr;
static std::mutex g_default_mu;
extern "C" int secp256k1_rangeproof_verify_amd(const void* ctx, uint64_t* min_value, uint64_t* max_value, const void* commit,
                                               const unsigned char* proof, size_t plen, const unsigned char* extra_commit,
                                               size_t extra_commit_len, const void* gen) {
    (void)ctx;
    if (!min_value || !max_value || !commit || !proof || !gen || (!extra_commit && extra_commit_len)) return s2k_fail("secp256k1_rangeproof_verify_amd", "illegal argument (ARG_CHECK)");
    {
        std::lock_guard<std::mutex> lock(g_default_mu);
        if (!g_default_engine) {
            const char* d = getenv("S2K_DEVICE");
            g_default_engine = s2k_engine_create(d ? atoi(d) : 0);
            if (!g_default_engine) return 0;
        }
    }
    int32_t res = 0; uint64_t off[2] = {0, plen}, eoff[2] = {0, extra_commit_len};
    if (!secp256k1_rangeproof_verify_batch(g_default_engine, &res, min_value, max_value, (const unsigned char*)commit, proof, off,
                                           extra_commit_len ? extra_commit : nullptr, extra_commit_len ? eoff : nullptr, (const unsigned char*)gen, 1)) return 0;
    return res;
}

// ------------------------------------------------------------------------------------------------------------
// BIP-340 batch verification (schnorr.h): one signature per lane
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
k_schnorr_verify(int32_t* __restrict__ results, schnorr_midstate mid, const unsigned char* __restrict__ sigs, const unsigned char* __restrict__ msgs,
                 size_t msglen, const unsigned char* __restrict__ pks, int pk_format, const u32* __restrict__ gtab, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int live = i < n;
    const size_t ii = live ? i : 0;
    const int r = schnorr_verify_lane(mid, sigs + 64 * ii, msgs + msglen * ii, msglen, pks + (pk_format ? 64 : 32) * ii, pk_format, live, gtab);
    if (live) results[i] = r;
}
extern "C" int secp256k1_schnorrsig_verify_batch_dev(s2k_engine* e, void* stream, int32_t* results, const unsigned char* sigs,
                                                     const unsigned char* msgs, size_t msglen, const unsigned char* pubkeys, int pk_format, size_t n) {
    if (!e) return s2k_fail("secp256k1_schnorrsig_verify_batch_dev", "null engine");
    if (n == 0) return 1;
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    HIPCHK(hipEventRecord(e->ev[0], st)); HIPCHK(hipEventRecord(e->ev[2], st));
    hipLaunchKernelGGL(k_schnorr_verify, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, results, e->bip340, sigs, msgs, msglen, pubkeys, pk_format, e->gtab, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->ev[3], st)); HIPCHK(hipEventRecord(e->ev[1], st));
    return 1;
}
extern "C" int secp256k1_schnorrsig_verify_batch(s2k_engine* e, int32_t* results, const unsigned char* sigs, const unsigned char* msgs,
                                                 size_t msglen, const unsigned char* pubkeys, int pk_format, size_t n) {
    if (!e) return s2k_fail("secp256k1_schnorrsig_verify_batch", "null engine");
    if (n == 0) return 1;
    std::lock_guard<std::mutex> lock(e->mu);
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
// ---- not yet implemented (filled in below as the round progresses) -----------------------------------------------
#define S2K_TODO(name) return s2k_fail(name, "not implemented yet")
extern "C" int s2k_engine_reserve(s2k_engine* e, size_t n_items) { if (!e) return 0; std::lock_guard<std::mutex> lock(e->mu); HIPCHK(hipSetDevice(e->device)); return engine_workspace(e, n_items * 16384); }
extern "C" int s2k_ecmult_multi(s2k_engine*, unsigned char*, int32_t*, const unsigned char*, const unsigned char*, const unsigned char*, const unsigned char*, size_t) { S2K_TODO("s2k_ecmult_multi"); }
extern "C" int s2k_ecmult_multi_dev(s2k_engine*, void*, unsigned char*, int32_t*, const unsigned char*, const unsigned char*, const unsigned char*, const unsigned char*, size_t) { S2K_TODO("s2k_ecmult_multi_dev"); }
extern "C" int s2k_ecmult_multi_partial_dev(s2k_engine*, void*, uint32_t*, const unsigned char*, const unsigned char*, const unsigned char*, const unsigned char*, size_t) { S2K_TODO("s2k_ecmult_multi_partial_dev"); }
extern "C" int s2k_gej_sum_dev(s2k_engine*, void*, unsigned char*, int32_t*, const uint32_t*, size_t) { S2K_TODO("s2k_gej_sum_dev"); }
extern "C" int secp256k1_bppp_norm_product_verify_batch(s2k_engine*, int32_t*, const unsigned char*, size_t, const unsigned char*, const unsigned char*, const unsigned char*, size_t, size_t, const unsigned char*, size_t, const unsigned char*, size_t) { S2K_TODO("secp256k1_bppp_norm_product_verify_batch"); }
