#include "engine_internal.h"

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
// Chain states that came from the CALLER (secp256k1_schnorrsig_aggverify_dev_chain) are checked before they are believed: with every
// intermediate state given, block j's state is compress(state[j - 1], block j) -- one compression per lane, all blocks in parallel -- and any
// mismatch raises the flag that makes the verdict 0.  The randomizers z_i come out of these states: unchecked, a wrong array could turn
// a forged aggregate into an accepted one (the reference always derives z_i from the inputs it verifies, main_impl.h:153-163).
__global__ void __launch_bounds__(256)
k_ha_check_chain(u32* flags, const u32* __restrict__ states, const unsigned char* aggsig, const unsigned char* pkx32, const unsigned char* msgs32, size_t nblocks) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nblocks) return;
    u32 wk[64]; ha_schedule(wk, aggsig, pkx32, msgs32, j);
    u32 st[8];
    if (j == 0) ha_tag_midstate(st); else { for (int k = 0; k < 8; k++) st[k] = states[8 * (j - 1) + k]; }
    ha_rounds(st, wk);
    u32 diff = 0;
    for (int k = 0; k < 8; k++) diff |= st[k] ^ states[8 * j + k];
    if (diff) flags[0] = 1u;
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
                     const unsigned char* d_agg, const ha_host_src* host = nullptr, const u32* d_states_in = nullptr) {
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
    if (nblocks && d_states_in) {
        d_states = const_cast<u32*>(d_states_in);              // the caller has walked the chain (secp256k1_schnorrsig_aggverify_dev_chain): checked, in parallel
        hipLaunchKernelGGL(k_ha_check_chain, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, st, d_flags, (const u32*)d_states_in, d_agg, (const unsigned char*)d_pkx, d_msg, nblocks);
    } else if (nblocks && host) {
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
// The `_dev` form with the randomizer hash's chain states supplied by the caller.  z_i hashes the whole prefix r_0|x(P_0)|m_0|...|r_i|x(P_i)|m_i
// (src/modules/schnorrsig_halfagg/main_impl.h:153-163): a Merkle-Damgard chain of 1.5 blocks per signature that no second lane can help
// with -- 118 ms on the device for 2^15 signatures, 2.6 ms of SHA extensions on one host core.  A caller whose inputs are in HBM usually
// still HAS them on the host (it received them there): it walks the chain with s2k_halfagg_chain_states (or its own SHA-256), uploads the
// 32 bytes per block, and the device only finalises every z_i in parallel.  chain_states: ((3 n) >> 1) x 8 words in HBM, state after every
// full 64-byte block of the tagged hash's input behind the tag midstate; NULL: the device walks the chain itself (the plain `_dev` form).
// The states are CHECKED on the device (k_ha_check_chain: state[j] == compress(state[j - 1], block j) for every j, in parallel -- the chain is
// serial to walk but not to verify), so a wrong or malicious array can only turn the verdict to 0, never to 1.
extern "C" int secp256k1_schnorrsig_aggverify_dev_chain(s2k_engine* e, void* stream, int32_t* result_dev, const unsigned char* pubkeys, int pk_format,
                                                        const unsigned char* msgs32, size_t n, const unsigned char* aggsig, size_t aggsig_len,
                                                        const uint32_t* chain_states) {
    if (!e) return s2k_fail("secp256k1_schnorrsig_aggverify_dev_chain", "null engine");
    if (!result_dev || !aggsig || ((!pubkeys || !msgs32) && n)) return s2k_fail_arg("secp256k1_schnorrsig_aggverify_dev_chain", "illegal argument (ARG_CHECK)");
    std::lock_guard<std::recursive_mutex> lock(e->mu);
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    stream_guard sg(e, st);
    if ((aggsig_len / 32) == 0 || (aggsig_len / 32) - 1 != n || (aggsig_len % 32) != 0) { HIPCHK(hipMemsetAsync(result_dev, 0, 4, st)); return 1; }     // main_impl.h:122-125
    if (!engine_workspace(e, ha_ws_bytes(e, n))) return 0;
    ws_carver c{e->ws, 0};
    return ha_launch(e, st, c, result_dev, pubkeys, pk_format, msgs32, n, aggsig, nullptr, chain_states);
}
// host helper for the call above: the chain states of one aggregate, from HOST copies of its inputs (SHA extensions when the CPU has them)
extern "C" int s2k_halfagg_chain_states(uint32_t* states_out, const unsigned char* pubkeys, int pk_format, const unsigned char* msgs32, size_t n,
                                        const unsigned char* aggsig) {
    if (!states_out || ((!pubkeys || !msgs32 || !aggsig) && n)) return s2k_fail_arg("s2k_halfagg_chain_states", "illegal argument (ARG_CHECK)");
    const ha_host_src h{aggsig, pubkeys, pk_format, msgs32};
    ha_host_chain(states_out, h, (3 * n) >> 1);
    return 1;
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

