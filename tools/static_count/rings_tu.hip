// Static instruction accounting for the hot kernel: this TU holds only k_rp_rings (same body as engine.hip -- regenerate with
// tools/static_count/sync.py after touching the kernel), so that
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -I../../secp256k1_zkp_amd/csrc rings_tu.hip -o rings.s
// takes ~1 min and tools/static_count/loops.py can count the instructions of every loop body.  Not part of the product.
#include "gtable.h"
#include "sha256.h"
#include "rangeproof.h"
#include <hip/hip_runtime.h>
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
