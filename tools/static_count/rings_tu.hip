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
// K3 comes as two kernels over the same grid (1 lane / ring, lane t = proof t >> 5, ring t & 31):
//   k_rp_rings_shared  the shared-generator form (rangeproof.h: rp_ring_shared) for every wavefront all of whose working lanes have a cached
//                      table for their proof's generator (lanes may name different slots); a wavefront it does not serve -- no table, or
//                      a suspect ring -- raises its word of `todo`;
//   k_rp_rings         the general form; with `todo` it only works on the wavefronts flagged there (the others leave at once).
// Two kernels rather than one with both bodies: each gets its own register allocation (the combined kernel spilled 325 VGPRs) and the
// hot loops of one form do not share the instruction cache with the other's.
__global__ void __launch_bounds__(256, S2K_RINGS_WAVES)
k_rp_rings_shared(rp_ws ws, const unsigned char* __restrict__ proofs, const uint64_t* __restrict__ proof_off, const u32* __restrict__ gtab, u32* __restrict__ ptab, size_t n, u32* ev,
                  rp_gen_dev gc, u32* __restrict__ todo, u32 stagger) {
    if (stagger & 255u) {                                                    // diagnostic: start the workgroups out of phase
        const u32 d = ((blockIdx.x * 2654435761u) >> 26) * (stagger & 255u);
        for (u32 i = 0; i < d; i++) __builtin_amdgcn_s_sleep(127);
    }
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t p = t >> 5; const u32 ring = (u32)(t & 31);
    int live = p < n;
    if (!live) p = 0;
    const rp_rec& rec = ws.rec[p];
    live &= (ring < rec.rings);
    __shared__ u32 s_dig[S2K_RING_DIG_WORDS * 256];
    __shared__ u32 s_inv[RP_INV_LDS_WORDS];
    const int idle = !(live && rec.ok);
    const u32 slot = idle ? gc.any : rec.gslot;
    int served = 0;
    const int part = S2K_WAVE_ANY(!idle) && S2K_WAVE_ALL(slot < RP_GEN_SLOTS);
    rp_inv_join(s_inv, part);                                              // (the wavefronts of the workgroup share their inversions)
    if (!S2K_WAVE_ANY(!idle)) served = 1;                                  // nothing to do for this wavefront in either form
    else if (part) {
        const u32 sl = slot < RP_GEN_SLOTS ? slot : gc.any;
        served = rp_ring_shared(rec, ws.pub0 + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS,
                                ws.ring_out + p * RP_RING_OUT_BYTES + ring * 33, ws.ring_ok + p * RP_MAX_RINGS + ring, proofs + proof_off[p], ring, live, gtab,
                                gc.tab[sl], gc.xmul[sl], ptab + t * S2K_RTAB_WORDS,
                                ptab + (size_t)gridDim.x * 256 * S2K_RTAB_WORDS + (t >> 6) * S2K_RRAW_WAVE_WORDS + (t & 63), S2K_LANE_DIG(s_dig), ev ? ev + (p * RP_MAX_RINGS + ring) * 32 : nullptr, stagger >> 8,
                                (stagger & 0x80u) ? (u32*)nullptr : s_inv);
    }
    if ((threadIdx.x & 63) == 0) todo[t >> 6] = served ? 0u : 1u;
}
__global__ void __launch_bounds__(256, S2K_RINGS_WAVES)
k_rp_rings(rp_ws ws, const unsigned char* __restrict__ proofs, const uint64_t* __restrict__ proof_off, const u32* __restrict__ gtab, u32* __restrict__ ptab, size_t n, u32* ev, int split,
           const u32* __restrict__ todo) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (todo && !S2K_UNIFORM(todo[t >> 6])) return;
    size_t p = t >> 5; const u32 ring = (u32)(t & 31);
    int live = p < n;
    if (!live) p = 0;
    const rp_rec& rec = ws.rec[p];
    live &= (ring < rec.rings);
    __shared__ u32 s_dig[S2K_DIG_WORDS * 256];
    const lane_mem lm{ptab + t * S2K_PTAB_WORDS, S2K_LANE_DIG(s_dig)};
    rp_ring(rec, ws.bases + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS, ws.pub0 + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS,
            ws.ring_out + p * RP_RING_OUT_BYTES + ring * 33, ws.ring_ok + p * RP_MAX_RINGS + ring, proofs + proof_off[p], ring, live, gtab, lm, ev ? ev + (p * RP_MAX_RINGS + ring) * 32 : nullptr,
            split ? ws.dbases + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS : (const u32*)nullptr, split ? ws.tcur + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS : (u32*)nullptr);
}
