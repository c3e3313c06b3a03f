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
__global__ void __launch_bounds__(256, S2K_RINGS_WAVES)
k_rp_rings(rp_ws ws, const unsigned char* __restrict__ proofs, const uint64_t* __restrict__ proof_off, const u32* __restrict__ gtab, u32* __restrict__ ptab, size_t n, u32* ev, int split,
           rp_gen_dev gc) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t p = t >> 5; const u32 ring = (u32)(t & 31);
    int live = p < n;
    if (!live) p = 0;
    const rp_rec& rec = ws.rec[p];
    live &= (ring < rec.rings);
    __shared__ u32 s_dig[S2K_RING_DIG_WORDS * 256];
    u32* const lane_tab = ptab + t * S2K_RTAB_WORDS;
    // Shared-generator form when every working lane of the wavefront has a cached table for its proof's generator (lanes may name
    // different slots); otherwise -- or when that form hands the wavefront back (a suspect ring) -- the general form below.
    if (gc.valid) {
        const int idle = !(live && rec.ok);
        const u32 slot = idle ? gc.any : rec.gslot;
        if (S2K_WAVE_ALL(slot < RP_GEN_SLOTS) && S2K_WAVE_ANY(!idle)) {
            const u32 sl = slot < RP_GEN_SLOTS ? slot : gc.any;
            if (rp_ring_shared(rec, ws.bases + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS, ws.pub0 + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS,
                               ws.ring_out + p * RP_RING_OUT_BYTES + ring * 33, ws.ring_ok + p * RP_MAX_RINGS + ring, proofs + proof_off[p], ring, live, gtab,
                               gc.tab[sl], gc.xmul[sl], lane_tab, S2K_LANE_DIG(s_dig), ev ? ev + (p * RP_MAX_RINGS + ring) * 32 : nullptr)) return;
        }
    }
    const lane_mem lm{lane_tab, S2K_LANE_DIG(s_dig)};
    rp_ring(rec, ws.bases + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS, ws.pub0 + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS,
            ws.ring_out + p * RP_RING_OUT_BYTES + ring * 33, ws.ring_ok + p * RP_MAX_RINGS + ring, proofs + proof_off[p], ring, live, gtab, lm, ev ? ev + (p * RP_MAX_RINGS + ring) * 32 : nullptr,
            split ? ws.dbases + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS : (const u32*)nullptr, split ? ws.tcur + (p * RP_MAX_RINGS + ring) * RP_GEJ_WORDS : (u32*)nullptr);
}
