// gtable.h -- device-resident fixed-base table for the generator G.
//
// Role of the reference's secp256k1_pre_g / secp256k1_pre_g_128 (src/precomputed_ecmult.h:30-33, built by
// src/ecmult_compute_table_impl.h:14-46): odd multiples for wNAF(15).  Here the table is organised for a machine with
// 288 GB of HBM that would rather gather 64 bytes than execute doublings: entry (w, v) = v * 2^(D w) * G for every magnitude
// v = 1 .. 2^(D-1) of a SIGNED D-bit digit of every window of a scalar (D = S2K_GTAB_BITS = 26: 10 windows; ecmult.h "generator
// table"), as one aligned 64-byte sector of canonical words (affine x, y), so ng*G is 10 mixed additions and zero doublings.
// 10 x 2^25 entries x 64 B = 21.5 GB; the gathers are issued one addition ahead, so their HBM latency is covered.  The table is
// *computed on the device* by the first call that needs it (two kernels, ~0.6 s), never shipped as data.
#pragma once
#include "ecmult.h"

S2K_HD void ge_set_generator(ge& g) {
    const u32 gx[9] = {0x16F81798u, 0x0F940AD8u, 0x138A3656u, 0x17F9B65Bu, 0x10B07029u, 0x114AE743u, 0x0EB15681u, 0x0FDF3B97u, 0x0079BE66u};
    const u32 gy[9] = {0x1B10D4B8u, 0x023E847Fu, 0x01550667u, 0x0F68914Du, 0x108A8FD1u, 0x1DFE0708u, 0x11957693u, 0x0EE4D478u, 0x00483ADAu};
#pragma unroll
    for (int i = 0; i < 9; i++) { g.x.n[i] = gx[i]; g.y.n[i] = gy[i]; }
}
S2K_HD void gtab_store(u32* gtab, u32 w, u32 v, const ge& a) {
    u32* p = gtab + S2K_GTAB_SLOT(w, v) * S2K_GTAB_ENTRY_WORDS;
    u32 wx[8], wy[8];
    fe_to_words(wx, a.x); fe_to_words(wy, a.y);                     // `a` is normalised (ge_set_gej)
    for (int i = 0; i < 8; i++) { p[i] = wx[i]; p[8 + i] = wy[i]; }
}
// step 1 (one thread per window w): base[w] = 2^(B w) * G, affine, stored as entry (w, 1).  `base`: any other point than G (the
// fixed-base tables of the rangeproof generators, rangeproof.h "shared-generator form", have exactly this layout).
S2K_HD void gtab_build_base(u32* gtab, u32 w, const ge* base = nullptr) {
    ge g; if (base) g = *base; else ge_set_generator(g);
    gej j; gej_set_ge(j, g);
    for (u32 i = 0; i < S2K_GTAB_BITS * w; i++) { gej t; gej_double(t, j); j = t; }
    ge a; ge_set_gej(a, j);
    gtab_store(gtab, w, 1, a);
}
// step 2 (one thread per (w, v), v = 2..2^(D-1)): entry = v * base[w] by left-to-right double-and-add.
S2K_HD void gtab_build_entry(u32* gtab, u32 w, u32 v) {
    ge base; gtab_load(base, gtab, w, 1);
    gej acc; gej_set_infinity(acc);
    for (int bit = S2K_GTAB_BITS - 1; bit >= 0; bit--) {
        gej t; gej_double(t, acc); acc = t;
        if ((v >> bit) & 1u) {
            const int f = gej_add_ge(t, acc, base);
            acc = t;
            if (f == GEJ_ADD_NEEDS_DOUBLE) { gej_double(t, acc); acc = t; }
        }
    }
    ge a; ge_set_gej(a, acc);
    gtab_store(gtab, w, v, a);
}
