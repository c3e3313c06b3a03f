// pedersen.h -- balance check of Pedersen commitments: sum(positive) - sum(negative) == infinity
// (secp256k1_pedersen_verify_tally, src/modules/generator/main_impl.h:371-396; commitment decoding
// secp256k1_pedersen_commitment_load :266-273).  SURVEY.md section 8f, rank 4.
//
// One lane per commitment lifts it (one square root) and signs it by the list it belongs to; the per-tally sums are
// bounded-run partial sums (the MSM's k_msm_roundN), so a 2-commitment tally and a 10^5-commitment tally both stay
// spread over lanes.  A tally is accepted iff its sum is the point at infinity; an empty tally is accepted (:384, :395).
#pragma once
#include "group.h"

// commit33: the serialised commitment (equivalently the first 33 bytes of the 64-byte secp256k1_pedersen_commitment object):
// byte 0 = 8 | sign bit, then x.  Returns 0 for an encoding secp256k1_pedersen_commitment_parse refuses (wrong prefix, x >= p,
// x not on the curve): such an object cannot come out of the reference's parse or commit functions.
S2K_HD int pedersen_load(ge& c, const unsigned char* commit33) {
    fe x;
    int ok = ((commit33[0] & 0xFE) == 8);                  // what secp256k1_pedersen_commitment_parse checks (:289-293)
    ok &= fe_set_b32_limit(x, commit33 + 1);
    ok &= ge_set_xquad(c, x);
    fe_normalize(c.y);
    if (commit33[0] & 1) { fe_neg(c.y, c.y, 1); fe_normalize(c.y); }
    return ok;
}
// largest t with off[t] <= i  (off non-decreasing, n+1 entries, i < off[n])
S2K_HD size_t pedersen_find_tally(const unsigned long long* off, size_t n, unsigned long long i) {
    size_t lo = 0, hi = n;
    while (hi - lo > 1) { const size_t mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}
