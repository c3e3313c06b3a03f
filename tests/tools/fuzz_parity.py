#!/usr/bin/env python3
"""Differential fuzzing of the GPU verifiers against the unmodified reference (oracle/_ref): large batches of valid proofs of mixed
shapes with aggressive mutations (header bytes, sign bits, lengths, scalars at the group order, trailing / missing bytes).
Not part of the test suite (it needs minutes of reference CPU time); run on a GPU box:  python tests/tools/fuzz_parity.py [seed] [n]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from secp256k1_zkp_amd import Engine  # noqa: E402
from tests.refapi import Ref, N  # noqa: E402


def mutate(p, rng, k):
    q = bytearray(p)
    if not q:
        return bytes(q)
    if k == 0:   q[int(rng.integers(0, min(len(q), 12)))] ^= 1 << int(rng.integers(0, 8))          # header / sign bytes
    elif k == 1: q[int(rng.integers(0, len(q)))] ^= 1 << int(rng.integers(0, 8))                   # anywhere
    elif k == 2: q = q[:int(rng.integers(0, len(q)))]                                               # truncated
    elif k == 3: q += bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))         # trailing bytes
    elif k == 4: o = len(q) - 32 * int(rng.integers(1, max(2, min(8, len(q) // 32)))); q[o:o + 32] = N.to_bytes(32, "big")   # a scalar == n
    elif k == 5: o = len(q) - 32 * int(rng.integers(1, max(2, min(8, len(q) // 32)))); q[o:o + 32] = bytes(32)              # a scalar == 0
    elif k == 6: q[0] = int(rng.integers(0, 256))                                                   # random first header byte
    elif k == 7: q[1] = int(rng.integers(0, 256))
    return bytes(q)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    rng = np.random.default_rng(seed); ref = Ref(); eng = Engine(0)
    # ---- rangeproofs
    C, P, G = [], [], []
    shapes = [(64, 0, 0), (32, 0, 0), (5, 2, 17), (1, 0, 0), (13, 3, 1000), (63, 0, 5), (7, 18, 0), (2, 0, 0), (3, 1, 9), (0, -1, 0)]
    per = max(1, n // len(shapes))
    for (mb, exp, mv) in shapes:
        vals = rng.integers(0, 2**max(mb, 1) if mb < 63 else 2**62, per, dtype=np.uint64) + np.uint64(mv) if mb else rng.integers(0, 2**40, per, dtype=np.uint64)
        c, p, g, _ = ref.make_rangeproofs(per, rng, min_bits=mb, exp=exp, min_value=mv, values=vals, threads=16)
        C.append(c); P += p; G.append(g)
    C = np.concatenate(C); G = np.concatenate(G)
    for i in range(len(P)):
        if i % 4: P[i] = mutate(P[i], rng, int(rng.integers(0, 8)))
        if i % 16 == 5: C[i, int(rng.integers(0, 33))] ^= 1 << int(rng.integers(0, 8))
        if i % 16 == 9: G[i, int(rng.integers(0, 64))] ^= 1 << int(rng.integers(0, 8))
        # (5 and 9 are odd: those proofs are always mutated as well.  The same on UNMUTATED proofs -- i % 4 == 0 -- is what finds a verifier that is
        #  more permissive than parse + verify about the OTHER inputs: round 4, a commitment prefix with bit 7 set on a valid proof)
        if i % 16 == 4: C[i, int(rng.integers(0, 33)) if rng.integers(0, 4) else 0] ^= 1 << int(rng.integers(0, 8))
        if i % 16 == 8: G[i, int(rng.integers(0, 64))] ^= 1 << int(rng.integers(0, 8))
    e = ref.rangeproof_verify_many(C, P, G, threads=16)
    r = eng.rangeproof_verify_batch(C, P, G)
    # a commitment that secp256k1_pedersen_commitment_parse refuses never reaches secp256k1_rangeproof_verify in the reference
    # (there is no object to pass); for those only the verdict 0 is compared, not the header-derived min/max
    parses = np.array([int(ref.pedersen_verify_tally_many([(C[i:i + 1], C[i:i + 1])])[0]) >= 0 for i in range(len(P))])
    bad = np.nonzero((e[0] != r[0]) | (parses & ((e[1] != r[1]) | (e[2] != r[2]))))[0]
    print("rangeproof: %d items, %d accepted, %d unparseable commitments, mismatches: %d" % (len(P), int(e[0].sum()), int((~parses).sum()), len(bad)), bad[:10])
    for i in bad[:5]:
        print("   item", i, "ref", e[0][i], e[1][i], e[2][i], "gpu", r[0][i], r[1][i], r[2][i], "len", len(P[i]), "hdr", P[i][:11].hex())
    # ---- crafted rangeproofs (tests/adversarial.py): forgeries with ring keys at infinity, results at infinity, suspect x coordinates -- as
    #      they are and with the mutations above on top (a flipped bit in a forged proof moves it off the crafted case in every possible way)
    from tests.adversarial import Crafter
    from tests.refapi import GENERATOR_H
    cr = Crafter(ref)
    gh = np.frombuffer(GENERATOR_H, np.uint8)
    CC, CP = [], []
    for t in range(max(4, n // 128)):
        rings = int(rng.choice([1, 2, 3, 4, 7, 16, 32]))
        kind = t % 4
        if kind == 0: c, p = cr.forge_infinity_keys(rng, rings)
        elif kind == 1: c, p = cr.forge_infinity_keys(rng, rings, neg=True)
        elif kind == 2: c, p = cr.forge_r_infinity(rng, max(rings, 2), int(rng.integers(0, max(rings, 2) - 1)))
        else: c, p = cr.sign(rng, min(rings, 4), int(rng.integers(0, 4 ** min(rings, 4))))
        for k in range(9):
            CC.append(np.frombuffer(c, np.uint8)); CP.append(p if k == 0 else mutate(p, rng, int(rng.integers(0, 8))))
    CC = np.stack(CC); CG = np.tile(gh, (len(CP), 1))
    e = ref.rangeproof_verify_many(CC, CP, CG, threads=16)
    r = eng.rangeproof_verify_batch(CC, CP, CG)
    bad = np.nonzero((e[0] != r[0]) | (e[1] != r[1]) | (e[2] != r[2]))[0]
    print("crafted rangeproofs: %d items, %d accepted, hand-backs %s, mismatches: %d" % (len(CP), int(e[0].sum()), eng.rp_handback()[2:], len(bad)), bad[:10])
    # ---- BIP-340
    sigs, msgs, pks = ref.make_schnorr(4 * n, rng, threads=16)
    for i in range(4 * n):
        k = i % 8
        if k == 1: sigs[i, int(rng.integers(0, 64))] ^= 1 << int(rng.integers(0, 8))
        elif k == 2: pks[i, int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
        elif k == 3: msgs[i, int(rng.integers(0, 32))] ^= 1
        elif k == 4: sigs[i, 32:] = np.frombuffer(N.to_bytes(32, "big"), np.uint8)
        elif k == 5: sigs[i, :32] = rng.integers(0, 256, 32, dtype=np.uint8)
        elif k == 6: pks[i] = rng.integers(0, 256, 32, dtype=np.uint8)
    e = ref.schnorr_verify_many(sigs, msgs, pks, threads=16)
    r = eng.schnorrsig_verify_batch(sigs, msgs, pks)
    print("bip340: %d items, %d accepted, mismatches: %d" % (4 * n, int(e.sum()), int((e != r).sum())))
    # ---- surjection proofs
    proofs, tags, outs = [], [], []
    base = [ref.make_surjection(rng, ni, nu) for (ni, nu) in ((1, 1), (2, 1), (3, 2), (3, 3), (5, 3), (8, 8), (17, 4), (40, 9))]
    for i in range(n):
        p, t, o = base[i % len(base)]
        t = t.copy(); o = o.copy()
        if i % 3 == 1: p = mutate(p, rng, int(rng.integers(0, 6)))
        if i % 7 == 2: t[int(rng.integers(0, t.shape[0])), int(rng.integers(0, 64))] ^= 1 << int(rng.integers(0, 8))
        if i % 11 == 3: o[int(rng.integers(0, 64))] ^= 1 << int(rng.integers(0, 8))
        proofs.append(p); tags.append(t); outs.append(o)
    e = np.array([ref.surjection_verify(p, t, o) for p, t, o in zip(proofs, tags, outs)], np.int32)
    r = eng.surjectionproof_verify_batch(proofs, tags, np.stack(outs))
    print("surjection: %d items, %d accepted, mismatches: %d" % (n, int(e.sum()), int((e != r).sum())))
    # ---- BP++ norm arguments
    for (gl, hl) in ((64, 8), (16, 16), (4, 32)):
        m = max(16, n // 16)
        pr, trs, rhos, gens, g_len, cvs, commits = ref.make_bppp(m, rng, gl, hl)
        pr = pr.copy(); rhos = rhos.copy(); cvs = cvs.copy(); commits = commits.copy(); trs = trs.copy()
        for i in range(m):
            k = i % 6
            if k == 1: pr[i, int(rng.integers(0, pr.shape[1]))] ^= 1 << int(rng.integers(0, 8))
            elif k == 2: rhos[i, int(rng.integers(0, 32))] ^= 1
            elif k == 3: cvs[i, int(rng.integers(0, cvs.shape[1])), int(rng.integers(0, 32))] ^= 1
            elif k == 4: commits[i, int(rng.integers(0, 33))] ^= 1 << int(rng.integers(0, 8))
            elif k == 5: pr[i, 65 * int(rng.integers(0, (pr.shape[1] - 64) // 65))] = int(rng.integers(0, 8))          # sign byte of an (X, R) pair
        e = ref.bppp_verify_many(pr, trs, rhos, gens, g_len, cvs, commits)
        r = eng.bppp_norm_product_verify_batch(pr, trs, rhos, gens, g_len, cvs, commits)
        print("bppp %d/%d: %d items, %d accepted, mismatches: %d" % (gl, hl, m, int(e.sum()), int((e != r).sum())))


def more(seed, n):
    """MSM with colliding points (P+P and P-P inside buckets), rewinding of mutated proofs / wrong nonces, tallies, half-aggregates"""
    rng = np.random.default_rng(seed + 1000); ref = Ref(); eng = Engine(0)
    P = 2**256 - 2**32 - 977
    pts = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(3)])
    neg = pts.copy()
    for i in range(3):
        neg[i, 32:] = np.frombuffer(((P - int.from_bytes(pts[i, 32:].tobytes(), "big")) % P).to_bytes(32, "big"), np.uint8)
    pool = np.concatenate([pts, neg])
    bad = 0
    for m in (300, 2000, 20000):
        sel = pool[rng.integers(0, 6, m)]
        sc = rng.integers(0, 256, (m, 32), dtype=np.uint8)
        sc[rng.integers(0, m, m // 8)] = sc[0]                       # repeated scalars on repeated points
        sc[rng.integers(0, m, m // 16), :24] = 0                     # short scalars
        g = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
        e_xy, e_inf = ref.ecmult_multi(sc, sel, g, None)
        r_xy, r_inf = eng.ecmult_multi(sc, sel, g, None)
        bad += int(e_inf != r_inf or not np.array_equal(e_xy, r_xy))
    print("msm with colliding points: mismatches:", bad)
    # K independent sums in one call (s2k_ecmult_multi_many): ragged sizes, colliding points, zero scalars, infinite points, with and without G terms --
    # every sum against the reference's single call; and single sums cut into several launches (S2K_OPT_MSM_MAX_TERMS lowered)
    bad = 0; nsums = 0
    for rep in range(3):
        K = int(rng.integers(1, 48))
        sizes = [int(x) for x in rng.choice([0, 1, 2, 7, 31, 32, 33, 87, 88, 89, 300, 1000, 2500], K)]
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64); m = int(off[-1])
        fresh = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(24)])
        sel = np.concatenate([pool, fresh])[rng.integers(0, 30, max(m, 1))][:m]
        sc = rng.integers(0, 256, (m, 32), dtype=np.uint8)
        if m:
            sc[rng.integers(0, m, m // 8 + 1)] = sc[0]
            sc[rng.integers(0, m, m // 16 + 1)] = 0
        inf = (rng.integers(0, 20, m) == 0).astype(np.uint8)
        gs = rng.integers(0, 256, (K, 32), dtype=np.uint8) if rep != 1 else None
        r_xy, r_inf = eng.ecmult_multi_many(sc, sel, off, gs, inf)
        for s_ in range(K):
            lo, hi = int(off[s_]), int(off[s_ + 1])
            e_xy, e_inf = ref.ecmult_multi(sc[lo:hi], sel[lo:hi], None if gs is None else bytes(gs[s_]), inf[lo:hi])
            bad += int(e_inf != int(r_inf[s_]) or not np.array_equal(e_xy, r_xy[s_])); nsums += 1
    print("many sums in one call: %d sums, mismatches: %d" % (nsums, bad))
    bad = 0
    for m, cap in ((5000, 700), (40000, 9973)):
        sel = np.concatenate([pool, fresh])[rng.integers(0, 30, m)]
        sc = rng.integers(0, 256, (m, 32), dtype=np.uint8); sc[rng.integers(0, m, m // 8)] = sc[0]
        g = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
        e_xy, e_inf = ref.ecmult_multi(sc, sel, g, None)
        eng.set_option(10, cap)                                      # S2K_OPT_MSM_MAX_TERMS
        try:
            r_xy, r_inf = eng.ecmult_multi(sc, sel, g, None)
        finally:
            eng.set_option(10, 0)
        bad += int(e_inf != r_inf or not np.array_equal(e_xy, r_xy))
    print("sums cut into several launches: mismatches:", bad)
    # rewind
    C, PR, G, NN = [], [], [], []
    for kw in (dict(msg_len=100, min_bits=64), dict(msg_len=0, min_bits=0, exp=-1, values=rng.integers(0, 2**50, n // 8, dtype=np.uint64)), dict(msg_len=33, min_bits=9, exp=1, min_value=3),
               dict(msg_len=500, min_bits=32)):
        c, p, g, v, b, nn, msg = ref.make_rangeproofs_msg(n // 8, rng, threads=16, **kw)
        C.append(c); PR += p; G.append(g); NN.append(nn)
    C = np.concatenate(C); G = np.concatenate(G); NN = np.concatenate(NN)
    for i in range(len(PR)):
        if i % 5 == 1: NN[i, int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
        if i % 5 == 2: PR[i] = mutate(PR[i], rng, int(rng.integers(0, 8)))
    e = ref.rangeproof_rewind_many(C, PR, G, NN, msg_capacity=600, threads=16)
    r = eng.rangeproof_rewind_batch(C, PR, G, NN, msg_capacity=600)
    ok = e[0] == 1
    mism = int((e[0] != r[0]).sum()) + int((e[1][ok] != r[1][ok]).any(axis=1).sum()) + int((e[2][ok] != r[2][ok]).sum()) + sum(1 for a, b, o in zip(e[3], r[3], ok) if o and a != b)
    print("rewind: %d items, %d rewound, mismatches: %d" % (len(PR), int(ok.sum()), mism))
    # tallies
    tallies = []
    for i in range(n // 4):
        a, b = ref.make_balanced_tally(rng, int(rng.integers(1, 6)), int(rng.integers(1, 6)))
        k = i % 6
        if k == 1: b = b[:-1]
        elif k == 2: a = a.copy(); a[0, int(rng.integers(0, 33))] ^= 1 << int(rng.integers(0, 8))
        elif k == 3: a, b = b, a
        elif k == 4: a = np.concatenate([a, b[:1]]); b = np.concatenate([b, b[:1]])
        tallies.append((a, b))
    e = np.maximum(ref.pedersen_verify_tally_many(tallies), 0)
    r = eng.pedersen_verify_tally_batch(tallies)
    print("tallies: %d items, %d balanced, mismatches: %d" % (len(tallies), int(e.sum()), int((e != r).sum())))
    # half-aggregates
    mism = 0; acc = 0
    for t in range(24):
        m = int(rng.integers(1, 300))
        sigs, msgs, pks = ref.make_schnorr(m, rng, threads=8)
        agg = bytearray(ref.halfagg_aggregate(pks, msgs, sigs))
        k = t % 4
        if k == 1: agg[int(rng.integers(0, len(agg)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 2: msgs[int(rng.integers(0, m)), 3] ^= 1
        elif k == 3: pks[int(rng.integers(0, m)), int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
        ee = max(0, ref.halfagg_verify(pks, msgs, bytes(agg), m)); rr = eng.schnorrsig_aggverify(pks, msgs, bytes(agg), n=m)
        mism += int(ee != rr); acc += ee
    print("half-aggregates: 24 items, %d accepted, mismatches: %d" % (acc, mism))


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "more":
        more(int(sys.argv[1]), int(sys.argv[2])); sys.exit(0)
    main()
