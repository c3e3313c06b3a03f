"""GPU: batches larger than the engine's lanes-per-launch cap run as several launches over sub-ranges that reuse the same
per-lane table arena and scratch records.  An engine created with a tiny cap (S2K_OPT_MAX_LANES = 512: 16 rangeproofs / 512
signatures per launch) must reproduce the reference (and therefore the default engine) item for item, in order, with ragged
offsets and failing items straddling the chunk boundaries."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_engine(engine):
    from secp256k1_zkp_amd import Engine
    e = Engine(0)
    e.set_option(Engine.OPT_MAX_LANES, 512)
    return e


def test_ecmult_and_schnorr_split(small_engine, ref):
    rng = np.random.default_rng(11)
    n = 1500
    a = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(16)])[rng.integers(0, 16, n)]
    na = rng.integers(0, 256, (n, 32), dtype=np.uint8); ng = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    inf = (rng.integers(0, 50, n) == 0).astype(np.uint8)
    r_ref, i_ref = ref.ecmult_batch(a, na, ng, inf)
    r, i = small_engine.ecmult_batch(a, na, ng, inf)
    assert np.array_equal(i, i_ref) and np.array_equal(r, r_ref)
    r_ref, i_ref = ref.ecmult_batch(a, na, None, None)
    r, i = small_engine.ecmult_batch(a, na, None, None)
    assert np.array_equal(i, i_ref) and np.array_equal(r, r_ref)

    sigs, msgs, pks = ref.make_schnorr(n, rng)
    bad = rng.choice(n, 100, replace=False)
    sigs[bad, 40] ^= 1
    exp = ref.schnorr_verify_many(sigs, msgs, pks, threads=8)
    assert np.array_equal(small_engine.schnorrsig_verify_batch(sigs, msgs, pks), exp)
    assert exp.sum() == n - 100


def test_rangeproof_split(small_engine, ref):
    rng = np.random.default_rng(12)
    commits, proofs, gens = [], [], []
    for (mb, exp, minv, k) in ((64, 0, 0, 20), (8, 0, 3, 15), (32, 1, 0, 15), (1, 0, 0, 3)):
        c, p, g, _ = ref.make_rangeproofs(k, rng, min_bits=mb, exp=exp, min_value=minv)
        commits.append(c); proofs += p; gens.append(g)
    commits = np.concatenate(commits); gens = np.concatenate(gens)
    order = rng.permutation(len(proofs))
    commits, gens, proofs = commits[order], gens[order], [proofs[i] for i in order]
    for i in (0, 15, 16, 17, 31, 32, 52):          # failures either side of the 16-proof chunk boundaries
        q = bytearray(proofs[i]); q[len(q) // 2] ^= 4; proofs[i] = bytes(q)
    proofs[5] = b""; proofs[33] = proofs[33][:-1]
    e_res, e_mn, e_mx = ref.rangeproof_verify_many(commits, proofs, gens)
    res, mn, mx = small_engine.rangeproof_verify_batch(commits, proofs, gens)
    assert np.array_equal(res, e_res) and np.array_equal(mn, e_mn) and np.array_equal(mx, e_mx)
    assert 0 < e_res.sum() < len(proofs)


def test_surjection_and_bppp_split(small_engine, ref):
    rng = np.random.default_rng(13)
    proofs, tags, outs = [], [], []
    for k in range(700):
        if k < 12:
            p, t, o = ref.make_surjection(rng, 1 + k, 1 + k % 3 if k else 1)
        else:                                       # replicate (valid) proofs to cross the 512-lane boundary, corrupt some
            j = k % 12
            p, t, o = proofs[j], tags[j], outs[j]
            if k % 29 == 0:
                q = bytearray(p); q[-1] ^= 1; p = bytes(q)
        proofs.append(p); tags.append(t); outs.append(o)
    exp = np.array([ref.surjection_verify(p, t, o) for p, t, o in zip(proofs, tags, outs)], np.int32)
    assert np.array_equal(small_engine.surjectionproof_verify_batch(proofs, tags, np.stack(outs)), exp)
    assert 0 < exp.sum() < len(proofs)

    # 64+8 generators: 2 + 6*2 + 72 = 86 terms per proof -> 5 proofs per 512-lane launch
    pr, trs, rhos, gens, gl, cvs, commits = ref.make_bppp(13, rng, 64, 8)
    pr = pr.copy(); pr[4, 3] ^= 1; pr[5, 70] ^= 1; pr[12, 0] ^= 2
    exp = ref.bppp_verify_many(pr, trs, rhos, gens, gl, cvs, commits)
    assert np.array_equal(small_engine.bppp_norm_product_verify_batch(pr, trs, rhos, gens, gl, cvs, commits), exp)
    assert 0 < exp.sum() < 13


def test_rewind_split(small_engine, ref):
    """rewinding 45 proofs with 16 proofs per launch: the scratch (challenges, pads, ring nonces) is per launch, the outputs per batch"""
    rng = np.random.default_rng(14)
    C, P, G, N = [], [], [], []
    for kw in (dict(msg_len=50, min_bits=32), dict(msg_len=0, min_bits=0, exp=-1, values=np.arange(15, dtype=np.uint64)), dict(msg_len=9, min_bits=6, exp=1, min_value=3)):
        c, p, g, v, b, nn, m = ref.make_rangeproofs_msg(15, rng, **kw)
        C.append(c); P += p; G.append(g); N.append(nn)
    C = np.concatenate(C); G = np.concatenate(G); N = np.concatenate(N)
    order = rng.permutation(len(P)); C, G, N, P = C[order], G[order], N[order], [P[i] for i in order]
    N[[3, 16, 31]] ^= 1
    e = ref.rangeproof_rewind_many(C, P, G, N, msg_capacity=200, threads=8)
    r = small_engine.rangeproof_rewind_batch(C, P, G, N, msg_capacity=200)
    ok = e[0] == 1
    assert np.array_equal(r[0], e[0]) and ok.sum() == len(P) - 3
    assert np.array_equal(r[1][ok], e[1][ok]) and np.array_equal(r[2][ok], e[2][ok]) and [m for m, o in zip(r[3], ok) if o] == [m for m, o in zip(e[3], ok) if o]
