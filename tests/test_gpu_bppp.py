"""GPU parity: secp256k1_bppp_norm_product_verify_batch vs the reference's (static) secp256k1_bppp_rangeproof_norm_product_verify:
the 13 accept/reject vectors of src/modules/bppp/test_vectors/verify.h, and freshly proven arguments (reference prover) of the
sizes BASELINE config 4 names (g_len 64, h_len 8) and the tested maximum (64/64), with mutations."""
import ctypes
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

SHA256_INIT_STATE = (np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], "<u4").tobytes()
                     + b"\0" * 64 + b"\0" * 8)      # secp256k1_sha256_initialize(): the transcript the vector driver uses (tests_impl.h:550)


def test_verify_vectors(engine):
    g = json.load(open(os.path.join(HERE, "golden", "bppp_verify_vectors.json")))
    gens = np.frombuffer(bytes.fromhex(g["gens"]), np.uint8)
    tr = np.frombuffer(SHA256_INIT_STATE, np.uint8)
    for v in g["vectors"]:
        proof = np.frombuffer(bytes.fromhex(v["proof"]), np.uint8)
        cvec = np.frombuffer(b"".join(bytes.fromhex(c) for c in v["c_vec"]), np.uint8)
        nlen, clen = v["n_vec_len"], len(v["c_vec"])
        res = engine.bppp_norm_product_verify_batch(proof, tr, np.frombuffer(bytes.fromhex(v["rho"]), np.uint8), gens[:33 * (nlen + clen)], nlen, cvec,
                                                    np.frombuffer(bytes.fromhex(v["commit33"]), np.uint8))
        assert res[0] == v["result"], v["index"]


@pytest.mark.parametrize("g_len,h_len,n", [(64, 8, 24), (64, 64, 8), (1, 1, 4), (2, 16, 4), (8, 1, 4), (32, 8, 6)])      # (72 / 128 / <= 36 / 40 generators: 18- / 17- / 20- / 19-bit set tables)
def test_random_proofs(engine, ref, g_len, h_len, n):
    rng = np.random.default_rng(g_len * 100 + h_len)
    proofs, trs, rhos, gens, gl, cvs, commits = ref.make_bppp(n, rng, g_len, h_len)
    # mutate a third: proof bytes, rho, c_vec, commitment
    proofs = proofs.copy(); rhos = rhos.copy(); cvs = cvs.copy(); commits = commits.copy()
    for i in range(n):
        if i % 3 == 1:
            k = i // 3 % 4
            if k == 0: proofs[i, int(rng.integers(0, proofs.shape[1]))] ^= 1 << int(rng.integers(0, 8))
            elif k == 1: rhos[i, 31] ^= 1
            elif k == 2: cvs[i, 0, 31] ^= 1
            else: commits[i, 5] ^= 1
    exp = ref.bppp_verify_many(proofs, trs, rhos, gens, gl, cvs, commits)
    res = engine.bppp_norm_product_verify_batch(proofs, trs, rhos, gens, gl, cvs, commits)
    assert np.array_equal(res, exp)
    assert exp[0] == 1 and exp.sum() < n


def test_generator_set_cache_switching(engine, ref):
    """the fixed-base table is keyed by the serialised generator set: alternating between two sets (and a shape change with the same
    prefix of generators) must rebuild it each time and never reuse stale entries"""
    rng = np.random.default_rng(404)
    a = ref.make_bppp(6, rng, 8, 2)
    b = ref.make_bppp(6, rng, 4, 4)            # different shape, its own generators
    for args in (a, b, a, b, a):
        proofs, trs, rhos, gens, gl, cvs, commits = args
        proofs = proofs.copy(); proofs[1, 7] ^= 1
        exp = ref.bppp_verify_many(proofs, trs, rhos, gens, gl, cvs, commits)
        res = engine.bppp_norm_product_verify_batch(proofs, trs, rhos, gens, gl, cvs, commits)
        assert np.array_equal(res, exp) and exp.sum() == 5
    # same length, one generator replaced by another valid point: every proof must now fail, and verify again with the original set
    proofs, trs, rhos, gens, gl, cvs, commits = a
    g2 = gens.copy(); g2[3] = g2[4]              # gens: (n_gens, 33) serialised points
    assert not engine.bppp_norm_product_verify_batch(proofs, trs, rhos, g2, gl, cvs, commits).any()
    assert not ref.bppp_verify_many(proofs, trs, rhos, g2, gl, cvs, commits).any()
    assert engine.bppp_norm_product_verify_batch(proofs, trs, rhos, gens, gl, cvs, commits).all()


def test_config4_full_size(engine, ref):
    """BASELINE config 4 at full size: 2^12 norm-argument proofs over 64 + 8 generators (256 distinct proofs, replicated), every 19th
    one corrupted in the proof bytes, rho, c_vec or the commitment"""
    rng = np.random.default_rng(412)
    base = ref.make_bppp(256, rng, 64, 8)
    reps = 16; n = 256 * reps
    proofs = np.concatenate([base[0]] * reps); trs = np.concatenate([base[1]] * reps); rhos = np.concatenate([base[2]] * reps)
    cvs = np.concatenate([base[5]] * reps); commits = np.concatenate([base[6]] * reps)
    for i in range(0, n, 19):
        k = (i // 19) % 4
        if k == 0: proofs[i, int(rng.integers(0, proofs.shape[1]))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1: rhos[i, int(rng.integers(0, 32))] ^= 1
        elif k == 2: cvs[i, int(rng.integers(0, 8)), 31] ^= 1
        else: commits[i, int(rng.integers(1, 33))] ^= 1
    exp = ref.bppp_verify_many(proofs, trs, rhos, base[3], base[4], cvs, commits)
    res = engine.bppp_norm_product_verify_batch(proofs, trs, rhos, base[3], base[4], cvs, commits)
    assert np.array_equal(res, exp)
    assert exp.sum() == n - len(range(0, n, 19))


def test_bppp_commit_batch(engine, ref):
    """secp256k1_bppp_commit (static, bppp_norm_product_impl.h:105-151) for a batch, on the generator set's fixed-base tables, byte for
    byte against the reference's own function (through oracle/ref_shim.c): random vectors, zero vectors (v*G only / infinity), and a
    table-free generator set size."""
    import ctypes
    rng = np.random.default_rng(88)
    sc = lambda *shape: (rng.integers(0, 256, shape + (32,), dtype=np.uint8) & np.array([0x7F] + [0xFF] * 31, np.uint8))
    for g_len, h_len, n in ((16, 4, 9), (64, 8, 5), (1, 1, 3)):
        gens = ref.bppp_generators(g_len + h_len)
        nv, lv, cv, mu = sc(n, g_len), sc(n, h_len), sc(n, h_len), sc(n)
        nv[1] = 0; lv[1] = 0                                   # commitment 1 = 0*G...: infinity -> 33 zero bytes
        if n > 2:
            nv[2] = 0; lv[2, 1:] = 0                           # one H term + v*G
        exp = np.zeros((n, 33), np.uint8)
        for i in range(n):
            cm = np.zeros(33, np.uint8)
            assert ref.lib.ref_bppp_commit(cm.ctypes.data_as(ctypes.c_void_p), gens.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(g_len + h_len),
                                           np.ascontiguousarray(nv[i]).ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(g_len),
                                           np.ascontiguousarray(lv[i]).ctypes.data_as(ctypes.c_void_p), np.ascontiguousarray(cv[i]).ctypes.data_as(ctypes.c_void_p),
                                           ctypes.c_size_t(h_len), np.ascontiguousarray(mu[i]).ctypes.data_as(ctypes.c_void_p)) == 1
            exp[i] = cm
        got, ok = engine.bppp_commit_batch(gens, g_len, nv, lv, cv, mu)
        assert ok.all() and np.array_equal(got, exp), (g_len, h_len)
        assert not got[1].any()
    # a generator that does not parse: no commitment is produced for the batch
    bad = ref.bppp_generators(8).copy(); bad[3, 0] = 7
    got, ok = engine.bppp_commit_batch(bad, 4, sc(2, 4), sc(2, 4), sc(2, 4), sc(2))
    assert not ok.any() and not got.any()
