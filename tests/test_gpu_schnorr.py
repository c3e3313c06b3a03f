"""GPU parity: secp256k1_schnorrsig_verify_batch vs the reference's secp256k1_schnorrsig_verify per item, on the BIP-340
vectors the reference's tests carry (src/modules/schnorrsig/tests_impl.h:208-807) and on random signatures with a
pseudo-random fraction corrupted (BASELINE config 2 construction)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_bip340_vectors(engine):
    vecs = json.load(open(os.path.join(HERE, "golden", "bip340_vectors.json")))["vectors"]
    by_len = {}
    for v in vecs:
        by_len.setdefault(len(v["msg"]) // 2, []).append(v)
    for msglen, vs in by_len.items():
        sigs = np.stack([np.frombuffer(bytes.fromhex(v["sig"]), np.uint8) for v in vs])
        pks = np.stack([np.frombuffer(bytes.fromhex(v["pk"]), np.uint8) for v in vs])
        msgs = np.stack([np.frombuffer(bytes.fromhex(v["msg"]), np.uint8) for v in vs]) if msglen else np.zeros((len(vs), 0), np.uint8)
        res = engine.schnorrsig_verify_batch(sigs, msgs, pks, msglen=msglen)
        assert list(res) == [v["result"] for v in vs], msglen


def test_random_batch(engine, ref):
    rng = np.random.default_rng(99)
    n = 4096
    sigs, msgs, pks = ref.make_schnorr(n, rng)
    # corrupt a fixed pseudo-random 1/16: flip a bit of s, of r, of the message, or of the key
    idx = rng.choice(n, n // 16, replace=False)
    for k, i in enumerate(idx):
        if k % 4 == 0: sigs[i, 32 + int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
        elif k % 4 == 1: sigs[i, int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
        elif k % 4 == 2: msgs[i, int(rng.integers(0, 32))] ^= 1
        else: pks[i, int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
    sigs[7, 32:] = 0xFF      # s >= n
    sigs[9, :32] = 0xFF      # r >= p
    exp = ref.schnorr_verify_many(sigs, msgs, pks, threads=8)
    res = engine.schnorrsig_verify_batch(sigs, msgs, pks)
    assert np.array_equal(res, exp)
    assert exp.sum() >= n - n // 16 - 2 and exp.sum() < n
    # the same keys as 64-byte secp256k1_xonly_pubkey objects (pk_format 1); keys that do not parse have no such form
    good = np.nonzero(ref.xonly_valid(pks[:600]))[0]
    res1 = engine.schnorrsig_verify_batch(sigs[good], msgs[good], ref.xonly_objects(pks[good]), pk_format=1)
    assert np.array_equal(res1, exp[good]) and len(good) > 500


def test_config2_full_size(engine, ref):
    """BASELINE config 2 at full size: 2^16 signatures, a fixed pseudo-random 1/256 corrupted (one bit of s or r)"""
    rng = np.random.default_rng(216)
    n = 1 << 16
    sigs, msgs, pks = ref.make_schnorr(n, rng, threads=16)
    bad = rng.choice(n, n // 256, replace=False)
    for k, i in enumerate(bad):
        sigs[i, (32 if k & 1 else 0) + int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
    exp = ref.schnorr_verify_many(sigs, msgs, pks, threads=16)
    res = engine.schnorrsig_verify_batch(sigs, msgs, pks)
    assert np.array_equal(res, exp) and exp.sum() == n - n // 256
