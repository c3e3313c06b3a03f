"""GPU parity: secp256k1_schnorrsig_aggverify_amd (half-aggregate verification as one (2n+1)-term MSM) vs the reference's
secp256k1_schnorrsig_aggverify: the spec vectors (modules/schnorrsig_halfagg/tests_impl.h:73-168), reference-generated aggregates
of every size class (per-lane small path, bucket path, odd/even n for the randomizer hash's block alignment) and mutations."""
import json
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_golden_vectors(engine):
    for v in json.load(open(os.path.join(HERE, "golden", "halfagg_vectors.json")))["vectors"]:
        r = engine.schnorrsig_aggverify(bytes.fromhex(v["pks"]), bytes.fromhex(v["msgs"]), bytes.fromhex(v["aggsig"]), n=v["n"])
        assert r == v["result"], v["name"]


@pytest.mark.parametrize("n", [1, 2, 3, 50, 95, 96, 97, 1000, 4097])
def test_aggregates_and_mutations(engine, ref, n):
    rng = np.random.default_rng(900 + n)
    sigs, msgs, pks = ref.make_schnorr(n, rng)
    agg = ref.halfagg_aggregate(pks, msgs, sigs)
    assert engine.schnorrsig_aggverify(pks, msgs, agg) == 1
    for k in range(7):
        a = bytearray(agg); m = msgs.copy(); p = pks.copy()
        i = int(rng.integers(0, n))
        if k == 0: a[32 * i + int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))        # some r_i
        elif k == 1: a[32 * n + int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))      # s
        elif k == 2: m[i, int(rng.integers(0, 32))] ^= 1
        elif k == 3: p[i, int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))            # key (may stop being a valid x)
        elif k == 4: a[32 * n:] = b"\xff" * 32                                                 # s >= group order
        elif k == 5: a = a[:-32]                                                               # wrong length
        elif k == 6 and n >= 2: m[[0, 1]] = m[[1, 0]]                                          # order matters
        exp = max(0, ref.halfagg_verify(p, m, bytes(a), n))
        assert engine.schnorrsig_aggverify(p, m, bytes(a), n=n) == exp, k
    # 64-byte key objects (pk_format 1): x | y little-endian limbs as secp256k1_xonly_pubkey holds them
    if n <= 100:
        objs = ref.xonly_objects(pks)
        assert engine.schnorrsig_aggverify(objs, msgs, agg, pk_format=1) == 1


def test_large_aggregate_throughput(engine, ref):
    rng = np.random.default_rng(4)
    n = 1 << 15
    sigs, msgs, pks = ref.make_schnorr(n, rng, threads=16)
    agg = ref.halfagg_aggregate(pks, msgs, sigs)
    assert engine.schnorrsig_aggverify(pks, msgs, agg) == 1
    t = time.perf_counter(); r = engine.schnorrsig_aggverify(pks, msgs, agg); dt = time.perf_counter() - t
    assert r == 1
    bad = bytearray(agg); bad[32 * (n - 1)] ^= 1
    assert engine.schnorrsig_aggverify(pks, msgs, bytes(bad)) == max(0, ref.halfagg_verify(pks, msgs, bytes(bad), n))
    print(f"\nhalf-aggregate verify, n={n}: {dt * 1e3:.1f} ms ({n / dt:.3g} signatures/s incl. H2D)")
