"""GPU tier: the reference's result-pinning known-answer tests for THIS path, reproduced through the engine's C ABI
(BASELINE.md section 3.7 names them as the parity gate):

  * run_ecmult_chain        (reference src/tests.c:4617-4674): X <- xn*X + gn*G iterated from a fixed point with
                            xn *= 0x1337, gn *= 0x7113; after 20000 iterations X must be the constant the reference hard-codes,
                            and must equal (prod xn)*A + (accumulated gn)*G computed directly;
  * run_ecmult_constants    (src/tests.c:5792-5951): SHA-256 digests (computed by an independent implementation and hard-coded in
                            the reference) over the uncompressed serialisations of x*G for three families of scalars, where x*G is
                            computed "in many different ways" that must all agree (test_ecmult_accumulate :5800-5830).
Both are run through s2k_ecmult_batch (= secp256k1_ecmult) AND s2k_ecmult_multi (= secp256k1_ecmult_multi_var).
The constants below are test vectors quoted from those reference lines."""
import hashlib

import numpy as np
import pytest

from tests.refapi import G_XY, N

pytestmark = pytest.mark.gpu

CHAIN_A = bytes.fromhex("8b30bbe9ae2a990696b22f670709dff3727fd8bc04d3362c6c7bf458e2846004"
                        "a357ae915c4a65281309edf20504740f0eb3343990216b4f81063cb65f2f7e0f")
CHAIN_XN = 0x84cc5452f7fde1edb4d38a8ce9b1b84ccef31f146e569be9705d357a42985407
CHAIN_GN = 0xa1e58d22553dcd42b23980625d4c57a96e9323d42b3152e5ca2c3990edc7c9de
CHAIN_RP = bytes.fromhex("D6E96687F9B10D092A6F35439D86CEBEA4535D0D409F53586440BD74B933E830"
                         "B95CBCA2C77DA786539BE8FD53354D2D3B4F566AE658045407ED6015EE1B2A88")
EXP_6BIT20 = "68b6ed6f28cac97f8e8bd6c06179346e5a8f2bbc3e1fc52e2ad045677f95958e"
EXP_8BIT8 = "8b658eea86ae3c9590b677a48c76d9ecf5ab8a2ffddb19121aeee6b76e053fc6"
EXP_2BIT = "e4711b4d141e6848b7af472b4cd204143a7587601af96360d0cb1faa859ab7b4"


def _b(v):
    return int(v % N).to_bytes(32, "big")


@pytest.mark.parametrize("via", ["ecmult_batch", "ecmult_multi"])
def test_run_ecmult_chain_constant(engine, via):
    x = CHAIN_A; xn, gn = CHAIN_XN, CHAIN_GN; ae, ge = 1, 0
    for i in range(20000):
        if via == "ecmult_batch":
            r, inf = engine.ecmult_batch(np.frombuffer(x, np.uint8), np.frombuffer(_b(xn), np.uint8), ng=np.frombuffer(_b(gn), np.uint8))
            assert inf[0] == 0; x = r[0].tobytes()
        else:
            r, inf = engine.ecmult_multi(np.frombuffer(_b(xn), np.uint8), np.frombuffer(x, np.uint8), g_sc=_b(gn))
            assert inf == 0; x = r.tobytes()
        ae = ae * xn % N; ge = (ge * xn + gn) % N
        xn = xn * 0x1337 % N; gn = gn * 0x7113 % N
    assert x == CHAIN_RP                                                # tests.c:4659-4666 (i == 19999)
    r, inf = engine.ecmult_batch(np.frombuffer(CHAIN_A, np.uint8), np.frombuffer(_b(ae), np.uint8), ng=np.frombuffer(_b(ge), np.uint8))
    assert inf[0] == 0 and r[0].tobytes() == x                          # tests.c:4669-4671
    r, inf = engine.ecmult_multi(np.frombuffer(_b(ae), np.uint8), np.frombuffer(CHAIN_A, np.uint8), g_sc=_b(ge))
    assert inf == 0 and r.tobytes() == x


def _xg_many_ways(engine, scalars, multi_every=1):
    """x*G for every scalar, computed the ways test_ecmult_accumulate does that touch this path; returns the serialisations"""
    n = len(scalars)
    sc = np.frombuffer(b"".join(_b(s) for s in scalars), np.uint8).reshape(n, 32)
    g = np.tile(np.frombuffer(G_XY, np.uint8), (n, 1)); zero = np.zeros((n, 32), np.uint8)
    r1, i1 = engine.ecmult_batch(g, sc, ng=None)                                         # ecmult(gj, x, NULL)
    r2, i2 = engine.ecmult_batch(g, sc, ng=zero)                                         # ecmult(gj, x, 0)
    r3, i3 = engine.ecmult_batch(g, zero, ng=sc, a_inf=np.ones(n, np.uint8))             # ecmult(inf, 0, x)
    assert np.array_equal(r1, r2) and np.array_equal(r1, r3) and np.array_equal(i1, i2) and np.array_equal(i1, i3)
    for k in range(0, n, multi_every):
        r4, i4 = engine.ecmult_multi(np.zeros((0, 32), np.uint8), np.zeros((0, 64), np.uint8), g_sc=_b(scalars[k]))      # multi_var(x, no terms)
        r5, i5 = engine.ecmult_multi(sc[k], g[k], g_sc=_b(0))                                                            # multi_var(0, (x, G))
        assert i4 == i1[k] and i5 == i1[k] and r4.tobytes() == r1[k].tobytes() and r5.tobytes() == r1[k].tobytes(), k
    return [b"\x00" if i1[k] else b"\x04" + r1[k].tobytes() for k in range(n)]


def _sha_family(prefix, iters):
    xs = [0, 1, N - 1]
    for i in range(iters):
        xs.append(int.from_bytes(hashlib.sha256(prefix.to_bytes(4, "little") + i.to_bytes(2, "little")).digest(), "big") % N)
    return xs


def test_run_ecmult_constants_sha_digests(engine):
    for prefix, iters, exp in ((4808378, 1024, EXP_6BIT20), (1607366309, 2048, EXP_8BIT8)):       # tests.c:5937-5946
        ser = _xg_many_ways(engine, _sha_family(prefix, iters))
        assert hashlib.sha256(b"".join(ser)).hexdigest() == exp


def test_run_ecmult_constants_2bit_digest(engine):
    xs = []
    for i in range(37):
        xs += [i, (N - i) % N]
    for i in range(256):
        for j in range(1, 256, 2):
            xs.append((j << i) % N)
    ser = _xg_many_ways(engine, xs, multi_every=64)                     # the multi_var forms on a 1/64 sample (one MSM call each)
    assert hashlib.sha256(b"".join(ser)).hexdigest() == EXP_2BIT         # tests.c:5847-5852
