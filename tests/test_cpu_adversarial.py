"""CPU tier: the crafted proofs of tests/adversarial.py do what they claim -- against the reference, against a verifier that lacks the
reference's infinity rejections, and through the host build of the device code (both forms of the rings stage).  The GPU tier
(tests/test_gpu_rangeproof_adversarial.py) then drives the same proofs through the real kernels."""
import ctypes
import json
import os

import numpy as np
import pytest

from tests.adversarial import Crafter, SurjectionCrafter
from tests.refapi import GENERATOR_H
from tests.test_cpu_oracle import _emu_rp, _emu_rp_shared, emu  # noqa: F401  (fixture)

HERE = os.path.dirname(os.path.abspath(__file__))
GH = np.frombuffer(GENERATOR_H, np.uint8).reshape(1, 64)


def _ref1(ref, c, p, g=GH):
    r, mn, mx = ref.rangeproof_verify_many(np.frombuffer(c, np.uint8).reshape(1, 33), [p], g)
    return int(r[0]), int(mn[0]), int(mx[0])


def test_signer_matches_the_reference(ref):
    """the Python signer's hashing and serialisations are the reference's: what it signs verifies there (and mutations do not)"""
    rng = np.random.default_rng(41); cr = Crafter(ref)
    for rings, value in ((1, 0), (1, 3), (2, 9), (4, 201), (6, 4095)):
        c, p = cr.sign(rng, rings, value)
        assert _ref1(ref, c, p) == (1, 0, 4**rings - 1) and cr.unchecked_verify(c, p) == 1
        q = bytearray(p); q[-1] ^= 1
        assert _ref1(ref, c, bytes(q))[0] == 0 and cr.unchecked_verify(c, bytes(q)) == 0


def test_forgeries_need_the_infinity_rejection(ref, emu):
    """every ring of a forged proof has a key at infinity: it satisfies all verification equations (a verifier that evaluates
    e*infinity instead of rejecting it says 1), the reference says 0 (borromean_impl.h:78), and so do both forms of the rings stage"""
    rng = np.random.default_rng(42); cr = Crafter(ref)
    for rings, js in ((1, None), (2, [1, 3]), (3, [3, 2, 1]), (4, None), (32, None)):
        c, p = cr.forge_infinity_keys(rng, rings, js)
        if rings <= 4:
            assert cr.unchecked_verify(c, p) == 1
        want = _ref1(ref, c, p)
        assert want[0] == 0
        ca = np.frombuffer(c, np.uint8)
        assert _emu_rp(emu, ca, p, GH[0]) == want
        r, fast = _emu_rp_shared(emu, ca, p, GH[0])
        assert r == want and fast == 0                              # every ring is suspect: none may be served by the shared form
    # the mirrored lift (C = +j*B: same x, no infinite key), a result at infinity
    for c, p in (cr.forge_infinity_keys(rng, 3, neg=True), cr.forge_r_infinity(rng, 3, 1), cr.forge_r_infinity(rng, 2, 0)):
        want = _ref1(ref, c, p); ca = np.frombuffer(c, np.uint8)
        assert want[0] == 0 and _emu_rp(emu, ca, p, GH[0]) == want and _emu_rp_shared(emu, ca, p, GH[0])[0] == want
    # another generator
    g2 = ref.rand_point(rng); cr2 = Crafter(ref, g2); g2a = np.frombuffer(g2, np.uint8).reshape(1, 64)
    c, p = cr2.forge_infinity_keys(rng, 2)
    want = _ref1(ref, c, p, g2a); ca = np.frombuffer(c, np.uint8)
    assert want[0] == 0 and cr2.unchecked_verify(c, p) == 1 and _emu_rp(emu, ca, p, g2a[0]) == want and _emu_rp_shared(emu, ca, p, g2a[0])[0] == want


def test_exceptional_fixture(ref, emu):
    """tests/golden/rangeproof_exceptional.json (made by tests/golden/make_exceptional.py): the reference accepts it, and its s_0 has
    the stated property"""
    from tests.refapi import N
    fx = json.load(open(os.path.join(HERE, "golden", "rangeproof_exceptional.json")))
    c = bytes.fromhex(fx["commit33"]); p = bytes.fromhex(fx["proof"])
    assert _ref1(ref, c, p) == (1, fx["min_value"], fx["max_value"]) and fx["result"] == 1
    s0 = int.from_bytes(p[2 + 32:2 + 64], "big")                    # header (2 bytes, no sign byte for one ring), e0, then s_0
    assert Crafter.fixed_base_top(s0, fx["digit_bits"])[0] == fx["top_digit"] and fx["digit_bits"] == 26
    ca = np.frombuffer(c, np.uint8)
    assert _emu_rp(emu, ca, p, GH[0])[0] == 1 and _emu_rp_shared(emu, ca, p, GH[0])[0][0] == 1


def test_surjection_forgeries(ref, emu):
    rng = np.random.default_rng(43); sc = SurjectionCrafter(ref)
    p, t, o = ref.make_surjection(rng, 5, 3)
    assert ref.surjection_verify(p, t, o) == 1 and sc.unchecked_verify(p, t, o) == 1
    ev = lambda p, t, o: emu.emu_surjection_verify(p, ctypes.c_size_t(len(p)), t.tobytes(), ctypes.c_size_t(t.shape[0]), o.tobytes())
    for args in ((3, [0, 2], 0), (3, [0, 2], 1), (8, [1, 4, 7], 1), (8, [1, 4, 7], 2), (1, [0], 0)):
        p, t, o = sc.forge_infinity(rng, *args)
        assert sc.unchecked_verify(p, t, o) == 1 and ref.surjection_verify(p, t, o) == 0 and ev(p, t, o) == 0
    p, t, o = sc.forge_r_infinity(rng, 4, [1, 3])
    assert ref.surjection_verify(p, t, o) == 0 and ev(p, t, o) == 0
