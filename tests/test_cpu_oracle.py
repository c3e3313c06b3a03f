"""CPU tier (-m "not gpu").  (1) pins the oracle: oracle/_ref (the unmodified reference) must reproduce every golden vector
extracted from the reference's own tests; (2) runs the per-lane *device* arithmetic compiled for the host with S2K_VERIFY on
(tests/host_emul) against the oracle: field / scalar / group / ecmult primitives, the five rangeproof stages, BIP-340, the
bucket MSM and the BP++ norm argument; (3) checks the C ABI: the product library loads and exports every symbol declared in
include/secp256k1_zkp_amd.h, and fails loudly (no CPU fallback) when no GPU is present."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from tests.refapi import GENERATOR_H, G_XY, N, P

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def emu():
    path = os.path.join(HERE, "host_emul", "libs2k_hostemu.so")
    if not os.path.exists(path):
        pytest.skip("host emulation library not built (run __graft_entry__.build())")
    return ctypes.CDLL(path)


def _b(v):
    return int(v).to_bytes(32, "big")


def _golden(name):
    return json.load(open(os.path.join(HERE, "golden", name)))


# ---- (1) the oracle reproduces the reference's own known answers ------------------------------------------------------
def test_ref_rangeproof_golden(ref):
    vecs = _golden("rangeproof_vectors.json")["vectors"]
    n = len(vecs)
    commits = np.stack([np.frombuffer(bytes.fromhex(v["commit33"]), np.uint8) for v in vecs])
    gens = np.frombuffer(GENERATOR_H * n, np.uint8).reshape(n, 64)
    res, mn, mx = ref.rangeproof_verify_many(commits, [bytes.fromhex(v["proof"]) for v in vecs], gens)
    for i, v in enumerate(vecs):
        assert res[i] == v["result"] and int(mn[i]) == int(v["min_value"]) and int(mx[i]) == int(v["max_value"]), v["name"]


def test_ref_bip340_golden(ref):
    vecs = _golden("bip340_vectors.json")["vectors"]
    for v in vecs:
        msg = bytes.fromhex(v["msg"])
        r = ref.schnorr_verify_many(np.frombuffer(bytes.fromhex(v["sig"]), np.uint8), np.frombuffer(msg, np.uint8) if msg else np.zeros(1, np.uint8),
                                    np.frombuffer(bytes.fromhex(v["pk"]), np.uint8), msglen=len(msg))
        assert r[0] == v["result"]


def test_ref_bppp_golden(ref):
    g = _golden("bppp_verify_vectors.json")
    gens = bytes.fromhex(g["gens"])
    st = ctypes.create_string_buffer(104)
    assert ref.lib.ref_sha256_state_size() == 104
    ref.lib.ref_sha256_state_from_prefix(st, b"", ctypes.c_size_t(0))
    for v in g["vectors"]:
        proof = bytes.fromhex(v["proof"]); cvec = b"".join(bytes.fromhex(c) for c in v["c_vec"]); nlen = v["n_vec_len"]; clen = len(v["c_vec"])
        r = ref.lib.ref_bppp_norm_verify(proof, ctypes.c_size_t(len(proof)), st.raw, bytes.fromhex(v["rho"]), gens[:33 * (nlen + clen)],
                                         ctypes.c_size_t(nlen + clen), ctypes.c_size_t(nlen), cvec, ctypes.c_size_t(clen), bytes.fromhex(v["commit33"]))
        assert r == v["result"], v["index"]


# ---- (2) device arithmetic, host-compiled with magnitude checks, vs the oracle -----------------------------------------------
def _call(lib, name, nout, *args):
    outs = [ctypes.create_string_buffer(n) for n in nout]
    r = getattr(lib, name)(*outs, *args)
    return r, [o.raw for o in outs]


def test_emu_field_scalar(emu, ref):
    rng = np.random.default_rng(7)
    edge = [0, 1, 2, P - 1, P - 2, P, P + 1, 2**256 - 1, 2**255, 977, 2**32 + 977, N, N - 1, N + 1, (P + 1) // 2, 2**232, 2**256 - 2**32]
    cs = [_b(e % 2**256) for e in edge] + [bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(200)]
    for i in range(len(cs)):
        a, c = cs[i], cs[(i * 7 + 3) % len(cs)]
        for name, args in (("fe_mul", (a, c)), ("fe_sqr", (a,)), ("fe_add", (a, c)), ("fe_negate", (a,)), ("fe_inv", (a,)), ("fe_sqrt", (a,)),
                           ("scalar_mul", (a, c)), ("scalar_add", (a, c)), ("scalar_negate", (a,)), ("scalar_set_b32", (a,))):
            r1, o1 = _call(ref.lib, "ref_" + name, [32], *args); r2, o2 = _call(emu, "emu_" + name, [32], *args)
            assert o1 == o2, name
            if name in ("fe_sqrt", "scalar_set_b32"):
                assert r1 == r2, name
        assert _call(ref.lib, "ref_scalar_split_lambda", [32, 32], a)[1] == _call(emu, "emu_scalar_split_lambda", [32, 32], a)[1]
        if i < 30:
            assert _call(ref.lib, "ref_scalar_inverse", [32], a)[1] == _call(emu, "emu_scalar_inverse", [32], a)[1]


def test_emu_group_ecmult(emu, ref):
    rng = np.random.default_rng(8)
    pts = [ref.rand_point(rng) for _ in range(12)] + [G_XY]
    neg = lambda p: p[:32] + _b((P - int.from_bytes(p[32:], "big")) % P)
    for i, a in enumerate(pts):
        for c in (pts[(i + 1) % len(pts)], a, neg(a)):
            for ai in (0, 1):
                for bi in (0, 1):
                    e1 = _call(ref.lib, "ref_ge_add", [64], a, ai, c, bi)
                    assert e1 == _call(emu, "emu_ge_add", [64], a, ai, c, bi)
                    za, zb = bytes(rng.integers(0, 256, 32, dtype=np.uint8)), bytes(rng.integers(0, 256, 32, dtype=np.uint8))
                    assert e1 == _call(emu, "emu_gej_add_var", [64], a, ai, c, bi, za, zb)
        assert _call(ref.lib, "ref_ge_double", [64], a, 0) == _call(emu, "emu_ge_double", [64], a, 0)
    sc_edge = [_b(v) for v in (0, 1, 2, 3, N - 1, N - 2, 255, 256, 257, 2**128, 2**128 - 1, N // 2, N // 2 + 1)]
    cases = []
    for i in range(40):
        na = bytes(rng.integers(0, 256, 32, dtype=np.uint8)) if i % 3 else sc_edge[i % len(sc_edge)]
        ng = bytes(rng.integers(0, 256, 32, dtype=np.uint8)) if i % 4 else sc_edge[(i * 5) % len(sc_edge)]
        cases.append((pts[i % len(pts)], 0, na, ng))
    for sa in (1, 2, 3, 255, 256):          # accumulator meets table points: P+P / P-P inside the loop
        for sg in (1, 2, 255, 256, N - 1, N - 2):
            cases.append((G_XY, 0, _b(sa), _b(sg))); cases.append((neg(G_XY), 0, _b(sa), _b(sg)))
    cases += [(G_XY, 1, _b(5), _b(7)), (G_XY, 0, _b(0), _b(0)), (G_XY, 0, _b(5), None), (pts[0], 0, sc_edge[4], None)]
    for (a, ai, na, ng) in cases:
        e1 = _call(ref.lib, "ref_ecmult", [64], a, ai, na, ng)
        for z in (None, bytes(rng.integers(0, 256, 32, dtype=np.uint8))):
            assert e1 == _call(emu, "emu_ecmult", [64], a, ai, na, ng, z)


LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
SPLIT_EDGE_SCALARS = [1, 2, 3, 15, 16, 17, 2**64 - 1, 2**64, 2**64 + 1, 2**65, 2**65 - 1, 2**127, 2**128, 2**128 + 2**64, 2**129 - 1, N - 1, N - 2, (N - 1) // 2,
                      LAMBDA, LAMBDA + 1, N - LAMBDA, (LAMBDA * 2**64) % N, 2**255, N - 2**64, 0]


def test_emu_ecmult_two_piece_form(emu, ref):
    """ecmult_lane_split (given T = 2^64 A; used by the rangeproof rings) with its fallback, on edge scalars of the piece split --
    multiples of 2^64, the lambda values, 0 (no variable part: the form refuses and the caller falls back) -- and random ones."""
    rng = np.random.default_rng(44)
    pts = [ref.rand_point(rng) for _ in range(6)] + [G_XY]
    took = ctypes.c_int(0)
    n_split = 0
    scal = [_b(v) for v in SPLIT_EDGE_SCALARS] + [bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(12)]
    for i, na in enumerate(scal):
        a = pts[i % len(pts)]
        for ng in (bytes(rng.integers(0, 256, 32, dtype=np.uint8)), _b(0), None):
            e1 = _call(ref.lib, "ref_ecmult", [64], a, 0, na, ng)
            z = bytes(rng.integers(0, 256, 32, dtype=np.uint8)) if i % 2 else None
            got = _call(emu, "emu_ecmult_split", [64], ctypes.byref(took), a, na, ng, z)
            assert got == e1, (i, ng is None)
            n_split += took.value
            if int.from_bytes(na, "big") == 0:
                assert took.value == 0
    assert n_split >= 2 * len(scal)          # the two-piece form itself produced (nearly) all of these


def fixed_base_edge_scalars(D, rng, count):
    """scalars whose SIGNED D-bit fixed-base digits (csrc/ecmult.h, "generator table") sit on every edge: the recoded scalar s' = s + K is built
    window by window from {0, 1, half - 1, half, half + 1, 2^D - 1} (digits -half, -half + 1, -1, 0, +1, half - 1) and random values, and
    s = s' - K is kept when it is a scalar; plus n - 1, 1, and the values around 2^256 - K where s' needs its 257th bit."""
    W = (256 + D - 1) // D
    half = 1 << (D - 1)
    K = sum(1 << (D - 1 + D * w) for w in range(W - 1))
    out = [N - 1, N - 2, 1, 2, K % N, (K + 1) % N, (N - K) % N, (1 << 256) - K - 1 if (1 << 256) - K - 1 < N else N - 3]
    edges = [0, 1, half - 1, half, half + 1, (1 << D) - 1]
    top_bits = 256 - D * (W - 1)
    while len(out) < count:
        sp = 0
        for w in range(W - 1):
            t = edges[int(rng.integers(0, 6))] if rng.integers(0, 4) else int(rng.integers(0, 1 << D))
            sp |= t << (D * w)
        sp |= int(rng.integers(0, 1 << top_bits)) << (D * (W - 1))
        s = sp - K
        if 0 < s < N:
            out.append(s)
    return out


def test_emu_signed_fixed_base_digits(emu, ref):
    """the generator part of the double multiplication with every edge of the signed digit recoding (host build: 12-bit digits), alone and
    next to a variable point, through both forms"""
    rng = np.random.default_rng(46)
    a = ref.rand_point(rng)
    took = ctypes.c_int(0)
    for s in fixed_base_edge_scalars(12, rng, 60):
        ng = _b(s)
        for na in (_b(0), bytes(rng.integers(0, 256, 32, dtype=np.uint8))):
            want = _call(ref.lib, "ref_ecmult", [64], a, 0, na, ng)
            assert _call(emu, "emu_ecmult", [64], a, 0, na, ng, None) == want, hex(s)
            assert _call(emu, "emu_ecmult_split", [64], ctypes.byref(took), a, na, ng, None) == want, hex(s)


def _emu_rp(emu, c, p, g, extra=b""):
    mn = ctypes.c_ulonglong(0); mx = ctypes.c_ulonglong(0)
    r = emu.emu_rangeproof_verify(ctypes.byref(mn), ctypes.byref(mx), c.tobytes(), p, ctypes.c_size_t(len(p)), extra, ctypes.c_size_t(len(extra)), g.tobytes())
    return r, mn.value, mx.value


def test_emu_rangeproof(emu, ref):
    rng = np.random.default_rng(3)
    vecs = _golden("rangeproof_vectors.json")["vectors"]
    gh = np.frombuffer(GENERATOR_H, np.uint8)
    emu.emu_split_count.restype = ctypes.c_ulonglong
    split0 = emu.emu_split_count()
    for v in vecs:
        r = _emu_rp(emu, np.frombuffer(bytes.fromhex(v["commit33"]), np.uint8), bytes.fromhex(v["proof"]), gh)
        assert r == (v["result"], int(v["min_value"]), int(v["max_value"])), v["name"]
    assert emu.emu_split_count() > split0          # the ring steps really took the two-piece form of the double multiplication (ecmult_lane_split)
    for (mb, exp, minv, n) in ((64, 0, 0, 1), (5, 2, 17, 1), (1, 0, 0, 1), (13, 3, 1000, 1)):
        commits, plist, gens, _ = ref.make_rangeproofs(n, rng, min_bits=mb, exp=exp, min_value=minv)
        res, mn, mx = ref.rangeproof_verify_many(commits, plist, gens)
        assert _emu_rp(emu, commits[0], plist[0], gens[0]) == (res[0], mn[0], mx[0]) and res[0] == 1
        for k in range(4):
            p = bytearray(plist[0]); p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8)); p = bytes(p)
            if k == 2: p = plist[0] + b"\x00"
            if k == 3: p = plist[0][:-1]
            rr, rmn, rmx = ref.rangeproof_verify_many(commits[:1], [p], gens[:1])
            assert _emu_rp(emu, commits[0], p, gens[0]) == (rr[0], rmn[0], rmx[0])


def test_emu_refused_commitment_encodings(emu, ref):
    """a serialised commitment secp256k1_pedersen_commitment_parse refuses makes an otherwise valid proof invalid (host build of the stages)"""
    rng = np.random.default_rng(32)
    commits, plist, gens, _ = ref.make_rangeproofs(1, rng, min_bits=6)
    assert _emu_rp(emu, commits[0], plist[0], gens[0])[0] == 1
    for (byte, mask) in ((0, 0x80), (0, 0x02), (0, 0x10)):
        c = commits[0].copy(); c[byte] ^= mask
        assert _emu_rp(emu, c, plist[0], gens[0])[0] == 0
        assert ref.rangeproof_verify_many(c[None], plist, gens)[0][0] == 0
    c = commits[0].copy(); c[1:] = 0xFF
    assert _emu_rp(emu, c, plist[0], gens[0])[0] == 0


def _emu_rp_shared(emu, c, p, g, extra=b"", k=None):
    """K3 in the shared-generator form with k rings per lane; k = None: 1, 2 and 4 must agree"""
    outs = []
    for kk in ((1, 2, 4) if k is None else (k,)):
        mn = ctypes.c_ulonglong(0); mx = ctypes.c_ulonglong(0); fast = ctypes.c_int(0)
        r = emu.emu_rangeproof_verify_shared(ctypes.byref(mn), ctypes.byref(mx), c.tobytes(), p, ctypes.c_size_t(len(p)), extra, ctypes.c_size_t(len(extra)),
                                             g.tobytes(), ctypes.byref(fast), ctypes.c_int(kk))
        outs.append(((r, mn.value, mx.value), fast.value))
    assert all(o[0] == outs[0][0] for o in outs), outs
    return outs[0][0], min(o[1] for o in outs)


def test_emu_rangeproof_shared_generator_form(emu, ref):
    """K3 in its shared-generator form (rp_ring_shared: tables of the ring's point built once, f_j*H from a fixed-base table of the
    generator): same accept/reject, min and max as the reference on its fixed vectors, on reference-signed proofs of several shapes
    (other generators included) and on mutated proofs; rings whose point collides with a multiple of the ring base (a key at
    infinity, which the reference rejects) must leave the fast form and still agree."""
    rng = np.random.default_rng(31)
    vecs = _golden("rangeproof_vectors.json")["vectors"]
    gh = np.frombuffer(GENERATOR_H, np.uint8)
    emu.emu_ring_step_count.restype = ctypes.c_ulonglong
    steps0 = emu.emu_ring_step_count()
    for v in vecs:
        r, fast = _emu_rp_shared(emu, np.frombuffer(bytes.fromhex(v["commit33"]), np.uint8), bytes.fromhex(v["proof"]), gh)
        assert r == (v["result"], int(v["min_value"]), int(v["max_value"])), v["name"]
    assert emu.emu_ring_step_count() > steps0
    for (mb, exp, minv) in ((64, 0, 0), (5, 2, 17), (1, 0, 0), (13, 3, 1000), (52, 0, 0)):
        commits, plist, gens, _ = ref.make_rangeproofs(1, rng, min_bits=mb, exp=exp, min_value=minv)
        res, mn, mx = ref.rangeproof_verify_many(commits, plist, gens)
        r, fast = _emu_rp_shared(emu, commits[0], plist[0], gens[0])
        assert r == (res[0], mn[0], mx[0]) and res[0] == 1 and fast >= 1
        for k in range(3):
            p = bytearray(plist[0]); p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8)); p = bytes(p)
            rr, rmn, rmx = ref.rangeproof_verify_many(commits[:1], [p], gens[:1])
            assert _emu_rp_shared(emu, commits[0], p, gens[0])[0] == (rr[0], rmn[0], rmx[0])
    # another generator than H
    g2 = np.frombuffer(ref.rand_point(rng), np.uint8).reshape(1, 64).copy()
    commits, plist, gens, _ = ref.make_rangeproofs(1, rng, min_bits=12, gens64=g2)
    res, mn, mx = ref.rangeproof_verify_many(commits, plist, gens)
    r, fast = _emu_rp_shared(emu, commits[0], plist[0], gens[0])
    assert r == (res[0], mn[0], mx[0]) and res[0] == 1 and fast >= 1
    # a ring commitment equal to +-j * 4^i * H: the key P_j = C + j*B is infinity (or 2jB): those rings must not be served by the fast form
    commits, plist, gens, _ = ref.make_rangeproofs(1, rng, min_bits=8)
    p = bytearray(plist[0])
    # header: byte 0 (exp | flags), byte 1 (mantissa - 1); 4 rings -> 1 sign byte, then 3 ring commitments x
    for (ring, j) in ((0, 1), (1, 2), (2, 3)):
        k = (j * 4**ring) % N
        pt = ref.ecmult_batch(gh.reshape(1, 64), np.frombuffer(_b(k), np.uint8).reshape(1, 32))[0][0]
        for sign in (0, 1):          # one of the two lifts is C = -j*B (key at infinity), the other C = +j*B
            q = bytearray(p); q[3 + 32 * ring:3 + 32 * ring + 32] = pt[:32].tobytes()
            q[2] = (q[2] & ~(1 << ring)) | (sign << ring)
            rr, rmn, rmx = ref.rangeproof_verify_many(commits[:1], [bytes(q)], gens[:1])
            r, fast = _emu_rp_shared(emu, commits[0], bytes(q), gens[0])
            assert r == (rr[0], rmn[0], rmx[0]) and fast <= 3, (ring, j, sign)


def test_emu_schnorr(emu, ref):
    for v in _golden("bip340_vectors.json")["vectors"]:
        msg = bytes.fromhex(v["msg"])
        assert emu.emu_schnorr_verify(bytes.fromhex(v["sig"]), msg, ctypes.c_size_t(len(msg)), bytes.fromhex(v["pk"]), 0) == v["result"]
    rng = np.random.default_rng(5)
    sigs, msgs, pks = ref.make_schnorr(24, rng, threads=4)
    sigs[::5, 40] ^= 1; sigs[1::7, 3] ^= 0x80; pks[2::9, 5] ^= 1
    exp = ref.schnorr_verify_many(sigs, msgs, pks)
    got = [emu.emu_schnorr_verify(sigs[i].tobytes(), msgs[i].tobytes(), ctypes.c_size_t(32), pks[i].tobytes(), 0) for i in range(24)]
    assert list(exp) == got


def test_emu_msm(emu, ref):
    rng = np.random.default_rng(9)
    pts = [ref.rand_point(rng) for _ in range(32)]
    for (n, g, c) in ((1, 0, 0), (1, 1, 0), (2, 1, 0), (5, 0, 4), (17, 1, 5), (100, 1, 0), (300, 1, 6), (600, 0, 0), (40, 1, 13), (40, 0, 11), (64, 1, 8)):
        Pn = np.frombuffer(b"".join(pts[i % 32] for i in range(n)), np.uint8).reshape(n, 64)
        S = rng.integers(0, 256, (n, 32), dtype=np.uint8); inf = np.zeros(n, np.uint8)
        if n > 3:
            S[1] = 0; inf[2] = 1
        gs = bytes(rng.integers(0, 256, 32, dtype=np.uint8)) if g else None
        r1, i1 = ref.ecmult_multi(S, Pn, gs, inf)
        out = ctypes.create_string_buffer(64)
        i2 = emu.emu_msm(out, gs, S.tobytes(), Pn.tobytes(), inf.tobytes(), ctypes.c_size_t(n), c)
        assert i1 == i2 and r1.tobytes() == out.raw, (n, g, c)


def test_emu_msm_lean_accumulation_hands_back_exceptional_runs(emu, ref):
    """the lean bucket accumulation (msm.h: no case analysis per addition, one zero test of ZZ per run) must refuse exactly the runs that
    meet P + P or P - P -- equal / opposite points with equal digits in one bucket -- and the exact form then gives the reference's sum"""
    rng = np.random.default_rng(77)
    emu.emu_msm_lean_refused.restype = ctypes.c_ulong
    P_FIELD = P
    A = np.frombuffer(ref.rand_point(rng), np.uint8); Q = np.frombuffer(ref.rand_point(rng), np.uint8)
    negA = A.copy(); y = (P_FIELD - int.from_bytes(A[32:].tobytes(), "big")) % P_FIELD; negA[32:] = np.frombuffer(y.to_bytes(32, "big"), np.uint8)
    k = rng.integers(0, 256, 32, dtype=np.uint8); k2 = rng.integers(0, 256, 32, dtype=np.uint8)
    for pts, scs in (([A, A, Q], [k, k, k2]), ([A, negA, Q], [k, k, k2]), ([A, A, A, negA, Q, Q], [k, k, k, k, k2, k2]), ([A, negA], [k, k])):
        n = len(pts)
        Pn = np.stack(pts); S = np.stack(scs)
        r1, i1 = ref.ecmult_multi(S, Pn, None, np.zeros(n, np.uint8))
        for c in (0, 4, 9):
            before = emu.emu_msm_lean_refused()
            out = ctypes.create_string_buffer(64)
            i2 = emu.emu_msm(out, None, S.tobytes(), Pn.tobytes(), bytes(n), ctypes.c_size_t(n), c)
            assert i2 >= 0 and i1 == i2 and (i1 or r1.tobytes() == out.raw), (n, c, i2)
            assert emu.emu_msm_lean_refused() > before, "equal / opposite points in one bucket must be handed back to the exact form"


def test_msm_digit_forms_agree(emu):
    """msm.h's three ways to the signed window digits of a half-scalar (carry recurrence; one window from its own constant; all windows
    from one addition, the binning pass's form) give the same digits for every width the plans use, and the digits add up to the value."""
    rng = np.random.default_rng(77)
    edge = [0, 1, 2**128 - 1, 2**128, 2**127, 2**127 - 1, int("5" * 32, 16), int("a" * 32, 16), int("7f" * 16, 16), int("80" * 16, 16)]
    vals = edge + [int.from_bytes(rng.integers(0, 256, 16, dtype=np.uint8).tobytes(), "big") for _ in range(200)]
    for c in range(4, 17):
        W = (129 + c - 1) // c
        for v in vals + [(1 << (c * j + c - 1)) + d for j in range(0, W - 1, 3) for d in (-1, 0, 1)] + [sum(((1 << (c - 1)) + e) << (c * j) for j in range(W - 1)) & (2**128 - 1) for e in (-1, 0, 1)]:
            k5 = (ctypes.c_uint32 * 5)(*[(v >> (32 * i)) & 0xFFFFFFFF for i in range(5)])
            out = (ctypes.c_int * (3 * W))()
            assert emu.emu_msm_digit_forms(out, k5, c) == W
            a, b, f = list(out[:W]), list(out[W:2 * W]), list(out[2 * W:])
            assert a == b == f, (c, hex(v))
            assert sum(d << (c * w) for w, d in enumerate(a)) == v and all(abs(d) <= 1 << (c - 1) for d in a)


def _emu_rewind(emu, commit33, proof, gen64, nonce, capacity):
    bl = ctypes.create_string_buffer(32); val = ctypes.c_ulonglong(0); msg = ctypes.create_string_buffer(4096); ol = ctypes.c_ulonglong(capacity)
    mn = ctypes.c_ulonglong(0); mx = ctypes.c_ulonglong(0)
    r = emu.emu_rangeproof_rewind(bl, ctypes.byref(val), msg if capacity else None, ctypes.byref(ol) if capacity else None, nonce, ctypes.byref(mn), ctypes.byref(mx),
                                  commit33, proof, ctypes.c_size_t(len(proof)), gen64)
    return r, bl.raw, val.value, msg.raw[:ol.value] if (r and capacity) else b""


def test_emu_rangeproof_rewind(emu, ref):
    """device code of rangeproof_rewind.h (HMAC-DRBG replay, value / blind / message recovery, commitment check) on the host: the
    reference tests' own rewind expectations, then reference-signed proofs of several shapes with right and wrong nonces"""
    from tests.refapi import GENERATOR_H
    for v in _golden("rangeproof_vectors.json")["vectors"]:
        rw = v["rewind"]
        r, bl, val, msg = _emu_rewind(emu, bytes.fromhex(v["commit33"]), bytes.fromhex(v["proof"]), GENERATOR_H, bytes.fromhex(rw["nonce"]), rw["capacity"])
        assert r == 1 and bl.hex() == rw["blind"] and val == int(rw["value"]) and msg.hex() == rw["message"], v["name"]
    rng = np.random.default_rng(809)
    for kw in (dict(msg_len=64, min_bits=12), dict(msg_len=0, min_bits=0, exp=-1, values=np.array([3, 4], np.uint64)), dict(msg_len=10, min_bits=3, exp=1, min_value=5)):
        c, p, g, v, b, nn, m = ref.make_rangeproofs_msg(2, rng, **kw)
        nn[1, 31] ^= 0x80
        res, bl, val, msgs, mn, mx = ref.rangeproof_rewind_many(c, p, g, nn, msg_capacity=300)
        for i in range(2):
            got = _emu_rewind(emu, c[i].tobytes(), p[i], g[i].tobytes(), nn[i].tobytes(), 300)
            assert got[0] == res[i]
            if res[i]:
                assert got[1] == bl[i].tobytes() and got[2] == int(val[i]) and got[3] == msgs[i]
        assert res[0] == 1 and res[1] == 0


def test_emu_halfagg(emu, ref):
    """device code of halfagg.h (points, schedules, chain, scalars) + the bucket MSM, on the host, against the golden verdicts
    and against the reference on a 120-signature aggregate (bucket path) with mutations"""
    for v in _golden("halfagg_vectors.json")["vectors"]:
        agg = bytes.fromhex(v["aggsig"])
        r = emu.emu_halfagg_verify(bytes.fromhex(v["pks"]), 0, bytes.fromhex(v["msgs"]), ctypes.c_size_t(v["n"]), agg, ctypes.c_size_t(len(agg)))
        assert r == v["result"], v["name"]
    rng = np.random.default_rng(77)
    n = 120
    sigs, msgs, pks = ref.make_schnorr(n, rng)
    agg = ref.halfagg_aggregate(pks, msgs, sigs)
    for k in range(4):
        a = bytearray(agg); m = msgs.copy()
        if k == 1: a[32 * 57 + 3] ^= 1
        if k == 2: a[-5] ^= 8
        if k == 3: m[119, 0] ^= 1
        exp = max(0, ref.halfagg_verify(pks, m, bytes(a), n))
        assert emu.emu_halfagg_verify(pks.tobytes(), 0, m.tobytes(), ctypes.c_size_t(n), bytes(a), ctypes.c_size_t(len(a))) == exp, k
        assert exp == (k == 0)


def test_emu_bppp(emu, ref):
    g = _golden("bppp_verify_vectors.json")
    gens = bytes.fromhex(g["gens"])
    st = ctypes.create_string_buffer(104); ref.lib.ref_sha256_state_from_prefix(st, b"", ctypes.c_size_t(0))
    for v in g["vectors"]:
        proof = bytes.fromhex(v["proof"]); cvec = b"".join(bytes.fromhex(c) for c in v["c_vec"]); nlen = v["n_vec_len"]; clen = len(v["c_vec"])
        r = emu.emu_bppp_verify(proof, ctypes.c_size_t(len(proof)), st.raw, bytes.fromhex(v["rho"]), gens[:33 * (nlen + clen)], ctypes.c_size_t(nlen + clen),
                                ctypes.c_size_t(nlen), cvec, ctypes.c_size_t(clen), bytes.fromhex(v["commit33"]))
        assert r == v["result"], v["index"]
    rng = np.random.default_rng(10)
    proofs, trs, rhos, gens, gl, cvs, commits = ref.make_bppp(3, rng, 8, 2)
    proofs = proofs.copy(); proofs[1, 7] ^= 4
    exp = ref.bppp_verify_many(proofs, trs, rhos, gens, gl, cvs, commits)
    for i in range(3):
        r = emu.emu_bppp_verify(proofs[i].tobytes(), ctypes.c_size_t(proofs.shape[1]), trs[i].tobytes(), rhos[i].tobytes(), gens.tobytes(), ctypes.c_size_t(gens.shape[0]),
                                ctypes.c_size_t(gl), cvs[i].tobytes(), ctypes.c_size_t(cvs.shape[1]), commits[i].tobytes())
        assert r == exp[i]
    assert list(exp) == [1, 0, 1]


def test_emu_surjection(emu, ref):
    from tests.test_cpu_restatement import _sj_golden
    for name, proof, ins, n_in, out, result in _sj_golden():
        assert emu.emu_surjection_verify(proof, ctypes.c_size_t(len(proof)), ins, ctypes.c_size_t(n_in), out) == result, name
    rng = np.random.default_rng(42)
    for (n_in, n_used) in ((1, 1), (3, 2), (9, 3)):
        proof, tags, out = ref.make_surjection(rng, n_in, n_used)
        assert emu.emu_surjection_verify(proof, ctypes.c_size_t(len(proof)), tags.tobytes(), ctypes.c_size_t(n_in), out.tobytes()) == 1
        for k in range(3):
            p = bytearray(proof); p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8)); p = bytes(p)
            assert emu.emu_surjection_verify(p, ctypes.c_size_t(len(p)), tags.tobytes(), ctypes.c_size_t(n_in), out.tobytes()) == ref.surjection_verify(p, tags, out)


# ---- (3) the C ABI ---------------------------------------------------------------------------------------------------------
def test_abi_symbols_and_loud_failure():
    from secp256k1_zkp_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip("product library not built")
    lib = _native.load()
    hdr = open(os.path.join(ROOT, "include", "secp256k1_zkp_amd.h")).read()
    declared = set(re.findall(r"S2K_API\s+[\w\s\*]+?\b(\w+)\s*\(", hdr))
    assert declared, "no S2K_API declarations found"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    import torch
    if not torch.cuda.is_available():
        # no device: creating an engine must fail with a message, and nothing may silently compute on the CPU
        from secp256k1_zkp_amd import Engine, S2KError
        with pytest.raises(S2KError):
            Engine(0)
        assert "HIP" in _native.last_error() or "device" in _native.last_error()


def test_header_is_plain_c(tmp_path):
    """include/secp256k1_zkp_amd.h must compile as C (c89 with the usual extensions off) and examples/ must compile against it"""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "secp256k1_zkp_amd.h"\nint main(void) { return sizeof(s2k_engine*) == 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "t.o")], check=True)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "examples", "rangeproof_verify.c"),
                    "-o", str(tmp_path / "e.o")], check=True)


def test_emu_seeded_table_construction(emu, ref):
    """the device's fixed-base table construction (csrc/gtable.h: window bases, seeds, ONE affine addition per remaining entry with a
    shared inversion per run of rows) run sequentially at small widths, for G and for another point: every entry a digit can address equals
    v * 2^(D w) * point, and none is left unwritten"""
    rng = np.random.default_rng(5)
    for D in (9, 10, 11, 12, 13, 14, 15):           # (round 6: every denominator yields R + C and R - C; odd and even widths split rows / columns differently)
        assert emu.emu_gtab_seeded_construction(D, None) == 0, D
    assert emu.emu_gtab_seeded_construction(12, ref.rand_point(rng)) == 0


def test_emu_msm_exhaustive_edge_group(emu, ref):
    """the shape of the reference's test_exhaustive_ecmult_multi (src/tests_exhaustive.c:198-227) through the host build of the bucket pipeline
    (tests/exhaustive_msm.py; the full edge lists run on the device, tests/test_gpu_msm_exhaustive.py): every i*P_x + j*P_y + k*G over a
    sub-list of the edge scalars and points, at the plan's window width and at a forced narrow one"""
    from tests import exhaustive_msm as xm
    S = [0, 1, 2, N - 1, xm.LAMBDA, (1 << 128) - 1]; Pm = [0, 1, N - 1, 2, xm.LAMBDA]
    pxy, pinf = xm.group_points(ref, Pm)
    combos = list(xm.cases(S, range(len(Pm))))
    want = xm.expected_points(ref, [(i * Pm[x] + j * Pm[y] + k) % N for (i, j, k, x, y) in combos])
    out = ctypes.create_string_buffer(64)
    for t, (i, j, k, x, y) in enumerate(combos):
        sc = np.stack([xm._b(i), xm._b(j)]); pts = np.stack([pxy[x], pxy[y]]); inf = np.array([pinf[x] != 0, pinf[y] != 0], np.uint8)
        rinf = emu.emu_msm(out, bytes(xm._b(k)), sc.tobytes(), pts.tobytes(), inf.tobytes(), ctypes.c_size_t(2), 4 if t % 2 else 0)
        exy, einf = want[(i * Pm[x] + j * Pm[y] + k) % N]
        assert rinf >= 0 and bool(rinf) == einf and (einf or out.raw == exy), (hex(i), hex(j), hex(k), x, y)
