"""Pins oracle/zkp_oracle.c (the plain-C restatement): (1) against the golden vectors recorded from the reference's own tests,
(2) against the unmodified reference (oracle/_ref) primitive by primitive on random and edge inputs."""
import ctypes
import json
import os

import numpy as np
import pytest

from tests.refapi import GENERATOR_H, G_XY, N, P

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def zo():
    path = os.path.join(ROOT, "oracle", "libzkp_oracle.so")
    if not os.path.exists(path):
        pytest.skip("oracle restatement not built (make -C oracle oracle)")
    return ctypes.CDLL(path)


def _golden(name):
    return json.load(open(os.path.join(HERE, "golden", name)))


def _b(v):
    return int(v).to_bytes(32, "big")


def _zo_rp(zo, commit, proof, gen, extra=b""):
    mn = ctypes.c_uint64(0); mx = ctypes.c_uint64(0)
    r = zo.zo_rangeproof_verify(ctypes.byref(mn), ctypes.byref(mx), commit, proof, ctypes.c_size_t(len(proof)), extra if extra else None,
                                ctypes.c_size_t(len(extra)), gen)
    return r, mn.value, mx.value


def test_golden_rangeproof(zo):
    for v in _golden("rangeproof_vectors.json")["vectors"]:
        assert _zo_rp(zo, bytes.fromhex(v["commit33"]), bytes.fromhex(v["proof"]), GENERATOR_H) == (v["result"], int(v["min_value"]), int(v["max_value"])), v["name"]


def test_golden_bip340(zo):
    for v in _golden("bip340_vectors.json")["vectors"]:
        msg = bytes.fromhex(v["msg"])
        assert zo.zo_schnorrsig_verify(bytes.fromhex(v["sig"]), msg, ctypes.c_size_t(len(msg)), bytes.fromhex(v["pk"])) == v["result"]


def test_golden_bppp(zo):
    g = _golden("bppp_verify_vectors.json")
    gens = bytes.fromhex(g["gens"])
    st = np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], "<u4").tobytes() + b"\0" * 72
    for v in g["vectors"]:
        proof = bytes.fromhex(v["proof"]); cvec = b"".join(bytes.fromhex(c) for c in v["c_vec"]); nlen = v["n_vec_len"]; clen = len(v["c_vec"])
        r = zo.zo_bppp_norm_verify(proof, ctypes.c_size_t(len(proof)), st, bytes.fromhex(v["rho"]), gens[:33 * (nlen + clen)], ctypes.c_size_t(nlen + clen),
                                   ctypes.c_size_t(nlen), cvec, ctypes.c_size_t(clen), bytes.fromhex(v["commit33"]))
        assert r == v["result"], v["index"]


def _call(lib, name, nout, *args):
    outs = [ctypes.create_string_buffer(n) for n in nout]
    r = getattr(lib, name)(*outs, *args)
    return r, [o.raw for o in outs]


def test_vs_reference_primitives(zo, ref):
    rng = np.random.default_rng(31)
    edge = [0, 1, 2, P - 1, P, P + 1, 2**256 - 1, N, N - 1, N + 1, 2**128, (P + 1) // 2]
    cs = [_b(e % 2**256) for e in edge] + [bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(100)]
    for i, a in enumerate(cs):
        c = cs[(i * 7 + 3) % len(cs)]
        assert _call(zo, "zo_fe_mul", [32], a, c)[1] == _call(ref.lib, "ref_fe_mul", [32], a, c)[1]
        assert _call(zo, "zo_fe_inv", [32], a)[1] == _call(ref.lib, "ref_fe_inv", [32], a)[1]
        assert _call(zo, "zo_fe_sqrt", [32], a) == _call(ref.lib, "ref_fe_sqrt", [32], a)
        assert _call(zo, "zo_scalar_mul", [32], a, c)[1] == _call(ref.lib, "ref_scalar_mul", [32], a, c)[1]
        assert _call(zo, "zo_scalar_split_lambda", [32, 32], a)[1] == _call(ref.lib, "ref_scalar_split_lambda", [32, 32], a)[1]
    pts = [ref.rand_point(rng) for _ in range(10)] + [G_XY]
    neg = lambda p: p[:32] + _b((P - int.from_bytes(p[32:], "big")) % P)
    for i, a in enumerate(pts):
        for c in (pts[(i + 1) % len(pts)], a, neg(a)):
            for ai in (0, 1):
                for bi in (0, 1):
                    assert _call(zo, "zo_ge_add", [64], a, ai, c, bi) == _call(ref.lib, "ref_ge_add", [64], a, ai, c, bi)
    sc_edge = [_b(v) for v in (0, 1, 2, N - 1, 255, 2**128, N // 2 + 1)]
    for i in range(30):
        na = bytes(rng.integers(0, 256, 32, dtype=np.uint8)) if i % 3 else sc_edge[i % len(sc_edge)]
        ng = bytes(rng.integers(0, 256, 32, dtype=np.uint8)) if i % 4 else sc_edge[(i * 5) % len(sc_edge)]
        a = pts[i % len(pts)]
        assert _call(zo, "zo_ecmult", [64], a, 0, na, ng) == _call(ref.lib, "ref_ecmult", [64], a, 0, na, ng)
    assert _call(zo, "zo_ecmult", [64], G_XY, 0, _b(5), None) == _call(ref.lib, "ref_ecmult", [64], G_XY, 0, _b(5), None)
    for n in (0, 1, 5, 87, 88, 200, 1500):
        Pn = np.frombuffer(b"".join(pts[k % len(pts)] for k in range(n)), np.uint8).reshape(n, 64) if n else np.zeros((0, 64), np.uint8)
        S = rng.integers(0, 256, (n, 32), dtype=np.uint8); inf = np.zeros(n, np.uint8)
        if n > 4:
            S[1] = 0; inf[2] = 1; S[3] = np.frombuffer(_b(2), np.uint8)
        g = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
        exp, einf = ref.ecmult_multi(S, Pn, g, inf)
        out = ctypes.create_string_buffer(64)
        gi = zo.zo_ecmult_multi(out, g, S.tobytes(), Pn.tobytes(), inf.tobytes(), ctypes.c_size_t(n))
        assert gi == einf and out.raw == exp.tobytes(), n


def test_vs_reference_protocols(zo, ref):
    rng = np.random.default_rng(32)
    for (mb, exp, minv) in ((64, 0, 0), (5, 2, 17), (1, 0, 0), (13, 3, 1000)):
        commits, plist, gens, _ = ref.make_rangeproofs(1, rng, min_bits=mb, exp=exp, min_value=minv)
        res, mn, mx = ref.rangeproof_verify_many(commits, plist, gens)
        assert _zo_rp(zo, commits[0].tobytes(), plist[0], gens[0].tobytes()) == (res[0], mn[0], mx[0])
        for k in range(3):
            p = bytearray(plist[0]); p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8)); p = bytes(p)
            if k == 2: p = plist[0] + b"\x00"
            rr, rmn, rmx = ref.rangeproof_verify_many(commits, [p], gens)
            assert _zo_rp(zo, commits[0].tobytes(), p, gens[0].tobytes()) == (rr[0], rmn[0], rmx[0])
    sigs, msgs, pks = ref.make_schnorr(20, rng, threads=2)
    sigs[::5, 40] ^= 1; pks[2::9, 5] ^= 1
    exp = ref.schnorr_verify_many(sigs, msgs, pks)
    assert [zo.zo_schnorrsig_verify(sigs[i].tobytes(), msgs[i].tobytes(), ctypes.c_size_t(32), pks[i].tobytes()) for i in range(20)] == list(exp)
    proofs, trs, rhos, gens, gl, cvs, commits = ref.make_bppp(3, rng, 16, 4)
    proofs = proofs.copy(); proofs[1, 9] ^= 2
    exp = ref.bppp_verify_many(proofs, trs, rhos, gens, gl, cvs, commits)
    got = [zo.zo_bppp_norm_verify(proofs[i].tobytes(), ctypes.c_size_t(proofs.shape[1]), trs[i].tobytes(), rhos[i].tobytes(), gens.tobytes(),
                                  ctypes.c_size_t(gens.shape[0]), ctypes.c_size_t(gl), cvs[i].tobytes(), ctypes.c_size_t(cvs.shape[1]), commits[i].tobytes()) for i in range(3)]
    assert got == list(exp) == [1, 0, 1]


def _sj_golden():
    from tests.refapi import lift_generator33
    g = _golden("surjection_vectors.json")
    tags = [lift_generator33(bytes.fromhex(t)) for t in g["tags33"]]
    out = lift_generator33(bytes.fromhex(g["output_tag33"]))
    cases = []
    for v in g["vectors"]:
        ins = b"".join(tags[v["tag_first"]:v["tag_first"] + v["n_inputs"]])
        cases.append((v["name"], bytes.fromhex(v["proof"]), ins, v["n_inputs"], out if v["output"] == "out" else tags[0], v["result"]))
    return cases


def test_golden_surjection(zo, ref):
    for name, proof, ins, n_in, out, result in _sj_golden():
        assert zo.zo_surjectionproof_verify(proof, ctypes.c_size_t(len(proof)), ins, ctypes.c_size_t(n_in), out) == result, name
        assert ref.lib.ref_surjectionproof_verify_ser(proof, ctypes.c_size_t(len(proof)), ins, ctypes.c_size_t(n_in), out) == result, name


def test_surjection_vs_reference(zo, ref):
    rng = np.random.default_rng(41)
    for (n_in, n_used) in ((1, 1), (3, 1), (3, 3), (8, 3), (20, 5)):
        proof, tags, out = ref.make_surjection(rng, n_in, n_used)
        assert ref.surjection_verify(proof, tags, out) == 1
        assert zo.zo_surjectionproof_verify(proof, ctypes.c_size_t(len(proof)), tags.tobytes(), ctypes.c_size_t(n_in), out.tobytes()) == 1
        for k in range(4):
            p = bytearray(proof); p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8)); p = bytes(p)
            t2 = tags.copy()
            if k == 3: p = proof; t2[0, 40] ^= 1
            assert zo.zo_surjectionproof_verify(p, ctypes.c_size_t(len(p)), t2.tobytes(), ctypes.c_size_t(n_in), out.tobytes()) == ref.surjection_verify(p, t2, out)


def _ha_golden():
    return [(v["name"], bytes.fromhex(v["pks"]), bytes.fromhex(v["msgs"]), v["n"], bytes.fromhex(v["aggsig"]), v["result"])
            for v in _golden("halfagg_vectors.json")["vectors"]]


def test_golden_halfagg(zo, ref):
    """spec vectors of modules/schnorrsig_halfagg/tests_impl.h:73-168 + reference-generated aggregates and their mutations"""
    for name, pks, msgs, n, agg, result in _ha_golden():
        assert zo.zo_schnorrsig_aggverify(pks, msgs, ctypes.c_size_t(n), agg, ctypes.c_size_t(len(agg))) == result, name
        assert max(0, ref.halfagg_verify(pks, msgs, agg, n)) == result, name


def test_halfagg_vs_reference(zo, ref):
    rng = np.random.default_rng(55)
    for n in (1, 4, 11):
        sigs, msgs, pks = ref.make_schnorr(n, rng)
        agg = ref.halfagg_aggregate(pks, msgs, sigs)
        assert zo.zo_schnorrsig_aggverify(pks.tobytes(), msgs.tobytes(), ctypes.c_size_t(n), agg, ctypes.c_size_t(len(agg))) == 1
        for k in range(6):
            a = bytearray(agg); m = msgs.copy(); p = pks.copy()
            if k < 3: a[int(rng.integers(0, len(a)))] ^= 1 << int(rng.integers(0, 8))
            elif k == 3: m[int(rng.integers(0, n)), 7] ^= 2
            else: p[int(rng.integers(0, n)), int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
            exp = max(0, ref.halfagg_verify(p.tobytes(), m.tobytes(), bytes(a), n))
            assert zo.zo_schnorrsig_aggverify(p.tobytes(), m.tobytes(), ctypes.c_size_t(n), bytes(a), ctypes.c_size_t(len(a))) == exp


def _pt_golden():
    out = []
    for v in _golden("pedersen_tally_vectors.json")["vectors"]:
        out.append((v["name"], np.frombuffer(bytes.fromhex("".join(v["pos"])), np.uint8).reshape(-1, 33), np.frombuffer(bytes.fromhex("".join(v["neg"])), np.uint8).reshape(-1, 33), v["result"]))
    return out


def test_golden_pedersen_tally(zo, ref):
    for name, pos, neg, result in _pt_golden():
        got = zo.zo_pedersen_verify_tally(pos.tobytes(), ctypes.c_size_t(pos.shape[0]), neg.tobytes(), ctypes.c_size_t(neg.shape[0]))
        assert got == result, name
        assert int(ref.pedersen_verify_tally_many([(pos, neg)])[0]) == result, name


def _zo_rewind(zo, commit33, proof, gen64, nonce, capacity):
    bl = ctypes.create_string_buffer(32); val = ctypes.c_uint64(0); msg = ctypes.create_string_buffer(4096); ol = ctypes.c_size_t(capacity)
    mn = ctypes.c_uint64(0); mx = ctypes.c_uint64(0)
    r = zo.zo_rangeproof_rewind(bl, ctypes.byref(val), msg if capacity else None, ctypes.byref(ol) if capacity else None, nonce, ctypes.byref(mn), ctypes.byref(mx),
                                commit33, proof, ctypes.c_size_t(len(proof)), None, ctypes.c_size_t(0), gen64)
    return r, bl.raw, val.value, msg.raw[:ol.value] if (r and capacity) else b"", mn.value, mx.value


def test_golden_rangeproof_rewind(zo):
    """what src/modules/rangeproof/tests_impl.h:666-687,737-757,795-810,843-880,1241-1346 assert about secp256k1_rangeproof_rewind"""
    for v in _golden("rangeproof_vectors.json")["vectors"]:
        rw = v["rewind"]
        r, bl, val, msg, mn, mx = _zo_rewind(zo, bytes.fromhex(v["commit33"]), bytes.fromhex(v["proof"]), GENERATOR_H, bytes.fromhex(rw["nonce"]), rw["capacity"])
        assert r == 1 and bl.hex() == rw["blind"] and val == int(rw["value"]) and msg.hex() == rw["message"], v["name"]
        assert mn == int(v["min_value"]) and mx == int(v["max_value"])
        bad = bytearray(bytes.fromhex(rw["nonce"])); bad[0] ^= 1
        assert _zo_rewind(zo, bytes.fromhex(v["commit33"]), bytes.fromhex(v["proof"]), GENERATOR_H, bytes(bad), rw["capacity"])[0] == 0, v["name"]


def test_rangeproof_rewind_vs_reference(zo, ref):
    rng = np.random.default_rng(808)
    for kw in (dict(msg_len=70, min_bits=16), dict(msg_len=0, min_bits=0, exp=-1, values=np.array([9, 10], np.uint64)), dict(msg_len=33, min_bits=5, exp=2, min_value=17),
               dict(msg_len=200, min_bits=7)):
        c, p, g, v, b, nn, m = ref.make_rangeproofs_msg(2, rng, **kw)
        nn2 = nn.copy(); nn2[1, 5] ^= 1                                   # wrong nonce on the second one
        for cap in (4096, 50, 0):
            res, bl, val, msgs, mn, mx = ref.rangeproof_rewind_many(c, p, g, nn2, msg_capacity=cap)
            for i in range(2):
                got = _zo_rewind(zo, c[i].tobytes(), p[i], g[i].tobytes(), nn2[i].tobytes(), cap)
                assert got[0] == res[i]
                if res[i]:
                    assert got[1] == bl[i].tobytes() and got[2] == int(val[i]) and got[3] == msgs[i]
            assert res[0] == 1 and res[1] == 0
