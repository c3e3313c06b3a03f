"""CPU tier: the reference-side half of the drop-in boundary (integration/secp256k1_amd_hook.{h,c}), compiled against the
reference into oracle/_ref/libsecp256k1_hooked.so.

What is proven here without a GPU:
  * no backend installed  -> the batch calls are the library's own per-item calls;
  * a backend that FAILS  -> every adapter falls back to the CPU path and returns the reference's verdicts (a failed
                             engine is never a verdict); this includes the *real* product library on this GPU-less box,
                             whose engine cannot be created -- exactly the "HIP failure" case of SURVEY.md section 8b;
  * a checking backend    -> the packed arrays (commitments, offsets, ragged tags, tally lists, drained MSM terms) decode to
                             exactly the caller's items: the backend re-verifies every item with the reference from the
                             packed form and the adapter hands its verdicts back;
  * the MSM seam          -> the reference's own BP++ vectors run through secp256k1_ecmult_multi_var_amd; a callback that
                             returns 0 makes the call return 0 (and stops the drain); an engine failure falls back.
The same adapters run against the real engine on the GPU in tests/test_gpu_hook.py.
"""
import ctypes
import json
import os

import numpy as np
import pytest

from tests import hookapi
from tests.refapi import GENERATOR_H, G_XY

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hk():
    if not os.path.exists(hookapi.HOOKED_PATH):
        pytest.skip("oracle/_ref/libsecp256k1_hooked.so not built (make -C oracle hooked)")
    h = hookapi.Hooked()
    yield h
    h.set_backend()


def _arr(ptr, nbytes, dtype=np.uint8):
    if not ptr or nbytes == 0:
        return np.zeros(0, dtype)
    return np.frombuffer((ctypes.c_uint8 * nbytes).from_address(ptr), dtype=dtype)


def _failing(fntype):
    return fntype(lambda *a: 0)


# ---- checking backends: decode the packed arrays and let the plain reference judge every item -------------------------
def _rp_checker(ref, seen):
    ref.lib.secp256k1_context_create.restype = ctypes.c_void_p
    ctx = ref.lib.secp256k1_context_create(ctypes.c_uint(1)); ref.lib.secp256k1_rangeproof_verify.restype = ctypes.c_int

    def fn(engine, results, mn, mx, commits33, proofs, proof_off, extra, extra_off, gens64, n):
        off = _arr(proof_off, 8 * (n + 1), np.uint64); eoff = _arr(extra_off, 8 * (n + 1), np.uint64) if extra_off else None
        pdata = _arr(proofs, int(off[n])); c = _arr(commits33, 33 * n).reshape(n, 33); g = _arr(gens64, 64 * n).reshape(n, 64)
        res = _arr(results, 4 * n, np.int32); omn = _arr(mn, 8 * n, np.uint64); omx = _arr(mx, 8 * n, np.uint64)
        for i in range(n):
            cobj = c[i].tobytes() + b"\0" * 31
            p = pdata[int(off[i]):int(off[i + 1])].tobytes()
            e = b"" if eoff is None else _arr(extra, int(eoff[n]))[int(eoff[i]):int(eoff[i + 1])].tobytes()
            a, b = ctypes.c_uint64(int(omn[i])), ctypes.c_uint64(int(omx[i]))
            res[i] = ref.lib.secp256k1_rangeproof_verify(ctypes.c_void_p(ctx), ctypes.byref(a), ctypes.byref(b), cobj, p, ctypes.c_size_t(len(p)),
                                                         e if e else None, ctypes.c_size_t(len(e)), g[i].tobytes())
            omn[i], omx[i] = a.value, b.value
        seen.append(n)
        return 1
    return hookapi.RP_FN(fn)


def _msm_checker(ref, seen):
    def fn(engine, r_xy, r_inf, g_sc, sc, pt_xy, pt_inf, n):
        s = _arr(sc, 32 * n); p = _arr(pt_xy, 64 * n); inf = _arr(pt_inf, n) if pt_inf else None
        g = _arr(g_sc, 32).tobytes() if g_sc else None
        xy, fl = ref.ecmult_multi(s.copy(), p.copy(), g_sc=g, pt_inf=None if inf is None else inf.copy())
        _arr(r_xy, 64)[:] = xy; _arr(r_inf, 4, np.int32)[0] = fl
        seen.append(n)
        return 1
    return hookapi.MSM_FN(fn)


def test_rangeproof_batch_cpu_failing_and_checking_backends(hk, ref):
    rng = np.random.default_rng(501)
    commits, proofs, gens, _ = ref.make_rangeproofs(6, rng, min_bits=8)
    v = json.load(open(os.path.join(HERE, "golden", "rangeproof_vectors.json")))
    assert v["generator"] == "secp256k1_generator_h"
    gh = GENERATOR_H
    fixed = v["vectors"]
    plist = list(proofs); c = [commits[i] for i in range(6)]; g = [gens[i] for i in range(6)]
    for x in fixed:                                                     # the reference's own fixed proofs (tests_impl.h:589-812,883-1349)
        plist.append(bytes.fromhex(x["proof"])); c.append(np.frombuffer(bytes.fromhex(x["commit33"]), np.uint8)); g.append(np.frombuffer(gh, np.uint8))
    # negatives: bit flips, truncation, trailing byte, empty proof
    bad = bytearray(proofs[0]); bad[20] ^= 4
    plist += [bytes(bad), proofs[1][:-1], proofs[2] + b"\0", b""]; c += [commits[0], commits[1], commits[2], commits[3]]; g += [gens[0]] * 4
    c = np.stack(c); g = np.stack(g); n = len(plist)
    extra = [b"" if i % 3 else b"xyz" * (i + 1) for i in range(n)]
    exp = ref.rangeproof_verify_many(c, plist, g)
    for ex in (None, extra):
        # (1) plain CPU library
        hk.set_backend()
        r0 = hk.rangeproof_verify_batch(c, plist, g, extra=ex)
        if ex is None:
            assert np.array_equal(r0[0], exp[0]) and np.array_equal(r0[1], exp[1]) and np.array_equal(r0[2], exp[2])
        # (2) failing backend -> fallback, identical results
        s0 = hk.stats()
        hk.set_backend(rangeproof=_failing(hookapi.RP_FN))
        r1 = hk.rangeproof_verify_batch(c, plist, g, extra=ex)
        assert hk.stats() == (s0[0], s0[1] + 1)
        assert all(np.array_equal(a, b) for a, b in zip(r0, r1))
        # (3) checking backend -> served, identical results: the packing is exact
        seen = []
        hk.set_backend(rangeproof=_rp_checker(ref, seen))
        r2 = hk.rangeproof_verify_batch(c, plist, g, extra=ex)
        assert seen == [n] and hk.stats() == (s0[0] + 1, s0[1] + 1)
        assert all(np.array_equal(a, b) for a, b in zip(r0, r2))
    assert r0[0].sum() < n and exp[0][:6].all()
    hk.set_backend()


def test_real_library_without_device_falls_back(hk, ref):
    """The product library on a box without a HIP device: engine creation fails, every call returns 0 with
    S2K_STATUS_ENGINE_FAILURE -> the adapters answer from the CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this is the GPU-less half; tests/test_gpu_hook.py covers the forced failure on a GPU box")
    from secp256k1_zkp_amd import _native
    L = _native.load()
    assert not L.s2k_engine_create(0)
    addr = lambda name: ctypes.cast(getattr(L, name), ctypes.c_void_p).value
    hk.set_backend(engine=None, rangeproof=addr("secp256k1_rangeproof_verify_batch"), msm=addr("s2k_ecmult_multi"),
                   schnorr=addr("secp256k1_schnorrsig_verify_batch"), surjection=addr("secp256k1_surjectionproof_verify_batch"),
                   tally=addr("secp256k1_pedersen_verify_tally_batch"))
    rng = np.random.default_rng(502)
    commits, proofs, gens, _ = ref.make_rangeproofs(3, rng, min_bits=8)
    s0 = hk.stats()
    res, mn, mx = hk.rangeproof_verify_batch(commits, proofs, gens)
    exp = ref.rangeproof_verify_many(commits, proofs, gens)
    assert np.array_equal(res, exp[0]) and np.array_equal(mx, exp[2]) and hk.stats()[1] == s0[1] + 1
    assert L.s2k_last_status() == 1                                     # S2K_STATUS_ENGINE_FAILURE
    sc = rng.integers(0, 256, (5, 32), dtype=np.uint8); pts = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(5)])
    xy, inf, _ = hk.ecmult_multi(sc, pts, g_sc=bytes(sc[0]))
    exy, einf = ref.ecmult_multi(sc, pts, g_sc=bytes(sc[0]))
    assert inf == einf and np.array_equal(xy, exy)
    # single-item forms: 0 + engine-failure status, never 1
    mnv, mxv = ctypes.c_uint64(0), ctypes.c_uint64(0)
    assert L.secp256k1_rangeproof_verify_amd(None, ctypes.byref(mnv), ctypes.byref(mxv), commits[0].tobytes() + b"\0" * 31, proofs[0], len(proofs[0]), None, 0, gens[0].tobytes()) == 0
    assert L.s2k_last_status() == 1
    assert L.secp256k1_rangeproof_verify_amd(None, None, ctypes.byref(mxv), commits[0].tobytes() + b"\0" * 31, proofs[0], len(proofs[0]), None, 0, gens[0].tobytes()) == 0
    assert L.s2k_last_status() == 2                                     # S2K_STATUS_ILLEGAL_ARGUMENT
    hk.set_backend()


def test_msm_seam_callback_rules(hk, ref):
    rng = np.random.default_rng(503)
    n = 40
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8); sc[3] = 0
    pts = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(n)])
    inf = np.zeros(n, np.uint8); inf[7] = 1
    g = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    exy, einf = ref.ecmult_multi(sc, pts, g_sc=g, pt_inf=inf)
    for gs in (g, None):
        exy, einf = ref.ecmult_multi(sc, pts, g_sc=gs, pt_inf=inf)
        hk.set_backend()
        xy, fl, calls = hk.ecmult_multi(sc, pts, g_sc=gs, pt_inf=inf)
        assert fl == einf and np.array_equal(xy, exy)
        seen = []
        hk.set_backend(msm=_msm_checker(ref, seen))
        xy, fl, calls = hk.ecmult_multi(sc, pts, g_sc=gs, pt_inf=inf)
        assert seen == [n] and calls == n and fl == einf and np.array_equal(xy, exy)
        # a callback that returns 0: the call returns 0, the drain stops there, the backend is never reached
        seen.clear()
        xy, fl, calls = hk.ecmult_multi(sc, pts, g_sc=gs, pt_inf=inf, fail_at=11)
        assert fl == -1 and calls == 12 and seen == []
        # engine failure: CPU path (which pulls the callback again), same answer
        hk.set_backend(msm=_failing(hookapi.MSM_FN))
        xy, fl, calls = hk.ecmult_multi(sc, pts, g_sc=gs, pt_inf=inf)
        assert fl == einf and np.array_equal(xy, exy) and calls >= 2 * n
        xy, fl, calls = hk.ecmult_multi(sc, pts, g_sc=gs, pt_inf=inf, fail_at=0)
        assert fl == -1
    # result at infinity: k*G - k*G
    k = rng.integers(0, 256, (1, 32), dtype=np.uint8); k[0, 0] &= 0x7F
    two = np.concatenate([k, k]); gp = np.frombuffer(G_XY, np.uint8)
    from tests.refapi import P
    negy = ((P - int.from_bytes(G_XY[32:], "big")) % P).to_bytes(32, "big")
    pp = np.stack([gp, np.frombuffer(G_XY[:32] + negy, np.uint8)])
    hk.set_backend(msm=_msm_checker(ref, []))
    xy, fl, _ = hk.ecmult_multi(two, pp)
    assert fl == 1
    hk.set_backend()


def test_bppp_vectors_through_the_msm_adapter(hk, ref):
    """The reference's own norm-argument verifier (bppp_norm_product_impl.h:425-552) with its three ecmult_multi_var call
    sites redirected to the adapter: all 13 vectors of modules/bppp/test_vectors/verify.h, with a checking backend, a failing
    backend and no backend."""
    g = json.load(open(os.path.join(HERE, "golden", "bppp_verify_vectors.json")))
    gens = bytes.fromhex(g["gens"])
    st = np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], "<u4").tobytes() + b"\0" * 72

    def run():
        out = []
        for v in g["vectors"]:
            proof = bytes.fromhex(v["proof"]); cvec = b"".join(bytes.fromhex(c) for c in v["c_vec"]); nlen = v["n_vec_len"]; clen = len(v["c_vec"])
            out.append(hk.lib.ref_bppp_norm_verify(proof, len(proof), st, bytes.fromhex(v["rho"]), gens[:33 * (nlen + clen)], nlen + clen, nlen, cvec, clen,
                                                   bytes.fromhex(v["commit33"])))
        return out
    exp = [v["result"] for v in g["vectors"]]
    hk.set_backend()
    assert run() == exp
    seen = []
    hk.set_backend(msm=_msm_checker(ref, seen))
    assert run() == exp and len(seen) > 0                  # the adapter really carried the verifier's MSMs
    s0 = hk.stats()
    hk.set_backend(msm=_failing(hookapi.MSM_FN))
    assert run() == exp and hk.stats()[1] > s0[1]
    hk.set_backend()


def test_schnorr_surjection_tally_adapters(hk, ref):
    rng = np.random.default_rng(504)
    # BIP-340
    sigs, msgs, pks = ref.make_schnorr(9, rng)
    sigs[2, 5] ^= 1; msgs[4, 0] ^= 1
    objs = ref.xonly_objects(pks)
    exp = ref.schnorr_verify_many(sigs, msgs, pks)
    seen = []

    def sch(engine, results, s, m, msglen, pk, fmt, n):
        assert fmt == 1 and msglen == 32
        ss = _arr(s, 64 * n).reshape(n, 64); mm = _arr(m, 32 * n).reshape(n, 32); po = _arr(pk, 64 * n).reshape(n, 64)
        assert np.array_equal(ss, sigs) and np.array_equal(mm, msgs) and np.array_equal(po, objs)
        _arr(results, 4 * n, np.int32)[:] = exp; seen.append(n)
        return 1
    for be in (None, _failing(hookapi.SCH_FN), hookapi.SCH_FN(sch)):
        hk.set_backend(schnorr=be)
        assert np.array_equal(hk.schnorrsig_verify_batch(sigs, msgs, objs), exp)
    assert seen == [9] and exp.sum() == 7
    # surjection proofs: ragged tag lists
    items = [ref.make_surjection(rng, k, min(k, 3)) for k in (1, 3, 8, 20)]
    bad = bytearray(items[1][0]); bad[-1] ^= 1
    items.append((bytes(bad), items[1][1], items[1][2]))
    exps = np.array([ref.surjection_verify(p, t, o) for p, t, o in items], np.int32)
    seen.clear()

    def sj(engine, results, proofs, proof_off, tags, tag_off, outs, n):
        po = _arr(proof_off, 8 * (n + 1), np.uint64); to = _arr(tag_off, 8 * (n + 1), np.uint64)
        pd = _arr(proofs, int(po[n])); td = _arr(tags, 64 * int(to[n])).reshape(-1, 64); od = _arr(outs, 64 * n).reshape(n, 64)
        res = _arr(results, 4 * n, np.int32)
        for i in range(n):
            assert pd[int(po[i]):int(po[i + 1])].tobytes() == items[i][0]          # re-serialisation is byte-exact
            res[i] = ref.surjection_verify(items[i][0], td[int(to[i]):int(to[i + 1])].copy(), od[i].copy())
        seen.append(n)
        return 1
    for be in (None, _failing(hookapi.SJ_FN), hookapi.SJ_FN(sj)):
        hk.set_backend(surjection=be)
        assert np.array_equal(hk.surjectionproof_verify_batch(items), exps)
    assert seen == [5] and list(exps) == [1, 1, 1, 1, 0]
    # tallies
    tallies = [ref.make_balanced_tally(rng, 2, 3), ref.make_balanced_tally(rng, 1, 1), ref.make_balanced_tally(rng, 4, 2)]
    a, b = ref.make_balanced_tally(rng, 2, 2)
    tallies.append((a, b[:1]))                                              # unbalanced
    tallies.append((np.zeros((0, 33), np.uint8), np.zeros((0, 33), np.uint8)))      # empty tally: accepted
    expt = ref.pedersen_verify_tally_many(tallies)
    seen.clear()

    def tl(engine, results, c33, off, npos, n):
        o = _arr(off, 8 * (n + 1), np.uint64); npv = _arr(npos, 8 * n, np.uint64); cd = _arr(c33, 33 * int(o[n])).reshape(-1, 33)
        back = []
        for t in range(n):
            seg = cd[int(o[t]):int(o[t + 1])]
            back.append((seg[:int(npv[t])].copy(), seg[int(npv[t]):].copy()))
        _arr(results, 4 * n, np.int32)[:] = ref.pedersen_verify_tally_many(back); seen.append(n)
        return 1
    for be in (None, _failing(hookapi.TALLY_FN), hookapi.TALLY_FN(tl)):
        hk.set_backend(tally=be)
        assert np.array_equal(hk.pedersen_verify_tally_batch(tallies), expt)
    assert seen == [5] and list(expt) == [1, 1, 1, 0, 1]
    hk.set_backend()


def test_halfagg_adapter(hk, ref):
    """secp256k1_amd_schnorrsig_aggverify: the reference's argument list; no backend / failing backend / checking backend all give
    the reference's verdict, the checking backend sees the xonly_pubkey objects and the aggregate exactly as handed in."""
    rng = np.random.default_rng(505)
    n = 6
    sigs, msgs, pks = ref.make_schnorr(n, rng)
    agg = ref.halfagg_aggregate(pks, msgs, sigs)
    objs = ref.xonly_objects(pks)
    bad = bytearray(agg); bad[40] ^= 1
    seen = []

    def chk(engine, result, pk, fmt, m, cnt, a, alen):
        assert fmt == 1 and cnt == n
        assert np.array_equal(_arr(pk, 64 * n).reshape(n, 64), objs) and np.array_equal(_arr(m, 32 * n).reshape(n, 32), msgs)
        abytes = _arr(a, alen).tobytes()
        _arr(result, 4, np.int32)[0] = ref.halfagg_verify(pks, msgs, abytes)
        seen.append(abytes)
        return 1
    for be in (None, _failing(hookapi.AGG_FN), hookapi.AGG_FN(chk)):
        hk.set_backend(aggverify=be)
        f0 = hk.stats()
        assert hk.schnorrsig_aggverify(objs, msgs, agg) == 1
        assert hk.schnorrsig_aggverify(objs, msgs, bytes(bad)) == 0
        assert hk.schnorrsig_aggverify(objs, msgs, agg[:-1]) == 0                  # wrong length: the reference's verdict is 0
        if be is not None:
            d = (hk.stats()[0] - f0[0], hk.stats()[1] - f0[1])
            assert d == ((3, 0) if seen else (0, 3))
    assert seen == [agg, bytes(bad), agg[:-1]]
    hk.set_backend()


def test_rewind_adapter_cpu_paths(hk, ref):
    """secp256k1_amd_rangeproof_rewind_batch without a backend and with a backend that fails: per item what
    secp256k1_rangeproof_rewind returns and writes (right nonce, wrong nonce, corrupted proof; with and without a message buffer)."""
    rng = np.random.default_rng(506)
    commits, plist, gens, values, blinds, nonces, msgs = ref.make_rangeproofs_msg(5, rng, msg_len=48, min_bits=32)
    nn = nonces.copy(); nn[1, 0] ^= 1                                  # wrong nonce: the reference rejects
    pl = list(plist); b = bytearray(pl[3]); b[len(b) // 2] ^= 2; pl[3] = bytes(b)
    for cap in (4096, 0):
        exp = ref.rangeproof_rewind_many(commits, pl, gens, nn, msg_capacity=cap)
        for be in (None, _failing(hookapi.REWIND_FN)):
            hk.set_backend(rewind=be)
            got = hk.rangeproof_rewind_batch(commits, pl, gens, nn, msg_capacity=cap)
            assert np.array_equal(got[0], exp[0]) and list(exp[0]) == [1, 0, 1, 0, 1]
            ok = exp[0] == 1
            assert np.array_equal(got[1][ok], exp[1][ok]) and np.array_equal(got[2][ok], exp[2][ok])
            assert not got[1][~ok].any() and not got[2][~ok].any()            # rejected items: zeroed outputs
            assert got[3] == exp[3]
            assert np.array_equal(got[4][ok], exp[4][ok]) and np.array_equal(got[5][ok], exp[5][ok])
            if cap:
                assert all(got[3][i][:48] == msgs[i].tobytes() for i in range(5) if ok[i])
    hk.set_backend()


def test_ecmult_batch_and_bppp_batch_adapters_cpu_paths(hk, ref):
    """the two adapters added for the single double multiplication (src/ecmult.h:47) and for the BP++ norm-argument verifier
    (bppp_norm_product_impl.h:425) with the reference's own types: without a backend they are the library's per-item calls; a checking
    backend sees exactly the caller's items in the engine's byte formats; a failing backend falls back and is counted"""
    rng = np.random.default_rng(511)
    n = 24
    a = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(n)])
    na = rng.integers(0, 256, (n, 32), dtype=np.uint8); ng = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    inf = np.zeros(n, np.uint8); inf[3] = 1; na[5] = 0; ng[5] = 0
    exp, einf = ref.ecmult_batch(a, na, ng, inf)
    hk.set_backend()
    r, ri = hk.ecmult_batch(a, na, ng, inf)
    assert np.array_equal(ri, einf) and np.array_equal(r, exp)
    e2, ei2 = ref.ecmult_batch(a, na, None, inf)
    r, ri = hk.ecmult_batch(a, na, None, inf)
    assert np.array_equal(ri, ei2) and np.array_equal(r, e2)
    seen = []
    EB_FN = ctypes.CFUNCTYPE(ctypes.c_int, *([ctypes.c_void_p] * 7 + [ctypes.c_size_t]))

    def eb(engine, r_xy, r_inf, a_xy, a_inf, pna, png, m):
        A = _arr(a_xy, 64 * m).reshape(m, 64).copy(); AI = _arr(a_inf, m).copy()
        x, fl = ref.ecmult_batch(A, _arr(pna, 32 * m).reshape(m, 32).copy(), None if not png else _arr(png, 32 * m).reshape(m, 32).copy(), AI)
        _arr(r_xy, 64 * m)[:] = x.reshape(-1); _arr(r_inf, 4 * m, np.int32)[:] = fl
        seen.append(m)
        return 1
    cb = EB_FN(eb)
    hk.set_backend(ecmult_batch=cb)
    s0 = hk.stats()
    r, ri = hk.ecmult_batch(a, na, ng, inf)
    assert seen == [n] and hk.stats() == (s0[0] + 1, s0[1]) and np.array_equal(ri, einf) and np.array_equal(r, exp)
    hk.set_backend(ecmult_batch=_failing(EB_FN))
    r, ri = hk.ecmult_batch(a, na, ng, inf)
    assert hk.stats() == (s0[0] + 1, s0[1] + 1) and np.array_equal(ri, einf) and np.array_equal(r, exp)
    # BP++: reference-made proofs, one corrupted
    proofs, trs, rhos, gens, gl, cvs, commits = ref.make_bppp(5, rng, 8, 4)
    proofs[2, 40] ^= 1
    want = np.array(ref.bppp_verify_many(proofs, trs, rhos, gens, gl, cvs, commits), np.int32)
    hk.set_backend()
    assert np.array_equal(hk.bppp_verify_batch(proofs, trs, rhos, gens, gl, cvs, commits), want) and list(want) == [1, 1, 0, 1, 1]
    BP_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                             ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t)
    got_args = []

    def bp(engine, results, pr, plen, tr, rh, gs, n_gens, g_len, cv, c_len, cm, m):
        P_ = _arr(pr, plen * m).reshape(m, plen).copy()
        res = ref.bppp_verify_many(P_, _arr(tr, 104 * m).reshape(m, 104).copy(), _arr(rh, 32 * m).reshape(m, 32).copy(), _arr(gs, 33 * n_gens).reshape(n_gens, 33).copy(), g_len,
                                   _arr(cv, 32 * c_len * m).reshape(m, c_len, 32).copy(), _arr(cm, 33 * m).reshape(m, 33).copy())
        _arr(results, 4 * m, np.int32)[:] = np.array(res, np.int32)
        got_args.append((m, plen, n_gens, g_len, c_len))
        return 1
    cbp = BP_FN(bp)
    hk.set_backend(bppp_batch=cbp)
    s0 = hk.stats()
    assert np.array_equal(hk.bppp_verify_batch(proofs, trs, rhos, gens, gl, cvs, commits), want)
    assert got_args == [(5, proofs.shape[1], gens.shape[0], gl, cvs.shape[1])] and hk.stats() == (s0[0] + 1, s0[1])
    hk.set_backend(bppp_batch=_failing(BP_FN))
    assert np.array_equal(hk.bppp_verify_batch(proofs, trs, rhos, gens, gl, cvs, commits), want) and hk.stats() == (s0[0] + 1, s0[1] + 1)
    hk.set_backend()


def test_msm_min_terms_default_keeps_small_sums_on_the_cpu(hk, ref):
    """SECP256K1_AMD_MSM_MIN_TERMS_DEFAULT: below it secp256k1_ecmult_multi_var_amd does not go to the backend at all"""
    rng = np.random.default_rng(512)
    seen = []
    hk.set_backend(msm=_msm_checker(ref, seen))
    hk.set_msm_min_terms(hookapi.Hooked.MSM_MIN_TERMS_DEFAULT)
    try:
        for n, goes in ((13, False), (255, False), (256, True), (300, True)):
            sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            pts = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(16)])[np.arange(n) % 16]
            exy, einf = ref.ecmult_multi(sc, pts)
            seen.clear()
            xy, fl, _ = hk.ecmult_multi(sc, pts)
            assert fl == einf and np.array_equal(xy, exy) and (seen == [n]) == goes, n
    finally:
        hk.set_msm_min_terms(0)
        hk.set_backend()


def test_asynchronous_adapter_without_a_backend(hk, ref):
    """secp256k1_amd_rangeproof_verify_batch_submit / _wait with no backend installed (and with only half of the pair): the library's
    own loop runs at submission time, the ticket is 0, waiting for it is a no-op -- same verdicts as the synchronous adapter."""
    rng = np.random.default_rng(515)
    commits, proofs, gens, _ = ref.make_rangeproofs(5, rng, min_bits=8)
    plist = list(proofs); bad = bytearray(plist[2]); bad[40] ^= 1; plist[2] = bytes(bad)
    exp = ref.rangeproof_verify_many(commits, plist, gens)
    hk.set_backend()
    t = hk.rangeproof_verify_batch_submit(commits, plist, gens)
    assert t[0] == 0
    r = hk.rangeproof_verify_batch_wait(t)
    assert np.array_equal(r[0], exp[0]) and np.array_equal(r[1], exp[1]) and np.array_equal(r[2], exp[2]) and r[0].sum() == 4
    calls = []
    def sub(engine, ticket, *a):
        calls.append(1)
        return 0
    SUB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, *([ctypes.c_void_p] * 9), ctypes.c_size_t)
    WAIT = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64)
    fs, fw = SUB(sub), WAIT(lambda e, t: 0)
    s0 = hk.stats()
    hk.set_backend(rangeproof_submit=fs)                                 # half a pair: not used
    t = hk.rangeproof_verify_batch_submit(commits, plist, gens)
    assert t[0] == 0 and not calls and hk.stats() == s0
    hk.set_backend(rangeproof_submit=fs, rangeproof_wait=fw)             # a backend that refuses: fallback at submission time
    t = hk.rangeproof_verify_batch_submit(commits, plist, gens)
    assert t[0] == 0 and calls == [1] and hk.stats() == (s0[0], s0[1] + 1)
    assert np.array_equal(hk.rangeproof_verify_batch_wait(t)[0], exp[0])
    hk.set_backend()
