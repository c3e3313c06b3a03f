"""GPU tier: the reference-side adapters (integration/secp256k1_amd_hook.c inside oracle/_ref/libsecp256k1_hooked.so) driving the
REAL engine through its C ABI -- the complete drop-in path a maintainer would ship: reference types in, packing, HIP kernels,
verdicts out -- compared item by item with the unmodified reference; plus the forced engine failure on a box that has a GPU
(the registered entry points are handed a NULL engine), which must fall back to the CPU path with the reference's verdicts."""
import ctypes
import json
import os

import numpy as np
import pytest

from tests import hookapi
from tests.refapi import GENERATOR_H, G_XY, P

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hk(engine):
    if not os.path.exists(hookapi.HOOKED_PATH):
        pytest.fail("-m gpu needs oracle/_ref/libsecp256k1_hooked.so (make -C oracle hooked)")
    h = hookapi.Hooked()
    yield h
    h.set_backend()


def _install(hk, engine, handle, ptrs=False):
    L = engine._lib
    addr = lambda name: ctypes.cast(getattr(L, name), ctypes.c_void_p).value
    hk.set_backend(engine=handle, rangeproof=addr("secp256k1_rangeproof_verify_batch"), msm=addr("s2k_ecmult_multi"),
                   rangeproof_ptrs=addr("secp256k1_rangeproof_verify_batch_ptrs") if ptrs else None,
                   schnorr=addr("secp256k1_schnorrsig_verify_batch"), surjection=addr("secp256k1_surjectionproof_verify_batch"),
                   tally=addr("secp256k1_pedersen_verify_tally_batch"), aggverify=addr("secp256k1_schnorrsig_aggverify_amd"), rewind=addr("secp256k1_rangeproof_rewind_batch"))


def _workload(ref, rng):
    commits, proofs, gens, _ = ref.make_rangeproofs(24, rng, min_bits=64)
    c2, p2, g2, _ = ref.make_rangeproofs(8, rng, min_bits=12, exp=1, min_value=5)
    plist = list(proofs) + list(p2); c = np.concatenate([commits, c2]); g = np.concatenate([gens, g2])
    v = json.load(open(os.path.join(HERE, "golden", "rangeproof_vectors.json")))
    assert v["generator"] == "secp256k1_generator_h"
    for x in v["vectors"]:
        plist.append(bytes.fromhex(x["proof"])); c = np.concatenate([c, np.frombuffer(bytes.fromhex(x["commit33"]), np.uint8)[None]])
        g = np.concatenate([g, np.frombuffer(GENERATOR_H, np.uint8)[None]])
    bad = bytearray(plist[0]); bad[100] ^= 1
    plist += [bytes(bad), plist[1][:-32], plist[2] + b"\0", b"", plist[3]]
    c = np.concatenate([c, c[[0, 1, 2, 3]], c[[4]]]); g = np.concatenate([g, g[[0, 1, 2, 3, 3]]])      # last: proof 3 against commitment 4
    return c, plist, g


def test_rangeproofs_through_the_hook(hk, engine, ref):
    rng = np.random.default_rng(601)
    c, plist, g = _workload(ref, rng)
    exp = ref.rangeproof_verify_many(c, plist, g)
    _install(hk, engine, engine._h)
    s0 = hk.stats()
    res, mn, mx = hk.rangeproof_verify_batch(c, plist, g)
    assert hk.stats() == (s0[0] + 1, s0[1])
    assert np.array_equal(res, exp[0]) and np.array_equal(mn, exp[1]) and np.array_equal(mx, exp[2])
    assert 0 < res.sum() < len(plist)
    # the pointer-array seam (secp256k1_rangeproof_verify_batch_ptrs): the engine gathers from the library's own objects, with extra_commit
    _install(hk, engine, engine._h, ptrs=True)
    s0 = hk.stats()
    res, mn, mx = hk.rangeproof_verify_batch(c, plist, g)
    assert hk.stats() == (s0[0] + 1, s0[1])
    assert np.array_equal(res, exp[0]) and np.array_equal(mn, exp[1]) and np.array_equal(mx, exp[2])
    extra = [b"" if i % 3 else b"script %d" % i for i in range(len(plist))]
    ce, pe, ge = ref.make_rangeproofs_extra(len(plist), rng, extra, min_bits=20)
    extra[4] = extra[4] + b"x"; extra[6] = b""
    exp_e = ref.rangeproof_verify_many_extra(ce, pe, ge, extra)
    res, mn, mx = hk.rangeproof_verify_batch(ce, pe, ge, extra=extra)
    assert np.array_equal(res, exp_e[0]) and np.array_equal(mx, exp_e[2]) and 0 < res.sum() < len(pe)
    # forced engine failure (NULL engine): CPU fallback, same verdicts, counted
    _install(hk, engine, None, ptrs=True)
    s0 = hk.stats()
    res2, _, _ = hk.rangeproof_verify_batch(c, plist, g)
    assert hk.stats() == (s0[0], s0[1] + 1) and np.array_equal(res2, exp[0])
    _install(hk, engine, None)
    s0 = hk.stats()
    res2, mn2, mx2 = hk.rangeproof_verify_batch(c, plist, g)
    assert hk.stats() == (s0[0], s0[1] + 1)
    assert engine._lib.s2k_last_status() == 1
    assert np.array_equal(res2, exp[0]) and np.array_equal(mn2, exp[1]) and np.array_equal(mx2, exp[2])


def test_msm_seam_and_bppp_vectors(hk, engine, ref):
    rng = np.random.default_rng(602)
    _install(hk, engine, engine._h)
    for n in (1, 7, 150, 400, 3000):
        sc = rng.integers(0, 256, (n, 32), dtype=np.uint8); sc[0] = 0
        pts = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(min(n, 64))])[np.arange(n) % min(n, 64)]
        inf = np.zeros(n, np.uint8); inf[n // 2] = 1
        g = bytes(rng.integers(0, 256, 32, dtype=np.uint8)) if n % 2 else None
        exy, einf = ref.ecmult_multi(sc, pts, g_sc=g, pt_inf=inf)
        s0 = hk.stats()
        xy, fl, calls = hk.ecmult_multi(sc, pts, g_sc=g, pt_inf=inf)
        assert hk.stats()[0] == s0[0] + 1 and calls == n and fl == einf and np.array_equal(xy, exy), n
        xy, fl, calls = hk.ecmult_multi(sc, pts, g_sc=g, pt_inf=inf, fail_at=n - 1)
        assert fl == -1 and calls == n
    # k*G - k*G
    k = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    negy = ((P - int.from_bytes(G_XY[32:], "big")) % P).to_bytes(32, "big")
    xy, fl, _ = hk.ecmult_multi(np.concatenate([k, k]), np.stack([np.frombuffer(G_XY, np.uint8), np.frombuffer(G_XY[:32] + negy, np.uint8)]))
    assert fl == 1
    # the reference's BP++ verifier with its MSMs on the GPU: all vectors of modules/bppp/test_vectors/verify.h
    g = json.load(open(os.path.join(HERE, "golden", "bppp_verify_vectors.json")))
    gens = bytes.fromhex(g["gens"])
    st = np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], "<u4").tobytes() + b"\0" * 72
    s0 = hk.stats()
    for v in g["vectors"]:
        proof = bytes.fromhex(v["proof"]); cvec = b"".join(bytes.fromhex(c) for c in v["c_vec"]); nlen = v["n_vec_len"]; clen = len(v["c_vec"])
        r = hk.lib.ref_bppp_norm_verify(proof, len(proof), st, bytes.fromhex(v["rho"]), gens[:33 * (nlen + clen)], nlen + clen, nlen, cvec, clen, bytes.fromhex(v["commit33"]))
        assert r == v["result"], v["index"]
    assert hk.stats()[0] > s0[0]
    # random proofs made by the reference prover, g_len 16 / h_len 4
    proofs, trs, rhos, gens_, gl, cvs, commits = ref.make_bppp(4, rng, 16, 4)
    proofs[1, 70] ^= 1
    exp = ref.bppp_verify_many(proofs, trs, rhos, gens_, gl, cvs, commits)
    got = [hk.lib.ref_bppp_norm_verify(proofs[i].tobytes(), proofs.shape[1], trs[i].tobytes(), rhos[i].tobytes(), gens_.tobytes(), gens_.shape[0], gl,
                                       cvs[i].tobytes(), cvs.shape[1], commits[i].tobytes()) for i in range(4)]
    assert got == list(exp) and list(exp) == [1, 0, 1, 1]
    # forced failure: CPU path
    _install(hk, engine, None)
    sc = rng.integers(0, 256, (20, 32), dtype=np.uint8); pts = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(20)])
    exy, einf = ref.ecmult_multi(sc, pts)
    f0 = hk.stats()[1]
    xy, fl, _ = hk.ecmult_multi(sc, pts)
    assert fl == einf and np.array_equal(xy, exy) and hk.stats()[1] == f0 + 1


def test_schnorr_surjection_tally_through_the_hook(hk, engine, ref):
    rng = np.random.default_rng(603)
    sigs, msgs, pks = ref.make_schnorr(70, rng)
    sigs[3, 9] ^= 1; msgs[11, 2] ^= 1; sigs[40, 40] ^= 0x80
    objs = ref.xonly_objects(pks); exp = ref.schnorr_verify_many(sigs, msgs, pks)
    items = [ref.make_surjection(rng, k, min(k, 3)) for k in (1, 2, 3, 9, 30)]
    bad = bytearray(items[2][0]); bad[-3] ^= 1; items.append((bytes(bad), items[2][1], items[2][2]))
    exps = np.array([ref.surjection_verify(p, t, o) for p, t, o in items], np.int32)
    tallies = [ref.make_balanced_tally(rng, 2, 3), ref.make_balanced_tally(rng, 5, 1), ref.make_balanced_tally(rng, 1, 9)]
    a, b = ref.make_balanced_tally(rng, 3, 2); tallies += [(a, b[:1]), (np.zeros((0, 33), np.uint8), np.zeros((0, 33), np.uint8))]
    expt = ref.pedersen_verify_tally_many(tallies)
    for handle in (engine._h, None):
        _install(hk, engine, handle)
        s0 = hk.stats()
        assert np.array_equal(hk.schnorrsig_verify_batch(sigs, msgs, objs), exp)
        assert np.array_equal(hk.surjectionproof_verify_batch(items), exps)
        assert np.array_equal(hk.pedersen_verify_tally_batch(tallies), expt)
        assert hk.stats() == ((s0[0] + 3, s0[1]) if handle else (s0[0], s0[1] + 3))
    assert exp.sum() == 67 and list(exps) == [1, 1, 1, 1, 1, 0] and list(expt) == [1, 1, 1, 0, 1]


def test_single_item_forms_have_the_reference_argument_lists(engine, ref):
    """secp256k1_{schnorrsig_verify,pedersen_verify_tally,surjectionproof_verify}_amd (include/secp256k1_schnorrsig.h:178,
    secp256k1_generator.h:190, secp256k1_surjectionproof.h:256): opaque objects in, the reference's verdict out."""
    L = engine._lib
    rng = np.random.default_rng(604)
    sigs, msgs, pks = ref.make_schnorr(3, rng); sigs[1, 0] ^= 1
    objs = ref.xonly_objects(pks); exp = ref.schnorr_verify_many(sigs, msgs, pks)
    for i in range(3):
        assert L.secp256k1_schnorrsig_verify_amd(None, sigs[i].tobytes(), msgs[i].tobytes(), 32, objs[i].tobytes()) == exp[i]
        assert L.s2k_last_status() == 0
    assert L.secp256k1_schnorrsig_verify_amd(None, None, msgs[0].tobytes(), 32, objs[0].tobytes()) == 0 and L.s2k_last_status() == 2
    # tally
    pos, neg = ref.make_balanced_tally(rng, 2, 3)
    def objs33(c):
        o = np.zeros((c.shape[0], 64), np.uint8); o[:, :33] = c
        return o, hookapi._ptr_array([o[i] for i in range(c.shape[0])])
    po, pp = objs33(pos); no, npp = objs33(neg)
    assert L.secp256k1_pedersen_verify_tally_amd(None, pp, 2, npp, 3) == 1
    assert L.secp256k1_pedersen_verify_tally_amd(None, pp, 2, npp, 2) == 0 and L.s2k_last_status() == 0
    assert L.secp256k1_pedersen_verify_tally_amd(None, None, 0, None, 0) == 1
    # surjection proof object (parsed by the reference into its opaque struct)
    hkl = ctypes.CDLL(hookapi.HOOKED_PATH) if os.path.exists(hookapi.HOOKED_PATH) else ref.lib
    hkl.secp256k1_context_create.restype = ctypes.c_void_p
    ctx = hkl.secp256k1_context_create(ctypes.c_uint(1))
    for k, flip in ((1, 0), (4, 0), (12, 0), (4, 1)):
        ser, tags, out = ref.make_surjection(rng, k, min(k, 3))
        if flip:
            ser = bytearray(ser); ser[-2] ^= 1; ser = bytes(ser)
        obj = ctypes.create_string_buffer(8 + 32 + 32 * 257 + 64)
        assert hkl.secp256k1_surjectionproof_parse(ctypes.c_void_p(ctx), obj, ser, ctypes.c_size_t(len(ser))) == 1
        e = ref.surjection_verify(ser, tags, out)
        assert L.secp256k1_surjectionproof_verify_amd(None, obj, np.ascontiguousarray(tags).ctypes.data_as(ctypes.c_void_p), k,
                                                      np.ascontiguousarray(out).ctypes.data_as(ctypes.c_void_p)) == e == (0 if flip else 1)


def test_halfagg_through_the_hook(hk, engine, ref):
    """secp256k1_amd_schnorrsig_aggverify (the reference's argument list) on the real engine -- one (2n+1)-term MSM -- and, with the
    engine handle withheld, on the CPU: the reference's verdicts either way."""
    rng = np.random.default_rng(605)
    n = 300
    sigs, msgs, pks = ref.make_schnorr(n, rng)
    agg = ref.halfagg_aggregate(pks, msgs, sigs)
    objs = ref.xonly_objects(pks)
    bad = bytearray(agg); bad[32 * 7 + 3] ^= 4
    m2 = msgs.copy(); m2[100, 1] ^= 1
    for handle in (engine._h, None):
        _install(hk, engine, handle)
        s0 = hk.stats()
        assert hk.schnorrsig_aggverify(objs, msgs, agg) == 1 == ref.halfagg_verify(pks, msgs, agg)
        assert hk.schnorrsig_aggverify(objs, msgs, bytes(bad)) == 0 == ref.halfagg_verify(pks, msgs, bytes(bad))
        assert hk.schnorrsig_aggverify(objs, m2, agg) == 0 == ref.halfagg_verify(pks, m2, agg)
        assert hk.stats() == ((s0[0] + 3, s0[1]) if handle else (s0[0], s0[1] + 3))


def test_rewind_through_the_hook(hk, engine, ref):
    """secp256k1_amd_rangeproof_rewind_batch on the real engine and, with the handle withheld, on the CPU: results, blinds, values,
    messages, min/max of every item as secp256k1_rangeproof_rewind gives them (right / wrong nonce, corrupted proof, mixed shapes)."""
    rng = np.random.default_rng(606)
    c1, p1, g1, _, _, n1, m1 = ref.make_rangeproofs_msg(20, rng, msg_len=64, min_bits=64)
    c2, p2, g2, _, _, n2, m2 = ref.make_rangeproofs_msg(12, rng, msg_len=16, min_bits=10, exp=1, min_value=3)
    commits = np.concatenate([c1, c2]); plist = list(p1) + list(p2); gens = np.concatenate([g1, g2]); nonces = np.concatenate([n1, n2])
    nonces[4, 31] ^= 1; nonces[25, 0] ^= 0x80
    b = bytearray(plist[9]); b[200] ^= 1; plist[9] = bytes(b)
    for cap in (4096, 0):
        exp = ref.rangeproof_rewind_many(commits, plist, gens, nonces, msg_capacity=cap)
        ok = exp[0] == 1
        assert 0 < ok.sum() < len(plist)
        for handle in (engine._h, None):
            _install(hk, engine, handle)
            s0 = hk.stats()
            got = hk.rangeproof_rewind_batch(commits, plist, gens, nonces, msg_capacity=cap)
            assert hk.stats() == ((s0[0] + 1, s0[1]) if handle else (s0[0], s0[1] + 1))
            assert np.array_equal(got[0], exp[0])
            assert np.array_equal(got[1][ok], exp[1][ok]) and np.array_equal(got[2][ok], exp[2][ok]) and got[3] == exp[3]
            assert not got[1][~ok].any() and not got[2][~ok].any()
            assert np.array_equal(got[4][ok], exp[4][ok]) and np.array_equal(got[5][ok], exp[5][ok])


def test_ecmult_batch_and_bppp_batch_through_the_hook(hk, engine, ref):
    """secp256k1_ecmult_batch_amd and secp256k1_amd_bppp_norm_product_verify_batch (reference types) on the real engine"""
    L = engine._lib
    addr = lambda name: ctypes.cast(getattr(L, name), ctypes.c_void_p).value
    rng = np.random.default_rng(611)
    hk.set_backend(engine=engine._h, ecmult_batch=addr("s2k_ecmult_batch"), bppp_batch=addr("secp256k1_bppp_norm_product_verify_batch"))
    n = 300
    a = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(32)])[np.arange(n) % 32]
    na = rng.integers(0, 256, (n, 32), dtype=np.uint8); ng = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    inf = np.zeros(n, np.uint8); inf[9] = 1; na[11] = 0; ng[11] = 0; na[12] = 0
    s0 = hk.stats()
    for g in (ng, None):
        exp, einf = ref.ecmult_batch(a, na, g, inf)
        r, ri = hk.ecmult_batch(a, na, g, inf)
        assert np.array_equal(ri, einf) and np.array_equal(r, exp)
    assert hk.stats() == (s0[0] + 2, s0[1])
    proofs, trs, rhos, gens, gl, cvs, commits = ref.make_bppp(40, rng, 16, 4)
    proofs[7, 33] ^= 2; cvs[9, 0, 31] ^= 1
    want = np.array(ref.bppp_verify_many(proofs, trs, rhos, gens, gl, cvs, commits), np.int32)
    got = hk.bppp_verify_batch(proofs, trs, rhos, gens, gl, cvs, commits)
    assert np.array_equal(got, want) and want.sum() == 38 and hk.stats() == (s0[0] + 3, s0[1])
    hk.set_backend()


def test_concurrent_verifier_threads_on_one_engine(hk, engine, ref):
    """include/secp256k1.h:42-52: verification calls may run concurrently.  Four threads push different batches through the hook (one engine:
    its mutex makes them take turns, its scratch is theirs in turn) while another thread re-installs the backend table: every verdict equals
    the reference's."""
    import threading
    rng = np.random.default_rng(612)
    jobs = []
    for t in range(4):
        c, p, g, _ = ref.make_rangeproofs(24, rng, min_bits=(64, 12, 52, 30)[t])
        for i in range(0, 24, 5):
            q = bytearray(p[i]); q[int(rng.integers(0, len(q)))] ^= 1 << int(rng.integers(0, 8)); p[i] = bytes(q)
        jobs.append((c, p, g, ref.rangeproof_verify_many(c, p, g)))
    _install(hk, engine, engine._h, ptrs=True)
    errors = []
    stop = threading.Event()

    def worker(k):
        c, p, g, exp = jobs[k]
        try:
            for _ in range(6):
                res, mn, mx = hk.rangeproof_verify_batch(c, p, g)
                if not (np.array_equal(res, exp[0]) and np.array_equal(mn, exp[1]) and np.array_equal(mx, exp[2])):
                    errors.append(k)
        except Exception as ex:          # noqa: BLE001
            errors.append((k, repr(ex)))

    def reinstaller():
        import time
        while not stop.is_set():
            _install(hk, engine, engine._h, ptrs=True)
            time.sleep(0.02)

    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    ri = threading.Thread(target=reinstaller)
    s0 = hk.stats()
    ri.start()
    for t in th: t.start()
    for t in th: t.join()
    stop.set(); ri.join()
    assert errors == []
    # every call was served by the engine: synchronous callers queue for one of the engine's two staging sets, they are not turned away
    # (the counters are atomic: every call is counted)
    s1 = hk.stats()
    assert s1[1] == s0[1] and s1[0] == s0[0] + 24
    hk.set_backend()


def test_asynchronous_pair_through_the_hook_and_the_c_abi(hk, engine, ref):
    """submit / wait: two batches in flight give, batch by batch, the verdicts of the synchronous call; tickets are single-use, a third
    submission is refused, the inputs may be overwritten as soon as submit has returned; without a backend the adapter verifies at
    submission time and hands out ticket 0."""
    rng = np.random.default_rng(611)
    c, plist, g = _workload(ref, rng)
    exp = ref.rangeproof_verify_many(c, plist, g)
    c2, p2, g2, _ = ref.make_rangeproofs(40, rng, min_bits=10)
    p2 = list(p2); bad = bytearray(p2[7]); bad[60] ^= 2; p2[7] = bytes(bad)
    exp2 = ref.rangeproof_verify_many(c2, p2, g2)
    # C ABI, packed form, through the Python wrapper
    for rounds in range(2):
        t1 = engine.rangeproof_verify_batch_submit(c, plist, g)
        t2 = engine.rangeproof_verify_batch_submit(c2, p2, g2)
        with pytest.raises(Exception):
            engine.rangeproof_verify_batch_submit(c2, p2, g2)                  # two in flight already
        assert engine._lib.s2k_last_status() == 3                              # S2K_STATUS_BUSY, not an engine failure
        import time
        t0 = time.time()
        with pytest.raises(Exception):
            engine.rangeproof_verify_batch(c2, p2, g2)                         # a synchronous call cannot get a set either: busy, at once
        assert engine._lib.s2k_last_status() == 3 and time.time() - t0 < 5.0
        order = (t2, t1) if rounds else (t1, t2)
        got = {t[0]: engine.rangeproof_verify_batch_wait(t) for t in order}
        for t, e in ((t1, exp), (t2, exp2)):
            r = got[t[0]]
            assert np.array_equal(r[0], e[0]) and np.array_equal(r[1], e[1]) and np.array_equal(r[2], e[2])
        with pytest.raises(Exception):
            engine.rangeproof_verify_batch_wait(t1)                             # waited for already
    # a long chain, submit(k+1) before wait(k), alternating workloads; the synchronous call in between two chains
    work = [(c, plist, g, exp), (c2, p2, g2, exp2)]
    prev = None
    for k in range(7):
        w = work[k & 1]
        cur = (engine.rangeproof_verify_batch_submit(w[0], w[1], w[2]), w[3])
        if prev is not None:
            r = engine.rangeproof_verify_batch_wait(prev[0])
            assert np.array_equal(r[0], prev[1][0]) and np.array_equal(r[2], prev[1][2])
        prev = cur
    r = engine.rangeproof_verify_batch_wait(prev[0])
    assert np.array_equal(r[0], prev[1][0])
    r = engine.rangeproof_verify_batch(c2, p2, g2)
    assert np.array_equal(r[0], exp2[0])
    # the adapters, reference types
    L = engine._lib
    addr = lambda name: ctypes.cast(getattr(L, name), ctypes.c_void_p).value
    hk.set_backend(engine=engine._h, rangeproof_ptrs=addr("secp256k1_rangeproof_verify_batch_ptrs"),
                   rangeproof_submit=addr("secp256k1_rangeproof_verify_batch_ptrs_submit"), rangeproof_wait=addr("secp256k1_rangeproof_verify_batch_wait"))
    s0 = hk.stats()
    ta = hk.rangeproof_verify_batch_submit(c, plist, g)
    tb = hk.rangeproof_verify_batch_submit(c2, p2, g2)
    assert ta[0] != 0 and tb[0] != 0 and ta[0] != tb[0]
    ra = hk.rangeproof_verify_batch_wait(ta); rb = hk.rangeproof_verify_batch_wait(tb)
    assert hk.stats() == (s0[0] + 2, s0[1])
    assert np.array_equal(ra[0], exp[0]) and np.array_equal(ra[1], exp[1]) and np.array_equal(ra[2], exp[2])
    assert np.array_equal(rb[0], exp2[0]) and np.array_equal(rb[2], exp2[2])
    # refused submission (NULL engine) -> verified on the CPU at submission time, ticket 0
    hk.set_backend(engine=None, rangeproof_submit=addr("secp256k1_rangeproof_verify_batch_ptrs_submit"), rangeproof_wait=addr("secp256k1_rangeproof_verify_batch_wait"))
    s0 = hk.stats()
    tc = hk.rangeproof_verify_batch_submit(c2, p2, g2)
    assert tc[0] == 0 and hk.stats() == (s0[0], s0[1] + 1)
    rc = hk.rangeproof_verify_batch_wait(tc)
    assert np.array_equal(rc[0], exp2[0]) and np.array_equal(rc[2], exp2[2])
    hk.set_backend()


def test_asynchronous_pair_large_batches_and_extra_commit(engine, ref):
    """submit / wait with batches that span several launch groups (more than 32 768 proofs: the pipeline's chunks alternate the two scratch
    sets while the other submission's chunks are queued behind them), with extra_commit data, and waited for from another thread."""
    import threading
    rng = np.random.default_rng(613)
    base = 96
    extra = [b"" if i % 4 else b"commit-%d" % i for i in range(base)]
    c, p, g = ref.make_rangeproofs_extra(base, rng, extra, min_bits=6)
    p = list(p); bad = bytearray(p[5]); bad[-1] ^= 1; p[5] = bytes(bad); extra[9] = b"other"
    exp = ref.rangeproof_verify_many_extra(c, p, g, extra)
    assert 0 < exp[0].sum() < base
    reps = 400                                                        # 38 400 proofs per batch: two launch groups
    idx = np.tile(np.arange(base), reps)
    C = c[idx]; G = g[idx]; P = [p[i] for i in idx]; E = [extra[i] for i in idx]
    pk = engine.pack(P); ek = engine.pack(E)
    c2, p2, g2, _ = ref.make_rangeproofs(64, rng, min_bits=9)
    exp2 = ref.rangeproof_verify_many(c2, list(p2), g2)
    t1 = engine.rangeproof_verify_batch_submit(C, pk, G, extra=ek)
    t2 = engine.rangeproof_verify_batch_submit(c2, list(p2), g2)
    got = {}
    th = threading.Thread(target=lambda: got.setdefault(2, engine.rangeproof_verify_batch_wait(t2)))      # the younger ticket first, elsewhere
    th.start(); th.join()
    got[1] = engine.rangeproof_verify_batch_wait(t1)
    assert np.array_equal(got[2][0], exp2[0]) and np.array_equal(got[2][2], exp2[2])
    r = got[1]
    assert np.array_equal(r[0], exp[0][idx]) and np.array_equal(r[1], exp[1][idx]) and np.array_equal(r[2], exp[2][idx])
    # and the same big batch synchronously
    r = engine.rangeproof_verify_batch(C, pk, G, extra=ek)
    assert np.array_equal(r[0], exp[0][idx]) and np.array_equal(r[2], exp[2][idx])
