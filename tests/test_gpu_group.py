"""GPU tier: the per-device table pool and the C-ABI engine groups (include/secp256k1_zkp_amd.h).
* engines on one device share the device's tables: a second engine costs no table memory and no table build, a generator cached through
  one is served by the other, four verifier threads with an engine each give the reference's verdicts;
* a group (here: two engines on device 0 -- the 1-GPU boxes of this tier have no second device; on a node the entries are its GPUs)
  cuts batches of independent items into ranges and shards one multi-scalar multiplication by terms: results equal the single engine's and
  the reference's, bit for bit, for the packed and the pointer form, with failures propagated."""
import ctypes
import threading
import time

import numpy as np
import pytest

from tests.refapi import GENERATOR_H, G_XY

pytestmark = pytest.mark.gpu


def test_engines_share_the_device_tables(engine, ref):
    import torch
    from secp256k1_zkp_amd import Engine
    rng = np.random.default_rng(1201)
    L = engine._lib
    sz = ctypes.c_size_t(0)
    g0 = L.s2k_engine_gtable(engine._h, ctypes.byref(sz))                  # (builds the table if no call has needed it yet)
    engine.cache_generator(GENERATOR_H); engine.sync()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    t0 = time.time(); e2 = Engine(0); dt = time.time() - t0
    try:
        assert L.s2k_engine_gtable(e2._h, ctypes.byref(sz)) == g0            # one table of G per device
        assert e2.generator_cached(GENERATOR_H)                            # ... and one cache of generator tables
        free1, _ = torch.cuda.mem_get_info(0)
        assert free0 - free1 < (1 << 30), (free0 - free1)                   # no second 11.8 GB table
        assert dt < 2.0, dt                                                 # streams, events, pinned mailbox only (measured < 0.25 s; the memory check above is what rules out a second table)
        c, p, g, _ = ref.make_rangeproofs(40, rng, min_bits=64)
        q = bytearray(p[3]); q[100] ^= 1; p[3] = bytes(q)
        want = ref.rangeproof_verify_many(c, p, g)
        for e in (engine, e2):
            res, mn, mx = e.rangeproof_verify_batch(c, p, g)
            assert np.array_equal(res, want[0]) and np.array_equal(mx, want[2])
            assert e.rp_handback()[0] > 0                                    # the shared-generator form served it: H's table is the pool's
        ga = np.frombuffer(ref.rand_point(rng), np.uint8)
        e2.cache_generator(ga)
        assert engine.generator_cached(ga)
        c2, p2, g2, _ = ref.make_rangeproofs(12, rng, min_bits=16, gens64=np.tile(ga, (12, 1)))
        res, _, _ = engine.rangeproof_verify_batch(c2, p2, g2)               # built on e2's stream, used on engine's: ordered by the build's event
        assert res.all() and engine.rp_handback()[0] > 0
    finally:
        e2.close()


def test_four_verifier_threads_with_an_engine_each(engine, ref):
    from secp256k1_zkp_amd import Engine
    rng = np.random.default_rng(1202)
    engs = [Engine(0) for _ in range(4)]
    try:
        jobs = []
        for t in range(4):
            c, p, g, _ = ref.make_rangeproofs(48, rng, min_bits=(64, 20, 52, 64)[t])
            for i in range(0, 48, 7):
                q = bytearray(p[i]); q[int(rng.integers(0, len(q)))] ^= 1 << int(rng.integers(0, 8)); p[i] = bytes(q)
            jobs.append((c, p, g, ref.rangeproof_verify_many(c, p, g)))
        errors = []

        def worker(k):
            c, p, g, want = jobs[k]
            try:
                for _ in range(5):
                    res, mn, mx = engs[k].rangeproof_verify_batch(c, p, g)
                    if not (np.array_equal(res, want[0]) and np.array_equal(mn, want[1]) and np.array_equal(mx, want[2])):
                        errors.append(k)
            except Exception as ex:      # noqa: BLE001
                errors.append((k, repr(ex)))
        th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
        for t in th: t.start()
        for t in th: t.join()
        assert errors == []
    finally:
        for e in engs: e.close()


@pytest.fixture(scope="module")
def group2():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu needs a GPU")
    from secp256k1_zkp_amd import Group
    devs = [0, 1] if torch.cuda.device_count() > 1 else [0, 0]
    g = Group(devs)
    yield g
    g.close()


def test_group_replica_dispatch(group2, engine, ref):
    rng = np.random.default_rng(1203)
    assert len(group2) == 2
    c1, p1, g1, _ = ref.make_rangeproofs(70, rng, min_bits=64)
    c2, p2, g2, _ = ref.make_rangeproofs(31, rng, min_bits=12)
    C = np.concatenate([c1, c2]); P = p1 + p2; Gn = np.concatenate([g1, g2])
    for i in range(0, len(P), 9):
        q = bytearray(P[i]); q[int(rng.integers(0, len(q)))] ^= 1 << int(rng.integers(0, 8)); P[i] = bytes(q)
    extra = [b"" if i % 3 else bytes([i]) for i in range(len(P))]
    for ex in (None, extra):
        want = ref.rangeproof_verify_many(C, P, Gn) if ex is None else ref.rangeproof_verify_many_extra(C, P, Gn, ex)
        res, mn, mx = group2.rangeproof_verify_batch(C, P, Gn, extra=ex)
        one = engine.rangeproof_verify_batch(C, P, Gn, extra=ex)
        assert np.array_equal(res, want[0]) and np.array_equal(mn, want[1]) and np.array_equal(mx, want[2])
        assert np.array_equal(res, one[0]) and np.array_equal(mx, one[2])
    for n in (1, 2, 3):                                                     # fewer items than engines / uneven shares
        res, _, _ = group2.rangeproof_verify_batch(C[:n], P[:n], Gn[:n])
        assert np.array_equal(res, ref.rangeproof_verify_many(C[:n], P[:n], Gn[:n])[0])
    # the pointer form (the reference's own objects), through ctypes
    n = len(P)
    L = group2._lib
    objs = [ctypes.create_string_buffer(C[i].tobytes() + b"\0" * 31, 64) for i in range(n)]
    gobj = [ctypes.create_string_buffer(Gn[i].tobytes(), 64) for i in range(n)]
    pbuf = [ctypes.create_string_buffer(P[i], len(P[i])) for i in range(n)]
    arr = lambda xs: (ctypes.c_void_p * n)(*[ctypes.addressof(x) for x in xs])
    plens = (ctypes.c_size_t * n)(*[len(x) for x in P])
    res = np.zeros(n, np.int32); mn = np.zeros(n, np.uint64); mx = np.zeros(n, np.uint64)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert L.secp256k1_rangeproof_verify_batch_ptrs_group(group2._h, vp(res), vp(mn), vp(mx), arr(objs), arr(pbuf), plens, None, None, arr(gobj), n) == 1
    want = ref.rangeproof_verify_many(C, P, Gn)
    assert np.array_equal(res, want[0]) and np.array_equal(mx, want[2])
    # BIP-340
    sigs, msgs, pks = ref.make_schnorr(301, rng, threads=4)
    sigs[::13, 33] ^= 2; pks[5::31, 0] ^= 0x40
    assert np.array_equal(group2.schnorrsig_verify_batch(sigs, msgs, pks), ref.schnorr_verify_many(sigs, msgs, pks))
    # an argument error reaches the caller as such, and nothing is marked valid
    res[:] = 1
    assert L.secp256k1_rangeproof_verify_batch_ptrs_group(group2._h, vp(res), vp(mn), vp(mx), None, arr(pbuf), plens, None, None, arr(gobj), n) == 0
    assert L.s2k_last_status() == 2


def test_group_term_sharded_msm(group2, engine, ref):
    import torch
    rng = np.random.default_rng(1204)
    Gpt = np.frombuffer(G_XY, np.uint8)
    for n, with_g in ((1, True), (2, False), (63, True), (5000, True), (40000, False)):
        ks = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        pts, _ = engine.ecmult_batch(np.tile(Gpt, (n, 1)), np.zeros((n, 32), np.uint8), ng=ks)
        pts[::7] = np.frombuffer(ref.rand_point(rng), np.uint8)              # some points made by the reference
        sc = rng.integers(0, 256, (n, 32), dtype=np.uint8); sc[::11] = 0
        inf = np.zeros(n, np.uint8); inf[3::17] = 1
        gsc = rng.integers(0, 256, 32, dtype=np.uint8) if with_g else None
        want, winf = ref.ecmult_multi(sc, pts, None if gsc is None else gsc.tobytes(), inf)
        got, ginf = group2.ecmult_multi(sc, pts, gsc, inf)
        assert ginf == winf and got.tobytes() == want.tobytes(), (n, with_g)
        one, oinf = engine.ecmult_multi(sc, pts, gsc, inf)
        assert one.tobytes() == got.tobytes() and oinf == ginf
    # resident slices (every slice on its engine's GPU)
    n = 30000
    ks = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pts, _ = engine.ecmult_batch(np.tile(Gpt, (n, 1)), np.zeros((n, 32), np.uint8), ng=ks)
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    gsc = rng.integers(0, 256, 32, dtype=np.uint8)
    cut = [0, 17000, n]
    devs = group2.devices
    scl = [torch.tensor(sc[cut[i]:cut[i + 1]]).to(torch.device("cuda", devs[i])) for i in range(2)]
    ptl = [torch.tensor(pts[cut[i]:cut[i + 1]]).to(torch.device("cuda", devs[i])) for i in range(2)]
    g0 = torch.tensor(gsc).to(torch.device("cuda", devs[0]))
    torch.cuda.synchronize()
    got, ginf = group2.ecmult_multi_dev(scl, ptl, g_sc_dev0=g0)
    want, winf = ref.ecmult_multi(sc, pts, gsc.tobytes())
    assert ginf == winf and got.tobytes() == want.tobytes()


def test_group_behind_the_reference_side_hook(group2, ref):
    """the hook's backend table takes a group as its `engine`: the `_group` entry points have the single-engine prototypes"""
    from tests import hookapi
    try:
        hk = hookapi.Hooked()
    except OSError as ex:
        pytest.fail(f"-m gpu needs oracle/_ref/libsecp256k1_hooked.so: {ex}")
    rng = np.random.default_rng(1205)
    L = group2._lib
    addr = lambda name: ctypes.cast(getattr(L, name), ctypes.c_void_p).value
    hk.set_backend(engine=group2._h, rangeproof_ptrs=addr("secp256k1_rangeproof_verify_batch_ptrs_group"))
    try:
        c, p, g, _ = ref.make_rangeproofs(50, rng, min_bits=32)
        q = bytearray(p[10]); q[77] ^= 8; p[10] = bytes(q)
        want = ref.rangeproof_verify_many(c, p, g)
        s0 = hk.stats()
        res, mn, mx = hk.rangeproof_verify_batch(c, p, g)
        s1 = hk.stats()
        assert np.array_equal(res, want[0]) and np.array_equal(mx, want[2]) and s1[0] == s0[0] + 1 and s1[1] == s0[1]
    finally:
        hk.set_backend()


def test_group_many_sums(group2, engine, ref):
    """Group.ecmult_multi_many: K independent sums cut into contiguous ranges over the members (balanced by terms, one host thread per
    member, no exchange); every sum equals the one-engine call's and the reference's secp256k1_ecmult_multi_var (src/ecmult_impl.h:822-867).
    Ragged sizes with empty sums at both ends, a sum that is most of the terms, generator terms, infinite points; K below the group size."""
    from tests.test_gpu_msm import _points
    rng = np.random.default_rng(1207)
    for sizes, with_g in (([0, 5, 300, 0, 1, 88, 89, 1024, 2, 0], True), ([3000, 1, 1, 1], False), ([7], True), ([0], False), ([], False), ([64] * 50, True)):
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        n = int(off[-1]); k = len(sizes)
        pts = _points(engine, rng, max(n, 1))[:n]
        sc = rng.integers(0, 256, (n, 32), dtype=np.uint8); sc[::13] = 0
        inf = np.zeros(n, np.uint8); inf[5::29] = 1
        g = rng.integers(0, 256, (k, 32), dtype=np.uint8) if with_g else None
        got, ginf = group2.ecmult_multi_many(sc, pts, off, g, inf)
        assert got.shape == (k, 64) and ginf.shape == (k,)
        if k:
            one, oinf = engine.ecmult_multi_many(sc, pts, off, g, inf)
            assert np.array_equal(one, got) and np.array_equal(oinf, ginf), sizes
        for s in range(k):
            lo, hi = int(off[s]), int(off[s + 1])
            exp, einf = ref.ecmult_multi(sc[lo:hi], pts[lo:hi], None if g is None else bytes(g[s]), inf[lo:hi])
            assert int(ginf[s]) == einf and np.array_equal(got[s], exp), (sizes, s)
    with pytest.raises(Exception):
        group2.ecmult_multi_many(np.zeros((4, 32), np.uint8), np.zeros((4, 64), np.uint8), np.array([0, 3, 2, 4], np.uint64))
