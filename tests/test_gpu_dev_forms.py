"""GPU tier: the `_dev` entry points (every array already in HBM, stream-ordered, nothing read back) of BP++ verify, BP++ commit,
half-aggregate verify, Pedersen tallies and rangeproof rewind give exactly what the host forms give -- which the other GPU tests pin
against the reference."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _d(a):
    import torch
    return torch.tensor(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def test_bppp_verify_and_commit_dev(engine, ref):
    import torch
    L = engine._lib
    rng = np.random.default_rng(901)
    proofs, trs, rhos, gens, gl, cvs, commits = ref.make_bppp(6, rng, 16, 4)
    proofs[2, 80] ^= 1
    exp = engine.bppp_norm_product_verify_batch(proofs, trs, rhos, gens, gl, cvs, commits)
    assert list(exp) == [1, 1, 0, 1, 1, 1]
    res = torch.full((6,), 7, dtype=torch.int32, device="cuda")
    dp, dt, dr, dg, dc, dm = _d(proofs), _d(trs), _d(rhos), _d(gens), _d(cvs), _d(commits)
    torch.cuda.synchronize()
    g_host = np.ascontiguousarray(gens)
    ok = L.secp256k1_bppp_norm_product_verify_batch_dev(engine._h, None, _p(res), _p(dp), proofs.shape[1], _p(dt), _p(dr), _p(dg), g_host.ctypes.data_as(ctypes.c_void_p),
                                                        gens.shape[0], gl, _p(dc), cvs.shape[1], _p(dm), 6)
    assert ok == 1 and L.s2k_engine_sync(engine._h) == 1
    assert np.array_equal(res.cpu().numpy(), exp)
    # commit
    sc = lambda *shape: (rng.integers(0, 256, shape + (32,), dtype=np.uint8) & np.array([0x7F] + [0xFF] * 31, np.uint8))
    nv, lv, cv, mu = sc(5, 16), sc(5, 4), sc(5, 4), sc(5)
    ehost, okh = engine.bppp_commit_batch(gens, 16, nv, lv, cv, mu)
    out = torch.zeros(5 * 33, dtype=torch.uint8, device="cuda"); r2 = torch.zeros(5, dtype=torch.int32, device="cuda")
    dn, dl, dcv, dmu = _d(nv), _d(lv), _d(cv), _d(mu)
    torch.cuda.synchronize()
    ok = L.secp256k1_bppp_commit_batch_dev(engine._h, None, _p(out), _p(r2), _p(dg), g_host.ctypes.data_as(ctypes.c_void_p), gens.shape[0], 16, _p(dn), _p(dl), _p(dcv), 4, _p(dmu), 5)
    assert ok == 1 and L.s2k_engine_sync(engine._h) == 1
    assert np.array_equal(out.cpu().numpy().reshape(5, 33), ehost) and r2.cpu().numpy().all() and okh.all()


def test_halfagg_tally_rewind_dev(engine, ref):
    import torch
    L = engine._lib
    rng = np.random.default_rng(902)
    # half-aggregate
    n = 40
    sigs, msgs, pks = ref.make_schnorr(n, rng)
    agg = ref.halfagg_aggregate(pks, msgs, sigs)
    for mutate in (0, 1):
        a = bytearray(agg)
        if mutate:
            a[37] ^= 1
        a = bytes(a)
        exp = engine.schnorrsig_aggverify(pks, msgs, a)
        r = torch.full((4,), 9, dtype=torch.int32, device="cuda")
        dpk, dm, da = _d(pks), _d(msgs), _d(np.frombuffer(a, np.uint8))
        torch.cuda.synchronize()
        assert L.secp256k1_schnorrsig_aggverify_dev(engine._h, None, _p(r), _p(dpk), 0, _p(dm), n, _p(da), len(a)) == 1
        assert L.s2k_engine_sync(engine._h) == 1
        assert int(r[0].item()) == exp == (0 if mutate else 1)
        # the same with the randomizer hash's chain walked on the host (s2k_halfagg_chain_states) and only its states uploaded
        states = np.zeros(((3 * n) >> 1, 8), np.uint32)
        assert L.s2k_halfagg_chain_states(states.ctypes.data, pks.ctypes.data, 0, msgs.ctypes.data, n, a) == 1
        dst = torch.tensor(states.view(np.int32)).cuda(); r2 = torch.full((4,), 9, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        assert L.secp256k1_schnorrsig_aggverify_dev_chain(engine._h, None, _p(r2), _p(dpk), 0, _p(dm), n, _p(da), len(a), _p(dst)) == 1
        assert L.s2k_engine_sync(engine._h) == 1 and int(r2[0].item()) == exp
        if not mutate:                                  # states of ANOTHER aggregate: the randomizers are wrong, the equation fails
            dst2 = dst.clone(); dst2[3, 2] ^= 1; torch.cuda.synchronize()
            assert L.secp256k1_schnorrsig_aggverify_dev_chain(engine._h, None, _p(r2), _p(dpk), 0, _p(dm), n, _p(da), len(a), _p(dst2)) == 1
            assert L.s2k_engine_sync(engine._h) == 1 and int(r2[0].item()) == 0
    r = torch.full((4,), 9, dtype=torch.int32, device="cuda")
    assert L.secp256k1_schnorrsig_aggverify_dev(engine._h, None, _p(r), _p(dpk), 0, _p(dm), n, _p(da), len(a) - 1) == 1      # wrong length: verdict 0
    assert L.s2k_engine_sync(engine._h) == 1 and int(r[0].item()) == 0
    # tallies
    tallies = [ref.make_balanced_tally(rng, 2, 3), ref.make_balanced_tally(rng, 9, 1), ref.make_balanced_tally(rng, 1, 1)]
    a_, b_ = ref.make_balanced_tally(rng, 2, 2); tallies.append((a_, b_[:1]))
    exp = engine.pedersen_verify_tally_batch(tallies)
    parts, off, npos = [], [0], []
    for pos, neg in tallies:
        parts += [pos, neg]; npos.append(pos.shape[0]); off.append(off[-1] + pos.shape[0] + neg.shape[0])
    dcm = _d(np.concatenate(parts)); off = np.array(off, np.uint64); npos = np.array(npos, np.uint64)
    r = torch.full((len(tallies),), 5, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert L.secp256k1_pedersen_verify_tally_batch_dev(engine._h, None, _p(r), _p(dcm), off.ctypes.data_as(ctypes.c_void_p), npos.ctypes.data_as(ctypes.c_void_p), len(tallies)) == 1
    assert L.s2k_engine_sync(engine._h) == 1
    assert np.array_equal(r.cpu().numpy(), exp) and list(exp) == [1, 1, 1, 0]
    # rewind
    commits, plist, gens, values, blinds, nonces, msgs_ = ref.make_rangeproofs_msg(5, rng, msg_len=40, min_bits=32)
    nonces2 = nonces.copy(); nonces2[3, 0] ^= 1
    eh = engine.rangeproof_rewind_batch(commits, plist, gens, nonces2, msg_capacity=128)
    data, poff = engine.pack(plist)
    m = len(plist)
    res = torch.zeros(m, dtype=torch.int32, device="cuda"); bl = torch.zeros(m * 32, dtype=torch.uint8, device="cuda"); val = torch.zeros(m, dtype=torch.int64, device="cuda")
    msg = torch.zeros(m * 128, dtype=torch.uint8, device="cuda"); ol = torch.full((m,), 128, dtype=torch.int64, device="cuda")
    mn = torch.zeros(m, dtype=torch.int64, device="cuda"); mx = torch.zeros(m, dtype=torch.int64, device="cuda")
    dno, dc, dpr, dgen = _d(nonces2), _d(commits), _d(np.concatenate([data, np.zeros(64, np.uint8)])), _d(gens)
    doff = torch.tensor(poff.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    assert L.secp256k1_rangeproof_rewind_batch_dev(engine._h, None, _p(res), _p(bl), _p(val), _p(msg), _p(ol), 128, _p(dno), _p(mn), _p(mx), _p(dc), _p(dpr), _p(doff),
                                                   None, None, _p(dgen), m) == 1
    assert L.s2k_engine_sync(engine._h) == 1
    r = res.cpu().numpy()
    assert np.array_equal(r, eh[0]) and list(r) == [1, 1, 1, 0, 1]
    assert np.array_equal(bl.cpu().numpy().reshape(m, 32), eh[1]) and np.array_equal(val.cpu().numpy().view(np.uint64), eh[2])
    got_msgs = [msg.cpu().numpy().reshape(m, 128)[i, :int(ol[i].item())].tobytes() if r[i] else b"" for i in range(m)]
    assert got_msgs == eh[3]


def test_rangeproof_dev_calls_in_flight(engine, ref):
    """S2K_OPT_RP_INPUTS_READY: several `_dev` calls queued back to back (first stage of call k+1 underneath the ring kernel of call
    k, two scratch sets alternating) give, call by call, what the host form gives; mixed valid / corrupted proofs."""
    import torch
    from secp256k1_zkp_amd import Engine
    rng = np.random.default_rng(77)
    batches = []
    for b in range(5):
        n = 64 + 32 * b
        commits, proofs, gens, _ = ref.make_rangeproofs(n, rng, min_bits=64 if b % 2 == 0 else 16, threads=8)
        proofs = [bytearray(p) for p in proofs]
        for i in range(0, n, 7 + b):
            proofs[i][len(proofs[i]) // 2 + b] ^= 0x10
        proofs = [bytes(p) for p in proofs]
        want, wmin, wmax = engine.rangeproof_verify_batch(commits, proofs, gens)
        assert 0 < int(want.sum()) < n
        pdata, poff = Engine.pack(proofs)
        d = dict(n=n, want=want, wmin=wmin, wmax=wmax, commits=_d(commits), gens=_d(np.ascontiguousarray(gens)),
                 proofs=_d(np.concatenate([pdata, np.zeros(64, np.uint8)])), off=torch.tensor(poff.astype(np.int64)).cuda(),
                 res=torch.full((n,), 7, dtype=torch.int32, device="cuda"), mn=torch.zeros(n, dtype=torch.int64, device="cuda"),
                 mx=torch.zeros(n, dtype=torch.int64, device="cuda"))
        batches.append(d)
    torch.cuda.synchronize()
    engine.set_option(Engine.OPT_RP_INPUTS_READY, 1)
    try:
        for rep in range(3):
            for d in batches:
                engine.rangeproof_verify_batch_dev(d["res"], d["mn"], d["mx"], d["commits"], d["proofs"], d["off"], d["gens"], d["n"])
        engine.sync()
    finally:
        engine.set_option(Engine.OPT_RP_INPUTS_READY, 0)
    for d in batches:
        got = d["res"].cpu().numpy()
        assert (got == d["want"]).all()
        ok = d["want"] == 1
        assert (d["mn"].cpu().numpy().astype(np.uint64)[ok] == np.asarray(d["wmin"], dtype=np.uint64)[ok]).all()
        assert (d["mx"].cpu().numpy().astype(np.uint64)[ok] == np.asarray(d["wmax"], dtype=np.uint64)[ok]).all()


def test_dev_calls_on_two_streams_share_the_scratch_safely(engine, ref):
    """The workspace and the table arena are per engine: `_dev` calls issued alternately on two caller streams must be ordered by the
    engine itself (stream_guard in csrc/engine_internal.h), so that each result equals the reference's whatever the interleaving."""
    import torch
    from tests.refapi import G_XY
    rng = np.random.default_rng(5)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    jobs = []
    for k in range(6):
        n = 3000 + 517 * k
        a = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        base, _ = ref.ecmult_batch(np.tile(np.frombuffer(G_XY, np.uint8), (n, 1)), a)
        na = rng.integers(0, 256, (n, 32), dtype=np.uint8); ng = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        want, winf = ref.ecmult_batch(base, na, ng=ng)
        jobs.append(dict(n=n, a=_d(base), na=_d(na), ng=_d(ng), want=want, winf=winf,
                         r=torch.zeros(64 * n, dtype=torch.uint8, device="cuda"), inf=torch.full((n,), 9, dtype=torch.int32, device="cuda")))
    torch.cuda.synchronize()
    for rep in range(2):
        for k, j in enumerate(jobs):
            st = (s1, s2)[k & 1]
            engine.ecmult_batch_dev(j["r"], j["inf"], j["a"], j["na"], ng=j["ng"], stream=ctypes.c_void_p(st.cuda_stream))
    torch.cuda.synchronize()
    for j in jobs:
        assert (j["inf"].cpu().numpy() == np.asarray(j["winf"]).reshape(-1)).all()
        assert (j["r"].cpu().numpy().reshape(-1, 64) == np.asarray(j["want"]).reshape(-1, 64)).all()


def test_rewind_dev_behind_a_queued_msm_with_inputs_ready(engine, ref):
    """S2K_OPT_RP_INPUTS_READY promises that the caller's INPUT arrays are complete, nothing about the engine's own scratch: the rewind `_dev`
    form replays the prover's random stream on a side stream into the shared workspace, which an MSM queued just before on the same stream is
    still using.  The engine orders the two (the rewind form always waits for the caller's stream); both results must be right, repeatedly."""
    import torch
    from secp256k1_zkp_amd import Engine
    from tests.refapi import G_XY
    L = engine._lib
    rng = np.random.default_rng(88)
    n_msm = 1 << 16
    k = rng.integers(0, 256, (n_msm, 32), dtype=np.uint8)
    g = np.frombuffer(G_XY * n_msm, np.uint8).reshape(n_msm, 64)
    pts, _ = engine.ecmult_batch(g, np.zeros((n_msm, 32), np.uint8), k)
    sc = rng.integers(0, 256, (n_msm, 32), dtype=np.uint8)
    want_xy, want_inf = engine.ecmult_multi(sc, pts)
    commits, plist, gens, values, blinds, nonces, msgs_ = ref.make_rangeproofs_msg(48, rng, msg_len=24, min_bits=64)
    nonces[5, 1] ^= 2
    eh = engine.rangeproof_rewind_batch(commits, plist, gens, nonces, msg_capacity=64)
    data, poff = engine.pack(plist)
    m = len(plist)
    d_sc, d_pt = _d(sc), _d(pts)
    r_xy = torch.zeros(64, dtype=torch.uint8, device="cuda"); r_inf = torch.zeros(1, dtype=torch.int32, device="cuda")
    res = torch.zeros(m, dtype=torch.int32, device="cuda"); bl = torch.zeros(m * 32, dtype=torch.uint8, device="cuda"); val = torch.zeros(m, dtype=torch.int64, device="cuda")
    msg = torch.zeros(m * 64, dtype=torch.uint8, device="cuda"); ol = torch.full((m,), 64, dtype=torch.int64, device="cuda")
    mn = torch.zeros(m, dtype=torch.int64, device="cuda"); mx = torch.zeros(m, dtype=torch.int64, device="cuda")
    dno, dc, dpr, dgen = _d(nonces), _d(commits), _d(np.concatenate([data, np.zeros(64, np.uint8)])), _d(gens)
    doff = torch.tensor(poff.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    engine.set_option(Engine.OPT_RP_INPUTS_READY, 1)
    try:
        for rep in range(4):
            r_xy.zero_(); res.zero_(); ol.fill_(64)
            torch.cuda.synchronize()
            engine.ecmult_multi_dev(r_xy, r_inf, d_sc, d_pt)
            assert L.secp256k1_rangeproof_rewind_batch_dev(engine._h, None, _p(res), _p(bl), _p(val), _p(msg), _p(ol), 64, _p(dno), _p(mn), _p(mx), _p(dc), _p(dpr),
                                                           _p(doff), None, None, _p(dgen), m) == 1
            engine.ecmult_multi_dev(r_xy, r_inf, d_sc, d_pt)
            engine.sync()
            assert bytes(r_xy.cpu().numpy()) == bytes(want_xy) and int(r_inf.item()) == int(want_inf)
            r = res.cpu().numpy()
            assert np.array_equal(r, eh[0]) and r.sum() == m - 1
            assert np.array_equal(bl.cpu().numpy().reshape(m, 32), eh[1]) and np.array_equal(val.cpu().numpy().view(np.uint64), eh[2])
    finally:
        engine.set_option(Engine.OPT_RP_INPUTS_READY, 0)
