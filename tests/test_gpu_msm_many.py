"""GPU parity: s2k_ecmult_multi_many (K independent sums in one launch chain) -- every sum bit-exact against the reference's
secp256k1_ecmult_multi_var on the same terms (src/ecmult_impl.h:823-867); and sums larger than one launch indexes, run as several
launches whose partial sums add (the reference's own batching, src/ecmult_impl.h:804-820, :856-865)."""
import hashlib

import numpy as np
import pytest

from tests.refapi import G_XY, N, P
from tests.test_gpu_msm import _points

pytestmark = pytest.mark.gpu

S2K_OPT_MSM_MAX_TERMS = 10


def _check_many(engine, ref, sc, pts, off, g=None, inf=None):
    got, ginf = engine.ecmult_multi_many(sc, pts, off, g, inf)
    k = len(off) - 1
    assert got.shape == (k, 64) and ginf.shape == (k,)
    for s in range(k):
        lo, hi = int(off[s]), int(off[s + 1])
        exp, einf = ref.ecmult_multi(sc[lo:hi], pts[lo:hi], None if g is None else bytes(g[s]), None if inf is None else inf[lo:hi])
        assert int(ginf[s]) == einf and np.array_equal(got[s], exp), (s, lo, hi)


@pytest.mark.parametrize("with_g", [False, True])
def test_many_ragged(engine, ref, with_g):
    """ragged sizes including 0, 1, 2 and both sides of the reference's algorithm switch (88 points); zero scalars, infinite points,
    a cancelling pair and a repeated point inside single sums (src/tests.c:5039-5152 shapes)."""
    rng = np.random.default_rng(77 + with_g)
    sizes = [0, 1, 2, 3, 31, 32, 33, 87, 88, 89, 200, 0, 511, 512, 1000, 1024, 5, 2048, 1, 0]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(off[-1])
    pts = _points(engine, rng, n)
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    inf = np.zeros(n, np.uint8)
    for s, sz in enumerate(sizes):
        lo = int(off[s])
        if sz >= 31:
            sc[lo + 1] = 0; inf[lo + 2] = 1
            sc[lo + 3] = np.frombuffer((N - 1).to_bytes(32, "big"), np.uint8)
            pts[lo + 5] = pts[lo + 4]; pts[lo + 5, 32:] = np.frombuffer(((P - int.from_bytes(pts[lo + 4, 32:].tobytes(), "big")) % P).to_bytes(32, "big"), np.uint8); sc[lo + 5] = sc[lo + 4]
            pts[lo + 7] = pts[lo + 6]
    g = rng.integers(0, 256, (len(sizes), 32), dtype=np.uint8) if with_g else None
    if with_g: g[3] = 0
    _check_many(engine, ref, sc, pts, off, g, inf)


def test_many_degenerate(engine, ref):
    """all-zero scalars, all-infinite points, everything cancelling, every scalar equal (one bucket takes a whole window) -- per sum."""
    rng = np.random.default_rng(5)
    m = 300
    sizes = [m, m, m, m]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    pts = _points(engine, rng, 4 * m)
    sc = rng.integers(0, 256, (4 * m, 32), dtype=np.uint8)
    inf = np.zeros(4 * m, np.uint8)
    sc[:m] = 0                                                      # sum 0: zero scalars
    inf[m:2 * m] = 1                                                # sum 1: infinite points
    h = m // 2                                                      # sum 2: second half cancels the first
    pts[2 * m + h:3 * m] = pts[2 * m:2 * m + h]
    for i in range(h):
        k = int.from_bytes(sc[2 * m + i].tobytes(), "big") % N
        sc[2 * m + h + i] = np.frombuffer(((N - k) % N).to_bytes(32, "big"), np.uint8)
    sc[3 * m:] = sc[3 * m]                                          # sum 3: equal scalars
    got, ginf = engine.ecmult_multi_many(sc, pts, off, None, inf)
    assert list(ginf[:3]) == [1, 1, 1] and not got[:3].any()
    _check_many(engine, ref, sc, pts, off, None, inf)


def test_many_config1_batched(engine, ref):
    """BASELINE config 1's inputs (src/bench_ecmult.c:262-276, :362-371), 64 sums of 1 024 terms + G, each against the reference; the device
    form gives the same bytes."""
    import torch
    n, k = 1024, 64
    rng = np.random.default_rng(11)
    seeds = np.stack([np.frombuffer(hashlib.sha256(b"ecmult" + i.to_bytes(4, "little")).digest(), np.uint8) for i in range(n)])
    g = np.frombuffer(G_XY * n, np.uint8).reshape(n, 64)
    ks = np.stack([np.frombuffer(((1 << i) % N).to_bytes(32, "big"), np.uint8) for i in range(n)])
    base, _ = engine.ecmult_batch(g, np.zeros((n, 32), np.uint8), ks)
    pts = np.tile(base, (k, 1))
    sc = np.tile(seeds, (k, 1)); sc[:, 0] ^= np.repeat(np.arange(k, dtype=np.uint8), n)          # a different scalar set per sum
    gs = rng.integers(0, 256, (k, 32), dtype=np.uint8)
    off = (np.arange(k + 1) * n).astype(np.uint64)
    got, ginf = engine.ecmult_multi_many(sc, pts, off, gs)
    for s in range(0, k, 7):
        exp, einf = ref.ecmult_multi(sc[s * n:(s + 1) * n], pts[s * n:(s + 1) * n], bytes(gs[s]), None)
        assert int(ginf[s]) == einf and np.array_equal(got[s], exp), s
    r = torch.zeros(k, 64, dtype=torch.uint8, device="cuda"); ri = torch.zeros(k, dtype=torch.int32, device="cuda")
    engine.ecmult_multi_many_dev(r, ri, torch.tensor(sc).cuda(), torch.tensor(pts).cuda(), off, torch.tensor(gs).cuda())
    engine.sync()
    assert np.array_equal(r.cpu().numpy(), got) and np.array_equal(ri.cpu().numpy(), ginf)


def test_many_large_sums_take_single_path(engine, ref):
    """a batch holding a sum above 8 192 terms: every sum goes through the single-sum path; same results."""
    rng = np.random.default_rng(9)
    sizes = [9000, 100, 0, 3000]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(off[-1])
    pts = _points(engine, rng, n); sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    g = rng.integers(0, 256, (len(sizes), 32), dtype=np.uint8)
    _check_many(engine, ref, sc, pts, off, g, None)


def test_many_argument_checks(engine):
    from secp256k1_zkp_amd import S2KError
    sc = np.zeros((4, 32), np.uint8); pts = np.zeros((4, 64), np.uint8)
    with pytest.raises(S2KError):
        engine.ecmult_multi_many(sc, pts, np.array([1, 4], np.uint64))           # does not start at 0
    with pytest.raises(S2KError):
        engine.ecmult_multi_many(sc, pts, np.array([0, 3, 2, 4], np.uint64))     # decreasing
    r, inf = engine.ecmult_multi_many(sc[:0], pts[:0], np.array([0], np.uint64)) # no sums: nothing to do
    assert r.shape == (0, 64)


@pytest.mark.parametrize("n,cap", [(1000, 64), (5000, 999), (70000, 20000), (300000, 100001)])
def test_oversized_sum_runs_as_several_launches(engine, ref, n, cap):
    """S2K_OPT_MSM_MAX_TERMS lowered: the sum is cut into ceil(n / cap) launches whose Jacobian partials are added -- same bytes as the
    reference and as the one-launch result, with and without the G term (it rides with the first slice), host and device forms."""
    import torch
    rng = np.random.default_rng(n)
    pts = _points(engine, rng, n); sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    inf = np.zeros(n, np.uint8); inf[n // 2] = 1; sc[n // 3] = 0
    for g in (None, bytes(rng.integers(0, 256, 32, dtype=np.uint8))):
        exp, einf = ref.ecmult_multi(sc, pts, g, inf)
        engine.set_option(S2K_OPT_MSM_MAX_TERMS, cap)
        try:
            got, ginf = engine.ecmult_multi(sc, pts, g, inf)
            r = torch.zeros(64, dtype=torch.uint8, device="cuda"); ri = torch.zeros(1, dtype=torch.int32, device="cuda")
            engine.ecmult_multi_dev(r, ri, torch.tensor(sc).cuda(), torch.tensor(pts).cuda(), None if g is None else torch.tensor(np.frombuffer(g, np.uint8).copy()).cuda(),
                                    torch.tensor(inf).cuda())
            engine.sync()
        finally:
            engine.set_option(S2K_OPT_MSM_MAX_TERMS, 0)
        assert ginf == einf and np.array_equal(got, exp), (n, cap, g is not None)
        assert int(ri.cpu()[0]) == einf and np.array_equal(r.cpu().numpy(), exp)


def test_sum_of_2p28_terms(engine):
    """n = 2^28 terms: more than one launch indexes with 32-bit bucket references (2 * 9 windows * n >= 2^32), so the call runs as two
    launches.  The reference would need ~15 minutes for it; the check is the size-independent identity
        sum_i s_i * (k_(i mod m) * G)  ==  (sum_j k_j * (sum_{i = j mod m} s_i)) * G
    with the inner sums taken limb-wise on the GPU (torch int64) and the right-hand side by the engine's own, separately verified,
    single double multiplication."""
    import torch
    free, _total = torch.cuda.mem_get_info()
    if free < 150 * (1 << 30):
        pytest.skip("needs ~110 GB of free HBM")
    from secp256k1_zkp_amd import Engine
    n, m = 1 << 28, 1 << 16
    rng = np.random.default_rng(228)
    big = Engine(0)                 # an engine of its own (the device's tables are shared): its ~110 GB of workspace go back when it is closed
    ks = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    g = np.frombuffer(G_XY * m, np.uint8).reshape(m, 64)
    base, binf = engine.ecmult_batch(g, np.zeros((m, 32), np.uint8), ks)
    assert not binf.any()
    dev = torch.device("cuda")
    pts = torch.tensor(base, device=dev).repeat(n // m, 1)                      # term i carries point i mod m
    gen = torch.Generator(device=dev); gen.manual_seed(228)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=gen)
    # column sums of the scalars' bytes per residue class j = i mod m: (n / m) * 255 < 2^63
    col = torch.zeros(m, 32, dtype=torch.int64, device=dev)
    step = 1 << 24
    for lo in range(0, n, step):
        col += sc[lo:lo + step].view(step // m, m, 32).to(torch.int64).sum(0)
    col = col.cpu().numpy()
    total = 0
    for j in range(m):
        sj = 0
        for b in range(32):
            sj = (sj << 8) + int(col[j, b])
        total = (total + sj * int.from_bytes(ks[j].tobytes(), "big")) % N
    r = torch.zeros(64, dtype=torch.uint8, device=dev); ri = torch.zeros(1, dtype=torch.int32, device=dev)
    try:
        big.ecmult_multi_dev(r, ri, sc, pts)
        big.sync()
    finally:
        big.close()
    got, ginf = r.cpu().numpy(), int(ri.cpu()[0])
    del sc, pts, col
    torch.cuda.empty_cache()
    exp, einf = engine.ecmult_batch(np.frombuffer(G_XY, np.uint8).reshape(1, 64), np.zeros((1, 32), np.uint8), np.frombuffer(total.to_bytes(32, "big"), np.uint8).reshape(1, 32))
    assert ginf == int(einf[0]) == 0 and np.array_equal(got, exp[0])


def test_unaligned_device_arrays(engine, ref):
    """the caller's arrays are plain byte arrays: at odd addresses the decode takes its byte-wise form, at 16-byte aligned ones the vector form --
    same results (single sum and many sums)."""
    import torch
    rng = np.random.default_rng(31)
    n = 700
    pts = _points(engine, rng, n); sc = rng.integers(0, 256, (n, 32), dtype=np.uint8); g = rng.integers(0, 256, 32, dtype=np.uint8)
    exp, einf = ref.ecmult_multi(sc, pts, bytes(g), None)
    for shift in (0, 1, 7):
        buf_s = torch.zeros(32 * n + 64, dtype=torch.uint8, device="cuda"); buf_p = torch.zeros(64 * n + 64, dtype=torch.uint8, device="cuda"); buf_g = torch.zeros(96, dtype=torch.uint8, device="cuda")
        d_s = buf_s[shift:shift + 32 * n]; d_p = buf_p[shift:shift + 64 * n]; d_g = buf_g[shift:shift + 32]
        d_s.copy_(torch.tensor(sc.reshape(-1))); d_p.copy_(torch.tensor(pts.reshape(-1))); d_g.copy_(torch.tensor(g))
        r = torch.zeros(64, dtype=torch.uint8, device="cuda"); ri = torch.zeros(1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        engine.ecmult_multi_dev(r, ri, d_s, d_p, d_g); engine.sync()
        assert int(ri.item()) == einf and np.array_equal(r.cpu().numpy(), exp), shift
        off = np.array([0, 300, 700], np.uint64)
        r2 = torch.zeros(2, 64, dtype=torch.uint8, device="cuda"); ri2 = torch.zeros(2, dtype=torch.int32, device="cuda")
        engine.ecmult_multi_many_dev(r2, ri2, d_s, d_p, off); engine.sync()
        for s_, (lo, hi) in enumerate(((0, 300), (300, 700))):
            e2, i2 = ref.ecmult_multi(sc[lo:hi], pts[lo:hi], None, None)
            assert int(ri2[s_].item()) == i2 and np.array_equal(r2[s_].cpu().numpy(), e2), (shift, s_)


def test_many_edges(engine, ref):
    """one empty sum with only its G term, a batch of empty sums, 70 000 three-term sums (the batch dimension of every launch), and the exact
    boundaries of the multi-launch split (n == cap, cap + 1, 2 cap, 2 cap + 1; with the G term riding in the first slice)."""
    from secp256k1_zkp_amd import Engine
    rng = np.random.default_rng(41)
    g1 = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    r, inf = engine.ecmult_multi_many(np.zeros((0, 32), np.uint8), np.zeros((0, 64), np.uint8), np.array([0, 0], np.uint64), g1)
    exp, einf = ref.ecmult_multi(np.zeros((0, 32), np.uint8), np.zeros((0, 64), np.uint8), bytes(g1[0]), None)
    assert int(inf[0]) == einf == 0 and np.array_equal(r[0], exp)
    r, inf = engine.ecmult_multi_many(np.zeros((0, 32), np.uint8), np.zeros((0, 64), np.uint8), np.zeros(6, np.uint64))
    assert inf.tolist() == [1] * 5 and not r.any()
    k = 70000
    pts = np.tile(_points(engine, rng, 3000), (k // 1000, 1))[:3 * k]
    sc = rng.integers(0, 256, (3 * k, 32), dtype=np.uint8)
    off = (np.arange(k + 1) * 3).astype(np.uint64)
    r, inf = engine.ecmult_multi_many(sc, pts, off)
    for s_ in (0, 1, 999, 65535, 65536, k - 1):
        exp, einf = ref.ecmult_multi(sc[3 * s_:3 * s_ + 3], pts[3 * s_:3 * s_ + 3], None, None)
        assert int(inf[s_]) == einf and np.array_equal(r[s_], exp), s_
    cap = 1000
    engine.set_option(Engine.OPT_MSM_MAX_TERMS, cap)
    try:
        for n in (cap - 1, cap, cap + 1, 2 * cap, 2 * cap + 1):
            p2 = _points(engine, rng, n); s2 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            for g in (None, bytes(rng.integers(0, 256, 32, dtype=np.uint8))):
                exp, einf = ref.ecmult_multi(s2, p2, g, None)
                got, ginf = engine.ecmult_multi(s2, p2, g, None)
                assert ginf == einf and np.array_equal(got, exp), (n, g is not None)
    finally:
        engine.set_option(Engine.OPT_MSM_MAX_TERMS, 0)


def test_many_sums_through_the_distributed_layer(engine, ref):
    """parallel.msm_many_sharded with the engine's backend (one rank: the whole range as one s2k_ecmult_multi_many_dev chain on the backend's
    stream; the two-rank protocol runs on gloo in tests/test_cpu_distributed.py): same bytes as the host-buffer call, each sum the reference's."""
    import torch
    from secp256k1_zkp_amd import parallel
    rng = np.random.default_rng(91)
    sizes = [0, 17, 1, 900, 88, 0, 1024, 3]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(off[-1]); k = len(sizes)
    pts = _points(engine, rng, n)
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8); sc[::19] = 0
    inf = np.zeros(n, np.uint8); inf[7::31] = 1
    g = rng.integers(0, 256, (k, 32), dtype=np.uint8)
    dev = torch.device("cuda", engine.device)
    be = parallel.EngineBackend(engine)
    t = lambda a: torch.tensor(np.ascontiguousarray(a)).to(dev)
    d_sc, d_pt, d_g, d_inf = t(sc), t(pts), t(g), t(inf)
    torch.cuda.synchronize()
    xy, fl = parallel.msm_many_sharded(be, d_sc, d_pt, off, d_g, d_inf)
    be.stream.synchronize()
    want_xy, want_fl = engine.ecmult_multi_many(sc, pts, off, g, inf)
    assert np.array_equal(xy.cpu().numpy(), want_xy) and np.array_equal(fl.cpu().numpy(), want_fl)
    for s in (1, 3, 6):
        lo, hi = int(off[s]), int(off[s + 1])
        exp, einf = ref.ecmult_multi(sc[lo:hi], pts[lo:hi], bytes(g[s]), inf[lo:hi])
        assert int(want_fl[s]) == einf and np.array_equal(want_xy[s], exp)
