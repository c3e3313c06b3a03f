"""GPU parity of the shared-generator form of the rings kernel (rangeproof.h: rp_ring_shared, the engine's generator-table cache):
whatever the cache holds, accept/reject, min and max equal the reference's secp256k1_rangeproof_verify -- with one shared generator,
with a different generator per proof, with mixes that cross the cache's capacity, across evictions, and for generators that get their
table automatically (host-buffer calls: counted before the launch; `_dev` calls: through the device mailbox)."""
import numpy as np
import pytest

from tests.refapi import GENERATOR_H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng2():
    """an engine of its own: these tests change cache options"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu needs a GPU")
    from secp256k1_zkp_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def _gen(ref, rng):
    return np.frombuffer(ref.rand_point(rng), np.uint8).copy()


def _batch(ref, rng, gens_list, per=3, min_bits=(64, 12, 52, 1)):
    """`per` proofs for every (generator, shape) pair, plus mutated copies"""
    C, P, G = [], [], []
    for g in gens_list:
        for mb in min_bits:
            gg = np.tile(g, (per, 1))
            c, p, g2, _ = ref.make_rangeproofs(per, rng, min_bits=mb, gens64=gg)
            C.append(c); P += p; G.append(g2)
    C = np.concatenate(C); G = np.concatenate(G)
    n = len(P)
    mc, mp, mg = [C], list(P), [G]
    for i in range(0, n, 2):
        q = bytearray(P[i]); q[int(rng.integers(0, len(q)))] ^= 1 << int(rng.integers(0, 8))
        mp.append(bytes(q)); mc.append(C[i:i + 1]); mg.append(G[i:i + 1])
        mp.append(P[i]); mc.append(C[i:i + 1]); mg.append(G[(i + per * len(min_bits)) % n][None])      # right proof, another generator
    return np.concatenate(mc), mp, np.concatenate(mg)


def _same(engine, ref, C, P, G):
    e_res, e_mn, e_mx = ref.rangeproof_verify_many(C, P, G, threads=8)
    res, mn, mx = engine.rangeproof_verify_batch(C, P, G)
    assert np.array_equal(res, e_res) and np.array_equal(mn, e_mn) and np.array_equal(mx, e_mx)
    return e_res


def test_cache_on_off_and_capacity(eng2, ref):
    from secp256k1_zkp_amd import Engine
    rng = np.random.default_rng(901)
    H = np.frombuffer(GENERATOR_H, np.uint8)
    g1, g2, g3 = _gen(ref, rng), _gen(ref, rng), _gen(ref, rng)
    C, P, G = _batch(ref, rng, [H, g1, g2, g3])
    # per-item random generators on top (the Elements case: every output its own blinded asset generator)
    gd = np.stack([_gen(ref, rng) for _ in range(8)])
    c, p, gd, _ = ref.make_rangeproofs(8, rng, min_bits=64, gens64=gd)
    C = np.concatenate([C, c]); P = P + p; G = np.concatenate([G, gd])
    eng2.set_option(Engine.OPT_GEN_CACHE_MIN, 1 << 30)           # nothing gets a table unless asked for
    eng2.set_option(Engine.OPT_GEN_CACHE_SLOTS, 0)
    e_res = _same(eng2, ref, C, P, G)                              # general form only
    assert 0 < e_res.sum() < len(P) and not eng2.generator_cached(GENERATOR_H)
    eng2.set_option(Engine.OPT_GEN_CACHE_SLOTS, 2)
    _same(eng2, ref, C, P, G)                                      # H only (built at this call)
    assert eng2.generator_cached(GENERATOR_H)
    eng2.cache_generator(g1)
    _same(eng2, ref, C, P, G)                                      # H + g1
    eng2.cache_generator(g2)                                       # capacity 2: the least recently used table (H or g1) makes room
    assert eng2.generator_cached(g2) and (eng2.generator_cached(GENERATOR_H) + eng2.generator_cached(g1)) == 1
    _same(eng2, ref, C, P, G)
    eng2.cache_generator(g3); eng2.cache_generator(g1)
    assert eng2.generator_cached(g3) and eng2.generator_cached(g1) and not eng2.generator_cached(g2)
    _same(eng2, ref, C, P, G)
    # back and forth between two batches that each want "their" generator
    for gx in (g2, g3, g2):
        eng2.cache_generator(gx)
        sel = [i for i in range(len(P)) if G[i].tobytes() == gx.tobytes()]
        _same(eng2, ref, C[sel], [P[i] for i in sel], G[sel])


def test_automatic_tables(eng2, ref):
    """a generator that keeps coming ON VALID PROOFS gets a table by itself: the final kernel reports, the next call counts and builds"""
    import torch
    from secp256k1_zkp_amd import Engine
    rng = np.random.default_rng(902)
    eng2.set_option(Engine.OPT_GEN_CACHE_SLOTS, 3)
    eng2.set_option(Engine.OPT_GEN_CACHE_MIN, 10)
    ga, gb = _gen(ref, rng), _gen(ref, rng)
    c, p, g, _ = ref.make_rangeproofs(6, rng, min_bits=20, gens64=np.tile(ga, (6, 1)))
    _same(eng2, ref, c, p, g)
    assert not eng2.generator_cached(ga)
    _same(eng2, ref, c, p, g)
    assert not eng2.generator_cached(ga)                           # the second call has read the first one's report: 6 < 10
    _same(eng2, ref, c, p, g)
    assert eng2.generator_cached(ga)                               # 12 valid proofs seen: built at the start of the third call
    # `_dev`: the header kernel reports generators without a table; the host reads that report at the next call
    c, p, g, _ = ref.make_rangeproofs(12, rng, min_bits=20, gens64=np.tile(gb, (12, 1)))
    q = bytearray(p[3]); q[40] ^= 4; p[3] = bytes(q)
    e_res, e_mn, e_mx = ref.rangeproof_verify_many(c, p, g)
    dev = torch.device("cuda", 0)
    pdata, poff = Engine.pack(p)
    d_c = torch.tensor(c).to(dev); d_g = torch.tensor(np.ascontiguousarray(g)).to(dev)
    d_p = torch.tensor(np.concatenate([pdata, np.zeros(64, np.uint8)])).to(dev); d_off = torch.tensor(poff.astype(np.int64)).to(dev)
    d_res = torch.zeros(12, dtype=torch.int32, device=dev); d_mn = torch.zeros(12, dtype=torch.int64, device=dev); d_mx = torch.zeros(12, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for k in range(3):
        eng2.rangeproof_verify_batch_dev(d_res, d_mn, d_mx, d_c, d_p, d_off, d_g, 12)
        eng2.sync()
        assert np.array_equal(d_res.cpu().numpy(), e_res) and np.array_equal(d_mx.cpu().numpy().view(np.uint64), e_mx)
        assert eng2.generator_cached(gb) == (k >= 1)               # reported by call 0, built at the start of call 1
    eng2.set_option(Engine.OPT_GEN_CACHE_MIN, 1 << 16)


def test_junk_proofs_cannot_buy_a_table(eng2, ref):
    """only proofs that VERIFIED count towards an automatic table, and an automatic table never takes the slot of secp256k1_generator_h or
    of a generator the application cached itself: junk proofs that merely name a generator (not even a curve point) cost the engine
    nothing, and valid proofs over ever new generators cannot push H out"""
    from secp256k1_zkp_amd import Engine
    rng = np.random.default_rng(903)
    eng2.set_option(Engine.OPT_GEN_CACHE_SLOTS, 2)
    eng2.set_option(Engine.OPT_GEN_CACHE_MIN, 4)
    H = np.frombuffer(GENERATOR_H, np.uint8)
    eng2.cache_generator(GENERATOR_H)
    c, p, g, _ = ref.make_rangeproofs(4, rng, min_bits=8)
    _same(eng2, ref, c, p, g)
    assert eng2.generator_cached(GENERATOR_H)
    junk_gen = rng.integers(0, 256, 64, dtype=np.uint8)
    jc = np.tile(c, (16, 1))                                         # (commitments must parse: the reference's API only ever sees parsed objects)
    jp = [bytes([0x40, 1]) + bytes(rng.integers(0, 256, 160, dtype=np.uint8)) for _ in range(64)]      # parses as a 1-ring proof, cannot verify
    for _ in range(3):
        res = _same(eng2, ref, jc, jp, np.tile(junk_gen, (64, 1)))
        assert not res.any()
    assert not eng2.generator_cached(junk_gen) and eng2.generator_cached(GENERATOR_H)
    ga = _gen(ref, rng)
    eng2.cache_generator(ga)                                        # slots: H or ga (explicit requests may evict anything)
    eng2.cache_generator(GENERATOR_H)
    assert eng2.generator_cached(ga) and eng2.generator_cached(GENERATOR_H)
    for t in range(3):                                              # valid proofs over new generators: both slots are pinned, nothing is built
        gx = _gen(ref, rng)
        c, p, g, _ = ref.make_rangeproofs(6, rng, min_bits=8, gens64=np.tile(gx, (6, 1)))
        for _ in range(2):
            assert _same(eng2, ref, c, p, g).all()
        assert not eng2.generator_cached(gx)
    assert eng2.generator_cached(ga) and eng2.generator_cached(GENERATOR_H)
    eng2.set_option(Engine.OPT_GEN_CACHE_MIN, 1 << 16)


@pytest.mark.parametrize("kind", ["shared", "distinct", "mixed"])
def test_full_size_by_generator_kind(engine, ref, kind):
    """2^14 proofs: (shared) one generator for all, (distinct) every proof its own, (mixed) five generators for most proofs -- more
    than the cache holds -- and own generators for the rest; 52-bit proofs (26 rings) and an extra_commit among them"""
    rng = np.random.default_rng({"shared": 11, "distinct": 12, "mixed": 13}[kind])
    n = 1 << 14
    if kind == "shared":
        gens = np.tile(_gen(ref, rng), (n, 1))
        engine.cache_generator(gens[0])
    elif kind == "distinct":
        gens = np.stack([_gen(ref, rng) for _ in range(n)])
    else:
        pool = np.stack([_gen(ref, rng) for _ in range(5)] + [np.frombuffer(GENERATOR_H, np.uint8)])
        gens = pool[rng.integers(0, 6, n)]
        own = rng.random(n) < 0.2
        gens[own] = np.stack([_gen(ref, rng) for _ in range(int(own.sum()))])
        engine.cache_generator(pool[0]); engine.cache_generator(pool[1])
    h = n // 2
    # first half 64-bit proofs, second half 52-bit ones (26 rings: Elements' default) signed over an extra_commit (Elements: the output script)
    c1, p1, _, _ = ref.make_rangeproofs(h, rng, min_bits=64, gens64=gens[:h], threads=16)
    extra = [b""] * h + [bytes(rng.integers(0, 256, int(rng.integers(1, 80)), dtype=np.uint8)) for _ in range(n - h)]
    c2, p2, _ = ref.make_rangeproofs_extra(n - h, rng, extra[h:], min_bits=52, gens64=gens[h:], threads=16)
    commits = np.concatenate([c1, c2]); proofs = p1 + p2
    for i in range(0, n, 41):
        q = bytearray(proofs[i]); q[int(rng.integers(0, len(q)))] ^= 1 << int(rng.integers(0, 8)); proofs[i] = bytes(q)
    for i in range(h + 7, n, 97):
        extra[i] = extra[i] + b"!"                                 # right proof, wrong extra_commit
    e_res, e_mn, e_mx = ref.rangeproof_verify_many_extra(commits, proofs, gens, extra, threads=16)
    res, mn, mx = engine.rangeproof_verify_batch(commits, proofs, gens, extra=extra)
    assert np.array_equal(res, e_res) and np.array_equal(mn, e_mn) and np.array_equal(mx, e_mx)
    bad = set(range(0, n, 41)) | set(range(h + 7, n, 97))
    assert e_res.sum() == n - len(bad)


def test_eviction_waits_for_its_readers_not_for_the_device(eng2, ref):
    """Rebuilding a generator slot used to be hipDeviceSynchronize() under the engine's and the pool's locks: every verifier thread on the GPU
    stalled for the longest stream in flight.  Now the build's stream waits for the events of the engines that READ that slot and nothing
    waits on the host: with a second engine's long queue of batches in flight on the very table that is evicted, the evicting call returns
    at once, the queue keeps running, its verdicts (read from the old table) are right, and the new table serves its own proofs."""
    import time
    import torch
    from secp256k1_zkp_amd import Engine
    rng = np.random.default_rng(907)
    eng_b = Engine(0)
    try:
        eng2.set_option(Engine.OPT_GEN_CACHE_MIN, 1 << 30)
        eng2.set_option(Engine.OPT_GEN_CACHE_SLOTS, 2)
        gx, gy, gz = _gen(ref, rng), _gen(ref, rng), _gen(ref, rng)
        eng2.cache_generator(gx); eng2.cache_generator(gy)             # both slots taken
        n = 2048
        gxy = np.tile(gx, (n, 1)); gxy[1::2] = gy                       # the queue reads BOTH tables: whichever slot is evicted is in use
        c, p, g, _ = ref.make_rangeproofs(n, rng, min_bits=64, gens64=gxy, threads=8)
        for i in range(0, n, 97):
            q = bytearray(p[i]); q[100 + i % 50] ^= 1; p[i] = bytes(q)
        e_res, e_mn, e_mx = ref.rangeproof_verify_many(c, p, g, threads=8)
        dev = torch.device("cuda", 0)
        pdata, poff = Engine.pack(p)
        d_c = torch.tensor(c).to(dev); d_g = torch.tensor(np.ascontiguousarray(g)).to(dev)
        d_p = torch.tensor(np.concatenate([pdata, np.zeros(64, np.uint8)])).to(dev); d_off = torch.tensor(poff.astype(np.int64)).to(dev)
        rounds = 60
        d_res = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(rounds)]
        d_mn = torch.zeros(n, dtype=torch.int64, device=dev); d_mx = torch.zeros(n, dtype=torch.int64, device=dev)
        s = torch.cuda.Stream(device=dev)
        eng_b.rangeproof_verify_batch_dev(d_res[0], d_mn, d_mx, d_c, d_p, d_off, d_g, n, stream=s.cuda_stream)      # (warm-up: workspace, tables)
        s.synchronize(); torch.cuda.synchronize()
        for k in range(rounds):
            eng_b.rangeproof_verify_batch_dev(d_res[k], d_mn, d_mx, d_c, d_p, d_off, d_g, n, stream=s.cuda_stream)
        ev = torch.cuda.Event(); ev.record(s)
        t0 = time.perf_counter()
        eng2.cache_generator(gz)                                        # evicts one of the two slots: a table eng_b's queue is reading
        dt = time.perf_counter() - t0
        still_running = not ev.query()
        s.synchronize()
        assert still_running, "the second engine's queue had drained before the eviction returned (%.1f ms)" % (dt * 1e3)
        assert dt < 0.05, "evicting a slot took %.1f ms on the host with another engine's queue in flight" % (dt * 1e3)
        for k in range(rounds):
            assert np.array_equal(d_res[k].cpu().numpy(), e_res), k
        assert eng2.generator_cached(gz) and eng2.generator_cached(gx) + eng2.generator_cached(gy) == 1
        c2, p2, g2, _ = ref.make_rangeproofs(64, rng, min_bits=64, gens64=np.tile(gz, (64, 1)), threads=8)
        _same(eng2, ref, c2, p2, g2)
        _same(eng_b, ref, c[:64], p[:64], g[:64])                       # gx and gy again, one of them now without a table (general form)
    finally:
        eng_b.close()
        eng2.set_option(Engine.OPT_GEN_CACHE_MIN, 1 << 16)
