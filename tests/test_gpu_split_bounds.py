"""scalars_near_split_bounds (reference src/tests.c:4718-4739, run_ecmult_near_split_bound :4783-4793): the 20 scalars that drive
secp256k1_scalar_split_lambda to its largest outputs, through every place where this engine splits a scalar with its OWN (odd-halves)
lattice split: s2k_ecmult_batch (as na on random points and on +-G, and as ng), s2k_ecmult_multi (as term scalars), the two-piece form
(ecmult_lane_split) and the ring form (ecmult_ring_step) as the multiplier of the variable point -- all against the reference."""
import json
import os

import numpy as np
import pytest

from tests.refapi import G_XY, N, P

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _b(v):
    return int(v).to_bytes(32, "big")


def _scalars():
    vals = [int(x, 16) for x in json.load(open(os.path.join(HERE, "golden", "split_bounds.json")))["scalars"]]
    assert len(vals) == 20
    # the reference's test also exercises the negations implicitly (n1 + n2 = -target); add them and their neighbours explicitly
    return vals + [(N - v) % N for v in vals] + [(v + 1) % N for v in vals[::4]] + [(v - 1) % N for v in vals[::4]]


def test_split_bounds_ecmult_batch(engine, ref):
    rng = np.random.default_rng(61)
    sc = _scalars()
    g = np.frombuffer(G_XY, np.uint8)
    ng_neg = g.copy(); ng_neg[32:] = np.frombuffer(_b(P - int.from_bytes(G_XY[32:], "big")), np.uint8)
    pts = [np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(6)] + [g, ng_neg]
    A, NA, NG = [], [], []
    for k in sc:
        for p in pts:
            A.append(p); NA.append(_b(k)); NG.append(bytes(rng.integers(0, 256, 32, dtype=np.uint8)))      # as the multiplier of a point
            A.append(p); NA.append(bytes(rng.integers(0, 256, 32, dtype=np.uint8))); NG.append(_b(k))      # as the multiplier of G
        A.append(pts[0]); NA.append(_b(k)); NG.append(_b(0))
        A.append(pts[1]); NA.append(_b(k)); NG.append(_b(k))
    a = np.stack(A); na = np.stack([np.frombuffer(x, np.uint8) for x in NA]); ng = np.stack([np.frombuffer(x, np.uint8) for x in NG])
    r_ref, i_ref = ref.ecmult_batch(a, na, ng)
    r, i = engine.ecmult_batch(a, na, ng)
    assert np.array_equal(i, i_ref) and np.array_equal(r, r_ref)
    # the reference's own identity: n1*P + n2*P + target*P = infinity for n1 + n2 = -target
    for k in sc[:20]:
        n1 = int.from_bytes(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), "big") % N
        n2 = (-(n1 + k)) % N
        pt = np.tile(pts[2], (3, 1)); s3 = np.stack([np.frombuffer(_b(x), np.uint8) for x in (n1, n2, k)])
        xy, inf = engine.ecmult_multi(s3, pt)
        assert int(inf) == 1


def test_split_bounds_ecmult_multi(engine, ref):
    rng = np.random.default_rng(62)
    sc = _scalars()
    n = len(sc)
    pts = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(n)])
    s = np.stack([np.frombuffer(_b(k), np.uint8) for k in sc])
    for g_sc in (None, _b(sc[3]), _b(sc[17])):
        want, winf = ref.ecmult_multi(s, pts, g_sc=g_sc)
        got, ginf = engine.ecmult_multi(s, pts, g_sc=g_sc)
        assert int(ginf) == int(winf) and bytes(got) == want.tobytes()
    # the same scalars on ONE point and on +-G (sums collapse: (sum k) * P)
    one = np.tile(pts[0], (n, 1))
    want, winf = ref.ecmult_multi(s, one)
    got, ginf = engine.ecmult_multi(s, one)
    assert int(ginf) == int(winf) and bytes(got) == want.tobytes()
    # a bucket-sized sum (the bucket path, not the small-n direct one): every constant many times over random points
    m = 4096
    big_pts = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(64)])[rng.integers(0, 64, m)]
    big_s = s[rng.integers(0, n, m)]
    want, winf = ref.ecmult_multi(big_s, big_pts)
    got, ginf = engine.ecmult_multi(big_s, big_pts)
    assert int(ginf) == int(winf) and bytes(got) == want.tobytes()


def test_split_bounds_two_piece_and_ring_forms(engine, ref):
    """as the multiplier e of the variable point in ecmult_lane_split (prim 38) and in the ring form (prim 40: tables built once, 13
    signed odd 5-bit digits per 65-bit piece and no fixed top digit), one class of scalar per wavefront"""
    import ctypes
    import torch
    from tests.test_cpu_oracle import SPLIT_EDGE_SCALARS
    lib = ctypes.CDLL(os.path.join(HERE, "gpu_prims", "libs2k_gpuprims.so"))
    gsz = ctypes.c_size_t(0)
    gtab = engine._lib.s2k_engine_gtable(engine._h, ctypes.byref(gsz))

    def run(op, n, a, b, c, scratch_words):
        dev = lambda x: None if x is None else torch.tensor(np.ascontiguousarray(x, np.uint8).reshape(-1)).cuda()
        ta, tb, tc = dev(a), dev(b), dev(c)
        out = torch.zeros(n * 64, dtype=torch.uint8, device="cuda"); flag = torch.zeros(n + scratch_words, dtype=torch.int32, device="cuda")
        ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        torch.cuda.synchronize()
        assert lib.s2k_test_prim(op, ptr(out), ptr(flag), ptr(ta), ptr(tb), ptr(tc), ctypes.c_void_p(gtab), n) == 1
        return out.cpu().numpy().reshape(n, 64), flag.cpu().numpy()[:n]

    rng = np.random.default_rng(63)
    classes = _scalars() + [v for v in SPLIT_EDGE_SCALARS if v != 0] + [None] * 4
    n = 64 * len(classes)
    base = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    A, _ = ref.ecmult_batch(np.tile(np.frombuffer(G_XY, np.uint8), (n, 1)), base)
    A[::64] = np.frombuffer(G_XY, np.uint8)                      # one lane per wavefront multiplies G itself
    e = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for k, v in enumerate(classes):
        if v is not None:
            e[64 * k:64 * (k + 1)] = np.frombuffer(_b(v), np.uint8)
    s = rng.integers(0, 256, (n, 32), dtype=np.uint8); f = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    z = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    # two-piece form: e*A + s*G
    want, winf = ref.ecmult_batch(A, e, ng=s)
    got, flag = run(38, n, A, np.concatenate([e, s], axis=1), z, n * 544 + 64)
    ok = winf == 0
    assert ((flag & 1) == winf).all() and (got[ok] == want[ok]).all()
    # ring form: e*A + s*G + f*G  ==  e*A + (s + f)*G
    sf = np.stack([np.frombuffer(_b((int.from_bytes(s[i].tobytes(), "big") + int.from_bytes(f[i].tobytes(), "big")) % N), np.uint8) for i in range(n)])
    want, winf = ref.ecmult_batch(A, e, ng=sf)
    rtab_words, raw_wave_words = 528, 2 * 16 * 27 * 64          # S2K_RTAB_WORDS, S2K_RRAW_WAVE_WORDS (csrc/ecmult.h)
    got, flag = run(40, n, A, np.concatenate([e, s, f], axis=1), None, n * rtab_words + (n // 64) * raw_wave_words + n * 544 + 64)      # + S2K_PTAB_WORDS per lane: the fallback's table
    done = (flag >> 1) == 1
    assert done.reshape(-1, 64).all(axis=1).sum() >= len(classes) - 2      # (a wavefront may meet an exceptional addition and hand back: not expected here)
    chk = done & (winf == 0)
    assert (got[chk] == want[chk]).all()


def test_signed_fixed_base_digit_edges(engine, ref):
    """the signed 26-bit fixed-base digits (csrc/ecmult.h) on the device: scalars whose recoded windows sit on every edge (digit -2^25 -- the
    entry stored in the next window's unused slot --, -1, 0, +1, 2^25 - 1, the 257th bit of s + K), as ng of the double multiplication with and
    without a variable point, as the generator term of the MSM, and through BIP-340's s*G via crafted-but-invalid signatures"""
    from tests.test_cpu_oracle import fixed_base_edge_scalars
    rng = np.random.default_rng(64)
    D = int(engine._lib.s2k_engine_gtable_bits(engine._h))
    sc = fixed_base_edge_scalars(D, rng, 512)
    n = len(sc)
    ng = np.stack([np.frombuffer(_b(v), np.uint8) for v in sc])
    A, _ = ref.ecmult_batch(np.tile(np.frombuffer(G_XY, np.uint8), (n, 1)), rng.integers(0, 256, (n, 32), dtype=np.uint8))
    for na in (np.zeros((n, 32), np.uint8), rng.integers(0, 256, (n, 32), dtype=np.uint8)):
        want, winf = ref.ecmult_batch(A, na, ng=ng)
        got, ginf = engine.ecmult_batch(A, na, ng=ng)
        assert np.array_equal(ginf, winf) and np.array_equal(got[winf == 0], want[winf == 0])
    for i in range(0, n, 37):
        pts = A[:40]; scs = rng.integers(0, 256, (40, 32), dtype=np.uint8)
        want, winf = ref.ecmult_multi(scs, pts, ng[i].tobytes())
        got, ginf = engine.ecmult_multi(scs, pts, ng[i].tobytes())
        assert ginf == winf and np.array_equal(got, want), i
