"""GPU parity of the per-lane device primitives (fe / scalar / group / sha256 / generator table) against the reference,
byte-exact.  Mirrors the reference's primitive unit tests (src/tests.c:3375-3437 fe_mul/sqr vs independent mulmod,
:3468-3582 sqrt, :3584 inverse, :3926-4213 test_ge, :6018-6060 endomorphism split)."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

from tests.refapi import G_XY, N, P

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def prims(engine):
    import torch
    lib = ctypes.CDLL(os.path.join(HERE, "gpu_prims", "libs2k_gpuprims.so"))
    gsz = ctypes.c_size_t(0)
    gtab = engine._lib.s2k_engine_gtable(engine._h, ctypes.byref(gsz))

    def run(op, n, out_bytes, a=None, b=None, c=None, scratch_words=0):
        dev = lambda x: None if x is None else torch.tensor(np.ascontiguousarray(x, np.uint8).reshape(-1)).cuda()
        ta, tb, tc = dev(a), dev(b), dev(c)
        out = torch.zeros(n * out_bytes, dtype=torch.uint8, device="cuda"); flag = torch.zeros(n + scratch_words, dtype=torch.int32, device="cuda")
        ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        torch.cuda.synchronize()
        ok = lib.s2k_test_prim(op, ptr(out), ptr(flag), ptr(ta), ptr(tb), ptr(tc), ctypes.c_void_p(gtab), n)
        assert ok == 1
        return out.cpu().numpy().reshape(n, out_bytes), flag.cpu().numpy()[:n]
    return run


def _b(v):
    return int(v).to_bytes(32, "big")


def _fe_inputs(rng, n):
    edge = [0, 1, 2, P - 1, P - 2, P, P + 1, 2**256 - 1, 2**255, 977, 2**32 + 977, N, (P + 1) // 2, 2**256 - 2**32, 2**232, 2**232 - 1]
    rows = [np.frombuffer(_b(e % 2**256), np.uint8) for e in edge]
    arr = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for i, r in enumerate(rows):
        arr[i] = r
    return arr


def _ref_rows(ref, name, nout, *cols):
    n = cols[0].shape[0]
    out = np.zeros((n, nout), np.uint8); flags = np.zeros(n, np.int64)
    for i in range(n):
        r, o = ref.call(name, [nout], *[c[i].tobytes() for c in cols])
        out[i] = np.frombuffer(o[0], np.uint8); flags[i] = r if r is not None else 0
    return out, flags


def test_field_ops(prims, ref):
    rng = np.random.default_rng(11)
    n = 512
    a = _fe_inputs(rng, n); b = np.roll(_fe_inputs(rng, n), 5, axis=0)
    for op, name, cols in ((0, "ref_fe_mul", (a, b)), (1, "ref_fe_sqr", (a,)), (2, "ref_fe_inv", (a,))):
        got, _ = prims(op, n, 32, *cols)
        exp, _ = _ref_rows(ref, name, 32, *cols)
        assert np.array_equal(got, exp), name
    got, fl = prims(3, n, 32, a)
    exp, efl = _ref_rows(ref, "ref_fe_sqrt", 32, a)
    assert np.array_equal(got, exp) and np.array_equal(fl, efl)
    # half: 2*half(a) == a
    got, _ = prims(4, n, 32, a)
    for i in range(n):
        assert (2 * int.from_bytes(got[i].tobytes(), "big")) % P == int.from_bytes(a[i].tobytes(), "big") % P


def test_scalar_ops(prims, ref):
    rng = np.random.default_rng(12)
    n = 256
    a = _fe_inputs(rng, n); b = np.roll(_fe_inputs(rng, n), 3, axis=0)
    got, _ = prims(8, n, 32, a, b); exp, _ = _ref_rows(ref, "ref_scalar_mul", 32, a, b)
    assert np.array_equal(got, exp)
    got, _ = prims(9, n, 64, a)
    for i in range(n):
        r, o = ref.call("ref_scalar_split_lambda", [32, 32], a[i].tobytes())
        assert got[i].tobytes() == o[0] + o[1]
    got, fl = prims(12, n, 32, a)
    for i in range(n):
        v = int.from_bytes(a[i].tobytes(), "big")
        assert fl[i] == (1 if v >= N else 0)
        assert int.from_bytes(got[i].tobytes(), "big") == (-v) % N
    got, _ = prims(13, 64, 32, a[:64]); exp, _ = _ref_rows(ref, "ref_scalar_inverse", 32, a[:64])
    assert np.array_equal(got, exp)


def test_group_ops(prims, ref):
    rng = np.random.default_rng(13)
    pts = [ref.rand_point(rng) for _ in range(30)] + [G_XY]
    neg = lambda p: p[:32] + _b((P - int.from_bytes(p[32:], "big")) % P)
    A, B = [], []
    for i, p in enumerate(pts):
        for q in (pts[(i + 1) % len(pts)], p, neg(p)):
            A.append(p); B.append(q)
    n = len(A)
    a = np.frombuffer(b"".join(A), np.uint8).reshape(n, 64); b = np.frombuffer(b"".join(B), np.uint8).reshape(n, 64)
    exp = np.zeros((n, 64), np.uint8); einf = np.zeros(n, np.int32)
    for i in range(n):
        r, o = ref.call("ref_ge_add", [64], A[i], 0, B[i], 0)
        exp[i] = np.frombuffer(o[0], np.uint8); einf[i] = r
    got, inf = prims(5, n, 64, a, b)
    got[inf != 0] = 0
    assert np.array_equal(inf, einf) and np.array_equal(got, exp)
    z = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    got, inf = prims(7, n, 64, a, b, z)
    got[inf != 0] = 0
    assert np.array_equal(inf, einf) and np.array_equal(got, exp)
    got, inf = prims(6, n, 64, a)
    for i in range(n):
        r, o = ref.call("ref_ge_double", [64], A[i], 0)
        assert got[i].tobytes() == o[0] and inf[i] == r


def test_sha256(prims):
    rng = np.random.default_rng(14)
    n = 64
    m = rng.integers(0, 256, (n, 100), dtype=np.uint8)
    got, _ = prims(10, n, 32, m)
    for i in range(n):
        assert got[i].tobytes() == hashlib.sha256(m[i].tobytes()).digest()


def _gtab_bits(engine):
    return int(engine._lib.s2k_engine_gtable_bits(engine._h))         # the width of the table this device got (26 unless memory was short)


def test_generator_table(prims, ref, engine):
    """entries (w, v) of the device-built window table equal v*2^(B w)*G computed by the reference's ecmult (role of
    test_pre_g_table, src/tests.c:4543-4615): every window's edge values plus a random sample.  The table holds the magnitudes
    v = 1 .. 2^(B-1) of a signed B-bit digit; the top window what is left of a 256-bit scalar plus the recoding's carry."""
    rng = np.random.default_rng(16)
    B = _gtab_bits(engine); W = (256 + B - 1) // B; top = 256 - B * (W - 1)
    nv = lambda w: (1 << (B - 1)) + 1 if w + 1 < W else (1 << top) + 2
    idx = [(w, v) for w in range(W) for v in (1, 2, 3, 255, 256, 257, nv(w) // 2 - 1, nv(w) // 2, nv(w) - 2, nv(w) - 1)]
    idx += [(w, int(rng.integers(1, nv(w)))) for w in rng.integers(0, W, 12000)]
    # the seeded construction's own edges (csrc/gtable.h): v = a * Kc + b with Kc = 2^(B // 2) columns, rows in runs of 16
    Kc = 1 << (B // 2)
    for w in range(W):
        for a in (1, 2, 15, 16, 17, 31, 32, (nv(w) - 1) // Kc - 1, (nv(w) - 1) // Kc):
            for b in (0, 1, 2, 63, 64, 255, 256, Kc - 1):
                v = a * Kc + b
                if 1 <= v < nv(w):
                    idx.append((w, v))
    n = len(idx)
    sel = np.array([(int(w) << 26) | v for (w, v) in idx], np.uint32)
    got, _ = prims(11, n, 64, sel.view(np.uint8))
    ng = np.stack([np.frombuffer(_b((v << (B * int(w))) % N), np.uint8) for (w, v) in idx])      # (the top window's largest entries exceed 2^256: mod n)
    g = np.frombuffer(G_XY * n, np.uint8).reshape(-1, 64)
    exp, inf = ref.ecmult_batch(g, np.zeros((n, 32), np.uint8), ng)
    assert not inf.any()
    assert np.array_equal(got, exp)


def test_chained(prims, ref):
    """Chained inline field ops (the intermediates of one Jacobian doubling) against big-integer arithmetic.
    Regression test for the ROCm 7.2 mul24 mis-widening described at S2K_OPAQUE (csrc/s2k_common.h): single
    fe_mul/fe_sqr calls were right, a product whose operand was the *output* of a previous product was not."""
    rng = np.random.default_rng(15)
    pts = [ref.rand_point(rng) for _ in range(64)] + [G_XY]
    n = len(pts)
    a = np.frombuffer(b"".join(pts), np.uint8).reshape(n, 64)
    got, _ = prims(20, n, 256, a)
    inv2 = pow(2, -1, P)
    for i in range(n):
        x = int.from_bytes(pts[i][:32], "big"); y = int.from_bytes(pts[i][32:], "big")
        S = y * y % P; L = 3 * x * x * inv2 % P; T = (-x * S) % P; X3 = (L * L + 2 * T) % P; S2 = S * S % P; XT = (X3 + T) % P
        exp = [y, S, L, T, X3, S2, XT, XT * L % P]
        for k, v in enumerate(exp):
            assert int.from_bytes(got[i, 32 * k:32 * k + 32].tobytes(), "big") == v, (i, k)


def _lazy_limbs(rng, n, mag, extreme):
    """n field elements as 9 raw limbs at magnitude `mag` of the fe.h contract (n[i] <= mag*(2^29+2^20), n[8] <= mag*(2^24+2^10));
    extreme: every limb AT its bound for the first rows, random below it afterwards"""
    hi = np.array([mag * ((1 << 29) + (1 << 20))] * 8 + [mag * ((1 << 24) + (1 << 10))], np.uint64)
    lim = (rng.random((n, 9)) * (hi + 1)).astype(np.uint64)
    lim = np.minimum(lim, hi)
    k = min(extreme, n)
    lim[:k] = hi
    if n > k + 1:
        lim[k] = hi; lim[k, ::2] = 0
        lim[k + 1] = 0
    return lim.astype(np.uint32)


def _val(limbs):
    return [sum(int(x) << (29 * i) for i, x in enumerate(row)) for row in limbs]


def test_lazy_magnitudes_on_device(prims):
    """fe_mul / fe_sqr / the lockstep pairs / the fused product pair (fe_muladd) with LAZY inputs at the limits of the magnitude
    contract (product of magnitudes 7: the 64-bit column accumulators are within a few percent of wrapping), on the device, against
    Python integers.  The host emulation asserts the same bounds (S2K_VERIFY); this is the device-side half, where the one real
    miscompile of round 1 lived."""
    rng = np.random.default_rng(71)
    n = 256
    u8 = lambda l: l.view(np.uint8)
    for ma, mb in ((7, 1), (1, 7), (3, 2), (2, 3), (1, 1), (5, 1)):
        a, b = _lazy_limbs(rng, n, ma, 4), _lazy_limbs(rng, n, mb, 4)
        got, _ = prims(30, n, 64, u8(a), u8(b))
        va, vb = _val(a), _val(b)
        for i in range(n):
            assert int.from_bytes(got[i, :32].tobytes(), "big") == va[i] * vb[i] % P, ("mul", ma, mb, i)
    for m in (1, 2):
        a = _lazy_limbs(rng, n, m, 4); va = _val(a)
        got, _ = prims(31, n, 64, u8(a))
        for i in range(n):
            assert int.from_bytes(got[i, :32].tobytes(), "big") == va[i] * va[i] % P, ("sqr", m, i)
    # pairs: (a*b, c*b), (a*b, c^2), (a^2, c^2)
    for op, ma, mb, mc in ((32, 3, 2, 3), (32, 7, 1, 1), (35, 3, 2, 2), (35, 1, 7, 2), (36, 2, 1, 2)):
        a, b, c = _lazy_limbs(rng, n, ma, 4), _lazy_limbs(rng, n, mb, 4), _lazy_limbs(rng, n, mc, 4)
        got, _ = prims(op, n, 64, u8(a), u8(b), u8(c))
        va, vb, vc = _val(a), _val(b), _val(c)
        for i in range(n):
            e1 = va[i] * vb[i] % P if op != 36 else va[i] * va[i] % P
            e2 = vc[i] * vb[i] % P if op == 32 else vc[i] * vc[i] % P
            assert int.from_bytes(got[i, :32].tobytes(), "big") == e1 and int.from_bytes(got[i, 32:].tobytes(), "big") == e2, (op, ma, mb, mc, i)
    # fused pairs, sum of magnitude products exactly 7
    for op, ma, mb, mc in ((33, 3, 2, 0), (34, 3, 1, 2), (34, 1, 3, 2), (34, 6, 1, 1)):
        # op 33: a*b + c*a needs ma*mb + mc*ma <= 7 -> (1, 3, 4) below; op 34: a*b + c^2 needs ma*mb + mc^2 <= 7
        if op == 33:
            ma, mb, mc = 1, 3, 4
        a, b, c = _lazy_limbs(rng, n, ma, 4), _lazy_limbs(rng, n, mb, 4), _lazy_limbs(rng, n, mc, 4)
        got, _ = prims(op, n, 64, u8(a), u8(b), u8(c))
        va, vb, vc = _val(a), _val(b), _val(c)
        for i in range(n):
            e = (va[i] * vb[i] + (vc[i] * va[i] if op == 33 else vc[i] * vc[i])) % P
            assert int.from_bytes(got[i, :32].tobytes(), "big") == e, (op, ma, mb, mc, i)


def test_lean_point_ops(prims, ref):
    """gej_double_lean / gej_add_ge_lean (the lock-step fast path of ecmult_lane) against the reference group law:
    2*(4P + Q); and the same-x report for Q = +-4P."""
    rng = np.random.default_rng(72)
    pts = [ref.rand_point(rng) for _ in range(40)] + [G_XY]
    n = len(pts)
    A = np.frombuffer(b"".join(pts), np.uint8).reshape(n, 64)
    B = np.roll(A, 3, axis=0).copy()
    g4, _ = ref.ecmult_batch(A, np.tile(np.frombuffer(_b(4), np.uint8), (n, 1)))
    B[0] = g4[0]; B[1] = g4[1]; B[1, 32:] = np.frombuffer(_b((P - int.from_bytes(g4[1, 32:].tobytes(), "big")) % P), np.uint8)
    got, same = prims(37, n, 64, A, B)
    assert list(same[:2]) == [1, 1] and not same[2:].any()
    for i in range(2, n):
        r1 = ref.call("ref_ge_add", [64], g4[i].tobytes(), 0, B[i].tobytes(), 0)[1][0]
        r2 = ref.call("ref_ge_double", [64], r1, 0)[1][0]
        assert got[i].tobytes() == r2, i


def test_two_piece_double_multiplication_on_device(prims, ref):
    """ecmult_lane_split (the rangeproof rings' form: T = 2^64 A given) with the caller's fallback, one class of scalar per wavefront so
    that the lock-step form really runs: edge values of the piece split (multiples of 2^64, lambda, n-1, ...), 0 (the form refuses, the
    wavefront falls back) and random scalars; random Jacobian Z on the inputs; against the reference's secp256k1_ecmult."""
    from tests.test_cpu_oracle import SPLIT_EDGE_SCALARS
    rng = np.random.default_rng(45)
    classes = SPLIT_EDGE_SCALARS + [None] * 7
    n = 64 * len(classes)
    base = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    A, _ = ref.ecmult_batch(np.tile(np.frombuffer(G_XY, np.uint8), (n, 1)), base)
    na = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for k, v in enumerate(classes):
        if v is not None:
            na[64 * k:64 * (k + 1)] = np.frombuffer(_b(v), np.uint8)
    ng = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for k in range(0, len(classes), 5):
        ng[64 * k:64 * (k + 1)] = 0                     # a whole wavefront without a generator part (the form wants the lanes to agree)
    z = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    want, winf = ref.ecmult_batch(A, na, ng=ng)
    got, flag = prims(38, n, 64, A, np.concatenate([na, ng], axis=1), z, scratch_words=n * 544 + 64)
    took = flag >> 1
    for k, v in enumerate(classes):
        sl = slice(64 * k, 64 * (k + 1))
        assert ((flag[sl] & 1) == winf[sl]).all(), k
        ok = winf[sl] == 0
        assert (got[sl][ok] == want[sl][ok]).all(), k
        assert took[sl].all() == (v != 0), (k, v)        # the two-piece form produced every class but na = 0


def test_cooperative_field_arithmetic(prims, ref):
    """cofield.h (one field element spread over the lanes of a wavefront: DPP lane shifts, v_readlane, ds_bpermute) against the
    integers and the reference group law, with lazily reduced inputs up to the magnitude contract's limits (product of magnitudes 7
    for a product, 6-7 for the fused pair) and all-ones limbs."""
    rng = np.random.default_rng(91)
    M29, TOP = (1 << 29) - 1, (1 << 24) - 1
    plimbs = [(P >> (29 * i)) & M29 for i in range(9)]

    def rep(v, mag, full=False):
        if full:
            return [mag * M29] * 8 + [mag * TOP]
        return [((v >> (29 * i)) & M29) + (mag - 1) * plimbs[i] for i in range(9)]
    val = lambda l: sum(x << (29 * i) for i, x in enumerate(l))
    items = 256
    A = np.zeros((items, 9), np.uint32); B = np.zeros((items, 9), np.uint32); want = []
    pts = np.frombuffer(b"".join(ref.rand_point(rng) for _ in range(items)), np.uint8).reshape(items, 64)
    mags = [(1, 1), (2, 1), (1, 2), (2, 1), (1, 1), (1, 2)]
    for it in range(items):
        ma, mb = mags[it % len(mags)]
        assert ma * mb + mb * mb <= 7 and 3 * ma <= 7 and ma * ma <= 7
        a = int.from_bytes(rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), "big") % P
        b = int.from_bytes(rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), "big") % P
        if it < 8:
            a, b = [(0, 0), (1, 1), (P - 1, P - 1), (P - 1, 1), (2**255, 2**255), (0, P - 1), (2**224 - 1, 2**232), (977, 2**32)][it]
        la, lb = rep(a, ma, full=(it % 16 == 9)), rep(b, mb, full=(it % 16 == 9 or it % 16 == 10))
        A[it] = la; B[it] = lb
        a, b = val(la) % P, val(lb) % P
        want.append((a * b % P, (a * b + b * b) % P, (b - 3 * a * pow(2, -1, P)) % P, a * b % P, b * b % P, a * a % P))
    dbl, _ = ref.ecmult_batch(pts, np.tile(np.frombuffer(_b(2**13), np.uint8), (items, 1)))
    got, flag = prims(39, 64 * items, 5, A.view(np.uint8), B.view(np.uint8), pts)
    got = got.reshape(-1)[:288 * items].reshape(items, 288)
    assert (flag[::64] == 15).all()                      # bit 0: lanes >= 9 still hold zero after every routine; bits 1-3: the three addition cases
    for it in range(items):
        for k in range(3):
            assert got[it, 32 * k:32 * k + 32].tobytes() == _b(want[it][k]), (it, k)
        assert got[it, 96:160].tobytes() == dbl[it].tobytes(), it
        for k in range(3):                               # the grouped product (three different products in one pass)
            assert got[it, 192 + 32 * k:224 + 32 * k].tobytes() == _b(want[it][3 + k]), (it, k)
