"""GPU: the counterpart of the reference's test_exhaustive_ecmult_multi (src/tests_exhaustive.c:198-227) through s2k_ecmult_multi -- see
tests/exhaustive_msm.py.  Every (i, j, k) of the edge scalars with every (x, y) of the edge points, once as the three-term sum the
reference's test makes (the engine's bucket-free path below 32 terms) and once padded with forty inactive terms -- zero scalars, points at
infinity: the reference skips them, src/ecmult_impl.h:523 -- so that the same sums go through the bucket pipeline."""
import numpy as np
import pytest

from tests import exhaustive_msm as xm
from tests.refapi import N

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("padded", [0, 1])
def test_exhaustive_edge_group(engine, ref, padded):
    S = xm.EDGE_SCALARS[:9] if padded else xm.EDGE_SCALARS[:11]
    P = xm.EDGE_POINTS[:7] if padded else xm.EDGE_POINTS[:6]
    pxy, pinf = xm.group_points(ref, P)
    combos = list(xm.cases(S, range(len(P))))
    want = xm.expected_points(ref, [(i * P[x] + j * P[y] + k) % N for (i, j, k, x, y) in combos])
    rng = np.random.default_rng(5)
    pad_pts = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(40)])
    pad_sc = rng.integers(0, 256, (40, 32), dtype=np.uint8); pad_inf = np.zeros(40, np.uint8)
    pad_sc[::2] = 0; pad_inf[1::2] = 1                                    # every padding term is inactive: zero scalar or point at infinity
    bad = 0
    for (i, j, k, x, y) in combos:
        sc = np.stack([xm._b(i), xm._b(j)]); pts = np.stack([pxy[x], pxy[y]]); inf = np.array([pinf[x] != 0, pinf[y] != 0], np.uint8)
        if padded:
            sc = np.concatenate([sc[:1], pad_sc[:20], sc[1:], pad_sc[20:]]); pts = np.concatenate([pts[:1], pad_pts[:20], pts[1:], pad_pts[20:]])
            inf = np.concatenate([inf[:1], pad_inf[:20], inf[1:], pad_inf[20:]])
        got, ginf = engine.ecmult_multi(sc, pts, bytes(xm._b(k)), inf)
        exy, einf = want[(i * P[x] + j * P[y] + k) % N]
        if bool(ginf) != einf or (not einf and got.tobytes() != exy):
            bad += 1
            assert bad < 5, ("differs", hex(i), hex(j), hex(k), x, y, padded)
    assert bad == 0
    assert len(combos) == len(S) ** 3 * len(P) ** 2
