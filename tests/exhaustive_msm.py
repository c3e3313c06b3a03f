"""The reference's exhaustive small-group check of the MSM (src/tests_exhaustive.c:198-227: every  i*P_x + j*P_y + k*G  through
secp256k1_ecmult_multi_var, compared with group[(i x + j y + k) mod order]) has the same shape here, on the real curve: the "group" is a
list of multiples p_x*G (p = 0: the point at infinity) and the scalars come from a list of EDGE values of the arithmetic underneath -- 0, 1,
small values, n - 1, n - 2, the GLV constant lambda and its negative, the values around 2^128 where the halves of the split change size,
(n +- 1) / 2 -- so that every combination of zero scalars, points at infinity, equal and opposite points, cancelling terms and results at
infinity occurs; the expected point is ((s_i p_x + s_j p_y + s_k) mod n)*G by the reference's secp256k1_ecmult."""
import numpy as np

from tests.refapi import G_XY, N

LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
EDGE_SCALARS = [0, 1, 2, 3, N - 1, N - 2, LAMBDA, N - LAMBDA, 1 << 128, (1 << 128) - 1, (N - 1) // 2, (N + 1) // 2, 0xFFFFFFFF]
EDGE_POINTS = [0, 1, N - 1, 2, LAMBDA, N - LAMBDA, 3, (N - 1) // 2, 1 << 128]          # multiples of G; 0 = infinity, N - 1 = -G, lambda = the endomorphism's image of G


def _b(v):
    return np.frombuffer(int(v).to_bytes(32, "big"), np.uint8)


def group_points(ref, multiples):
    """(xy, inf) of p*G for every p"""
    g = np.tile(np.frombuffer(G_XY, np.uint8), (len(multiples), 1))
    xy, inf = ref.ecmult_batch(g, np.zeros((len(multiples), 32), np.uint8), np.stack([_b(p % N) for p in multiples]))
    return xy, np.asarray(inf)


def cases(scalars, points):
    for i in scalars:
        for j in scalars:
            for k in scalars:
                for x in range(len(points)):
                    for y in range(len(points)):
                        yield i, j, k, x, y


def expected_points(ref, values):
    """e*G for every distinct e (0 -> infinity)"""
    vals = sorted(set(values))
    xy, inf = group_points(ref, vals)
    return {v: (xy[t].tobytes(), int(inf[t]) != 0) for t, v in enumerate(vals)}
