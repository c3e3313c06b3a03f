"""GPU parity: secp256k1_pedersen_verify_tally_batch vs the reference's secp256k1_pedersen_verify_tally per tally: the committed
fixtures (reference-generated transactions and their broken variants), and one ragged batch of random transactions of every
size from empty to thousands of commitments (the partial-sum rounds), with failures mixed in."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fixtures(engine):
    from tests.test_cpu_restatement import _pt_golden
    cases = _pt_golden()
    res = engine.pedersen_verify_tally_batch([(c[1], c[2]) for c in cases])
    assert [int(r) for r in res] == [max(0, c[3]) for c in cases]          # -1 (unparseable) is reported as 0


def test_random_ragged_batch(engine, ref):
    rng = np.random.default_rng(61)
    tallies = []
    for (ni, no) in [(1, 1), (1, 2), (2, 3), (5, 4), (7, 9), (8, 8), (30, 70), (1, 64), (65, 1), (513, 700), (3000, 2500)] * 2:
        tallies.append(ref.make_balanced_tally(rng, ni, no))
    n = len(tallies)
    for k in range(n):                                         # broken variants
        a, b = tallies[k]
        if k % 4 == 0: tallies.append((a, b[:-1]))
        elif k % 4 == 1: c = b.copy(); c[int(rng.integers(0, len(c))), 0] ^= 1; tallies.append((a, c))
        elif k % 4 == 2: tallies.append((np.concatenate([a, a[:1]]), b))
        else: tallies.append((b, a))                            # still balanced
    e = np.zeros((0, 33), np.uint8)
    tallies += [(e, e), (tallies[0][0], e), (e, tallies[0][0]), (tallies[3][0][:1], tallies[3][0][:1])]
    order = rng.permutation(len(tallies))
    tallies = [tallies[i] for i in order]
    exp = np.maximum(ref.pedersen_verify_tally_many(tallies), 0)
    res = engine.pedersen_verify_tally_batch(tallies)
    assert np.array_equal(res, exp)
    assert 0 < exp.sum() < len(tallies)
