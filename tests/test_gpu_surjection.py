"""GPU parity: secp256k1_surjectionproof_verify_batch vs the reference's parse + secp256k1_surjectionproof_verify per item: the
fixed vectors of src/modules/surjection/tests_impl.h:488-632 (accept, wrong keys, malformed bitmaps / lengths) and freshly
generated proofs of mixed shapes with mutations, in one ragged batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fixed_vectors(engine):
    from tests.test_cpu_restatement import _sj_golden
    cases = _sj_golden()
    res = engine.surjectionproof_verify_batch([c[1] for c in cases], [np.frombuffer(c[2], np.uint8) for c in cases],
                                              np.stack([np.frombuffer(c[4], np.uint8) for c in cases]))
    assert list(res) == [c[5] for c in cases]


def test_random_ragged_batch(engine, ref):
    rng = np.random.default_rng(43)
    proofs, tags, outs = [], [], []
    for (n_in, n_used, cnt) in ((1, 1, 6), (3, 1, 6), (3, 3, 8), (8, 3, 8), (20, 5, 4), (256, 2, 2), (40, 16, 2)):
        for _ in range(cnt):
            p, t, o = ref.make_surjection(rng, n_in, n_used)
            proofs.append(p); tags.append(t); outs.append(o)
    n = len(proofs)
    for i in range(n):            # mutated copies
        p = bytearray(proofs[i]); p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8))
        proofs.append(bytes(p)); tags.append(tags[i]); outs.append(outs[i])
        t2 = tags[i].copy(); t2[int(rng.integers(0, t2.shape[0])), int(rng.integers(0, 64))] ^= 1
        proofs.append(proofs[i]); tags.append(t2); outs.append(outs[i])
    proofs += [b"", b"\x01", b"\x01\x00\x01" + b"\x00" * 10]
    tags += [tags[0], tags[0], tags[0]]; outs += [outs[0]] * 3
    exp = np.array([ref.surjection_verify(p, t, o) for p, t, o in zip(proofs, tags, outs)], np.int32)
    res = engine.surjectionproof_verify_batch(proofs, tags, np.stack(outs))
    assert np.array_equal(res, exp)
    assert exp[:n].all() and exp.sum() < len(proofs)
