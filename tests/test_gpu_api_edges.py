"""GPU: API edge behaviour of the C ABI -- empty batches succeed without touching the device buffers, illegal arguments and a null
engine fail loudly (return 0 + s2k_last_error), workspace reservation, the MSM of nothing is the point at infinity."""
import ctypes

import numpy as np
import pytest

from secp256k1_zkp_amd import _native

pytestmark = pytest.mark.gpu


def test_empty_batches(engine):
    z = np.zeros((0, 64), np.uint8)
    assert engine.ecmult_batch(z, np.zeros((0, 32), np.uint8), None, None)[0].shape[0] == 0
    assert engine.schnorrsig_verify_batch(z, np.zeros((0, 32), np.uint8), np.zeros((0, 32), np.uint8)).size == 0
    res, mn, mx = engine.rangeproof_verify_batch(np.zeros((0, 33), np.uint8), [], z)
    assert res.size == 0
    assert engine.surjectionproof_verify_batch([], [], z).size == 0
    assert engine.pedersen_verify_tally_batch([]).size == 0
    xy, inf = engine.ecmult_multi(np.zeros((0, 32), np.uint8), z, None, None)
    assert inf == 1 and not xy.any()
    xy, inf = engine.ecmult_multi(np.zeros((0, 32), np.uint8), z, bytes(32), None)          # 0*G
    assert inf == 1


def test_null_engine_and_illegal_arguments(engine):
    lib = _native.load()
    res = np.zeros(4, np.int32)
    assert lib.secp256k1_schnorrsig_verify_batch(None, res.ctypes.data_as(ctypes.c_void_p), None, None, 32, None, 0, 4) == 0
    assert b"null engine" in lib.s2k_last_error()
    assert lib.s2k_engine_sync(None) == 0
    # aggverify: NULL result pointer / NULL aggregate are ARG_CHECK failures in the reference (illegal callback), a loud 0 here
    assert lib.secp256k1_schnorrsig_aggverify_amd(engine._h, None, None, 0, None, 0, None, 32) == 0
    assert b"illegal argument" in lib.s2k_last_error()
    # tallies: offsets that run backwards
    off = np.array([0, 2, 1], np.uint64); npos = np.array([1, 0], np.uint64); c = np.zeros((2, 33), np.uint8); r2 = np.zeros(2, np.int32)
    assert lib.secp256k1_pedersen_verify_tally_batch(engine._h, r2.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p), off.ctypes.data_as(ctypes.c_void_p),
                                                     npos.ctypes.data_as(ctypes.c_void_p), 2) == 0
    assert b"malformed" in lib.s2k_last_error()
    assert lib.s2k_engine_create(10**6) in (None, 0)
    assert b"out of range" in lib.s2k_last_error()


def test_reserve_and_reuse(engine, ref):
    lib = _native.load()
    assert lib.s2k_engine_reserve(engine._h, 2048) == 1
    rng = np.random.default_rng(5)
    c, p, g, _ = ref.make_rangeproofs(8, rng, min_bits=16)
    a = engine.rangeproof_verify_batch(c, p, g)
    b = engine.rangeproof_verify_batch(c, p, g)                 # same buffers reused, same verdicts
    assert a[0].all() and np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])


def test_wrong_shaped_host_arrays_raise(engine, ref):
    """the host wrappers derive n from one array; every other array is checked against it before anything reaches hipMemcpy"""
    rng = np.random.default_rng(6)
    c, p, g, _ = ref.make_rangeproofs(3, rng, min_bits=8)
    with pytest.raises(ValueError):
        engine.rangeproof_verify_batch(c[:2], p, g)
    with pytest.raises(ValueError):
        engine.rangeproof_verify_batch(c, p, g[:, :63])
    data, off = engine.pack(list(p)); off2 = off.copy(); off2[1], off2[2] = off[2], off[1]
    with pytest.raises(ValueError):
        engine.rangeproof_verify_batch(c, (data, off2), g)
    with pytest.raises(ValueError):
        engine.rangeproof_verify_batch(c, (data[:-5], off), g)
    sigs, msgs, pks = ref.make_schnorr(4, rng)
    with pytest.raises(ValueError):
        engine.schnorrsig_verify_batch(sigs, msgs[:3], pks)
    with pytest.raises(ValueError):
        engine.schnorrsig_verify_batch(sigs, msgs, pks[:, :31])
    with pytest.raises(ValueError):
        engine.ecmult_multi(np.zeros((4, 32), np.uint8), np.zeros((3, 64), np.uint8))
    with pytest.raises(ValueError):
        engine.ecmult_batch(np.zeros((4, 64), np.uint8), np.zeros((4, 31), np.uint8))
