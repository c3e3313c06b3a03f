"""CPU tier: the shippable recipe for the hooked library -- integration/secp256k1_amd_hook.patch + integration/build_hooked.sh -- applies to
the reference tree (a COPY of it: the tree itself is never written to) and builds with the reference's own CMake; the resulting
libsecp256k1.so exports the hook next to the library's API, serves a batch on the CPU when no backend is installed and through the
backend table when one is.  (The test tier's own hooked library, oracle/_ref/libsecp256k1_hooked.so, is built differently: through the
oracle shim, with test drivers.)  Skipped where the reference tree or cmake is absent (the GPU box)."""
import ctypes
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("S2K_REFERENCE", "/root/reference")


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    if not os.path.isdir(os.path.join(REF, "src")) or shutil.which("cmake") is None or shutil.which("patch") is None:
        pytest.skip("reference tree / cmake / patch not present")
    out = tmp_path_factory.mktemp("hooked")
    r = subprocess.run(["bash", os.path.join(ROOT, "integration", "build_hooked.sh"), REF, str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr + open(os.path.join(out, "build.log")).read()[-2000:] if os.path.exists(os.path.join(out, "build.log")) else r.stderr
    return ctypes.CDLL(os.path.join(out, "build", "lib", "libsecp256k1.so"))


def test_patched_reference_builds_and_serves(built):
    L = built
    for name in ("secp256k1_amd_set_backend", "secp256k1_amd_stats", "secp256k1_amd_set_msm_min_terms", "secp256k1_amd_rangeproof_verify_batch",
                 "secp256k1_amd_rangeproof_verify_batch_submit", "secp256k1_amd_rangeproof_verify_batch_wait", "secp256k1_amd_rangeproof_rewind_batch",
                 "secp256k1_amd_schnorrsig_verify_batch", "secp256k1_amd_schnorrsig_aggverify", "secp256k1_amd_surjectionproof_verify_batch",
                 "secp256k1_amd_pedersen_verify_tally_batch", "secp256k1_rangeproof_verify", "secp256k1_context_create"):
        assert hasattr(L, name), name
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    L.secp256k1_context_create.restype = vp; L.secp256k1_context_create.argtypes = [ctypes.c_uint]
    L.secp256k1_amd_rangeproof_verify_batch.argtypes = [vp] * 10 + [sz]
    L.secp256k1_pedersen_commitment_parse.argtypes = [vp, vp, ctypes.c_char_p]
    L.secp256k1_amd_set_backend.argtypes = [vp]; L.secp256k1_amd_stats.argtypes = [ctypes.POINTER(sz), ctypes.POINTER(sz)]
    ctx = L.secp256k1_context_create(1)
    vecs = json.load(open(os.path.join(HERE, "golden", "rangeproof_vectors.json")))["vectors"]
    from tests.refapi import GENERATOR_H
    n = len(vecs)
    commits = [ctypes.create_string_buffer(64) for _ in vecs]
    for c, v in zip(commits, vecs):
        assert L.secp256k1_pedersen_commitment_parse(ctx, c, bytes.fromhex(v["commit33"])) == 1
    proofs = [ctypes.create_string_buffer(bytes.fromhex(v["proof"]), len(v["proof"]) // 2) for v in vecs]
    gens = [ctypes.create_string_buffer(GENERATOR_H, 64) for _ in vecs]
    arr = lambda xs: (vp * n)(*[ctypes.addressof(x) for x in xs])
    plens = (sz * n)(*[len(v["proof"]) // 2 for v in vecs])

    def run():
        res = (ctypes.c_int * n)(); mn = (ctypes.c_uint64 * n)(); mx = (ctypes.c_uint64 * n)()
        assert L.secp256k1_amd_rangeproof_verify_batch(ctx, res, mn, mx, arr(commits), arr(proofs), plens, None, None, arr(gens), n) == 1
        return list(res), list(mn), list(mx)
    want = ([v["result"] for v in vecs], [int(v["min_value"]) for v in vecs], [int(v["max_value"]) for v in vecs])
    assert run() == want                                       # no backend: the library's own verifier
    # a backend that answers from a table (it is the dispatch that is under test here, not an engine)
    calls = []
    FN = ctypes.CFUNCTYPE(ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz)

    def backend(engine, results, mn, mx, cobj, pr, pl, ex, el, gobj, cnt):
        calls.append(cnt)
        r = ctypes.cast(results, ctypes.POINTER(ctypes.c_int32)); a = ctypes.cast(mn, ctypes.POINTER(ctypes.c_uint64)); b = ctypes.cast(mx, ctypes.POINTER(ctypes.c_uint64))
        for i in range(cnt):
            r[i] = want[0][i]; a[i] = want[1][i]; b[i] = want[2][i]
        return 1
    cb = FN(backend)

    class Backend(ctypes.Structure):
        _fields_ = [(k, vp) for k in ("engine", "rangeproof_verify_batch", "ecmult_multi", "schnorrsig_verify_batch", "surjectionproof_verify_batch", "pedersen_verify_tally_batch",
                                      "schnorrsig_aggverify", "rangeproof_rewind_batch", "rangeproof_verify_batch_ptrs", "ecmult_batch", "bppp_norm_product_verify_batch",
                                      "rangeproof_verify_batch_ptrs_submit", "rangeproof_verify_batch_wait")]
    b = Backend(); b.rangeproof_verify_batch_ptrs = ctypes.cast(cb, vp).value
    L.secp256k1_amd_set_backend(ctypes.byref(b))
    s0, f0 = sz(0), sz(0); L.secp256k1_amd_stats(ctypes.byref(s0), ctypes.byref(f0))
    assert run() == want and calls == [n]
    s1, f1 = sz(0), sz(0); L.secp256k1_amd_stats(ctypes.byref(s1), ctypes.byref(f1))
    assert s1.value == s0.value + 1 and f1.value == f0.value
    L.secp256k1_amd_set_backend(None)
