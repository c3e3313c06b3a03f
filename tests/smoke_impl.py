"""smoke(): one small invocation of the hot path on cuda:0, checked against the oracle (golden vectors recorded from the
reference's own tests, plus -- when oracle/_ref or the C restatement is present -- a live per-item comparison)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def run():
    import torch
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    from secp256k1_zkp_amd import Engine
    from tests.refapi import GENERATOR_H, G_XY
    eng = Engine(0)
    vecs = json.load(open(os.path.join(HERE, "golden", "rangeproof_vectors.json")))["vectors"]
    n = len(vecs)
    commits = np.stack([np.frombuffer(bytes.fromhex(v["commit33"]), np.uint8) for v in vecs])
    proofs = [bytes.fromhex(v["proof"]) for v in vecs]
    bad = bytearray(proofs[3]); bad[1000] ^= 4
    proofs_all = proofs + [bytes(bad)]
    commits_all = np.concatenate([commits, commits[3:4]])
    gens = np.frombuffer(GENERATOR_H * (n + 1), np.uint8).reshape(n + 1, 64)
    res, mn, mx = eng.rangeproof_verify_batch(commits_all, proofs_all, gens)
    for i, v in enumerate(vecs):
        assert res[i] == v["result"] and int(mn[i]) == int(v["min_value"]) and int(mx[i]) == int(v["max_value"]), v["name"]
    assert res[n] == 0
    print(f"smoke: {n} reference rangeproof vectors verified on GPU, 1 mutated proof rejected; rings kernel {eng.last_ms(1):.2f} ms")
    try:
        from tests.refapi import Ref
        ref = Ref()
    except OSError:
        ref = None
    if ref is not None:
        rng = np.random.default_rng(1)
        k = 64
        a = np.frombuffer(G_XY * k, np.uint8).reshape(k, 64)
        na = rng.integers(0, 256, (k, 32), dtype=np.uint8); ng = rng.integers(0, 256, (k, 32), dtype=np.uint8)
        r, inf = eng.ecmult_batch(a, na, ng)
        r2, inf2 = ref.ecmult_batch(a, na, ng)
        assert np.array_equal(r, r2) and np.array_equal(inf, inf2)
        print("smoke: 64 double multiplications bit-exact vs oracle/_ref")
    else:
        import ctypes
        zpath = os.path.join(os.path.dirname(HERE), "oracle", "libzkp_oracle.so")
        if os.path.exists(zpath):
            zo = ctypes.CDLL(zpath)
            rng = np.random.default_rng(1)
            k = 16
            a = np.frombuffer(G_XY * k, np.uint8).reshape(k, 64)
            na = rng.integers(0, 256, (k, 32), dtype=np.uint8); ng = rng.integers(0, 256, (k, 32), dtype=np.uint8)
            r, inf = eng.ecmult_batch(a, na, ng)
            for i in range(k):
                out = ctypes.create_string_buffer(64)
                assert zo.zo_ecmult(out, a[i].tobytes(), 0, na[i].tobytes(), ng[i].tobytes()) == inf[i] and out.raw == r[i].tobytes()
            print("smoke: 16 double multiplications bit-exact vs oracle/zkp_oracle.c")
    eng.close()
