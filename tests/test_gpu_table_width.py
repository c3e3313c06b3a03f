"""GPU: the fixed-base tables at a narrower digit width (csrc/ecmult.h, csrc/gtable.h -- the role of the reference's ECMULT_WINDOW_SIZE knob,
src/ecmult.h:14-38).  The width is a property of the table, chosen when the device's first table is allocated: $S2K_GTAB_BITS at start-up, or
whatever fits when HBM is short (26 bits = 21.5 GB per table down to 20 bits = 0.44 GB).  Each case runs in a process of its own (the tables
belong to the device's pool for the life of the process): full-size config-3 batch with mutated proofs and per-proof generators, double
multiplications on the fixed-base digit edges, BIP-340, an MSM with a generator term, and the table's entries themselves -- all against the
reference, bit for bit, whatever the width."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.environ["S2K_ROOT"])
from tests import refapi
from tests.refapi import G_XY, N
from tests.test_cpu_oracle import fixed_base_edge_scalars
from secp256k1_zkp_amd import Engine
ref = refapi.Ref()
rng = np.random.default_rng(int(os.environ.get("S2K_SEED", "7")))
leave = float(os.environ.get("S2K_LEAVE_GB", "0"))
hog = None
if leave > 0:                                   # take all of the HBM but `leave` GB before the engine allocates anything
    free, total = torch.cuda.mem_get_info(0)
    hog = torch.empty(int(free - leave * 2**30), dtype=torch.uint8, device="cuda:0")
free0, _ = torch.cuda.mem_get_info(0)
eng = Engine(0)
t0 = time.time()
sz = ctypes.c_size_t(0)
gtab = eng._lib.s2k_engine_gtable(eng._h, ctypes.byref(sz))
build_s = time.time() - t0
D = int(eng._lib.s2k_engine_gtable_bits(eng._h))
out = {"bits": D, "table_bytes": int(sz.value), "table_build_s": build_s}
assert gtab, "no table"
# (1) rangeproofs: n 64-bit proofs on secp256k1_generator_h + 52-bit ones on per-proof generators, a share of them mutated
n = int(os.environ.get("S2K_N", "4096"))
c, p, g, _ = ref.make_rangeproofs(n, rng, min_bits=64)
gens2 = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(64)])[rng.integers(0, 64, 256)]
c2, p2, g2, _ = ref.make_rangeproofs(256, rng, min_bits=52, gens64=gens2)
C = np.concatenate([c, c2]); P = list(p) + list(p2); G = np.concatenate([g, g2])
for i in range(0, len(P), 7):
    q = bytearray(P[i]); q[int(rng.integers(0, len(q)))] ^= 1 << int(rng.integers(0, 8)); P[i] = bytes(q)
want = ref.rangeproof_verify_many(C, P, G, threads=8)
for rep in range(2):                           # (the second call finds secp256k1_generator_h's table in the cache, when there was room for one)
    got = eng.rangeproof_verify_batch(C, P, G)
    assert all(np.array_equal(a, b) for a, b in zip(got, want)), "rangeproof verdicts differ at %d bits" % D
out["rangeproofs"] = len(P); out["accepted"] = int(np.asarray(want[0]).sum())
out["generator_h_cached"] = int(eng._lib.s2k_engine_generator_cached(eng._h, refapi.GENERATOR_H))
# (2) double multiplications with ng on the signed-digit edges of THIS width
sc = fixed_base_edge_scalars(D, rng, 256)
m = len(sc)
ng = np.stack([np.frombuffer(int(v).to_bytes(32, "big"), np.uint8) for v in sc])
A, _ = ref.ecmult_batch(np.tile(np.frombuffer(G_XY, np.uint8), (m, 1)), rng.integers(0, 256, (m, 32), dtype=np.uint8))
for na in (np.zeros((m, 32), np.uint8), rng.integers(0, 256, (m, 32), dtype=np.uint8)):
    w_xy, w_inf = ref.ecmult_batch(A, na, ng=ng)
    g_xy, g_inf = eng.ecmult_batch(A, na, ng=ng)
    assert np.array_equal(g_inf != 0, np.asarray(w_inf) != 0) and np.array_equal(g_xy[np.asarray(w_inf) == 0], w_xy[np.asarray(w_inf) == 0]), "ecmult differs at %d bits" % D
# (3) BIP-340 and an MSM with a generator term
sigs, msgs, pks = ref.make_schnorr(512, rng, threads=4)
sigs[::5, 40] ^= 1
assert np.array_equal(eng.schnorrsig_verify_batch(sigs, msgs, pks), ref.schnorr_verify_many(sigs, msgs, pks))
scs = rng.integers(0, 256, (3000, 32), dtype=np.uint8)
pts = np.tile(A, (3000 // m + 1, 1))[:3000]
gs = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
w = ref.ecmult_multi(scs, pts, gs); gt = eng.ecmult_multi(scs, pts, gs)
assert gt[1] == w[1] and np.array_equal(gt[0], w[0])
# (4) the table's own entries: edges of every window and of the seeded construction, and a sample
W = (256 + D - 1) // D; top = 256 - D * (W - 1); Kc = 1 << (D // 2)
nv = lambda w: (1 << (D - 1)) + 1 if w + 1 < W else (1 << top) + 2
idx = [(w, v) for w in range(W) for v in (1, 2, 3, Kc - 1, Kc, Kc + 1, 16 * Kc - 1, 16 * Kc, 16 * Kc + 1, nv(w) // 2, nv(w) - 2, nv(w) - 1) if 1 <= v < nv(w)]
idx += [(int(w), int(rng.integers(1, nv(w)))) for w in rng.integers(0, W, 3000)]
prim = ctypes.CDLL(os.path.join(os.environ["S2K_ROOT"], "tests", "gpu_prims", "libs2k_gpuprims.so"))
sel = torch.tensor(np.array([(w << 26) | v for (w, v) in idx], np.uint32).view(np.uint8)).cuda()
o = torch.zeros(len(idx), 64, dtype=torch.uint8, device="cuda"); fl = torch.zeros(len(idx), dtype=torch.int32, device="cuda")
prim.s2k_test_prim.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int]
torch.cuda.synchronize()
prim.s2k_test_prim(11, o.data_ptr(), fl.data_ptr(), sel.data_ptr(), None, None, gtab, len(idx))
torch.cuda.synchronize()
ngs = np.stack([np.frombuffer(((v << (D * w)) % N).to_bytes(32, "big"), np.uint8) for (w, v) in idx])
exp, inf = ref.ecmult_batch(np.tile(np.frombuffer(G_XY, np.uint8), (len(idx), 1)), np.zeros((len(idx), 32), np.uint8), ngs)
assert not np.asarray(inf).any() and np.array_equal(o.cpu().numpy(), exp), "table entries differ at %d bits" % D
free1, _ = torch.cuda.mem_get_info(0)
out["hbm_used_gb"] = (free0 - free1) / 2**30
print("RESULT " + json.dumps(out))
'''


def _child(env_extra, timeout=900):
    env = dict(os.environ, S2K_ROOT=ROOT)
    env.pop("S2K_GTAB_BITS", None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert len(line) == 1, r.stdout[-2000:]
    return json.loads(line[0][7:])


@pytest.mark.parametrize("bits", [20, 24])
def test_asked_for_width(bits, engine):
    j = _child({"S2K_GTAB_BITS": str(bits)})
    assert j["bits"] == bits
    W = (256 + bits - 1) // bits
    assert j["table_bytes"] == ((W << (bits - 1)) + 1) * 64
    if bits == 20:
        assert j["table_bytes"] < 2**29 and j["hbm_used_gb"] < 4.0          # G's table, H's table, scratch: an engine in a few GB
    print("\n%d-bit tables: %s" % (bits, j))


def test_width_follows_the_memory_that_is_there(engine):
    """an engine under an artificial cap: everything but 9 GB of the HBM is taken before the first table is allocated -- the table of G comes
    out narrower than 26 bits (24: 5.9 GB), the generator's table may not fit at all (general form), and every result is still the reference's"""
    j = _child({"S2K_LEAVE_GB": "9", "S2K_N": "2048"})
    assert 20 <= j["bits"] < 26 and j["table_bytes"] < 9 * 2**30
    print("\nunder a 9 GB cap: %s" % j)


def test_width_changed_at_run_time(engine, ref):
    """S2K_OPT_GTAB_BITS: the device's tables are given back and rebuilt at another width by the next call; every result stays the reference's
    (rangeproofs on the shared-generator form, single multiplications through the table of G), whatever the width and in whatever order."""
    import time
    import numpy as np
    import torch
    from secp256k1_zkp_amd import Engine
    from tests.refapi import G_XY
    # (the child of the test above took all but 9 GB of the HBM; the driver gives a dead process's memory back from a work queue, not
    #  before the process is reaped -- an engine that sizes its tables in that window rightly picks a narrower width, which is not what
    #  this test is about)
    t0 = time.time()
    while torch.cuda.mem_get_info(0)[0] < (64 << 30) and time.time() - t0 < 60: time.sleep(0.2)
    assert torch.cuda.mem_get_info(0)[0] >= (64 << 30), "the device's memory is still held %.0f s after the previous test's process ended" % (time.time() - t0)
    rng = np.random.default_rng(2611)
    c, p, g, _ = ref.make_rangeproofs(96, rng, min_bits=64)
    q = bytearray(p[5]); q[300] ^= 2; p[5] = bytes(q)
    e_res, e_mn, e_mx = ref.rangeproof_verify_many(c, p, g, threads=8)
    k = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    gpts = np.frombuffer(G_XY * 64, np.uint8).reshape(64, 64)
    w_xy, w_inf = ref.ecmult_batch(gpts, np.zeros((64, 32), np.uint8), ng=k, a_inf=np.ones(64, np.uint8))
    try:
        for bits in (24, 20, 26, 22):
            engine.set_option(Engine.OPT_GTAB_BITS, bits)
            res, mn, mx = engine.rangeproof_verify_batch(c, p, g)
            assert int(engine._lib.s2k_engine_gtable_bits(engine._h)) == bits
            assert np.array_equal(res, e_res) and np.array_equal(mn, e_mn) and np.array_equal(mx, e_mx), bits
            xy, inf = engine.ecmult_batch(gpts, np.zeros((64, 32), np.uint8), ng=k, a_inf=np.ones(64, np.uint8))
            assert np.array_equal(xy, w_xy) and not inf.any(), bits
    finally:
        engine.set_option(Engine.OPT_GTAB_BITS, 26)
