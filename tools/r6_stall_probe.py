#!/usr/bin/env python3
"""Are there sporadic long calls?  Each operation 300 times, every call timed on the host with a wait (enqueue + completion): median, p99, max and
the number of calls slower than 3x the median.  S2K_LIB selects the library (A/B against round 5)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from secp256k1_zkp_amd import Engine
from tests.refapi import Ref, G_XY
ref = Ref(); eng = Engine(0); dev = torch.device("cuda:0"); rng = np.random.default_rng(1)
def probe(name, fn, reps=300, group=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(group): fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts); med = np.median(ts)
    print("%-28s median %8.3f ms  p99 %8.3f  max %8.3f  groups of %d calls slower than 3x median: %d of %d" % (name, med, np.percentile(ts, 99), ts.max(), group, int((ts > 3 * med).sum()), reps), flush=True)
n = 1 << 16
sigs, msgs, pks = ref.make_schnorr(n, rng, threads=16)
d = [torch.tensor(x).to(dev) for x in (sigs, msgs, pks)]; res = torch.zeros(n, dtype=torch.int32, device=dev)
probe("bip340 2^16", lambda: eng.schnorrsig_verify_batch_dev(res, d[0], d[1], d[2]))
if hasattr(eng._lib, "s2k_ecmult_multi_many_dev") and eng._lib.s2k_ecmult_multi_many_dev.argtypes is not None:
    K, nm = 256, 1024
    ks = rng.integers(0, 256, (nm, 32), dtype=np.uint8)
    pts, _ = ref.ecmult_batch(np.tile(np.frombuffer(G_XY, np.uint8), (nm, 1)), ks)
    d_s = torch.tensor(rng.integers(0, 256, (K * nm, 32), dtype=np.uint8)).to(dev); d_p = torch.tensor(np.tile(pts, (K, 1))).to(dev)
    r_xy = torch.zeros(K, 64, dtype=torch.uint8, device=dev); r_inf = torch.zeros(K, dtype=torch.int32, device=dev); offs = (np.arange(K + 1) * nm).astype(np.uint64)
    try:
        probe("many 256 x 1024", lambda: eng.ecmult_multi_many_dev(r_xy, r_inf, d_s, d_p, offs))
    except Exception as ex: print("many: n/a", ex)
nm = 1 << 16
ks = torch.tensor(rng.integers(0, 256, (nm, 32), dtype=np.uint8)).to(dev)
gp = torch.tensor(np.frombuffer(G_XY, np.uint8).copy()).to(dev).repeat(nm, 1); pt = torch.zeros(nm, 64, dtype=torch.uint8, device=dev); pi = torch.zeros(nm, dtype=torch.int32, device=dev); z = torch.zeros(nm, 32, dtype=torch.uint8, device=dev)
probe("ecmult_batch 2^16", lambda: eng.ecmult_batch_dev(pt, pi, gp, z, ks))
r = torch.zeros(64, dtype=torch.uint8, device=dev); ri = torch.zeros(1, dtype=torch.int32, device=dev)
probe("msm 2^16", lambda: eng.ecmult_multi_dev(r, ri, ks, pt))
