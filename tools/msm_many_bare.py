#!/usr/bin/env python3
"""K independent sums through s2k_ecmult_multi_many_dev (inputs resident) against K calls of s2k_ecmult_multi_dev.  python tools/msm_many_bare.py [K n]..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from secp256k1_zkp_amd import Engine
from tests.refapi import G_XY
args = [int(x) for x in sys.argv[1:]] or [256, 1024, 4096, 64, 1024, 100, 64, 4096, 16, 8192]
cases = list(zip(args[0::2], args[1::2]))
eng = Engine(0); dev = torch.device("cuda:0"); rng = np.random.default_rng(3)
nmax = max(k * n for k, n in cases)
ks = torch.tensor(rng.integers(0, 256, (nmax, 32), dtype=np.uint8)).to(dev)
gpts = torch.tensor(np.frombuffer(G_XY, np.uint8).copy()).to(dev).repeat(nmax, 1)
pts = torch.zeros(nmax, 64, dtype=torch.uint8, device=dev); pinf = torch.zeros(nmax, dtype=torch.int32, device=dev); z = torch.zeros(nmax, 32, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
eng.ecmult_batch_dev(pts, pinf, gpts, z, ks); eng.sync()
scs = torch.tensor(rng.integers(0, 256, (nmax, 32), dtype=np.uint8)).to(dev)
for K, n in cases:
    off = (np.arange(K + 1) * n).astype(np.uint64)
    r = torch.zeros(K, 64, dtype=torch.uint8, device=dev); ri = torch.zeros(K, dtype=torch.int32, device=dev)
    g = scs[:K].clone()
    eng.ecmult_multi_many_dev(r, ri, scs[:K * n], pts[:K * n], off, g); eng.sync()
    reps = 5
    t = time.perf_counter()
    for _ in range(reps): eng.ecmult_multi_many_dev(r, ri, scs[:K * n], pts[:K * n], off, g)
    eng.sync(); tm = (time.perf_counter() - t) / reps
    r1 = torch.zeros(K, 64, dtype=torch.uint8, device=dev); ri1 = torch.zeros(K, dtype=torch.int32, device=dev)
    t = time.perf_counter()
    for s in range(K): eng.ecmult_multi_dev(r1[s], ri1[s], scs[s * n:(s + 1) * n], pts[s * n:(s + 1) * n], g[s])
    eng.sync(); t1 = time.perf_counter() - t
    same = bool((r == r1).all()) and bool((ri == ri1).all())
    print("K=%6d n=%6d  many %9.3f ms (%8.2f Mpoint-scalar/s)   K single calls %9.3f ms (%7.2f M/s)   same results: %s   device events %8.3f ms"
          % (K, n, tm * 1e3, K * n / tm / 1e6, t1 * 1e3, K * n / t1 / 1e6, same, eng.last_ms(0)))
