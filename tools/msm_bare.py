#!/usr/bin/env python3
"""MSM through the engine's device entry point alone (s2k_ecmult_multi_dev, inputs resident): K calls queued back to back and waited for
once (throughput), and single calls with a wait each (latency).  python tools/msm_bare.py [sizes...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from secp256k1_zkp_amd import Engine
from tests.refapi import G_XY
sizes = [int(x) for x in sys.argv[1:]] or [64, 1024, 16384, 65536, 1 << 18, 1 << 20]
eng = Engine(0); dev = torch.device("cuda:0"); rng = np.random.default_rng(3)
nmax = max(sizes)
ks = torch.tensor(rng.integers(0, 256, (nmax, 32), dtype=np.uint8)).to(dev)
gpts = torch.tensor(np.frombuffer(G_XY, np.uint8).copy()).to(dev).repeat(nmax, 1)
pts = torch.zeros(nmax, 64, dtype=torch.uint8, device=dev); pinf = torch.zeros(nmax, dtype=torch.int32, device=dev); z = torch.zeros(nmax, 32, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
eng.ecmult_batch_dev(pts, pinf, gpts, z, ks); eng.sync()
scs = torch.tensor(rng.integers(0, 256, (nmax, 32), dtype=np.uint8)).to(dev)
r = torch.zeros(64, dtype=torch.uint8, device=dev); ri = torch.zeros(1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for n in sizes:
    eng.ecmult_multi_dev(r, ri, scs[:n], pts[:n]); eng.sync()
    K = 8
    t = time.perf_counter()
    for _ in range(K): eng.ecmult_multi_dev(r, ri, scs[:n], pts[:n])
    eng.sync(); tq = (time.perf_counter() - t) / K
    t = time.perf_counter()
    for _ in range(K):
        eng.ecmult_multi_dev(r, ri, scs[:n], pts[:n]); eng.sync()
    tl = (time.perf_counter() - t) / K
    print("n=%8d  queued %8.3f ms (%8.2f Mterm/s)   single call + wait %8.3f ms   device events %8.3f ms   result %s.. inf %d" % (n, tq * 1e3, n / tq / 1e6, tl * 1e3, eng.last_ms(0), bytes(r.cpu().numpy()[:6]).hex(), int(ri.item())))
