#!/usr/bin/env python3
"""MSM time against the number of terms (device-resident inputs), to spot cliffs at the algorithm switch points."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from secp256k1_zkp_amd import Engine, parallel
from tests.refapi import G_XY
eng = Engine(0); dev = torch.device("cuda:0"); rng = np.random.default_rng(3)
nmax = 1 << 22
ks = torch.tensor(rng.integers(0, 256, (nmax, 32), dtype=np.uint8)).to(dev)
gpts = torch.tensor(np.frombuffer(G_XY, np.uint8).copy()).to(dev).repeat(nmax, 1)
pts = torch.zeros(nmax, 64, dtype=torch.uint8, device=dev); pinf = torch.zeros(nmax, dtype=torch.int32, device=dev); z = torch.zeros(nmax, 32, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
eng.ecmult_batch_dev(pts, pinf, gpts, z, ks); torch.cuda.synchronize()
scs = torch.tensor(rng.integers(0, 256, (nmax, 32), dtype=np.uint8)).to(dev)
be = parallel.EngineBackend(eng)
for n in (16, 31, 32, 64, 128, 256, 512, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 1 << 22):
    parallel.msm_sharded(be, scs[:n], pts[:n]); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): parallel.msm_sharded(be, scs[:n], pts[:n])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print("n=%8d  %8.3f ms  %8.2f Mterm/s  fallback=%d" % (n, dt * 1e3, n / dt / 1e6, eng.last_msm_fallback()))
