#!/usr/bin/env python3
"""What one rank of an N-GPU term-sharded sum does, measured on ONE GPU: the Jacobian partial of a 1/N slice (s2k_ecmult_multi_partial_dev)
followed by the sum of N exchanged partials (s2k_gej_sum_dev); the all-gather of N x 112 bytes between them is the only part not executed.
python tools/msm_shard_probe.py [total_terms ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from secp256k1_zkp_amd import Engine
from tests.refapi import G_XY
totals = [int(x) for x in sys.argv[1:]] or [1 << 20, 1 << 24]
eng = Engine(0); dev = torch.device("cuda:0"); rng = np.random.default_rng(3)
nmax = max(totals)
ks = torch.tensor(rng.integers(0, 256, (nmax, 32), dtype=np.uint8)).to(dev)
gpts = torch.tensor(np.frombuffer(G_XY, np.uint8).copy()).to(dev).repeat(nmax, 1)
pts = torch.zeros(nmax, 64, dtype=torch.uint8, device=dev); pinf = torch.zeros(nmax, dtype=torch.int32, device=dev); z = torch.zeros(nmax, 32, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
eng.ecmult_batch_dev(pts, pinf, gpts, z, ks); eng.sync()
del gpts, z, ks
scs = torch.tensor(rng.integers(0, 256, (nmax, 32), dtype=np.uint8)).to(dev)
r = torch.zeros(64, dtype=torch.uint8, device=dev); ri = torch.zeros(1, dtype=torch.int32, device=dev)
K = 8
for total in totals:
    eng.ecmult_multi_dev(r, ri, scs[:total], pts[:total]); eng.sync()
    t = time.perf_counter()
    for _ in range(K): eng.ecmult_multi_dev(r, ri, scs[:total], pts[:total])
    eng.sync(); whole = (time.perf_counter() - t) / K
    want = bytes(r.cpu().numpy())
    for world in (2, 4, 8):
        n = total // world
        parts = torch.zeros(world, 28, dtype=torch.int32, device=dev)
        for k in range(world): eng.ecmult_multi_partial_dev(parts[k], scs[k * n:(k + 1) * n], pts[k * n:(k + 1) * n])
        eng.gej_sum_dev(r, ri, parts, world); eng.sync()
        assert bytes(r.cpu().numpy()) == want, "the shards' sum differs from the whole"
        t = time.perf_counter()
        for _ in range(K):
            eng.ecmult_multi_partial_dev(parts[0], scs[:n], pts[:n]); eng.gej_sum_dev(r, ri, parts, world)
        eng.sync(); rank = (time.perf_counter() - t) / K
        print("total %9d  whole %8.3f ms   %d ranks: slice partial + sum of %d partials %8.3f ms  (x%.2f before the all-gather of %d bytes)" % (total, whole * 1e3, world, world, rank * 1e3, whole / rank, world * 112))
