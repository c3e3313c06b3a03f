# Double-multiplication throughput against the batch size (device-resident): one wavefront per SIMD up to 65 536 items finishes in the
# latency of one multiplication (0.69-0.74 ms, 89 M/s at 2^16), two per SIMD from 131 072 on give 108-118 M/s -- so splitting one
# multiplication over two lanes to fill an under-occupied launch would not pay (measured before building it).  python tools/occupancy_probe.py
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from secp256k1_zkp_amd import Engine
from tests.refapi import G_XY
eng = Engine(0); dev = torch.device("cuda:0"); rng = np.random.default_rng(3)
nmax = 1 << 19
a = torch.tensor(np.frombuffer(G_XY, np.uint8).copy()).to(dev).repeat(nmax, 1)
na = torch.tensor(rng.integers(0, 256, (nmax, 32), dtype=np.uint8)).to(dev); ng = torch.tensor(rng.integers(0, 256, (nmax, 32), dtype=np.uint8)).to(dev)
r = torch.zeros(nmax, 64, dtype=torch.uint8, device=dev); ri = torch.zeros(nmax, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for n in (16384, 32768, 65536, 98304, 131072, 196608, 262144, 524288):
    eng.ecmult_batch_dev(r[:n], ri[:n], a[:n], na[:n], ng[:n]); eng.sync()
    ms = []
    for _ in range(5):
        eng.ecmult_batch_dev(r[:n], ri[:n], a[:n], na[:n], ng[:n]); eng.sync(); ms.append(eng.last_ms(0))
    print("n=%7d  %.3f ms  %.1f M/s" % (n, min(ms), n / min(ms) / 1e3))
