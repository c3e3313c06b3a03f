# Four MSM calls of n terms under rocprofv3 (--kernel-trace --stats) give the per-kernel breakdown in profiles/r02u_msm_*: python tools/msm_kernel_breakdown.py <n>
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from secp256k1_zkp_amd import Engine, parallel
from tests.refapi import G_XY
eng = Engine(0); dev = torch.device("cuda:0"); rng = np.random.default_rng(3)
n = int(sys.argv[1])
ks = torch.tensor(rng.integers(0, 256, (n, 32), dtype=np.uint8)).to(dev)
gpts = torch.tensor(np.frombuffer(G_XY, np.uint8).copy()).to(dev).repeat(n, 1)
pts = torch.zeros(n, 64, dtype=torch.uint8, device=dev); pinf = torch.zeros(n, dtype=torch.int32, device=dev); z = torch.zeros(n, 32, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
eng.ecmult_batch_dev(pts, pinf, gpts, z, ks); torch.cuda.synchronize()
scs = torch.tensor(rng.integers(0, 256, (n, 32), dtype=np.uint8)).to(dev)
be = parallel.EngineBackend(eng)
for _ in range(4): parallel.msm_sharded(be, scs, pts)
torch.cuda.synchronize()
