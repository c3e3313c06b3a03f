#!/usr/bin/env python3
"""Diagnostic: wall time of the parts of the shared-generator rings kernel, by launching it with parts switched off ($S2K_RP_DEBUG bits:
1 = no 2^64 chain, 2 = no table construction, 4 = no steps; results are meaningless in those launches).  The knob only exists in a
-DS2K_DIAG build of the library (the product library ignores the environment): build one and point S2K_LIB at it, on the GPU box:
    python -m secp256k1_zkp_amd.build_lib -o /tmp/libs2k_diag.so -DS2K_DIAG
    S2K_LIB=/tmp/libs2k_diag.so python tools/rings_parts.py [n]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from secp256k1_zkp_amd import Engine
from tests.refapi import Ref
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 14
ref = Ref(); rng = np.random.default_rng(1)
commits, proofs, gens, _ = ref.make_rangeproofs(n, rng, min_bits=64, threads=32)
dev = torch.device("cuda", 0)
pdata, poff = Engine.pack(proofs)
d_c = torch.tensor(commits).to(dev); d_g = torch.tensor(np.ascontiguousarray(gens)).to(dev)
d_p = torch.tensor(np.concatenate([pdata, np.zeros(64, np.uint8)])).to(dev); d_o = torch.tensor(poff.astype(np.int64)).to(dev)
d_res = torch.zeros(n, dtype=torch.int32, device=dev); d_mn = torch.zeros(n, dtype=torch.int64, device=dev); d_mx = torch.zeros(n, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
out = {}
for mode in (0, 1, 2, 3, 4, 5, 6, 7):
    os.environ["S2K_RP_DEBUG"] = str(mode)
    eng = Engine(0)
    for _ in range(4):
        eng.rangeproof_verify_batch_dev(d_res, d_mn, d_mx, d_c, d_p, d_o, d_g, n)
    eng.sync()
    ms = [eng.last_ms(16 + k) for k in range(3)]
    out["mode %d (%s)" % (mode, ", ".join(x for b, x in ((1, "no chain"), (2, "no tables"), (4, "no steps")) if mode & b) or "everything")] = round(float(np.mean(ms)), 3)
    if mode == 0:
        assert bool(d_res.all().item())
    eng.close()
print(json.dumps(out, indent=1))
