#!/usr/bin/env python3
"""Probe: does a second engine on the same GPU (own streams, own table arena, shared device tables) raise the resident rangeproof
throughput?  K `_dev` calls queued on each engine, all waited for once, against the same 2K calls on one engine.
    python tools/two_engines_probe.py [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from secp256k1_zkp_amd import Engine
from tests.refapi import Ref
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = 1 << 14
ref = Ref(); rng = np.random.default_rng(1)
commits, proofs, gens, _ = ref.make_rangeproofs(n, rng, min_bits=64, threads=32)
dev = torch.device("cuda", 0)
pdata, poff = Engine.pack(proofs)
d_c = torch.tensor(commits).to(dev); d_g = torch.tensor(np.ascontiguousarray(gens)).to(dev)
d_p = torch.tensor(np.concatenate([pdata, np.zeros(64, np.uint8)])).to(dev); d_o = torch.tensor(poff.astype(np.int64)).to(dev)
engs = [Engine(0), Engine(0)]
outs = [(torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)) for _ in engs]
torch.cuda.synchronize()
for e in engs:
    e.set_option(Engine.OPT_RP_INPUTS_READY, 1)
def run(which, k):
    for e in which:
        for _ in range(2):
            e.rangeproof_verify_batch_dev(*outs[engs.index(e)], d_c, d_p, d_o, d_g, n)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        for e in which:
            e.rangeproof_verify_batch_dev(*outs[engs.index(e)], d_c, d_p, d_o, d_g, n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    assert all(bool(outs[engs.index(e)][0].all().item()) for e in which)
    return n * k * len(which) / dt
for rep in range(2):
    print("one engine : %.0f verifies/s" % run(engs[:1], 2 * K))
    print("two engines: %.0f verifies/s" % run(engs, K))
