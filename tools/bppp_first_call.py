#!/usr/bin/env python3
"""BP++ norm-argument batches (BASELINE config 4 shape; also 2^14 and 2^16 proofs, where the generator terms are throughput): the first call
with a NEW generator set (it builds the set's fixed-base tables) and the steady state, inputs resident.  S2K_LIB selects the library (A/B)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from secp256k1_zkp_amd import Engine
from tests.refapi import Ref
ref = Ref(); eng = Engine(0); rng = np.random.default_rng(5); dev = torch.device("cuda:0")
base = ref.make_bppp(64, rng, 64, 8)
want = ref.bppp_verify_many(*base)
assert want.all()
gens = np.ascontiguousarray(base[3])
for n in [int(x) for x in sys.argv[1:]] or [1 << 12, 1 << 14, 1 << 16]:
    reps = n // 64
    tile = lambda a: torch.tensor(np.ascontiguousarray(np.concatenate([a] * reps))).to(dev)
    d_pr, d_tr, d_rho, d_cv, d_cm = tile(base[0]), tile(base[1]), tile(base[2]), tile(base[5]), tile(base[6])
    d_gens = torch.tensor(gens).to(dev); res = torch.zeros(n, dtype=torch.int32, device=dev)
    call = lambda: eng.bppp_norm_product_verify_batch_dev(res, d_pr, base[0].shape[1], d_tr, d_rho, d_gens, gens, base[4], d_cv, base[5].shape[1], d_cm, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); call(); torch.cuda.synchronize(); first = time.perf_counter() - t0
    assert res.cpu().numpy().all()
    t0 = time.perf_counter()
    for _ in range(10): call()
    torch.cuda.synchronize(); steady = (time.perf_counter() - t0) / 10
    print("n=%6d  first call %8.2f ms   steady %7.3f ms (%.2f M verifies/s)" % (n, first * 1e3, steady * 1e3, n / steady / 1e6))
