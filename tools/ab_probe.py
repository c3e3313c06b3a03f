#!/usr/bin/env python3
"""A/B of two builds of the library on ONE box: resident 64-bit rangeproof throughput (the bench's headline loop) and 2^16 BIP-340, alternating
between the libraries (each in its own subprocess: S2K_LIB).    python tools/ab_probe.py libA.so libB.so [libC.so ...] [rounds]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from secp256k1_zkp_amd import Engine
from tests.refapi import Ref
n = 1 << 14; K = 12
ref = Ref(); rng = np.random.default_rng(1)
commits, proofs, gens, _ = ref.make_rangeproofs(n, rng, min_bits=64, threads=32)
dev = torch.device("cuda", 0)
pdata, poff = Engine.pack(proofs)
d_c = torch.tensor(commits).to(dev); d_g = torch.tensor(np.ascontiguousarray(gens)).to(dev)
d_p = torch.tensor(np.concatenate([pdata, np.zeros(64, np.uint8)])).to(dev); d_o = torch.tensor(poff.astype(np.int64)).to(dev)
e = Engine(0)
o = (torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(n, dtype=torch.int64, device=dev))
torch.cuda.synchronize()
e.set_option(Engine.OPT_RP_INPUTS_READY, 1)
for _ in range(3): e.rangeproof_verify_batch_dev(*o, d_c, d_p, d_o, d_g, n)
torch.cuda.synchronize()
res = []
for rep in range(3):
    t = time.perf_counter()
    for _ in range(K): e.rangeproof_verify_batch_dev(*o, d_c, d_p, d_o, d_g, n)
    torch.cuda.synchronize()
    res.append(n * K / (time.perf_counter() - t))
assert bool(o[0].all().item())
kern = [e.last_ms(16 + k) for k in range(8)]
# every proof its own generator (general form of the rings kernel)
gens2 = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(256)])[rng.integers(0, 256, n)]
e.set_option(Engine.OPT_GEN_CACHE_SLOTS, 1)          # (only secp256k1_generator_h keeps a table: these 256 generators stay uncached)
c2, p2, g2, _ = ref.make_rangeproofs(n, rng, min_bits=64, gens64=gens2, threads=32)
pd2, po2 = Engine.pack(p2)
d_c2 = torch.tensor(c2).to(dev); d_g2 = torch.tensor(np.ascontiguousarray(g2)).to(dev)
d_p2 = torch.tensor(np.concatenate([pd2, np.zeros(64, np.uint8)])).to(dev); d_o2 = torch.tensor(po2.astype(np.int64)).to(dev)
torch.cuda.synchronize()
for _ in range(2): e.rangeproof_verify_batch_dev(*o, d_c2, d_p2, d_o2, d_g2, n)
torch.cuda.synchronize()
res2 = []
for rep in range(2):
    t = time.perf_counter()
    for _ in range(K): e.rangeproof_verify_batch_dev(*o, d_c2, d_p2, d_o2, d_g2, n)
    torch.cuda.synchronize()
    res2.append(n * K / (time.perf_counter() - t))
assert bool(o[0].all().item())
m = 1 << 16
sigs, msgs, pks = ref.make_schnorr(m, rng, threads=32)
d = [torch.tensor(x).to(dev) for x in (sigs, msgs, pks)]; r = torch.zeros(m, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
e.schnorrsig_verify_batch_dev(r, d[0], d[1], d[2]); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): e.schnorrsig_verify_batch_dev(r, d[0], d[1], d[2])
torch.cuda.synchronize()
sch = m * 10 / (time.perf_counter() - t)
assert bool(r.all().item())
print("RESULT rp %%s  kernel_ms %%.3f  distinct %%s  bip340 %%.3e" %% (" ".join("%%.0f" %% x for x in res), float(np.mean(kern)), " ".join("%%.0f" %% x for x in res2), sch))
''' % ROOT
libs = [a for a in sys.argv[1:] if a.endswith(".so")]; rounds = ([int(a) for a in sys.argv[1:] if a.isdigit()] or [2])[0]
for rd in range(rounds):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, S2K_LIB=os.path.abspath(lib)), capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        print(os.path.basename(lib), line[0] if line else ("FAILED: " + out.stderr[-800:]), flush=True)
