#!/usr/bin/env python3
"""What makes the first calls after a pause slow (the bench's occasional 4-10x outliers, tools/ab_probe.py's BIP-340 line)?  2^16 BIP-340
verifications, 10 calls per group, timed per call: back to back; after 3 s of sleep (GPU and CPU idle); after 3 s of 32 busy CPU threads
(the reference's provers, GPU idle); after 3 s of busy CPU threads while the GPU keeps working."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from secp256k1_zkp_amd import Engine
from tests.refapi import Ref
ref = Ref(); rng = np.random.default_rng(5); dev = torch.device("cuda", 0)
m = 1 << 16
sigs, msgs, pks = ref.make_schnorr(m, rng, threads=32)
d = [torch.tensor(x).to(dev) for x in (sigs, msgs, pks)]; r = torch.zeros(m, dtype=torch.int32, device=dev)
e = Engine(0)
for _ in range(20): e.schnorrsig_verify_batch_dev(r, *d)
torch.cuda.synchronize()
def group(label):
    ts = []
    for _ in range(10):
        t = time.perf_counter(); e.schnorrsig_verify_batch_dev(r, *d); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print("%-46s per call ms: %s" % (label, " ".join("%.2f" % x for x in ts)), flush=True)
def busy_cpu(): ref.make_schnorr(1 << 17, np.random.default_rng(9), threads=32)
for rep in range(3):
    group("back to back")
    time.sleep(3.0); group("after 3 s asleep")
    t0 = time.time()
    while time.time() - t0 < 3.0: busy_cpu()
    group("after ~3 s of 32 busy CPU threads, GPU idle")
    stop = False
    def keep():
        while not stop: e.schnorrsig_verify_batch_dev(r, *d); torch.cuda.synchronize()
    th = threading.Thread(target=keep); th.start()
    t0 = time.time()
    while time.time() - t0 < 3.0: busy_cpu()
    stop = True; th.join()
    group("after ~3 s of busy CPU threads, GPU kept busy")
