#!/usr/bin/env python3
"""Where do the ~50 ms holes after multi-threaded CPU legs sit (tools/r6_idle_probe.py found one in 120 calls)?  Per call: host time until
the launch returns, host time until the wait returns, the device's own elapsed time (the engine's events); the wait is either
torch.cuda.synchronize() (blocking) or a poll of an event (never sleeps).  Prints every call above 5 ms and the tallies."""
import os, sys, time
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
tally = {}
for mode in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("block", "poll")):
    n_calls = n_slow = 0
    for rep in range(reps):
        t0 = time.time()
        while time.time() - t0 < 0.4: ref.make_schnorr(1 << 16, np.random.default_rng(9), threads=32)
        for k in range(30):
            t = time.perf_counter(); e.schnorrsig_verify_batch_dev(r, *d); tl = time.perf_counter()
            if mode == "block": torch.cuda.synchronize()
            elif mode == "engine": e.sync()                          # the engine's own wait on its stream (s2k_engine_sync)
            else:
                ev = torch.cuda.Event(); ev.record()
                while not ev.query(): pass
            tw = time.perf_counter()
            n_calls += 1
            if (tw - t) > 5e-3:
                n_slow += 1
                print("%s rep %d call %d: launch %.2f ms, wait %.2f ms, device events %.2f ms" % (mode, rep, k, (tl - t) * 1e3, (tw - tl) * 1e3, e.last_ms(0)), flush=True)
    tally[mode] = (n_calls, n_slow)
print("calls / calls above 5 ms:", tally)
