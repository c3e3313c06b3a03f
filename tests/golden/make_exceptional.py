#!/usr/bin/env python3
"""Generates tests/golden/rangeproof_exceptional.json: a VALID one-ring Borromean rangeproof (the reference accepts it) whose verification
meets an exceptional addition -- P + P at the last window of a low-to-high evaluation of s*G by signed 26-bit fixed-base digits, the
engine's generator table (tests/adversarial.py: Crafter.grind_exceptional_doubling).  One nonce candidate in 2^22 fits, so the 2^22
candidates of a ring are spread over worker processes (a full sweep is ~5 core-minutes and finds a proof with probability 1 - 1/e; the
seed is bumped until one does).  Needs oracle/_ref (run in the build container):
    python tests/golden/make_exceptional.py [seed] [workers]"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import multiprocessing as mp
import numpy as np
from tests.refapi import Ref, GENERATOR_H
from tests.adversarial import Crafter

D = 26


def work(args):
    seed, lo, hi = args
    out = Crafter(Ref()).grind_exceptional_doubling(np.random.default_rng(seed), D=D, w_lo=lo, w_hi=hi)      # (same seed: same ring in every worker)
    return None if out is None else (out[0].hex(), out[1].hex(), out[2])


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2026
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, (os.cpu_count() or 2) - 1)
    top = 1 << (256 - D * ((256 + D - 1) // D - 1))
    found = None
    while found is None:
        step = 1 << 15
        jobs = [(seed, lo, min(lo + step, top)) for lo in range(1, top, step)]
        with mp.Pool(workers) as pool:
            for r in pool.imap_unordered(work, jobs):
                if r is not None:
                    found = r; pool.terminate(); break
        if found is None:
            print("seed", seed, ": no candidate fits, next seed", flush=True); seed += 1
    c, p, w = bytes.fromhex(found[0]), bytes.fromhex(found[1]), found[2]
    ref = Ref()
    res, mn, mx = ref.rangeproof_verify_many(np.frombuffer(c, np.uint8).reshape(1, 33), [p], np.frombuffer(GENERATOR_H, np.uint8).reshape(1, 64))
    assert res[0] == 1
    json.dump({"about": "valid proof, reference verdict 1; with s_0 cut into signed 26-bit digits d_w (csrc/ecmult.h), e*C + sum_{w<9} d_w 2^(26w) G equals d_9 2^234 G",
               "seed": seed, "digit_bits": D, "commit33": c.hex(), "proof": p.hex(), "generator": GENERATOR_H.hex(), "top_digit": w, "result": 1,
               "min_value": int(mn[0]), "max_value": int(mx[0])}, open(os.path.join(HERE, "rangeproof_exceptional.json"), "w"), indent=1)
    print("ok", seed, w)
