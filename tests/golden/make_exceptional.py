#!/usr/bin/env python3
"""Generates tests/golden/rangeproof_exceptional.json: a VALID one-ring Borromean rangeproof (the reference accepts it) whose verification
meets an exceptional addition -- P + P at the last generator window of a low-to-high 24-bit fixed-base evaluation of s*G + e*P
(tests/adversarial.py: Crafter.grind_exceptional_doubling).  Needs oracle/_ref (run in the build container):
    python tests/golden/make_exceptional.py [seed]"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
from tests.refapi import Ref, GENERATOR_H
from tests.adversarial import Crafter

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2024
ref = Ref(); rng = np.random.default_rng(seed)
out = Crafter(ref).grind_exceptional_doubling(rng, tries=16)
assert out is not None, "no candidate found; try another seed"
c, p, w = out
res, mn, mx = ref.rangeproof_verify_many(np.frombuffer(c, np.uint8).reshape(1, 33), [p], np.frombuffer(GENERATOR_H, np.uint8).reshape(1, 64))
assert res[0] == 1
json.dump({"about": "valid proof, reference verdict 1; s_0 of its only ring makes e*C + (s mod 2^240)*G equal to (s >> 240)*2^240*G", "seed": seed,
           "commit33": c.hex(), "proof": p.hex(), "generator": GENERATOR_H.hex(), "top_window": w, "result": 1, "min_value": int(mn[0]), "max_value": int(mx[0])},
          open(os.path.join(HERE, "rangeproof_exceptional.json"), "w"), indent=1)
print("ok", w)
