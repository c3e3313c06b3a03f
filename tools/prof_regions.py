#!/usr/bin/env python3
"""Region attribution of k_rp_rings (diagnostic): build the library with -DS2K_PROF into a side library, run one 2^14 batch,
print the share of wave-cycles spent in each region of a ring step.
    python tools/prof_regions.py            (on the GPU box)
"""
import ctypes, os, subprocess, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.environ.get("S2K_LIB") or os.path.join(ROOT, "secp256k1_zkp_amd", "libsecp256k1_zkp_amd_prof.so")   # S2K_LIB: a -DS2K_PROF build made elsewhere
if not os.path.exists(lib):
    subprocess.check_call([sys.executable, "-m", "secp256k1_zkp_amd.build_lib", "-o", lib, "-DS2K_PROF"], cwd=ROOT)
os.environ["S2K_LIB"] = lib
import torch
from secp256k1_zkp_amd import Engine, _native
from tests.refapi import Ref
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 14
ref = Ref(); rng = np.random.default_rng(1)
commits, proofs, gens, _ = ref.make_rangeproofs(n, rng, min_bits=64, threads=32)
eng = Engine(0)
L = _native.load()
pr = L.s2k_prof_read; pr.argtypes = [ctypes.c_void_p]; pr.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 16)()
res, mn, mx = eng.rangeproof_verify_batch(commits, proofs, gens)
assert res.all()
pr(buf)
res, mn, mx = eng.rangeproof_verify_batch(commits, proofs, gens)
pr(buf)
v = np.array(list(buf), dtype=np.float64)
names = ["step prologue (key load, next key, T update)", "ecmult: split + digits", "ecmult_lane only: table build", "ecmult_lane only: main loop rest", "to-affine (inversion)",
         "hash + bookkeeping", "main loop: 4 lean doublings", "main loop: lean addition + operand decode/locate", "split: 2 x co-Z table construction",
         "split: 2 x table rescale", "split: generator additions + leaving the isomorphic curve", "ring: 2^64 * key chain (64 doublings)"]
if os.environ.get("S2K_GEN_CACHE", "") != "0":        # the shared-generator form of the kernel (rp_ring_shared / ecmult_ring_step) uses the slots like this
    names = ["step prologue (scalars, f_j)", "ring step: split + digits", "-", "-", "to-affine (inversion)", "hash + bookkeeping", "ring step: 5 lean doublings",
             "ring step: lean additions + operand decode/locate (variable point)", "ring: two 16-entry tables (construction + rescale)", "-",
             "ring step: G and H table additions + leaving the isomorphic curve", "ring: 2^64 * C chain (64 doublings)"]
tot = v[:12].sum()
steps = n * 32 * 4 / 64
out = {("%d: " % i) + names[i]: {"wave_cycles": v[i], "share": round(v[i] / tot, 4), "cycles_per_wave_step": round(v[i] / steps)} for i in range(12)}
out["steps"] = steps
print(json.dumps(out, indent=1))
