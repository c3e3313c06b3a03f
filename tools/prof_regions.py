#!/usr/bin/env python3
"""Region attribution of k_rp_rings (diagnostic): build engine.hip with -DS2K_PROF into a side library, run one 2^14 batch,
print the share of wave-cycles spent in each region of a ring step.
    python tools/prof_regions.py            (on the GPU box)
"""
import ctypes, os, subprocess, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "secp256k1_zkp_amd", "libsecp256k1_zkp_amd_prof.so")
if not os.path.exists(lib):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-DS2K_PROF",
                           "-fvisibility=hidden", "-o", lib, os.path.join(ROOT, "secp256k1_zkp_amd", "csrc", "engine.hip")])
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
names = ["step prologue (key load, next key)", "ecmult: split + digits", "ecmult: table build", "ecmult: main loop (rest: first operand, general loop)", "to-affine (inversion)",
         "hash + bookkeeping", "main loop: 4 lean doublings", "main loop: lean addition + operand decode/locate"]
tot = v[:8].sum()
out = {names[i]: {"wave_cycles": v[i], "share": v[i] / tot, "cycles_per_wave_step": v[i] / (n * 32 * 4 / 64)} for i in range(8)}
out["steps"] = n * 32 * 4 / 64
print(json.dumps(out, indent=1))
