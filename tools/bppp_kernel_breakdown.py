# Per-kernel breakdown of the BP++ norm-argument batch (2^12 proofs, g_len 64, h_len 8: BASELINE config 4) under rocprofv3:
#   rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python tools/bppp_kernel_breakdown.py
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from secp256k1_zkp_amd import Engine
from tests.refapi import Ref
ref = Ref(); eng = Engine(0); rng = np.random.default_rng(5)
n = 1 << 12
base = ref.make_bppp(64, rng, 64, 8)
reps = n // 64
args = [np.concatenate([base[0]] * reps), np.concatenate([base[1]] * reps), np.concatenate([base[2]] * reps), base[3], base[4],
        np.concatenate([base[5]] * reps), np.concatenate([base[6]] * reps)]
for _ in range(5):
    assert eng.bppp_norm_product_verify_batch(*args).all()
