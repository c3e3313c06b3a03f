#!/usr/bin/env python3
"""Regenerates tools/static_count/rings_tu.hip from the k_rp_rings kernel of secp256k1_zkp_amd/csrc/engine_rangeproof.hip."""
import os
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "..", "..", "secp256k1_zkp_amd", "csrc", "engine_rangeproof.hip")).read()
i = src.index("#ifndef S2K_RINGS_WAVES")
j = src.index("__global__ void __launch_bounds__(64)\nk_rp_final")
tu = os.path.join(here, "rings_tu.hip")
head = open(tu).read().split("#include <hip/hip_runtime.h>\n")[0] + "#include <hip/hip_runtime.h>\n"
open(tu, "w").write(head + src[i:j])
