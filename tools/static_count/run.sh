#!/bin/bash
# Static instruction accounting of k_rp_rings: compile tools/static_count/rings_tu.hip to assembly, list every loop body.
#   tools/static_count/run.sh [extra hipcc flags]        -> /tmp/dis/rings_tu.s
set -e
cd "$(dirname "$0")"
mkdir -p /tmp/dis
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -I../../secp256k1_zkp_amd/csrc "$@" rings_tu.hip -o /tmp/dis/rings_tu.s -Wno-unused-value 2>&1 | grep -E "error" || true
python3 loops.py /tmp/dis/rings_tu.s
