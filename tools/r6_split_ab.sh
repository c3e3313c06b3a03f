#!/bin/bash
# round 6: the general form's table construction in the wavefront-interleaved parking area + packed finished sectors (S2K_SPLIT_INTERLEAVED)
# against the in-place layout of rounds 2-5, same box, alternating (tools/ab_probe.py prints the shared-generator rate, the ring kernel's
# time, the distinct-generator rate and 2^16 BIP-340)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${1:-r06aa}_split_interleaved_ab.txt
timeout 900 python -m pytest tests/test_gpu_rangeproof.py tests/test_gpu_rangeproof_adversarial.py tests/test_gpu_split_bounds.py tests/test_gpu_split_launch.py tests/test_gpu_gen_cache.py tests/test_gpu_prims.py -x -q 2>&1 | tail -3 > $OUT
timeout 1500 python tools/ab_probe.py tools/ab_libs/lib_r6_inplace.so tools/ab_libs/lib_r6_split_nt1.so secp256k1_zkp_amd/libsecp256k1_zkp_amd.so 3 >> $OUT 2>&1
cat $OUT
