# (needs a -DS2K_DIAG build: python -m secp256k1_zkp_amd.build_lib -o /tmp/libs2k_diag.so -DS2K_DIAG; export S2K_LIB=/tmp/libs2k_diag.so)
# MSM time through the device entry point against the window width (S2K_MSM_C overrides msm_make_plan's choice): bash tools/msm_c_sweep.sh
# -> what msm.h's plan was picked from (profiles/r03c_msm_c_sweep.txt)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for c in 4 6 7 8 9 10 12; do echo "c=$c"; S2K_MSM_C=$c python tools/msm_bare.py 64 256 1024 2>/dev/null | cut -c1-60,100-140; done
for c in 6 8 10 11 12 13; do echo "c=$c"; S2K_MSM_C=$c python tools/msm_bare.py 4096 16384 65536 2>/dev/null | cut -c1-60,100-140; done
for c in 10 11 12 13; do echo "c=$c"; S2K_MSM_C=$c python tools/msm_bare.py 262144 1048576 2>/dev/null | cut -c1-60,100-140; done
