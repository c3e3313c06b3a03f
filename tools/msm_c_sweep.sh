cd /root/repo
for c in 4 6 7 8 9 10 12; do echo "c=$c"; S2K_MSM_C=$c python tools/msm_bare.py 64 256 1024 2>/dev/null | cut -c1-60,100-140; done
for c in 6 8 10 11 12 13; do echo "c=$c"; S2K_MSM_C=$c python tools/msm_bare.py 4096 16384 65536 2>/dev/null | cut -c1-60,100-140; done
for c in 10 11 12 13; do echo "c=$c"; S2K_MSM_C=$c python tools/msm_bare.py 262144 1048576 2>/dev/null | cut -c1-60,100-140; done
