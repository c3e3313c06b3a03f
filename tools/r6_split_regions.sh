#!/bin/bash
# region attribution of the general form of the rings kernel (S2K_GEN_CACHE=0: no generator tables, every ring through k_rp_rings) with the
# table construction in the interleaved parking area (new) and in place (old): -DS2K_PROF side libraries built in the container
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in new old; do
  S2K_GEN_CACHE=0 S2K_LIB=$PWD/tools/ab_libs/lib_prof_$v.so timeout 600 python tools/prof_regions.py > gpurun_out/${1:-r06ab}_general_regions_$v.json 2> gpurun_out/${1:-r06ab}_general_regions_$v.err
  tail -2 gpurun_out/${1:-r06ab}_general_regions_$v.err
done
python - <<'P'
import json,sys
for v in ("new","old"):
    j=json.load(open("gpurun_out/%s_general_regions_%s.json" % (sys.argv[1] if len(sys.argv)>1 else "r06ab", v)))
    print(v, {k.split(":")[0]: (x["share"], x["cycles_per_wave_step"]) for k,x in j.items() if isinstance(x,dict) and x["share"]>0})
P
