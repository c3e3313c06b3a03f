# the reference's own test program as a differential driver, beyond what the test tier runs: 64 iterations of the protocol modules and the WHOLE
# program (every module) at its default 16 -- exit status 0 = the reference's assertions hold AND the engine agreed on every checked call
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/${RTAG:-r06r}
B=oracle/_ref/ref_tests_routed
(S2K_RT_ECMULT_EVERY=16 timeout 1500 $B -i=64 -t=rangeproof -t=generator -t=surjection -t=schnorrsig -t=schnorrsig_halfagg -t=bppp -t=musig -t=whitelist > gpurun_out/${RTAG:-r06r}/protocol_64.out 2> gpurun_out/${RTAG:-r06r}/protocol_64.err; echo "rc=$?" >> gpurun_out/${RTAG:-r06r}/protocol_64.err) &
(S2K_RT_ECMULT_EVERY=16 timeout 1500 $B > gpurun_out/${RTAG:-r06r}/whole.out 2> gpurun_out/${RTAG:-r06r}/whole.err; echo "rc=$?" >> gpurun_out/${RTAG:-r06r}/whole.err) &
wait
for f in protocol_64 whole; do echo "## $f"; tail -3 gpurun_out/${RTAG:-r06r}/$f.out; grep "s2k-route\|rc=" gpurun_out/${RTAG:-r06r}/$f.err; done > gpurun_out/${RTAG:-r06r}/summary.txt
cat gpurun_out/${RTAG:-r06r}/summary.txt
