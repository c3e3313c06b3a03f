cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05F
(timeout 1300 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r05F/gpu_tests_tail.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/r05F/smoke.txt
OUT=gpurun_out/r05F/fuzz_tally.txt; : > $OUT
for spec in "811 26" "812 26" "813 24" "814 20"; do
  set -- $spec
  for extra in "" more; do
    echo "seed $1 $extra, $2-bit tables" >> $OUT
    S2K_GTAB_BITS=$2 timeout 600 python tests/tools/fuzz_parity.py $1 3000 $extra 2>&1 | grep -i "mismatch\|error\|Traceback" >> $OUT
  done
done
echo "tallies: $(grep -c mismatches $OUT); with a mismatch: $(grep mismatches $OUT | grep -v 'mismatches: 0\|mismatches: \[\]' | wc -l)" | tee -a $OUT
cat gpurun_out/r05F/gpu_tests_tail.txt gpurun_out/r05F/smoke.txt
