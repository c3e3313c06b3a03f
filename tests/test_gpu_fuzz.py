"""GPU tier: a bounded run of the differential fuzzer (tests/tools/fuzz_parity.py) as a test -- mixed-shape valid proofs with header /
length / scalar-at-the-group-order / trailing-byte mutations, random signatures and keys, broken surjection and BP++ proofs,
colliding MSM points, rewinds with wrong nonces, tallies, half-aggregates -- every verdict (and min/max, blind, value, message)
against the unmodified reference.  The tally of the run is written to gpurun_out/fuzz_tally.txt (copied to profiles/ per round)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_bounded_differential_fuzz(engine, ref, seed):
    tool = os.path.join(ROOT, "tests", "tools", "fuzz_parity.py")
    out = ""
    for extra in ([], ["more"]):
        r = subprocess.run([sys.executable, tool, str(seed), "768"] + extra, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        out += r.stdout
    lines = [l for l in out.splitlines() if "mismatches" in l]
    assert len(lines) >= 11 and any("crafted" in l for l in lines) and any("many sums" in l for l in lines), out
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "fuzz_tally.txt"), "a") as f:
        f.write("seed %d\n" % seed + "\n".join(lines) + "\n")
    for l in lines:
        m = re.search(r"mismatches: (\d+|\[\])", l)
        assert m and m.group(1) in ("0", "[]"), l
