"""CPU tier: the routed build of the reference's test program (oracle/Makefile: routed; tests/integration/route_*.h) still compiles against
the reference tree and still intercepts every call it is meant to -- run in its count-only mode (S2K_RT_OFF=1: no engine is created, so this
says nothing about parity; tests/test_gpu_reference_suite.py is the parity run)."""
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_tests_routed")
REF = os.environ.get("S2K_REFERENCE", "/root/reference")


def test_routed_reference_tests_intercept_the_hot_path_calls():
    if os.path.isdir(os.path.join(REF, "src")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "routed"], check=True, timeout=600)
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/ref_tests_routed not built and no reference tree to build it from")
    env = dict(os.environ); env["S2K_RT_OFF"] = "1"
    r = subprocess.run([BIN, "-t=bppp", "-t=schnorrsig_halfagg", "-t=surjection", "-t=generator", "-t=schnorrsig"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    assert "engine OFF (count only)" in r.stderr
    rep = {m.group(1): int(m.group(2)) for m in re.finditer(r"s2k-route: (\S+)\s+calls\s+(\d+)", r.stderr)}
    for name in ("ecmult", "ecmult_multi_var", "schnorrsig_verify", "pedersen_verify_tally", "surjectionproof_verify", "schnorrsig_aggverify", "bppp_norm_product_verify"):
        assert rep.get(name, 0) > 0, (name, rep)
