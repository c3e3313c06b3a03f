"""GPU: a plain C program (examples/rangeproof_verify.c, gcc -std=c99) linked against the shared library verifies the reference's
fixed rangeproof vectors through secp256k1_rangeproof_verify_amd (the reference's argument list) and the batch entry point."""
import json
import os
import subprocess

import pytest

from tests.refapi import GENERATOR_H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_on_fixed_vectors(tmp_path):
    exe = str(tmp_path / "rp_verify")
    libdir = os.path.join(ROOT, "secp256k1_zkp_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "rangeproof_verify.c"), "-o", exe,
                    os.path.join(libdir, "libsecp256k1_zkp_amd.so"), "-Wl,-rpath," + libdir], check=True)
    (tmp_path / "gen").write_bytes(GENERATOR_H)
    for v in json.load(open(os.path.join(ROOT, "tests", "golden", "rangeproof_vectors.json")))["vectors"][:3]:
        (tmp_path / "commit").write_bytes(bytes.fromhex(v["commit33"])); (tmp_path / "proof").write_bytes(bytes.fromhex(v["proof"]))
        out = subprocess.run([exe, str(tmp_path / "commit"), str(tmp_path / "proof"), str(tmp_path / "gen")], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        lines = out.stdout.split("\n")
        assert lines[0] == "%d %s %s" % (v["result"], v["min_value"], v["max_value"]), v["name"]
        assert lines[1] == lines[0] and lines[2] == "0", v["name"]


def test_c_program_many_sums(tmp_path):
    """examples/ecmult_multi_many.c: K sums in one call against K single calls, from plain C (ragged: one empty sum, one of double length)"""
    exe = str(tmp_path / "mm")
    libdir = os.path.join(ROOT, "secp256k1_zkp_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "ecmult_multi_many.c"), "-o", exe,
                    os.path.join(libdir, "libsecp256k1_zkp_amd.so"), "-Wl,-rpath," + libdir], check=True)
    for args in (["8", "200"], ["3", "1"], ["40", "33"]):
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and out.stdout.startswith("OK %s sums" % args[0]), (out.stdout, out.stderr)
