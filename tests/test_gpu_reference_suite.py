"""GPU tier: the reference's OWN test program as a differential driver of the engine (tests/integration/route_modules.h, route_api.h).

oracle/_ref/ref_tests_routed is the reference's src/tests.c compiled from the sources where they lie, with every call of the two static
seams of the hot path (secp256k1_ecmult, secp256k1_ecmult_multi_var -- made by the modules and by the tests) and every call of a public
verifier on the path (rangeproof verify / rewind, BIP-340 verify, half-aggregate verify, Pedersen tally, surjection proof, the BP++ norm
argument) followed by the engine's form of the same call on the same operands; a difference aborts the program.  The reference's tests
bring what a hand-written parity test does not think of: proofs of every shape its signer can make (exponents, min_bits, min_value,
messages, extra_commit), its corrupted and truncated proofs, its fixed vectors, test_ecmult_multi's infinities / zero scalars / cancelling
terms / every batching size, the GLV edge scalars, MuSig key aggregation, whitelist and adaptor signatures running on secp256k1_ecmult.
The program's exit status is the verdict: 0 = the reference's own assertions hold AND the engine agreed on every checked call."""
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_tests_routed")

pytestmark = pytest.mark.gpu
# default tier: ALL 16 iterations of the test program for both selections (the two processes run side by side: ~4 minutes of mostly host
# time); S2K_TEST_SHORT=1 (builder's quick runs): half / a quarter of them
SHORT = os.environ.get("S2K_TEST_SHORT") == "1"
SELECTIONS = {
    "protocol": (["-i=8" if SHORT else "-i=16", "-t=rangeproof", "-t=generator", "-t=surjection", "-t=schnorrsig", "-t=schnorrsig_halfagg", "-t=bppp", "-t=musig", "-t=whitelist"],
                 {"S2K_RT_ECMULT_EVERY": "16"}),
    "ecmult": (["-i=4" if SHORT else "-i=16", "-t=ecmult"], {"S2K_RT_ECMULT_EVERY": "16"}),
}


def _parse(stderr):
    rep = {}
    for m in re.finditer(r"s2k-route: (\S+)\s+calls\s+(\d+)\s+checked\s+(\d+)\s+reference-accepted\s+(\d+)", stderr):
        rep[m.group(1)] = tuple(int(m.group(i)) for i in (2, 3, 4))
    return rep


@pytest.fixture(scope="module")
def runs(request):
    """Both selections of the routed test program, started together (each process has its own engine; the host side -- the reference's
    own tests -- is most of their time) and waited for once."""
    from tests.conftest import _gpu_tier
    if not os.path.exists(BIN):
        msg = "oracle/_ref/ref_tests_routed not built (make -C oracle routed, needs the reference tree)"
        if _gpu_tier(request.config):
            pytest.fail("-m gpu needs the routed reference test program: " + msg)
        pytest.skip(msg)
    procs = {}
    for name, (args, env) in SELECTIONS.items():
        e = dict(os.environ); e.update(env)
        procs[name] = subprocess.Popen([BIN] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)
    out = {}
    for name, p in procs.items():
        try:
            so, se = p.communicate(timeout=1500)
        except subprocess.TimeoutExpired:
            p.kill(); so, se = p.communicate()
            se += "\n(timed out)"
        out[name] = (p.returncode, so, se, _parse(se))
    return out


def test_protocol_modules_of_the_reference_suite(runs):
    """rangeproof, generator (Pedersen), surjection, schnorrsig, half-aggregate, BP++, MuSig, whitelist: the modules whose verifiers sit on
    the path.  Every public verifier call and every multi-scalar multiplication is checked, every 16th double multiplication."""
    rc, so, se, rep = runs["protocol"]
    assert rc == 0, so[-2000:] + se[-4000:]
    assert "engine on" in se
    # (floors well under what 8 iterations give: ~5 000 / 150 / 270 / 45 / 283 / 80 / 41 / 225 / 25 000)
    for name, floor in (("rangeproof_verify", 1000), ("rangeproof_rewind", 50), ("schnorrsig_verify", 100), ("pedersen_verify_tally", 15),
                        ("surjectionproof_verify", 50), ("schnorrsig_aggverify", 30), ("bppp_norm_product_verify", 20), ("ecmult_multi_var", 100), ("ecmult", 1000)):
        calls, checked, accepted = rep[name]
        assert checked >= floor, (name, rep[name])
    # both verdicts were seen: the reference's tests corrupt what they sign
    for name in ("rangeproof_verify", "schnorrsig_verify", "surjectionproof_verify", "schnorrsig_aggverify", "bppp_norm_product_verify", "pedersen_verify_tally"):
        calls, checked, accepted = rep[name]
        assert 0 < accepted < checked, (name, rep[name])
    print("\n" + "\n".join(l for l in se.splitlines() if l.startswith("s2k-route")))


def test_ecmult_module_of_the_reference_suite(runs):
    """The reference's ecmult tests (run_ecmult_chain, run_ecmult_constants, run_ecmult_near_split_bound, test_ecmult_multi over both
    algorithms and every batching size, ...): every multi-scalar multiplication checked (~85 000 at the default 16 iterations, ~19 000 with
    S2K_TEST_SHORT=1), double multiplications sampled."""
    rc, so, se, rep = runs["ecmult"]
    assert rc == 0, so[-2000:] + se[-4000:]
    calls, checked, accepted = rep["ecmult_multi_var"]
    assert checked >= 5000 and checked == accepted, rep          # ~18 900 at 4 iterations, ~85 400 at 16
    assert rep["ecmult"][1] >= 1000, rep                         # ~2 150 / ~8 700
    print("\n" + "\n".join(l for l in se.splitlines() if l.startswith("s2k-route")))
