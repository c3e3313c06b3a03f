#!/usr/bin/env python3
"""Extract the reference's own known-answer vectors for the hot path into small JSON fixtures.

Run in the build container (needs /root/reference):   python tests/golden/make_golden.py
Only *test data* (byte arrays and the expected results the reference's tests assert) is extracted; the fixtures are
what the GPU box uses, since /root/reference does not exist there.  Sources:
  rangeproof : src/modules/rangeproof/tests_impl.h:589-812 (3 fixed proofs) and :883-1349 (3 reproducible proofs,
               incl. the maximal 64-bit / exp=18 / 5126-byte one)
  bppp       : src/modules/bppp/test_vectors/verify.h (13 norm-argument accept/reject vectors)
  schnorrsig : src/modules/schnorrsig/tests_impl.h:208-807 (the BIP-340 vectors)
"""
import json
import os
import re
import sys

REF = os.environ.get("S2K_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def c_arrays(text):
    """name -> bytes for every `unsigned char NAME[...] = { 0x.., ... };` in text (in order of appearance, duplicates kept as list)."""
    out = []
    for m in re.finditer(r"unsigned char\s+(\w+)\s*\[[^\]]*\]\s*=\s*\{([^;]*?)\}\s*;", text, re.S):
        body = re.sub(r"/\*.*?\*/", "", m.group(2), flags=re.S)
        vals = re.findall(r"0[xX][0-9a-fA-F]{1,2}|\b\d+\b", body)
        out.append((m.group(1), bytes(int(v, 0) for v in vals), m.start()))
    return out


def rangeproof():
    path = os.path.join(REF, "src/modules/rangeproof/tests_impl.h")
    text = open(path).read()
    a = text.index("static void test_rangeproof_fixed_vectors(void)")
    b = text.index("static void print_vector_helper")
    c = text.index("static void test_rangeproof_fixed_vectors_reproducible(void)")
    fixed = {n: v for n, v, _ in c_arrays(text[a:b])}
    repro = {n: v for n, v, _ in c_arrays(text[c:])}
    U64 = 2**64 - 1
    I64 = 2**63 - 1
    vecs = [
        # (proof, commit, expected result, min, max) -- expectations are the CHECK()s at tests_impl.h:663-664,734-735,792-793
        ("fixed_1", fixed["vector_1"], fixed["commit_1"], 1, 86, 25586),
        ("fixed_2", fixed["vector_2"], fixed["commit_2"], 1, 0, 15),
        ("fixed_3", fixed["vector_3"], fixed["commit_3"], 1, U64, U64),
        # tests_impl.h:1245-1246, 1292-1293, 1343-1344
        ("repro_0_max64_exp18", repro["vector_0"], repro["commit_0"], 1, 0, U64),
        ("repro_1_minbits3", repro["vector_1"], repro["commit_1"], 1, 3, 73),
        ("repro_2_large_min", repro["vector_2"], repro["commit_2"], 1, I64 - 1, I64),
    ]
    data = [dict(name=n, proof=p.hex(), commit33=cm.hex(), result=r, min_value=str(mn), max_value=str(mx)) for n, p, cm, r, mn, mx in vecs]
    # what the same tests assert about secp256k1_rangeproof_rewind: (nonce, blind, value, recovered length, message) --
    # tests_impl.h:666-687, 737-757, 795-810 (nonce = the commitment bytes) and :843-880, 1241-1246, 1295-1300, 1341-1346
    # (vector_nonce / vector_blind, message = 0xFF bytes)
    glob = {n: v for n, v, _ in c_arrays(text[b:c])}
    MAXMSG = int(re.search(r"define SECP256K1_RANGEPROOF_MAX_MESSAGE_LEN\s+(\d+)", open(os.path.join(REF, "include/secp256k1_rangeproof.h")).read()).group(1))
    msg2 = re.search(r'message_2\[\] = "([^"]*)";', text).group(1).encode() + b"\0"      # a C string literal: sizeof() includes the NUL
    rew = [
        (fixed["commit_1"][:32], fixed["blind_1"], 86, b"\0" * 448),
        (fixed["commit_2"][:32], fixed["blind_2"], 11, msg2 + b"\0" * (192 - len(msg2))),
        (fixed["nonce_3"], fixed["blind_3"], U64, b""),
        (glob["vector_nonce"], glob["vector_blind"], U64, b"\xff" * MAXMSG),
        (glob["vector_nonce"], glob["vector_blind"], 13, b"\xff" * 128),
        (glob["vector_nonce"], glob["vector_blind"], I64, b""),
    ]
    for d, (nonce, blind, value, msg) in zip(data, rew):
        d["rewind"] = dict(nonce=nonce.hex(), blind=blind.hex(), value=str(value), capacity=MAXMSG, message=msg.hex())      # capacity: the tests' buffer size
    json.dump(dict(source="src/modules/rangeproof/tests_impl.h:589-812,883-1349", generator="secp256k1_generator_h", vectors=data),
              open(os.path.join(OUT, "rangeproof_vectors.json"), "w"), indent=0)
    print("rangeproof:", [(d["name"], len(d["proof"]) // 2) for d in data])


def bppp():
    path = os.path.join(REF, "src/modules/bppp/test_vectors/verify.h")
    text = open(path).read()
    arrs = c_arrays(text)
    byname = {}
    for n, v, _ in arrs:
        byname[n] = v
    gens = byname["verify_vector_gens"]
    vecs = []
    i = 0
    while f"verify_vector_{i}_commit33" in byname:
        nlen = int(re.search(rf"verify_vector_{i}_n_vec_len\s*=\s*(\d+)", text).group(1))
        res = int(re.search(rf"verify_vector_{i}_result\s*=\s*(\d+)", text).group(1))
        # c_vec32 is a 2-D array: parse rows
        m = re.search(rf"verify_vector_{i}_c_vec32\[(\d+)\]\[32\]\s*=\s*\{{(.*?)\}}\s*;", text, re.S)
        rows = re.findall(r"\{([^{}]*)\}", m.group(2))
        cvec = [bytes(int(v, 0) for v in re.findall(r"0[xX][0-9a-fA-F]{1,2}", r)) for r in rows]
        assert len(cvec) == int(m.group(1)) and all(len(c) == 32 for c in cvec)
        vecs.append(dict(index=i, commit33=byname[f"verify_vector_{i}_commit33"].hex(), n_vec_len=nlen, c_vec=[c.hex() for c in cvec],
                         rho=byname[f"verify_vector_{i}_r32"].hex(), proof=byname[f"verify_vector_{i}_proof"].hex(), result=res))
        i += 1
    json.dump(dict(source="src/modules/bppp/test_vectors/verify.h (driver tests_impl.h:540-588: transcript = plain sha256_initialize)",
                   gens=gens.hex(), vectors=vecs), open(os.path.join(OUT, "bppp_verify_vectors.json"), "w"), indent=0)
    print("bppp:", len(vecs), "vectors, gens", len(gens) // 33)


def bip340():
    path = os.path.join(REF, "src/modules/schnorrsig/tests_impl.h")
    text = open(path).read()
    a = text.index("static void test_schnorrsig_bip_vectors(void)")
    body = text[a:]
    end = body.index("\n}\n")
    body = body[:end]
    vecs = []
    # each vector is a { ... } block with pk / msg / sig arrays followed by a check helper call
    for blk in re.split(r"\n    \{\n", body)[1:]:
        arrs = {n: v for n, v, _ in c_arrays(blk)}
        if "pk" in arrs and "sig" not in arrs and re.search(r"CHECK\(!secp256k1_xonly_pubkey_parse", blk):
            # vectors 5 and 14: the key itself does not parse; any signature must be rejected
            vecs.append(dict(pk=arrs["pk"].hex(), msg="00" * 32, sig="00" * 64, result=0, pk_invalid=1))
            continue
        if "pk" not in arrs or "sig" not in arrs:
            continue
        msg = arrs.get("msg", b"")
        mm = re.search(r"unsigned char msg\[(\d+)\];\s*memset\(msg,\s*(0x[0-9a-fA-F]+|\d+),\s*sizeof\(msg\)\)", blk)
        if mm:                                      # vector 18: 100 bytes of 0x99 built with memset (tests_impl.h:761-762)
            msg = bytes([int(mm.group(2), 0)]) * int(mm.group(1))
        m = re.search(r"test_schnorrsig_bip_vectors_check_verify\(pk,\s*(?:msg|NULL),\s*(?:sizeof\(msg\)|\d+),\s*sig,\s*(\d)\)", blk)
        if m is None:
            if "secp256k1_xonly_pubkey_parse" in blk and "CHECK(!" in blk:
                vecs.append(dict(pk=arrs["pk"].hex(), msg=msg.hex(), sig=arrs["sig"].hex(), result=0, pk_invalid=1))
            continue
        vecs.append(dict(pk=arrs["pk"].hex(), msg=msg.hex(), sig=arrs["sig"].hex(), result=int(m.group(1)), pk_invalid=0))
    json.dump(dict(source="src/modules/schnorrsig/tests_impl.h:208-807 (BIP-340 test vectors)", vectors=vecs),
              open(os.path.join(OUT, "bip340_vectors.json"), "w"), indent=0)
    print("bip340:", len(vecs), "vectors", [v["result"] for v in vecs], "msg lens", sorted(set(len(v["msg"]) // 2 for v in vecs)))


def surjection():
    """src/modules/surjection/tests_impl.h:488-632: 5 accepted proofs over up to 5 fixed input tags + the rejection cases."""
    path = os.path.join(REF, "src/modules/surjection/tests_impl.h")
    text = open(path).read()
    a = text.index("static void test_fixed_vectors(void)")
    arrs = {n: v for n, v, _ in c_arrays(text[a:])}
    tags = [arrs["tag%d_ser" % i].hex() for i in range(5)]
    vecs = []
    for name, n_in in (("total1_used1", 1), ("total2_used1", 2), ("total3_used2", 3), ("total5_used3", 5), ("total5_used5", 5)):
        vecs.append(dict(name=name, proof=arrs[name].hex(), n_inputs=n_in, tag_first=0, output="out", result=1))
    # "check invalid keys fail" (:611-613)
    vecs.append(dict(name="total1_used1_wrong_input", proof=arrs["total1_used1"].hex(), n_inputs=1, tag_first=1, output="out", result=0))
    vecs.append(dict(name="total1_used1_wrong_output", proof=arrs["total1_used1"].hex(), n_inputs=1, tag_first=0, output="tag0", result=0))
    # parse failures (:609, :616-632): wrong length, extra / missing / out-of-range bitmap bits
    t55 = bytearray(arrs["total5_used5"]); t53 = bytearray(arrs["total5_used3"])
    vecs.append(dict(name="total5_used5_truncated", proof=bytes(t55[:len(t53)]).hex(), n_inputs=5, tag_first=0, output="out", result=0))
    for nm, base, val, extra in (("t55_6bits", t55, 0x3f, 0), ("t55_6bits_len", t55, 0x3f, 32), ("t55_bit_off", t55, 0x37, 0),
                                 ("t53_4bits", t53, 0x35, 0), ("t53_4bits_len", t53, 0x35, 32), ("t53_oor", t53, 0x34, 0)):
        b = bytearray(base); b[2] = val
        vecs.append(dict(name=nm, proof=(bytes(b) + b"\0" * extra).hex(), n_inputs=5, tag_first=0, output="out", result=0))
    json.dump(dict(source="src/modules/surjection/tests_impl.h:488-632", tags33=tags, output_tag33=arrs["output_tag_ser"].hex(), vectors=vecs),
              open(os.path.join(OUT, "surjection_vectors.json"), "w"), indent=0)
    print("surjection:", len(vecs), "vectors")


def halfagg():
    """src/modules/schnorrsig_halfagg/tests_impl.h:73-168: the three verification vectors of the half-aggregation spec, plus
    aggregates produced by the reference itself here (oracle/_ref: secp256k1_schnorrsig_aggregate on reference-signed BIP-340
    signatures, fixed seed) with the reference's verdict on each mutation."""
    import ctypes
    import numpy as np
    path = os.path.join(REF, "src/modules/schnorrsig_halfagg/tests_impl.h")
    text = open(path).read()
    a = text.index("void test_schnorrsig_aggverify_spec_vectors(void)")
    b = text.index("static void test_schnorrsig_aggregate_api_internal")
    vecs, cur = [], {}
    for name, val, _ in c_arrays(text[a:b]):
        cur[name] = val
        if name == "aggsig":
            n = len(cur.get("msgs32", b"")) // 32
            vecs.append(dict(name="spec_%d" % len(vecs), n=n, pks=cur.get("pubkeys_ser", b"").hex(), msgs=cur.get("msgs32", b"").hex(), aggsig=val.hex(), result=1))
            cur = {}
    assert [v["n"] for v in vecs] == [0, 1, 2]
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from tests.refapi import Ref
    ref = Ref(); rng = np.random.default_rng(340)
    for n in (1, 2, 3, 5, 8):
        sigs, msgs, pks = ref.make_schnorr(n, rng)
        agg = ref.halfagg_aggregate(pks, msgs, sigs)
        def add(name, pk, mg, ag):
            vecs.append(dict(name="gen_n%d_%s" % (n, name), n=n, pks=bytes(pk).hex(), msgs=bytes(mg).hex(), aggsig=bytes(ag).hex(),
                             result=max(0, ref.halfagg_verify(pk, mg, ag, n))))
        add("ok", pks.tobytes(), msgs.tobytes(), agg)
        m = bytearray(agg); m[-1] ^= 1; add("bad_s", pks.tobytes(), msgs.tobytes(), m)
        m = bytearray(agg); m[32 * (n - 1) + 5] ^= 4; add("bad_r_last", pks.tobytes(), msgs.tobytes(), m)
        m = bytearray(msgs.tobytes()); m[0] ^= 1; add("bad_msg0", pks.tobytes(), m, agg)
        if n >= 2:
            sw = pks.copy(); sw[[0, 1]] = sw[[1, 0]]; add("swapped_keys", sw.tobytes(), msgs.tobytes(), agg)
        add("s_overflow", pks.tobytes(), msgs.tobytes(), agg[:32 * n] + b"\xff" * 32)
        add("r_not_on_curve", pks.tobytes(), msgs.tobytes(), b"\x00" * 31 + b"\x05" + agg[32:])
        add("short", pks.tobytes(), msgs.tobytes(), agg[:-1])
        add("long", pks.tobytes(), msgs.tobytes(), agg + b"\x00" * 32)
    json.dump(dict(source="src/modules/schnorrsig_halfagg/tests_impl.h:73-168 (spec_*) + oracle/_ref generated (gen_*)", vectors=vecs),
              open(os.path.join(OUT, "halfagg_vectors.json"), "w"), indent=0)
    print("halfagg:", len(vecs), "vectors", sum(v["result"] for v in vecs), "accepted")


def pedersen():
    """The reference has no fixed vectors for secp256k1_pedersen_verify_tally (src/modules/generator/tests_impl.h:239-300 draws random
    transactions); these are transactions built the same way with oracle/_ref (fixed seed) and the reference's verdicts, plus
    unbalanced / reordered / duplicated / unparseable variants."""
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from tests.refapi import Ref
    ref = Ref(); rng = np.random.default_rng(2718)
    cases = []
    def add(name, pos, neg):
        res = int(ref.pedersen_verify_tally_many([(pos, neg)])[0])
        cases.append(dict(name=name, pos=[bytes(x).hex() for x in np.asarray(pos).reshape(-1, 33)], neg=[bytes(x).hex() for x in np.asarray(neg).reshape(-1, 33)], result=res))
    for k, (ni, no) in enumerate(((1, 1), (1, 2), (2, 2), (3, 5), (8, 9), (1, 17))):
        a, b = ref.make_balanced_tally(rng, ni, no)
        add("balanced_%d_%d" % (ni, no), a, b); add("swapped_%d_%d" % (ni, no), b, a)
        add("missing_output_%d_%d" % (ni, no), a, b[:-1])
        c = b.copy(); c[0, 0] ^= 1; add("flipped_sign_%d_%d" % (ni, no), a, c)
        if k == 3:
            add("both_sides_doubled", np.concatenate([a, a]), np.concatenate([b, b]))
            add("same_commitment_both_sides", a[:1], a[:1]); add("twice_vs_once", np.concatenate([a[:1], a[:1]]), a[:1])
            c = b.copy(); c[1, 0] = 2; add("bad_prefix", a, c)
            c = b.copy(); c[1, 1:] = 0xFF; add("x_overflow", a, c)
            c = b.copy(); c[1, 1:] = 0; c[1, 33 - 1] = 5; add("x_not_on_curve", a, c)
    add("empty", np.zeros((0, 33), np.uint8), np.zeros((0, 33), np.uint8))
    a, b = ref.make_balanced_tally(rng, 2, 2)
    add("only_positive", a, a[:0]); add("only_negative", a[:0], a)
    json.dump(dict(source="generated by oracle/_ref (secp256k1_pedersen_commit / _blind_sum / _verify_tally), cf. src/modules/generator/tests_impl.h:193-300; result -1 = a commitment does not parse",
                   vectors=cases), open(os.path.join(OUT, "pedersen_tally_vectors.json"), "w"), indent=0)
    print("pedersen tally:", len(cases), "vectors", [c["result"] for c in cases])


def split_bounds():
    """scalars_near_split_bounds (src/tests.c:4718-4739): the 20 scalars that reach the largest outputs of secp256k1_scalar_split_lambda,
    (a*LAMBDA + (ORDER + b)/2) % ORDER for a in -2..2, b in -3, -1, 1, 3 -- the reference feeds them to ecmult in run_ecmult_near_split_bound"""
    text = open(os.path.join(REF, "src", "tests.c")).read()
    body = text[text.index("scalars_near_split_bounds[20] = {"):]
    body = body[:body.index("};")]
    vals = []
    for m in re.finditer(r"SECP256K1_SCALAR_CONST\(([^)]*)\)", body):
        words = [int(x, 16) for x in m.group(1).split(",")]
        assert len(words) == 8
        v = 0
        for w in words:
            v = (v << 32) | w
        vals.append("%064x" % v)
    assert len(vals) == 20
    # (the values are taken as the array holds them; they are what the reference's run_ecmult_near_split_bound feeds to ecmult)
    json.dump(dict(source="src/tests.c:4718-4739 scalars_near_split_bounds", scalars=vals), open(os.path.join(OUT, "split_bounds.json"), "w"), indent=0)
    print("split bounds:", len(vals), "scalars")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference tree not found at " + REF)
    rangeproof(); bppp(); bip340(); surjection(); halfagg(); pedersen(); split_bounds()
