#!/usr/bin/env python3
"""Secondary throughput figures quoted in DESIGN.md (the headline lives in bench.py): BIP-340 batch (config 2), BP++ norm argument
(config 4), surjection proofs, bare double multiplications, Pedersen tallies, half-aggregate verification.  Inputs are made with
oracle/_ref (test infrastructure) and are resident in HBM where a _dev entry point exists; prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from secp256k1_zkp_amd import Engine  # noqa: E402
from tests.refapi import Ref, G_XY  # noqa: E402


def timed(fn, reps=3):
    torch.cuda.synchronize()              # the engine's stream is not ordered against torch's: inputs must be complete
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def main():
    eng = Engine(0); ref = Ref(); rng = np.random.default_rng(1); dev = torch.device("cuda:0"); out = {}
    # config 2: 2^16 BIP-340 verifications
    n = 1 << 16
    sigs, msgs, pks = ref.make_schnorr(n, rng, threads=16)
    sigs[::256, 40] ^= 1
    d = [torch.tensor(x).to(dev) for x in (sigs, msgs, pks)]; res = torch.zeros(n, dtype=torch.int32, device=dev)
    dt = timed(lambda: eng.schnorrsig_verify_batch_dev(res, d[0], d[1], d[2]))
    assert int(res.sum().item()) == n - n // 256
    out["bip340_verify_2p16"] = {"ms": dt * 1e3, "per_s": n / dt}
    # bare double multiplications, 2^21
    n = 1 << 21
    a = torch.tensor(np.frombuffer(G_XY, np.uint8).copy()).to(dev).repeat(n, 1)
    na = torch.tensor(rng.integers(0, 256, (n, 32), dtype=np.uint8)).to(dev); ng = torch.tensor(rng.integers(0, 256, (n, 32), dtype=np.uint8)).to(dev)
    r = torch.zeros(n, 64, dtype=torch.uint8, device=dev); ri = torch.zeros(n, dtype=torch.int32, device=dev)
    dt = timed(lambda: eng.ecmult_batch_dev(r, ri, a, na, ng))
    out["ecmult_2p21"] = {"ms": dt * 1e3, "per_s": n / dt}
    # config 4: 2^12 BP++ norm arguments (g_len 64, h_len 8), host-buffer API
    n = 1 << 12
    base = ref.make_bppp(64, rng, 64, 8)
    reps = n // 64
    args = [np.concatenate([base[0]] * reps), np.concatenate([base[1]] * reps), np.concatenate([base[2]] * reps), base[3], base[4],
            np.concatenate([base[5]] * reps), np.concatenate([base[6]] * reps)]
    dt = timed(lambda: eng.bppp_norm_product_verify_batch(*args))
    assert eng.bppp_norm_product_verify_batch(*args).all()
    out["bppp_norm_verify_2p12"] = {"ms": dt * 1e3, "per_s": n / dt, "note": "host buffers (H2D included)"}
    # surjection proofs 3-of-3, 2^16 (64 distinct proofs replicated)
    protos = [ref.make_surjection(rng, 3, 3) for _ in range(64)]
    n = 1 << 16
    proofs = [protos[i % 64][0] for i in range(n)]; tags = [protos[i % 64][1] for i in range(n)]; outs = np.stack([protos[i % 64][2] for i in range(n)])
    dt = timed(lambda: eng.surjectionproof_verify_batch(proofs, tags, outs), reps=1)
    out["surjection_3of3_2p16"] = {"ms": dt * 1e3, "per_s": n / dt, "note": "host buffers + python packing included; kernel time: last_ms", "kernel_ms": eng.last_ms(1)}
    # Pedersen tallies: 2^15 transactions of 2 inputs / 3 outputs
    protos = [ref.make_balanced_tally(rng, 2, 3) for _ in range(64)]
    n = 1 << 15
    tallies = [protos[i % 64] for i in range(n)]
    dt = timed(lambda: eng.pedersen_verify_tally_batch(tallies), reps=1)
    out["pedersen_tally_2in3out_2p15"] = {"ms": dt * 1e3, "per_s": n / dt, "note": "host buffers + python packing included", "call_ms": eng.last_ms(0)}
    # half-aggregate verification, n = 2^15
    n = 1 << 15
    sigs, msgs, pks = ref.make_schnorr(n, rng, threads=16)
    agg = ref.halfagg_aggregate(pks, msgs, sigs)
    dt = timed(lambda: eng.schnorrsig_aggverify(pks, msgs, agg), reps=2)
    out["halfagg_verify_2p15"] = {"ms": dt * 1e3, "signatures_per_s": n / dt}
    # the `_dev` forms of the same aggregate (inputs resident in HBM): the device walks the randomizer hash chain itself / the caller walks it
    # on the host (s2k_halfagg_chain_states) and uploads the states
    import ctypes
    L = eng._lib
    d_pk = torch.tensor(np.ascontiguousarray(pks)).cuda(); d_m = torch.tensor(np.ascontiguousarray(msgs)).cuda(); d_a = torch.tensor(np.frombuffer(agg, np.uint8).copy()).cuda()
    d_r = torch.zeros(4, dtype=torch.int32, device="cuda")
    p_ = lambda t: ctypes.c_void_p(t.data_ptr())
    torch.cuda.synchronize()
    def dev_chain_on_device():
        assert L.secp256k1_schnorrsig_aggverify_dev(eng._h, None, p_(d_r), p_(d_pk), 0, p_(d_m), n, p_(d_a), len(agg)) == 1
        assert L.s2k_engine_sync(eng._h) == 1
    states = np.zeros(((3 * n) >> 1, 8), np.uint32)
    pks_c = np.ascontiguousarray(pks, np.uint8); msgs_c = np.ascontiguousarray(msgs, np.uint8); agg_b = bytes(agg)
    def dev_chain_by_caller():
        assert L.s2k_halfagg_chain_states(states.ctypes.data, pks_c.ctypes.data, 0, msgs_c.ctypes.data, n, agg_b) == 1
        d_s = torch.tensor(states.view(np.int32)).cuda()
        assert L.secp256k1_schnorrsig_aggverify_dev_chain(eng._h, None, p_(d_r), p_(d_pk), 0, p_(d_m), n, p_(d_a), len(agg), p_(d_s)) == 1
        assert L.s2k_engine_sync(eng._h) == 1
    dt1 = timed(dev_chain_on_device, reps=2); assert int(d_r[0].item()) == 1
    dt2 = timed(dev_chain_by_caller, reps=2); assert int(d_r[0].item()) == 1
    out["halfagg_verify_2p15_dev"] = {"ms_device_chain": dt1 * 1e3, "ms_chain_states_from_the_caller": dt2 * 1e3, "signatures_per_s_chain_states_from_the_caller": n / dt2,
                                      "note": "_dev forms; the second figure includes walking the chain on the host and uploading the states"}
    # rangeproof rewind (wallet scan): 2^12 64-bit proofs with 64-byte messages, right nonce
    n = 1 << 12
    c, p, g, v, b, nn, m = ref.make_rangeproofs_msg(n, rng, msg_len=64, min_bits=64, threads=16)
    packed = eng.pack(p)
    dt = timed(lambda: eng.rangeproof_rewind_batch(c, packed, g, nn, msg_capacity=64), reps=2)
    r = eng.rangeproof_rewind_batch(c, packed, g, nn, msg_capacity=64)
    assert r[0].all() and np.array_equal(r[2], v)
    out["rangeproof_rewind_2p12"] = {"ms": dt * 1e3, "per_s": n / dt, "note": "host buffers (21 MB of proofs H2D included); verification + recovery"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
