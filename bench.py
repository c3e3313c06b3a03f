#!/usr/bin/env python3
"""bench.py -- headline benchmark: 64-bit Borromean rangeproof verifies/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1: starts the N ranks itself, one per GPU, through the line below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the verification hot path (secp256k1_rangeproof_verify semantics, five HIP kernels) over one batch
of 2^14 synthetic 64-bit proofs per GPU (BASELINE.json configs[2]; the batch the metric is quoted on), inputs already
resident in HBM.  Independent proofs shard with no exchange step, so N GPUs run N replicas of the batch ("weak" scaling).
One JSON line is printed by rank 0.  Also reported: the roofline of the dominant kernel (k_rp_rings) and the reference's
own CPU path timed on the host cores of the same box (oracle/_ref, bounded sample).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 1 << 14                    # proofs per GPU per step
PROOF_BYTES_ALGO = 5223            # 5126 B proof + 33 B commitment + 64 B generator
MAC64_PER_PROOF = 6.6e6            # reference schedule, 64x64->128 MACs per 64-bit proof
MAC64_PER_MSM_TERM = 6.3e3         # reference schedule at n >= 16384 (20 bucket additions per term)
MAC64_PER_MSM_TERM_SMALL = 10.6e3  # reference schedule at n = 1 024 (32 bucket additions per term)
MAC64_PER_SCHNORR = 45e3 + 267 * 21 + 267 * 21      # ~45 k MAC64 for the double multiplication + one inversion + the key's square root (~267 field operations of ~21 MAC64 each)
MSM_BYTES_PER_TERM = 96            # 32 B scalar + 64 B affine point
# integer-MAC peak of the chip: v_mad_u64_u32 lane-ops/s.  Measured with >= 10 ms launches at 8 waves/SIMD
# (tools/ubench/issue_model.hip -> profiles/r02a_issue_model.txt: 3.73e13 = 4.22 cycles per wave64 instruction at the nominal
# 2.4 GHz); the architectural half-rate figure is 256 CU x 4 SIMD x 16 lanes x 2.4 GHz = 3.93e13.  The measured one is used.
MAD32_PEAK = 3.73e13
MAD32_PEAK_ARCH = 3.93e13
HBM_PEAK_GBS = 8000.0


def make_inputs(n, seed, distinct_generators=False):
    """n 64-bit proofs (exp 0, min_value 0), signed with the reference when oracle/_ref is available (as src/bench_rangeproof.c:26-36
    does), otherwise tiled from the committed golden 64-bit vector.  Generator: secp256k1_generator_h for every proof (what
    bench_rangeproof uses), or -- distinct_generators -- a different random curve point per proof (the Elements case: every
    confidential output carries its own blinded asset generator)."""
    rng = np.random.default_rng(seed)
    try:
        from tests.refapi import Ref
        ref = Ref()
        gens_in = None
        if distinct_generators:
            gens_in = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(n)])
        commits, proofs, gens, _ = ref.make_rangeproofs(n, rng, min_bits=64, gens64=gens_in, threads=min(usable_cores() * 2, 64))
        return commits, proofs, gens, "synthetic (secp256k1_rangeproof_sign via oracle/_ref, %d unique 64-bit proofs)" % n, ref
    except OSError:
        from secp256k1_zkp_amd.constants import GENERATOR_H
        v = [x for x in json.load(open(os.path.join(ROOT, "tests", "golden", "rangeproof_vectors.json")))["vectors"] if x["name"].startswith("repro_0")][0]
        commits = np.tile(np.frombuffer(bytes.fromhex(v["commit33"]), np.uint8), (n, 1))
        proofs = [bytes.fromhex(v["proof"])] * n
        gens = np.frombuffer(GENERATOR_H * n, np.uint8).reshape(n, 64)
        return commits, proofs, gens, "synthetic (reference golden 64-bit proof tiled; oracle/_ref not present)", None


def device_wait(*engines):
    """How a timed region ends: the engines' own waits on their streams (hipStreamSynchronize; every `_dev` result is ordered on the stream
    of its call), THEN torch.cuda.synchronize() as the contract asks -- which finds an idle device.  torch.cuda.synchronize() alone, left to
    wait for an engine's work, returned 40-55 ms late 9 times in 3 600 calls after the multi-threaded CPU legs of a process like this one
    (the reference's provers) -- twice over: profiles/r06ae_idle_probe3.txt, _probe4.txt -- while the engine's wait (0 in 3 600) and a polled
    event (0 in 3 600) never did, with the device's own events showing the usual kernel time: a late wake-up of that blocking wait, not
    work."""
    import torch                          # (imported where it is used, like everywhere in this file: the launcher path never needs it)
    for e in engines:
        e.sync()
    torch.cuda.synchronize()


def usable_cores():
    """host cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU boxes expose
    256 hardware threads but grant the container 16 CPUs of quota; oversubscribing only slows the OpenMP team down)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def cpu_baseline_port(commits, proofs, gens):
    """fallback when oracle/_ref did not travel: the plain-C restatement (oracle/zkp_oracle.c), kind = "port"."""
    import ctypes
    path = os.path.join(ROOT, "oracle", "libzkp_oracle.so")
    if not os.path.exists(path):
        return None
    zo = ctypes.CDLL(path)
    cores = usable_cores()
    kn = min(len(proofs), 64 * cores)
    stride = max(len(p) for p in proofs[:kn])
    buf = np.zeros((kn, stride), np.uint8)
    for i in range(kn):
        buf[i, :len(proofs[i])] = np.frombuffer(proofs[i], np.uint8)
    plens = np.array([len(p) for p in proofs[:kn]], np.uint64)
    res = np.zeros(kn, np.int32); mn = np.zeros(kn, np.uint64); mx = np.zeros(kn, np.uint64)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    c = np.ascontiguousarray(commits[:kn]); g = np.ascontiguousarray(gens[:kn])
    t = time.time()
    zo.zo_rangeproof_verify_many(vp(res), vp(mn), vp(mx), vp(c), vp(buf), ctypes.c_size_t(stride), vp(plens), vp(g), ctypes.c_size_t(kn), ctypes.c_int(cores))
    dt = time.time() - t
    assert res.all()
    return {"value": kn / dt, "unit": "verifies/s", "cores": cores, "kind": "port", "sample": "%d 64-bit proofs on %d threads (%.2f s), oracle/zkp_oracle.c" % (kn, cores, dt)}


def measure_dropin(eng, commits, proofs, gens, steps):
    """host-memory entry points, wall clock per call (everything between handing over host buffers and having the verdicts)"""
    import ctypes
    from secp256k1_zkp_amd import Engine
    n = len(proofs)
    out = {"unit": "verifies/s", "batch": n, "includes": "packing into pinned staging (host threads), H2D, all kernels, D2H, result copy-out",
           "stage_threads": int(os.environ.get("S2K_STAGE_THREADS", "0")) or "default min(8, cores / 2)"}
    # (i) packed numpy arrays -> secp256k1_rangeproof_verify_batch
    pdata, poff = Engine.pack(proofs)
    c = np.ascontiguousarray(commits, np.uint8); g = np.ascontiguousarray(gens, np.uint8)
    res = np.zeros(n, np.int32); mn = np.zeros(n, np.uint64); mx = np.zeros(n, np.uint64)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    L = eng._lib
    def call_packed():
        assert L.secp256k1_rangeproof_verify_batch(eng._h, vp(res), vp(mn), vp(mx), vp(c), vp(pdata), vp(poff), None, None, vp(g), n) == 1
    call_packed(); call_packed()                 # (both staging sets of the engine get their pinned memory here)
    t = time.perf_counter()
    for _ in range(steps):
        call_packed()
    dtp = (time.perf_counter() - t) / steps
    assert res.all()
    out["host_buffers"] = {"entry": "secp256k1_rangeproof_verify_batch (packed arrays)", "value": n / dtp, "ms_per_call": dtp * 1e3}
    # (i') the same batches through the asynchronous pair, two in flight: submit(k+1) gathers and copies underneath the kernels of batch k
    L.secp256k1_rangeproof_verify_batch_submit.argtypes = [ctypes.c_void_p] * 11 + [ctypes.c_size_t]
    outs = [(np.zeros(n, np.int32), np.zeros(n, np.uint64), np.zeros(n, np.uint64)) for _ in range(2)]
    def submit(k):
        r, a, b = outs[k & 1]
        tk = ctypes.c_uint64(0)
        assert L.secp256k1_rangeproof_verify_batch_submit(eng._h, ctypes.byref(tk), vp(r), vp(a), vp(b), vp(c), vp(pdata), vp(poff), None, None, vp(g), n) == 1
        return tk.value
    def wait(tk):
        assert L.secp256k1_rangeproof_verify_batch_wait(eng._h, ctypes.c_uint64(tk)) == 1
    k2 = max(2 * steps, 6)
    ta, tb = submit(0), submit(1); wait(ta); wait(tb)      # (both staging sets at full size: the halved synchronous calls above left them at half)
    t = time.perf_counter()
    prev = submit(0)
    for k in range(1, k2):
        cur = submit(k)
        wait(prev)
        prev = cur
    wait(prev)
    dta = (time.perf_counter() - t) / k2
    assert outs[0][0].all() and outs[1][0].all() and int(outs[1][2].min()) == 2**64 - 1
    out["two_in_flight"] = {"entry": "secp256k1_rangeproof_verify_batch_submit / _wait (packed arrays; submit(k+1) before wait(k))", "value": n / dta, "ms_per_call": dta * 1e3,
                            "batches": k2}
    # (i'') the SYNCHRONOUS call from two verifier threads sharing the engine: each queues for one of the two staging sets, so the gathering
    # and the copies of one thread's batch run underneath the kernels of the other's (ctypes releases the GIL for the duration of a call)
    import threading
    def two_threads(call_with):
        errs = []
        def worker(k):
            try:
                for _ in range(steps):
                    call_with(k)
            except Exception as ex:          # noqa: BLE001
                errs.append(repr(ex))
        th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
        t = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        dt2 = (time.perf_counter() - t) / (2 * steps)
        assert not errs, errs
        return dt2
    def call_packed_k(k):
        r, a, b = outs[k]
        assert L.secp256k1_rangeproof_verify_batch(eng._h, vp(r), vp(a), vp(b), vp(c), vp(pdata), vp(poff), None, None, vp(g), n) == 1
    dtt = two_threads(call_packed_k)
    assert outs[0][0].all() and outs[1][0].all()
    out["two_threads_synchronous"] = {"entry": "secp256k1_rangeproof_verify_batch from two threads on one engine", "value": n / dtt, "ms_per_call": dtt * 1e3}
    # (ii) reference types through the hooked reference library
    try:
        from tests import hookapi
        if os.path.exists(hookapi.HOOKED_PATH):
            hk = hookapi.Hooked()
            addr = lambda name: ctypes.cast(getattr(L, name), ctypes.c_void_p).value
            hk.set_backend(engine=eng._h, rangeproof=addr("secp256k1_rangeproof_verify_batch"), rangeproof_ptrs=addr("secp256k1_rangeproof_verify_batch_ptrs"),
                           rangeproof_submit=addr("secp256k1_rangeproof_verify_batch_ptrs_submit"), rangeproof_wait=addr("secp256k1_rangeproof_verify_batch_wait"))
            cobj = np.zeros((n, 64), np.uint8); cobj[:, :33] = c.reshape(n, 33)
            gobj = g.reshape(n, 64).copy()
            pbufs = [np.frombuffer(p, np.uint8).copy() for p in proofs]            # every proof its own allocation, as a caller's objects would be
            plens = (ctypes.c_size_t * n)(*[len(p) for p in proofs])
            cp = hookapi._ptr_array([cobj[i] for i in range(n)]); gp = hookapi._ptr_array([gobj[i] for i in range(n)]); pp = hookapi._ptr_array(pbufs)
            r32 = (ctypes.c_int * n)(); mn2 = np.zeros(n, np.uint64); mx2 = np.zeros(n, np.uint64)
            def call_hook():
                assert hk.lib.secp256k1_amd_rangeproof_verify_batch(hk.ctx, r32, mn2.ctypes.data, mx2.ctypes.data, cp, pp, plens, None, None, gp, n) == 1
            s0 = hk.stats()
            call_hook()
            t = time.perf_counter()
            for _ in range(steps):
                call_hook()
            dth = (time.perf_counter() - t) / steps
            s1 = hk.stats()
            assert s1 == (s0[0] + steps + 1, s0[1]), "the hook fell back to the CPU"
            assert all(r32[i] == 1 for i in range(0, n, 97)) and int(mx2.min()) == 2**64 - 1
            out["hooked_reference_types"] = {"entry": "secp256k1_amd_rangeproof_verify_batch (libsecp256k1_hooked.so: the unmodified reference + integration/secp256k1_amd_hook.c)",
                                             "value": n / dth, "ms_per_call": dth * 1e3, "served": s1[0] - s0[0], "fell_back": s1[1] - s0[1]}
            # (ii'') the synchronous adapter from two verifier threads
            def call_hook_k(k):
                assert hk.lib.secp256k1_amd_rangeproof_verify_batch(hk.ctx, r32b[k], mnb[k].ctypes.data, mxb[k].ctypes.data, cp, pp, plens, None, None, gp, n) == 1
            r32b = [(ctypes.c_int * n)() for _ in range(2)]; mnb = [np.zeros(n, np.uint64) for _ in range(2)]; mxb = [np.zeros(n, np.uint64) for _ in range(2)]
            s0 = hk.stats()
            dtt = two_threads(call_hook_k)
            s1 = hk.stats()
            assert s1[1] == s0[1], "the hook fell back to the CPU"
            out["hooked_two_threads_synchronous"] = {"entry": "secp256k1_amd_rangeproof_verify_batch from two threads (one context, one engine)", "value": n / dtt, "ms_per_call": dtt * 1e3}
            # (ii') the asynchronous adapters, two batches in flight
            def hsubmit(k):
                tk = ctypes.c_uint64(0)
                assert hk.lib.secp256k1_amd_rangeproof_verify_batch_submit(hk.ctx, ctypes.byref(tk), r32b[k & 1], mnb[k & 1].ctypes.data, mxb[k & 1].ctypes.data,
                                                                           cp, pp, plens, None, None, gp, n) == 1
                assert tk.value != 0, "the hook verified on the CPU at submission time"
                return tk.value
            def hwait(tk):
                assert hk.lib.secp256k1_amd_rangeproof_verify_batch_wait(hk.ctx, ctypes.c_uint64(tk)) == 1
            ha, hb = hsubmit(0), hsubmit(1); hwait(ha); hwait(hb)
            t = time.perf_counter()
            prev = hsubmit(0)
            for k in range(1, k2):
                cur = hsubmit(k)
                hwait(prev)
                prev = cur
            hwait(prev)
            dtb = (time.perf_counter() - t) / k2
            assert all(r32b[j][i] == 1 for j in range(2) for i in range(0, n, 97)) and int(mxb[1].min()) == 2**64 - 1
            out["hooked_two_in_flight"] = {"entry": "secp256k1_amd_rangeproof_verify_batch_submit / _wait (reference types, submit(k+1) before wait(k))",
                                           "value": n / dtb, "ms_per_call": dtb * 1e3, "batches": k2}
            hk.set_backend()
    except OSError as ex:
        out["hooked_reference_types"] = {"skipped": str(ex)}
    return out


def measure_secondary(eng, ref, dev, steps, with_cpu=True):
    """BASELINE configs 2 and 4 and the GPU side of config 1, each with an in-run check of the verdicts / the result against the reference
    and a VALU roofline from SURVEY 8d's per-unit work (Schnorr: ~45 k MAC64 + one inversion and one square root per signature; norm
    argument: a (g_len + h_len) + 13-point multi-scalar multiplication per proof at ~10.6 k MAC64 per term; MSM of 1 024 terms: 10.6 k)."""
    import torch
    from secp256k1_zkp_amd import Engine
    from secp256k1_zkp_amd.constants import G_XY
    out = {}
    rng = np.random.default_rng(777)

    def loop(fn):
        # K calls queued back to back and waited for once -- twice, the faster pass counts: these loops are 5-60 ms long and run between the
        # CPU legs of this function (reference provers and verifiers on the host cores), after which one pass in ten showed a 40-60 ms hole --
        # not the engine: 1 500 consecutive groups of calls in a quiet process had none (profiles/r06y_stall_probe.txt, r06y_bench_outliers.txt);
        # later traced to torch.cuda.synchronize() waking up late (device_wait above, which these loops now end with; best-of-two kept)
        fn(); device_wait(eng)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            device_wait(eng)
            d = (time.perf_counter() - t0) / steps
            best = d if best is None or d < best else best
        return best

    def roof(mac64_per_unit, units, sec, what):
        rate = 4 * mac64_per_unit * units / sec
        return {"bound": "valu", "achieved": rate / 1e12, "peak": MAD32_PEAK / 1e12, "unit": "T lane-MAC/s (v_mad_u64_u32)", "frac": rate / MAD32_PEAK,
                "frac_of_architectural_peak": rate / MAD32_PEAK_ARCH, "note": what + " x 4 v_mad_u64_u32 x units / wall time of the whole call (K calls queued, waited for once)"}

    # ---- config 2: 2^16 BIP-340 signatures, some of them broken
    n = 1 << 16
    sigs, msgs, pks = ref.make_schnorr(n, rng, threads=usable_cores())
    sigs[::251, 40] ^= 1; pks[100::509, 7] ^= 4
    chk = np.arange(0, n, 61)                                                   # a sample of the batch through the reference (all of it would take ~3 s more)
    want = ref.schnorr_verify_many(sigs[chk], msgs[chk], pks[chk])
    d = [torch.tensor(x).to(dev) for x in (sigs, msgs, pks)]; res = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    sec = loop(lambda: eng.schnorrsig_verify_batch_dev(res, d[0], d[1], d[2]))
    got = res.cpu().numpy()
    assert np.array_equal(got[chk], want), "BIP-340 verdicts differ from the reference"
    bad = np.zeros(n, bool); bad[::251] = True; bad[100::509] = True
    assert not got[bad].any() and got[~bad].all()
    out["bip340_2p16"] = {"metric": "BIP-340 signature verifies/sec (BASELINE config 2)", "value": n / sec, "unit": "verifies/s", "ms": sec * 1e3, "batch": n,
                          "verified": True, "result_check": "every verdict as constructed (%d broken signatures / keys rejected), %d items compared with the reference's secp256k1_schnorrsig_verify" % (int(bad.sum()), chk.size),
                          "roofline": roof(MAC64_PER_SCHNORR, n, sec, "algorithmic %.1fe3 MAC64 per signature (SURVEY 8d: ~45 k for the double multiplication + one inversion + the key's square root)" % (MAC64_PER_SCHNORR / 1e3)),
                          "hbm_roofline": {"achieved": 160.0 * n / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 160.0 * n / sec / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_signature": 160},
                          "occupancy_note": "one call of BASELINE's size is occupancy bound: 2^16 signatures are 1 024 wavefronts, ONE per SIMD, where a wavefront issues an instruction every ~7.6 cycles whatever it is; "
                                            "`two_engines` (two submitting threads) or a larger batch fill the issue slots this figure leaves free"}
    # the same batches through TWO engines on the device (two streams): 2^16 signatures are 1 024 wavefronts -- exactly ONE per SIMD, where a
    # wavefront issues an instruction every ~7.6 cycles whatever it is -- so a second batch in flight runs in the issue slots the first leaves
    # free.  What a verifier with two submitting threads (an engine each) gets at this batch size; `value` above stays the one-engine figure.
    eng_s = Engine(eng.device)
    try:
        res_s = torch.zeros(n, dtype=torch.int32, device=dev)
        eng_s.schnorrsig_verify_batch_dev(res_s, d[0], d[1], d[2]); device_wait(eng, eng_s)
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.schnorrsig_verify_batch_dev(res, d[0], d[1], d[2]); eng_s.schnorrsig_verify_batch_dev(res_s, d[0], d[1], d[2])
        device_wait(eng, eng_s)
        sec_s = (time.perf_counter() - t0) / (2 * steps)
        assert np.array_equal(res_s.cpu().numpy(), got) and np.array_equal(res.cpu().numpy(), got), "BIP-340 verdicts differ with two engines"
        out["bip340_2p16"]["two_engines"] = {"value": n / sec_s, "ms_per_batch": sec_s * 1e3, "verified": True,
                                             "roofline_frac": 4 * MAC64_PER_SCHNORR * n / sec_s / MAD32_PEAK,
                                             "note": "batches submitted alternately to two engines on the device (two streams); per-batch time = wall / batches"}
    finally:
        eng_s.close()
    del d, res
    # ---- config 4: 2^12 BP++ norm arguments (g_len 64, h_len 8): 64 distinct proofs by the reference's prover, tiled; three of the 64 broken
    nb = 1 << 12
    base = ref.make_bppp(64, rng, 64, 8)
    pr = base[0].copy(); pr[5, 10] ^= 1; pr[21, 70] ^= 0x80; cm = base[6].copy(); cm[40, 12] ^= 2
    want = ref.bppp_verify_many(pr, base[1], base[2], base[3], base[4], base[5], cm)
    assert want.sum() == 61
    reps = nb // 64
    tile = lambda a: torch.tensor(np.ascontiguousarray(np.concatenate([a] * reps))).to(dev)
    d_pr, d_tr, d_rho, d_cv, d_cm = tile(pr), tile(base[1]), tile(base[2]), tile(base[5]), tile(cm)
    gens = np.ascontiguousarray(base[3]); d_gens = torch.tensor(gens).to(dev)
    res = torch.zeros(nb, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    call = lambda: eng.bppp_norm_product_verify_batch_dev(res, d_pr, pr.shape[1], d_tr, d_rho, d_gens, gens, base[4], d_cv, base[5].shape[1], d_cm, nb)
    sec = loop(call)
    got = res.cpu().numpy()
    assert np.array_equal(got, np.tile(want, reps)), "BP++ verdicts differ from the reference"
    terms = 64 + 8 + 13
    out["bppp_2p12"] = {"metric": "BP++ norm-argument verifies/sec (BASELINE config 4: the norm argument is the measurable unit, SURVEY 8d)", "value": nb / sec, "unit": "verifies/s",
                        "ms": sec * 1e3, "batch": nb, "g_len": 64, "h_len": 8, "verified": True,
                        "result_check": "== secp256k1_bppp_rangeproof_norm_product_verify of the reference on the 64 distinct proofs (61 valid, 3 broken), tiled %d times" % reps,
                        "roofline": roof(MAC64_PER_MSM_TERM_SMALL * terms, nb, sec, "algorithmic %d terms x 10.6e3 MAC64 per term per proof (SURVEY 8d, the 1 024-term schedule)" % terms),
                        "occupancy_note": "one call of BASELINE's size is occupancy bound: 2^12 proofs put 832 wavefronts (13 full double multiplications per proof) on 1 024 SIMDs; 2^16 proofs per call run at 3.6e6/s "
                                          "(profiles/r06n_bppp_tables_ab.txt), `two_engines` shows what a second submitting thread gets at this size"}
    # the same batches through TWO engines on the device (each its own stream and scratch; the tables of G are the device's): a call is a chain of
    # latency-bound stages -- 13 full double multiplications per proof on ~40 % of the lane slots -- so what a verifier with two submitting
    # threads sees is two chains side by side
    eng_b = Engine(eng.device)
    try:
        res_b = torch.zeros(nb, dtype=torch.int32, device=dev)
        call_b = lambda: eng_b.bppp_norm_product_verify_batch_dev(res_b, d_pr, pr.shape[1], d_tr, d_rho, d_gens, gens, base[4], d_cv, base[5].shape[1], d_cm, nb)
        call(); call_b(); device_wait(eng, eng_b)
        t0 = time.perf_counter()
        for _ in range(steps):
            call(); call_b()
        device_wait(eng, eng_b)
        sec2 = (time.perf_counter() - t0) / (2 * steps)
        assert np.array_equal(res_b.cpu().numpy(), np.tile(want, reps)) and np.array_equal(res.cpu().numpy(), np.tile(want, reps)), "BP++ verdicts differ with two engines"
        out["bppp_2p12"]["two_engines"] = {"value": nb / sec2, "ms_per_batch": sec2 * 1e3, "verified": True,
                                           "roofline_frac": 4 * MAC64_PER_MSM_TERM_SMALL * terms * nb / sec2 / MAD32_PEAK,
                                           "note": "batches submitted alternately to two engines on the device (two streams); per-batch time = wall / batches"}
    finally:
        eng_b.close()
    if with_cpu:
        k = 256
        idx = np.arange(k) % 64
        t0 = time.time(); r = ref.bppp_verify_many(pr[idx], base[1][idx], base[2][idx], base[3], base[4], base[5][idx], cm[idx]); t = time.time() - t0
        assert np.array_equal(r, want[idx])
        out["bppp_2p12"]["cpu_baseline"] = {"value": k / t, "unit": "verifies/s", "cores": 1, "kind": "reference",
                                            "sample": "secp256k1_bppp_rangeproof_norm_product_verify through oracle/ref_shim.c (src/bench_bppp.c is an empty stub), %d proofs on one thread in %.2f s" % (k, t)}
    del d_pr, d_tr, d_rho, d_cv, d_cm, res
    # ---- config 1 on the GPU: the 1 024 (scalar, point) pairs of bench_ecmult as ONE call (latency, not throughput: the CPU row is the baseline's)
    nm = 1024
    ks = rng.integers(0, 256, (nm, 32), dtype=np.uint8); scs = rng.integers(0, 256, (nm, 32), dtype=np.uint8); gsc = rng.integers(0, 256, 32, dtype=np.uint8)
    pts, pinf = ref.ecmult_batch(np.tile(np.frombuffer(G_XY, np.uint8), (nm, 1)), ks)
    d_s, d_p, d_g = torch.tensor(scs).to(dev), torch.tensor(pts).to(dev), torch.tensor(gsc).to(dev)
    r_xy = torch.zeros(64, dtype=torch.uint8, device=dev); r_inf = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    sec = loop(lambda: eng.ecmult_multi_dev(r_xy, r_inf, d_s, d_p, g_sc=d_g))
    exp_xy, exp_inf = ref.ecmult_multi(scs, pts, gsc.tobytes())
    assert bytes(r_xy.cpu().numpy()) == exp_xy.tobytes() and int(r_inf.item()) == exp_inf, "1 024-term MSM differs from the reference's ecmult_multi_var"
    # the same call with TWO in flight (S2K_OPT_MSM_PIPELINE: the engine alternates between two stream / workspace sets for small sums;
    # S2K_OPT_RP_INPUTS_READY: the inputs are resident and untouched): what a caller with a stream of small sums sees
    eng.set_option(Engine.OPT_RP_INPUTS_READY, 1); eng.set_option(Engine.OPT_MSM_PIPELINE, 1)
    o2 = [(torch.zeros(64, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)) for _ in range(2)]
    torch.cuda.synchronize()
    for k in range(2):
        eng.ecmult_multi_dev(o2[k][0], o2[k][1], d_s, d_p, g_sc=d_g)
    device_wait(eng)
    kq = 4 * steps
    t0 = time.perf_counter()
    for k in range(kq):
        eng.ecmult_multi_dev(o2[k & 1][0], o2[k & 1][1], d_s, d_p, g_sc=d_g)
    device_wait(eng)
    sec2 = (time.perf_counter() - t0) / kq
    eng.set_option(Engine.OPT_RP_INPUTS_READY, 0); eng.set_option(Engine.OPT_MSM_PIPELINE, 0)
    assert all(bytes(o[0].cpu().numpy()) == exp_xy.tobytes() and int(o[1].item()) == exp_inf for o in o2), "1 024-term MSM with two calls in flight differs"
    out["bench_ecmult_1023p_g"] = {"metric": "one 1 024-term multi-scalar multiplication incl. G (BASELINE config 1's input shape)", "value": nm / sec / 1e6, "unit": "Mpoint-scalar/s",
                                   "ms": sec * 1e3, "terms": nm, "verified": True, "result_check": "== secp256k1_ecmult_multi_var of the reference on the same inputs",
                                   "roofline": roof(MAC64_PER_MSM_TERM_SMALL, nm, sec, "algorithmic 10.6e3 MAC64 per term (SURVEY 8d, n = 1 024)"),
                                   "two_in_flight": {"ms": sec2 * 1e3, "mpoint_scalar_per_s": nm / sec2 / 1e6, "calls": kq},
                                   "note": "a single small MSM is latency bound on a GPU (DESIGN 4.3); config 1 is the CPU reference's row, kept beside it"}
    return out


def measure_next_rows(eng, ref, dev, steps, with_cpu=True):
    """SURVEY 8(f) -- the callers either side of the hot path that the engine also serves -- in the driver-run line: surjection proofs, the
    half-aggregate verifier, Pedersen tallies, rangeproof rewinding, bppp_commit, and K independent small sums in one launch chain.  Inputs by
    the reference's own provers (oracle/_ref), resident in HBM when the clock starts; every entry carries `verified` (results == the reference's
    on the distinct inputs), a VALU roofline from the reference schedule's MAC64 count (stated per entry) and the reference function timed on ONE
    host core through oracle/ref_shim.c on a bounded sample."""
    import ctypes
    import torch
    from secp256k1_zkp_amd import Engine
    from secp256k1_zkp_amd.constants import G_XY, GENERATOR_H
    L, H = eng._lib, eng._h
    out = {}
    rng = np.random.default_rng(4321)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    D = lambda a: torch.tensor(np.ascontiguousarray(a)).to(dev)

    def loop(fn):
        # K calls queued back to back and waited for once -- twice, the faster pass counts: these loops are 5-60 ms long and run between the
        # CPU legs of this function (reference provers and verifiers on the host cores), after which one pass in ten showed a 40-60 ms hole --
        # not the engine: 1 500 consecutive groups of calls in a quiet process had none (profiles/r06y_stall_probe.txt, r06y_bench_outliers.txt);
        # later traced to torch.cuda.synchronize() waking up late (device_wait above, which these loops now end with; best-of-two kept)
        fn(); device_wait(eng)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            device_wait(eng)
            d = (time.perf_counter() - t0) / steps
            best = d if best is None or d < best else best
        return best

    def roof(mac64_per_unit, units, sec, what):
        rate = 4 * mac64_per_unit * units / sec
        return {"bound": "valu", "achieved": rate / 1e12, "peak": MAD32_PEAK / 1e12, "unit": "T lane-MAC/s (v_mad_u64_u32)", "frac": rate / MAD32_PEAK,
                "frac_of_architectural_peak": rate / MAD32_PEAK_ARCH, "note": what + " x 4 v_mad_u64_u32 x units / wall time of the whole call (K calls queued, waited for once)"}

    def cpu(fn, units, what):
        t0 = time.time(); fn(); t = time.time() - t0
        return {"value": units / t, "unit": "per second", "cores": 1, "kind": "reference", "sample": "%s, %d on one thread in %.2f s" % (what, units, t)}

    ECMULT = 45e3                                   # one secp256k1_ecmult (a*P + b*G) of the reference schedule
    SQRT = 267 * 21                                 # one square root / inversion: ~267 field operations of ~21 MAC64
    # ---- surjection proofs: 2^16 proofs of the 3-inputs / 3-used shape (a confidential transaction's usual one), 64 distinct by the reference's prover
    n = 1 << 16
    base = [ref.make_surjection(rng, 3, 3) for _ in range(64)]
    proofs = [b[0] for b in base]; tags = [b[1] for b in base]; outs = [b[2] for b in base]
    for i in (5, 21, 40):
        q = bytearray(proofs[i]); q[10 + i] ^= 1; proofs[i] = bytes(q)
    want = np.array([ref.surjection_verify(p_, t_, o_) for p_, t_, o_ in zip(proofs, tags, outs)], np.int32)
    assert want.sum() == 61
    reps = n // 64
    data, off = Engine.pack(proofs * reps)
    toff = (np.arange(n + 1) * 3).astype(np.uint64)
    d_p, d_off = D(np.concatenate([data, np.zeros(64, np.uint8)])), D(off.astype(np.int64))
    d_t, d_toff, d_o = D(np.concatenate(tags * reps)), D(toff.astype(np.int64)), D(np.stack(outs * reps))
    res = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    def call():
        assert L.secp256k1_surjectionproof_verify_batch_dev(H, None, P(res), P(d_p), P(d_off), P(d_t), P(d_toff), P(d_o), n) == 1
    sec = loop(call)
    assert np.array_equal(res.cpu().numpy(), np.tile(want, reps)), "surjection verdicts differ from the reference"
    out["surjection_2p16"] = {"metric": "surjection-proof verifies/sec (3 inputs, 3 used; secp256k1_surjectionproof_verify)", "value": n / sec, "unit": "verifies/s", "ms": sec * 1e3, "batch": n,
                              "verified": True, "result_check": "== secp256k1_surjectionproof_parse + _verify of the reference on the 64 distinct proofs (61 valid, 3 broken), tiled %d times" % reps,
                              "roofline": roof(3 * ECMULT, n, sec, "algorithmic 3 ring keys x ~45e3 MAC64 (one secp256k1_ecmult each, src/modules/surjection/main_impl.h:360-402 -> borromean_verify)")}
    if with_cpu:
        k = 1024
        out["surjection_2p16"]["cpu_baseline"] = cpu(lambda: [ref.surjection_verify(proofs[i % 64], tags[i % 64], outs[i % 64]) for i in range(k)], k,
                                                     "secp256k1_surjectionproof_verify through oracle/ref_shim.c")
    del d_p, d_off, d_t, d_toff, d_o, res
    # ---- half-aggregated Schnorr: one aggregate of 2^15 signatures (host-buffer form: the serial randomizer hash chain walks on the host under the lifting kernel)
    n = 1 << 15
    sigs, msgs, pks = ref.make_schnorr(n, rng, threads=usable_cores())
    agg = ref.halfagg_aggregate(pks, msgs, sigs)
    bad = bytearray(agg); bad[32 * 77 + 5] ^= 1
    assert eng.schnorrsig_aggverify(pks, msgs, bytes(bad)) == 0
    sec = None
    for _ in range(2):                              # (the faster of two passes: see loop())
        t0 = time.perf_counter()
        for _ in range(steps):
            ok = eng.schnorrsig_aggverify(pks, msgs, agg)
        d_ = (time.perf_counter() - t0) / steps
        sec = d_ if sec is None or d_ < sec else sec
    k = 1 << 11
    agg_k = ref.halfagg_aggregate(pks[:k], msgs[:k], sigs[:k])
    assert ok == 1 and eng.schnorrsig_aggverify(pks[:k], msgs[:k], agg_k) == ref.halfagg_verify(pks[:k], msgs[:k], agg_k) == 1
    out["halfagg_2p15"] = {"metric": "half-aggregate signatures verified/sec (one aggregate of 2^15 signatures, secp256k1_schnorrsig_aggverify; host buffers, H2D included)", "value": n / sec, "unit": "signatures/s",
                           "ms": sec * 1e3, "batch": n, "verified": True,
                           "result_check": "accepts the reference's aggregate, rejects it with one bit of r_77 flipped; a %d-signature aggregate == the reference's verdict" % k,
                           "roofline": roof(2 * ECMULT + 2 * SQRT, n, sec, "algorithmic two single multiplications (z_i R_i, z_i e_i P_i) + two x-lifts per signature (src/modules/schnorrsig_halfagg/main_impl.h:108-198)")}
    if with_cpu:
        out["halfagg_2p15"]["cpu_baseline"] = cpu(lambda: ref.halfagg_verify(pks[:k], msgs[:k], agg_k), k, "secp256k1_schnorrsig_aggverify through oracle/ref_shim.c (one aggregate)")
    # ---- Pedersen tallies: 2^15 balances of 2 inputs / 3 outputs (secp256k1_pedersen_verify_tally), 64 distinct, 4 unbalanced
    n = 1 << 15
    tl = [ref.make_balanced_tally(rng, 2, 3) for _ in range(64)]
    for i in (3, 30, 31, 60):
        tl[i] = (tl[i][0], np.concatenate([tl[i][1][:2], tl[(i + 1) % 64][1][2:]]))
    want = ref.pedersen_verify_tally_many(tl)
    assert want.sum() == 60
    reps = n // 64
    cm = np.concatenate([np.concatenate([a, b]) for a, b in tl] * reps)
    toff = (np.arange(n + 1) * 5).astype(np.uint64); npos = np.full(n + 1, 2, np.uint64)
    d_c = D(cm); res = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    def call():
        assert L.secp256k1_pedersen_verify_tally_batch_dev(H, None, P(res), P(d_c), toff.ctypes.data_as(ctypes.c_void_p), npos.ctypes.data_as(ctypes.c_void_p), n) == 1
    sec = loop(call)
    assert np.array_equal(res.cpu().numpy(), np.tile(want, reps)), "tally verdicts differ from the reference"
    out["tally_2p15"] = {"metric": "Pedersen tallies verified/sec (2 inputs, 3 outputs; secp256k1_pedersen_verify_tally)", "value": n / sec, "unit": "tallies/s", "ms": sec * 1e3, "batch": n,
                         "verified": True, "result_check": "== secp256k1_pedersen_verify_tally of the reference on the 64 distinct tallies (60 balanced, 4 not), tiled %d times" % reps,
                         "roofline": roof(5 * SQRT + 5 * 16 * 21, n, sec, "algorithmic five commitment loads (a square root each) + five point additions per tally (src/modules/generator/main_impl.h:371-396)")}
    if with_cpu:
        k = 4096
        out["tally_2p15"]["cpu_baseline"] = cpu(lambda: ref.pedersen_verify_tally_many([tl[i % 64] for i in range(k)]), k, "secp256k1_pedersen_verify_tally through oracle/ref_shim.c")
    del d_c, res
    # ---- rangeproof rewinding: 2^12 64-bit proofs with a 64-byte message each, verification + recovery (secp256k1_rangeproof_rewind)
    n = 1 << 12
    c, p, g, vals, blinds, nonces, msgs_in = ref.make_rangeproofs_msg(64, rng, msg_len=64, min_bits=64, threads=usable_cores())
    nonces[7, 3] ^= 0x10
    q = bytearray(p[9]); q[len(q) // 2] ^= 1; p[9] = bytes(q)
    e_res, e_bl, e_val, e_msgs, e_mn, e_mx = ref.rangeproof_rewind_many(c, p, g, nonces, msg_capacity=128, threads=usable_cores())
    assert e_res.sum() == 62
    reps = n // 64
    data, off = Engine.pack(p * reps)
    d_c, d_p, d_off, d_g, d_n = D(np.tile(c, (reps, 1))), D(np.concatenate([data, np.zeros(64, np.uint8)])), D(off.astype(np.int64)), D(np.tile(g, (reps, 1))), D(np.tile(nonces, (reps, 1)))
    res = torch.zeros(n, dtype=torch.int32, device=dev); bl = torch.zeros(n, 32, dtype=torch.uint8, device=dev); val = torch.zeros(n, dtype=torch.int64, device=dev)
    mo = torch.zeros(n, 128, dtype=torch.uint8, device=dev); ol = torch.full((n,), 128, dtype=torch.int64, device=dev); mn = torch.zeros(n, dtype=torch.int64, device=dev); mx = torch.zeros(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    def call():
        ol.fill_(128)
        assert L.secp256k1_rangeproof_rewind_batch_dev(H, None, P(res), P(bl), P(val), P(mo), P(ol), 128, P(d_n), P(mn), P(mx), P(d_c), P(d_p), P(d_off), None, None, P(d_g), n) == 1
    sec = loop(call)
    got = res.cpu().numpy(); okm = np.tile(e_res, reps) == 1
    assert np.array_equal(got, np.tile(e_res, reps)) and np.array_equal(bl.cpu().numpy()[okm], np.tile(e_bl, (reps, 1))[okm]) and \
        np.array_equal(val.cpu().numpy().view(np.uint64)[okm], np.tile(e_val, reps)[okm]), "rewind results differ from the reference"
    out["rewind_2p12"] = {"metric": "rangeproofs rewound/sec (64-bit proofs, verification + recovery of value, blinding factor and message; secp256k1_rangeproof_rewind)", "value": n / sec,
                          "unit": "rewinds/s", "ms": sec * 1e3, "batch": n, "verified": True,
                          "result_check": "== secp256k1_rangeproof_rewind of the reference on the 64 distinct proofs (62 recovered, one wrong nonce, one broken proof): verdicts, blinding factors, values; tiled %d times" % reps,
                          "roofline": roof(MAC64_PER_PROOF, n, sec, "algorithmic 6.6e6 MAC64 per proof for the verification inside the rewind (SURVEY 8d); the replay of the prover's random stream (~1 500 SHA-256 compressions) is not MAC work")}
    if with_cpu:
        k = 32
        out["rewind_2p12"]["cpu_baseline"] = cpu(lambda: ref.rangeproof_rewind_many(c[:k], p[:k], g[:k], nonces[:k], msg_capacity=128, threads=1), k, "secp256k1_rangeproof_rewind through oracle/ref_shim.c")
    del d_c, d_p, d_off, d_g, d_n, res, bl, val, mo, ol, mn, mx
    # ---- bppp_commit: 2^12 commitments over the 64 + 8 generator set of config 4, on the set's fixed-base tables
    n = 1 << 12
    g_len, h_len = 64, 8
    gens = ref.bppp_generators(g_len + h_len)
    sc = lambda *shape: (rng.integers(0, 256, shape + (32,), dtype=np.uint8) & np.array([0x7F] + [0xFF] * 31, np.uint8))
    nv, lv, cv, mu = sc(n, g_len), sc(n, h_len), sc(n, h_len), sc(n)
    d_nv, d_lv, d_cv, d_mu, d_gens = D(nv), D(lv), D(cv), D(mu), D(gens)
    d_out = torch.zeros(n, 33, dtype=torch.uint8, device=dev); res = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    def call():
        assert L.secp256k1_bppp_commit_batch_dev(H, None, P(d_out), P(res), P(d_gens), gens.ctypes.data_as(ctypes.c_void_p), g_len + h_len, g_len, P(d_nv), P(d_lv), P(d_cv), h_len, P(d_mu), n) == 1
    sec = loop(call)
    got = d_out.cpu().numpy()
    def ref_commit(i):
        cmt = np.zeros(33, np.uint8)
        assert ref.lib.ref_bppp_commit(cmt.ctypes.data_as(ctypes.c_void_p), gens.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(g_len + h_len), np.ascontiguousarray(nv[i]).ctypes.data_as(ctypes.c_void_p),
                                       ctypes.c_size_t(g_len), np.ascontiguousarray(lv[i]).ctypes.data_as(ctypes.c_void_p), np.ascontiguousarray(cv[i]).ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(h_len),
                                       np.ascontiguousarray(mu[i]).ctypes.data_as(ctypes.c_void_p)) == 1
        return cmt
    for i in range(0, n, 128):
        assert np.array_equal(got[i], ref_commit(i)), "bppp_commit differs from the reference"
    assert res.cpu().numpy().all()
    terms = g_len + h_len + 1
    out["bppp_commit_2p12"] = {"metric": "BP++ commitments/sec (secp256k1_bppp_commit, g_len 64, h_len 8)", "value": n / sec, "unit": "commitments/s", "ms": sec * 1e3, "batch": n, "verified": True,
                               "result_check": "every 128th commitment == the reference's secp256k1_bppp_commit (static, through oracle/ref_shim.c), byte for byte",
                               "roofline": roof(MAC64_PER_MSM_TERM_SMALL * terms, n, sec, "algorithmic one %d-term secp256k1_ecmult_multi_var per commitment x 10.6e3 MAC64 per term (src/modules/bppp/bppp_norm_product_impl.h:105-151)" % terms)}
    if with_cpu:
        k = 128
        out["bppp_commit_2p12"]["cpu_baseline"] = cpu(lambda: [ref_commit(i) for i in range(k)], k, "secp256k1_bppp_commit through oracle/ref_shim.c")
    del d_nv, d_lv, d_cv, d_mu, d_out, res
    # ---- K independent small sums in one launch chain: 256 sums of bench_ecmult's shape (1 024 terms + G each)
    K, nm = 256, 1024
    ks = rng.integers(0, 256, (nm, 32), dtype=np.uint8)
    pts, pinf = ref.ecmult_batch(np.tile(np.frombuffer(G_XY, np.uint8), (nm, 1)), ks)
    scs = rng.integers(0, 256, (K * nm, 32), dtype=np.uint8); gs = rng.integers(0, 256, (K, 32), dtype=np.uint8)
    d_s, d_p, d_g = D(scs), D(np.tile(pts, (K, 1))), D(gs)
    r_xy = torch.zeros(K, 64, dtype=torch.uint8, device=dev); r_inf = torch.zeros(K, dtype=torch.int32, device=dev)
    offs = (np.arange(K + 1) * nm).astype(np.uint64)
    torch.cuda.synchronize()
    sec = loop(lambda: eng.ecmult_multi_many_dev(r_xy, r_inf, d_s, d_p, offs, d_g))
    got = r_xy.cpu().numpy(); gi = r_inf.cpu().numpy()
    for s_ in range(0, K, 37):
        exp_xy, exp_inf = ref.ecmult_multi(scs[s_ * nm:(s_ + 1) * nm], pts, gs[s_].tobytes())
        assert bytes(got[s_]) == exp_xy.tobytes() and int(gi[s_]) == exp_inf, "batched 1 024-term MSM differs from the reference's ecmult_multi_var"
    out["ecmult_multi_many_256x1024"] = {"metric": "256 independent 1 024-term multi-scalar multiplications incl. G in one launch chain (s2k_ecmult_multi_many_dev; BASELINE config 1's shape, batched)",
                                         "value": K * nm / sec / 1e6, "unit": "Mpoint-scalar/s", "ms": sec * 1e3, "sums": K, "terms_per_sum": nm, "verified": True,
                                         "result_check": "every 37th sum == secp256k1_ecmult_multi_var of the reference on the same terms",
                                         "roofline": roof(MAC64_PER_MSM_TERM_SMALL, K * nm, sec, "algorithmic 10.6e3 MAC64 per term (SURVEY 8d, n = 1 024)")}
    return out


def measure_group(devices, commits, proofs, gens, ref, steps):
    """The C-ABI engine group (s2k_group_*, include/secp256k1_zkp_amd.h) over `devices`, from ONE process: (i) replica dispatch -- `len(devices)`
    times the headline batch handed over in HOST memory, cut into one contiguous range per engine, wall clock of the whole call; (ii) one
    2^20-term multi-scalar multiplication sharded by terms, slices resident on their GPUs, 112-byte partials gathered on the first device
    (hipMemcpyPeerAsync) and summed there.  Both checked in-run."""
    import torch
    from secp256k1_zkp_amd import Group
    from secp256k1_zkp_amd.constants import G_XY, N as ORDER
    k = len(devices)
    g = Group(devices)
    out = {"devices": list(devices), "engines": k}
    try:
        n = len(proofs)
        C = np.ascontiguousarray(np.tile(commits, (k, 1))); G = np.ascontiguousarray(np.tile(gens, (k, 1))); P = Group_pack(proofs * k)
        for e in range(k):
            g.engine(e).cache_generator(bytes(gens[0]))
        res, mn, mx = g.rangeproof_verify_batch(C, P, G)                      # warm-up: staging buffers, tables
        assert res.all()
        t0 = time.perf_counter()
        for _ in range(steps):
            res, mn, mx = g.rangeproof_verify_batch(C, P, G)
        dt = (time.perf_counter() - t0) / steps
        assert res.all() and int(mx.min()) == 2**64 - 1
        out["rangeproof_host_buffers"] = {"proofs_per_call": n * k, "ms_per_call": dt * 1e3, "verifies_per_s": n * k / dt,
                                          "note": "secp256k1_rangeproof_verify_batch_group: packing into pinned staging, H2D, kernels, D2H on every device concurrently; synchronous calls"}
        # ---- one 2^20-term sum, term-sharded
        nm = 1 << 20
        rng = np.random.default_rng(4242)
        ks = rng.integers(0, 256, (nm, 32), dtype=np.uint8); sc = rng.integers(0, 256, (nm, 32), dtype=np.uint8)
        e0 = g.engine(0); d0 = torch.device("cuda", devices[0])
        gp = torch.tensor(np.frombuffer(G_XY, np.uint8).copy()).to(d0).repeat(nm, 1)
        pts = torch.zeros(nm, 64, dtype=torch.uint8, device=d0); pinf = torch.zeros(nm, dtype=torch.int32, device=d0)
        torch.cuda.synchronize(d0)
        e0.ecmult_batch_dev(pts, pinf, gp, torch.zeros(nm, 32, dtype=torch.uint8, device=d0), torch.tensor(ks).to(d0)); e0.sync()
        cut = [nm * i // k for i in range(k + 1)]
        pts_h = pts.cpu()
        scl = [torch.tensor(sc[cut[i]:cut[i + 1]]).to(torch.device("cuda", devices[i])) for i in range(k)]
        ptl = [pts_h[cut[i]:cut[i + 1]].to(torch.device("cuda", devices[i])) for i in range(k)]
        for d in set(devices):
            torch.cuda.synchronize(torch.device("cuda", d))
        xy, inf = g.ecmult_multi_dev(scl, ptl)
        t0 = time.perf_counter()
        for _ in range(steps):
            xy, inf = g.ecmult_multi_dev(scl, ptl)
        dtm = (time.perf_counter() - t0) / steps
        tot = sum(int.from_bytes(ks[i].tobytes(), "big") * int.from_bytes(sc[i].tobytes(), "big") for i in range(nm)) % ORDER
        verified = False
        if ref is not None:
            exp_xy, exp_inf = ref.ecmult_batch(np.frombuffer(G_XY, np.uint8), np.zeros(32, np.uint8), ng=np.frombuffer(tot.to_bytes(32, "big"), np.uint8), a_inf=np.ones(1, np.uint8))
            assert inf == int(exp_inf[0]) and xy.tobytes() == exp_xy[0].tobytes(), "group MSM differs from (sum s_i k_i)*G"
            verified = True
        out["msm_2p20_term_sharded"] = {"terms": nm, "ms": dtm * 1e3, "mpoint_scalar_per_s": nm / dtm / 1e6, "verified": verified,
                                        "frac": 4 * MAC64_PER_MSM_TERM * nm / dtm / (MAD32_PEAK * len(set(devices))),
                                        "note": "s2k_ecmult_multi_group_dev: host call to host result (one partial per engine, peer copies to the first device, sum there)"}
    finally:
        g.close()
    return out


def Group_pack(items):
    from secp256k1_zkp_amd import Engine
    return Engine.pack(items)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _usable_cpu_list():
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    return cpus[:usable_cores()]


def _run_ref_bench(binary, args, iters, cpu=None, timeout=300):
    """one run of a reference bench program (oracle/_ref/, built by `make -C oracle benches` from the reference's src/bench_*.c) ->
    {name: (min, avg, max)} in microseconds per iteration as src/bench.h:78-111 prints them"""
    import subprocess
    cmd = [binary] + list(args)
    if cpu is not None:
        cmd = ["taskset", "-c", str(cpu)] + cmd
    env = dict(os.environ, SECP256K1_BENCH_ITERS=str(iters))
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout).stdout
    res = {}
    for line in out.splitlines():
        parts = [x.strip() for x in line.split(",")]
        if len(parts) == 4:
            try:
                res[parts[0]] = tuple(float(x) for x in parts[1:])
            except ValueError:
                pass
    return res


def cpu_baseline_ref_benches():
    """The CPU column from the reference's OWN benchmark programs on this box's host cores:
    bench_rangeproof (as shipped: min_bits = 32, and the min_bits = 64 variant BASELINE's metric names), one process and one
    taskset-pinned process per usable core; bench_ecmult pippenger_wnaf, one process.  None when the binaries did not travel."""
    import concurrent.futures
    d = os.path.join(ROOT, "oracle", "_ref")
    b64, b32, bec = (os.path.join(d, x) for x in ("bench_rangeproof_64", "bench_rangeproof", "bench_ecmult"))
    if not all(os.path.exists(x) and os.access(x, os.X_OK) for x in (b64, b32, bec)):
        return None
    out = {"cpu_model": cpu_model(), "cores": usable_cores(), "hardware_threads": os.cpu_count()}
    iters = 12                                  # verifications per repetition (x 10 repetitions, src/bench.h): ~0.6 s per process
    r32 = _run_ref_bench(b32, [], iters).get("rangeproof_verify_bit")
    r64 = _run_ref_bench(b64, [], iters).get("rangeproof_verify_bit")
    if not r32 or not r64:
        return None
    out["bench_rangeproof"] = {"min_bits": 32, "us_per_bit_min_avg_max": r32, "verifies_per_s_one_process": 1e6 / (32 * r32[1])}
    out["bench_rangeproof_64"] = {"min_bits": 64, "us_per_bit_min_avg_max": r64, "verifies_per_s_one_process": 1e6 / (64 * r64[1])}
    cpus = _usable_cpu_list()
    t = time.time()
    with concurrent.futures.ThreadPoolExecutor(len(cpus)) as ex:
        rs = list(ex.map(lambda c: _run_ref_bench(b64, [], 3 * iters, cpu=c).get("rangeproof_verify_bit"), cpus))
    wall = time.time() - t
    rs = [r for r in rs if r]
    out["bench_rangeproof_64_all_cores"] = {"processes": len(rs), "pinned_to": cpus, "wall_s": wall,
                                            "verifies_per_s": sum(1e6 / (64 * r[1]) for r in rs),
                                            "us_per_bit_avg_min_max_over_processes": (min(r[1] for r in rs), max(r[1] for r in rs))}
    bmain = os.path.join(d, "bench")                                           # src/bench.c (public API): BIP-340 verification per item
    if os.path.exists(bmain) and os.access(bmain, os.X_OK):
        sv = _run_ref_bench(bmain, ["schnorrsig_verify"], 4000).get("schnorrsig_verify")
        if sv:
            out["bench_schnorrsig_verify"] = {"us_per_verify_min_avg_max": sv, "verifies_per_s_one_process": 1e6 / sv[1]}
    ec = _run_ref_bench(bec, ["pippenger_wnaf"], 400)
    big = ec.get("ecmult_multi_32767p_g")
    if big:
        out["bench_ecmult_pippenger_32767p_g"] = {"us_per_point_min_avg_max": big, "mpoint_scalar_per_s": 1.0 / big[1]}
    k1 = ec.get("ecmult_multi_1023p_g")                                        # BASELINE config 1: 1024 (scalar, point) pairs, Pippenger, CPU only
    if k1:
        out["bench_ecmult_pippenger_1023p_g"] = {"us_per_point_min_avg_max": k1, "mpoint_scalar_per_s": 1.0 / k1[1]}
    return out


def cpu_baseline(ref, commits, proofs, gens):
    """the reference's secp256k1_rangeproof_verify on host cores, bounded sample (~10-20 s of CPU work)."""
    if ref is None:
        return cpu_baseline_port(commits, proofs, gens)
    cores = usable_cores()
    k1 = min(256, len(proofs))
    t = time.time(); r, _, _ = ref.rangeproof_verify_many(commits[:k1], proofs[:k1], gens[:k1], threads=1); t1 = time.time() - t
    assert r.all()
    kn = min(len(proofs), max(256, 256 * cores))
    ref.rangeproof_verify_many(commits[:cores], proofs[:cores], gens[:cores], threads=cores)      # spin up the OpenMP team
    t = time.time(); r, _, _ = ref.rangeproof_verify_many(commits[:kn], proofs[:kn], gens[:kn], threads=cores); tn = time.time() - t
    assert r.all()
    return {"value": kn / tn, "unit": "verifies/s", "cores": cores, "kind": "reference",
            "per_core": k1 / t1,
            "sample": "%d 64-bit proofs on %d threads (%.2f s) = %.0f verifies/s; single thread: %d proofs in %.2f s = %.1f verifies/s per core (the reference is single-threaded; cores = cgroup CPU quota of the box)" % (kn, cores, tn, kn / tn, k1, t1, k1 / t1)}


def main():
    t_process = time.time()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-msm", action="store_true")
    ap.add_argument("--no-msm-big", action="store_true", help="skip the 2^24-term strong-scaling MSM entry")
    ap.add_argument("--no-distinct", action="store_true", help="skip the second timed loop (every proof its own generator)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the host-memory (drop-in path) timing")
    ap.add_argument("--no-group", action="store_true", help="skip the single-process C-ABI engine-group block (rank 0 over all the run's GPUs)")
    ap.add_argument("--no-secondary", action="store_true", help="skip BASELINE configs 1, 2 and 4 (bench_ecmult 1024 pairs, BIP-340 2^16, BP++ norm argument 2^12)")
    ap.add_argument("--no-widths", action="store_true", help="skip the headline at 24- and 20-bit fixed-base tables")
    ap.add_argument("--no-next", action="store_true", help="skip the SURVEY 8(f) rows (surjection, half-aggregate, tallies, rewind, bppp_commit, batched small sums)")
    ap.add_argument("--rp-split", type=int, choices=(0, 1), default=None, help="S2K_OPT_RP_SPLIT of the engine (A/B of the two forms of the ring kernel's double multiplication: tools/profile_mem_counters.sh)")
    args = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the hosts only support dmabuf IPC: RCCL between the ranks fails without it (set here, not at import: tests import this module)

    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    # one process per GPU over RCCL ("nccl"); S2K_DIST_BACKEND=gloo lets the N>1 code path be exercised on a box with
    # fewer GPUs than ranks (ranks then share devices) -- for testing only, never for reported numbers
    backend = os.environ.get("S2K_DIST_BACKEND", "nccl")
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if backend == "nccl" and torch.cuda.device_count() < args.gpus:
        sys.exit("bench.py: --gpus %d but this node has %d GPU(s): one rank per GPU over RCCL needs %d devices "
                 "(S2K_DIST_BACKEND=gloo lets ranks share a device, for testing only)" % (args.gpus, torch.cuda.device_count(), args.gpus))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher of N ranks, one per GPU (exactly the command the docstring shows);
        # rank 0's JSON line goes to this process's stdout.  Under torch.distributed.run (WORLD_SIZE set) this branch is not taken.
        # (--standalone: the launcher picks its own rendezvous port -- probing for a free one here and handing it over is a race when
        #  several benches start on one node)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush(); sys.stderr.flush()
        os.execvpe(cmd[0], cmd, dict(os.environ))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): the two must agree" % (args.gpus, world))
    local = local % torch.cuda.device_count()
    rank_devices, collective_world = [local], 1
    if world > 1:
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        # proof that the collective library saw every rank: a checked all-reduce (sum of rank + 1) and the ranks' device ids gathered
        chk = torch.tensor([rank + 1], dtype=torch.int64, device=torch.device("cuda", local)); dist.all_reduce(chk)
        collective_world = dist.get_world_size()
        assert int(chk.item()) == world * (world + 1) // 2 and collective_world == world, "all-reduce over the ranks gave %d for world %d" % (int(chk.item()), world)
        ids = [torch.zeros(1, dtype=torch.int64, device=torch.device("cuda", local)) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([local], dtype=torch.int64, device=torch.device("cuda", local)))
        rank_devices = [int(t.item()) for t in ids]
        if backend == "nccl":
            assert sorted(rank_devices) == list(range(world)), "ranks do not own distinct devices: %r" % (rank_devices,)
    from secp256k1_zkp_amd import Engine
    eng = Engine(local)
    if args.rp_split is not None: eng.set_option(Engine.OPT_RP_SPLIT, args.rp_split)
    torch.cuda.set_device(local)
    # the device's fixed-base table of G, built now (first use) so that its cost is a line of the record rather than part of a warm-up step
    import ctypes
    t_tab = time.perf_counter(); tab_sz = ctypes.c_size_t(0)
    assert eng._lib.s2k_engine_gtable(eng._h, ctypes.byref(tab_sz)), "no generator table"
    tables = {"digit_bits": int(eng._lib.s2k_engine_gtable_bits(eng._h)), "bytes_per_table": int(tab_sz.value),
              "first_use_ms_incl_allocation": (time.perf_counter() - t_tab) * 1e3, "build_ms_device": float(eng._lib.s2k_engine_gtable_build_ms(eng._h)),
              "note": "table of G: allocated and built on the device by the first call that needs it (csrc/gtable.h: seeds + one affine addition per entry with shared inversions); "
                      "build_ms_device = the construction kernels by HIP events; the wall figure also holds the allocation, which takes seconds when another process has just freed tens of GB on the device"}
    n = args.batch
    dev = torch.device("cuda", local)
    if world > 1 and n % world == 0:
        # signing 2^14 proofs costs ~10 s of host CPU: every rank signs n/world of them and the pieces are all-gathered, so
        # that each rank ends up with the same n unique proofs (its replica of the batch) without N-fold host work
        nl = n // world
        c_l, p_l, g_l, data_desc, ref = make_inputs(nl, seed=1234 + rank)
        plen = len(p_l[0]); assert all(len(p) == plen for p in p_l)
        loc = torch.tensor(np.concatenate([np.ascontiguousarray(c_l).reshape(nl, 33), np.frombuffer(b"".join(p_l), np.uint8).reshape(nl, plen),
                                           np.ascontiguousarray(g_l).reshape(nl, 64)], axis=1)).to(dev)
        parts = [torch.empty_like(loc) for _ in range(world)]
        dist.all_gather(parts, loc)
        allr = torch.cat(parts).reshape(n, 33 + plen + 64).cpu().numpy()
        commits = np.ascontiguousarray(allr[:, :33]); gens = np.ascontiguousarray(allr[:, 33 + plen:])
        proofs = [allr[i, 33:33 + plen].tobytes() for i in range(n)]
        data_desc = data_desc.replace("%d unique" % nl, "%d unique" % n)
    else:
        commits, proofs, gens, data_desc, ref = make_inputs(n, seed=1234 + rank)
    pdata, poff = Engine.pack(proofs)
    d_commits = torch.tensor(commits).to(dev); d_gens = torch.tensor(np.ascontiguousarray(gens)).to(dev)
    d_proofs = torch.tensor(np.concatenate([pdata, np.zeros(64, np.uint8)])).to(dev); d_off = torch.tensor(poff.astype(np.int64)).to(dev)
    d_res = torch.zeros(n, dtype=torch.int32, device=dev); d_min = torch.zeros(n, dtype=torch.int64, device=dev); d_max = torch.zeros(n, dtype=torch.int64, device=dev)
    stream = None                         # the engine's own stream (HIP events for the roofline are recorded on it)
    # Calls in flight.  The input arrays of this benchmark are resident and never touched again, which is exactly what the engine's
    # S2K_OPT_RP_INPUTS_READY contract asks the caller to promise: the first stage of step k+1 (header parse, lifts, message hash, key sum:
    # ~1 ms of small kernels) then runs on side streams underneath the ring kernel of step k instead of waiting for it.  Every step still
    # does all of its work inside the timed region.  That mode gives `value`; the same loop with the engine's default contract (the first
    # stage of a call waits for everything queued before it) gives `value_serialized_calls` and the ring kernel's undisturbed time for
    # the roofline.  S2K_BENCH_PIPELINE=0 makes the default contract the headline.
    pipeline = os.environ.get("S2K_BENCH_PIPELINE", "1") != "0"
    torch.cuda.synchronize()              # ...which is not ordered against torch's streams: inputs must be resident first

    def step():
        eng.rangeproof_verify_batch_dev(d_res, d_min, d_max, d_commits, d_proofs, d_off, d_gens, n, stream=stream)

    def timed(k_steps, in_flight):
        eng.set_option(Engine.OPT_RP_INPUTS_READY, 1 if in_flight else 0)
        for _ in range(args.warmup):
            step()
        device_wait(eng)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # the K steps are queued back to back and waited for once (stream-ordered `_dev` calls: no host round trip between steps)
        for _ in range(k_steps):
            step()
        device_wait(eng)                  # (= the engine's wait on its stream, then torch.cuda.synchronize())
        d = time.perf_counter() - t0
        return d, [eng.last_ms(16 + k) for k in range(min(k_steps, 32))]          # HIP events around the ring kernels on the launch stream

    ser_steps = max(2, min(args.steps, 10))
    dt_ser, kern_ms = timed(ser_steps, False)
    if pipeline:
        dt, kern_ms_pipe = timed(args.steps, True)
    else:
        dt, kern_ms_pipe = timed(args.steps, False)
        kern_ms = kern_ms_pipe
    eng.set_option(Engine.OPT_RP_INPUTS_READY, 0)
    if world > 1:
        tser = torch.tensor([dt_ser], dtype=torch.float64, device=dev); dist.all_reduce(tser, op=dist.ReduceOp.MAX); dt_ser = float(tser.item())
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        ok = torch.tensor([int(d_res.all().item())], device=dev); dist.all_reduce(ok, op=dist.ReduceOp.MIN); all_ok = bool(ok.item())
    else:
        all_ok = bool(d_res.all().item())
    assert all_ok, "a valid proof was rejected"
    assert int(d_max.min().item()) == -1          # max_value == 2^64-1 for every 64-bit proof

    # The same batch size with a DIFFERENT generator for every proof (the Elements case: one blinded asset generator per confidential
    # output).  No fixed-base generator table applies there (secp256k1_zkp_amd.h, s2k_engine_cache_generator), so every ring takes the
    # general form of the rings kernel: reported next to the headline so that the shared-generator figure cannot be misread.
    distinct = None
    if not args.no_distinct and ref is not None:
        c2, p2, g2, _, _ = make_inputs(n, seed=4321 + rank, distinct_generators=True)
        pd2, po2 = Engine.pack(p2)
        d_c2 = torch.tensor(c2).to(dev); d_g2 = torch.tensor(np.ascontiguousarray(g2)).to(dev)
        d_p2 = torch.tensor(np.concatenate([pd2, np.zeros(64, np.uint8)])).to(dev); d_o2 = torch.tensor(po2.astype(np.int64)).to(dev)
        torch.cuda.synchronize()
        for _ in range(max(1, args.warmup)):
            eng.rangeproof_verify_batch_dev(d_res, d_min, d_max, d_c2, d_p2, d_o2, d_g2, n, stream=stream)
        device_wait(eng)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0d = time.perf_counter()
        for _ in range(args.steps):
            eng.rangeproof_verify_batch_dev(d_res, d_min, d_max, d_c2, d_p2, d_o2, d_g2, n, stream=stream)
        device_wait(eng)
        dtd = time.perf_counter() - t0d
        if world > 1:
            tmax = torch.tensor([dtd], dtype=torch.float64, device=dev); dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dtd = float(tmax.item())
        assert bool(d_res.all().item()), "a valid proof (own generator) was rejected"
        distinct = {"value": world * n * args.steps / dtd, "ms_per_step": dtd / args.steps * 1e3}
        del d_c2, d_g2, d_p2, d_o2

    # The drop-in path: the same 2^14 proofs handed over in HOST memory, timed from the call to the return (packing into pinned staging,
    # H2D, the kernels, D2H all inside): (i) the engine's host-buffer entry point on packed numpy arrays, (ii) the reference-side adapter
    # secp256k1_amd_rangeproof_verify_batch of libsecp256k1_hooked.so (the unmodified reference + integration/secp256k1_amd_hook.c) with the
    # reference's own types -- arrays of pointers to secp256k1_pedersen_commitment / secp256k1_generator objects and to the proofs.
    dropin = None
    if rank == 0 and not args.no_dropin:
        dropin = measure_dropin(eng, commits, proofs, gens, steps=max(3, min(args.steps, 5)))
        dropin["resident_verifies_per_s"] = n * args.steps / dt

    # secondary figure of the BASELINE metric: one 2^20-term MSM (config 5), terms sharded over the ranks, partial
    # Jacobian sums all-gathered as raw limbs (RCCL) and summed locally -- strong scaling, reported next to the headline.
    msm = None
    if not args.no_msm:
        # (the loops above leave the board at its power limit; a different workload is timed from an idle board, as a caller would meet it)
        torch.cuda.synchronize(); time.sleep(1.0)
        from secp256k1_zkp_amd import parallel
        from secp256k1_zkp_amd.constants import G_XY, N as ORDER
        be = parallel.EngineBackend(eng)
        msm_fn = parallel.msm_sharded if (world == 1 or os.environ.get("S2K_MSM_SHARDING", "terms") == "terms") else parallel.msm_window_sharded

        def msm_inputs(nm, seed):
            rng = np.random.default_rng(seed)
            ks_h = rng.integers(0, 256, (nm, 32), dtype=np.uint8)
            ks = torch.tensor(ks_h).to(dev)
            gpts = torch.tensor(np.frombuffer(G_XY, np.uint8).copy()).to(dev).repeat(nm, 1)
            pts = torch.zeros(nm, 64, dtype=torch.uint8, device=dev); pinf = torch.zeros(nm, dtype=torch.int32, device=dev)
            zero_na = torch.zeros(nm, 32, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()          # the engine's stream is not ordered against torch's: inputs must be complete first
            eng.ecmult_batch_dev(pts, pinf, gpts, zero_na, ks, stream=stream)   # P_i = k_i*G
            scs_h = rng.integers(0, 256, (nm, 32), dtype=np.uint8)
            scs = torch.tensor(scs_h).to(dev)
            eng.sync(); torch.cuda.synchronize()
            del gpts, zero_na, ks
            return ks_h, scs_h, scs, pts

        def msm_time(scs, pts):
            """K sharded MSMs queued back to back (partial -> all-gather -> sum is one stream-ordered chain; nothing waits for the GPU inside
            the timed region), waited for once"""
            for _ in range(max(1, args.warmup)):          # (W untimed calls, as for the headline)
                msm_fn(be, scs, pts, to_host=False)
            device_wait(eng)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            tm = time.perf_counter()
            for _ in range(args.steps):
                xy_d, inf_d = msm_fn(be, scs, pts, to_host=False)
            device_wait(eng)
            dtm = time.perf_counter() - tm
            if world > 1:
                tmax = torch.tensor([dtm], dtype=torch.float64, device=dev); dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dtm = float(tmax.item())
            return dtm / args.steps * 1e3, bytes(xy_d.cpu().numpy()), int(inf_d.item())

        def block(nm, ms, xy):
            mad_rate = 4 * MAC64_PER_MSM_TERM * nm / (ms * 1e-3)
            return {"terms": nm, "terms_per_rank": (nm + world - 1) // world, "ms": ms, "mpoint_scalar_per_s": nm / (ms * 1e-3) / 1e6, "scaling": "strong",
                    "frac": mad_rate / (MAD32_PEAK * world),
                    "roofline": {"bound": "valu", "achieved": mad_rate / 1e12, "peak": MAD32_PEAK * world / 1e12, "unit": "T lane-MAC/s (v_mad_u64_u32)",
                                 "frac": mad_rate / (MAD32_PEAK * world), "note": "algorithmic 6.3e3 MAC64/term (reference schedule) x 4; whole call incl. the exchange, host to host"},
                    "hbm_roofline": {"achieved": MSM_BYTES_PER_TERM * nm / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                     "frac": MSM_BYTES_PER_TERM * nm / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), "algorithmic_bytes_per_term": MSM_BYTES_PER_TERM},
                    "result_x": xy[:8].hex()}

        # ---- 2^20 terms (BASELINE config 5)
        nm = 1 << 20
        ks_h, scs_h, scs, pts = msm_inputs(nm, 99)
        ms, xy, minf = msm_time(scs, pts)
        # in-run check of the timed result: P_i = k_i*G, so sum s_i*P_i must be (sum s_i*k_i mod n)*G -- one generator multiplication
        # by the reference (when oracle/_ref travelled; otherwise the block is marked unverified)
        checker = None
        if rank == 0:
            to_int = lambda a: [int.from_bytes(a[i].tobytes(), "big") for i in range(a.shape[0])]
            tot = sum(x * y for x, y in zip(to_int(ks_h), to_int(scs_h))) % ORDER
            tot_b = np.frombuffer(tot.to_bytes(32, "big"), np.uint8)
            if ref is not None:
                exp_xy, exp_inf = ref.ecmult_batch(np.frombuffer(G_XY, np.uint8), np.zeros(32, np.uint8), ng=tot_b, a_inf=np.ones(1, np.uint8)); checker = "reference secp256k1_ecmult"
                assert minf == int(exp_inf[0]) and xy == exp_xy[0].tobytes(), "MSM result differs from (sum s_i k_i)*G"
        msm = block(nm, ms, xy)
        msm["sharding"] = (("terms over ranks, all-gather of %d x 112 B Jacobian partials + local sum" % world) if msm_fn is parallel.msm_sharded
                           else ("bucket windows over ranks, all-gather of per-window Jacobian sums + local Horner, %d ranks" % world))
        msm["verified"] = checker is not None
        if rank == 0:
            # memory-side traffic and issued instructions of the MSM kernels: committed rocprofv3 passes (tools/profile_msm.sh), used only when the
            # file is stamped with this library's binary or sources (same rule as the ring kernel's counters below)
            import hashlib
            from secp256k1_zkp_amd import _native as _nat
            _lib = hashlib.sha256(open(_nat.LIB_PATH, "rb").read()).hexdigest(); _src = _nat.sources_sha256()
            mc = None
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*msm_counters.json")), reverse=True):
                try:
                    jj = json.load(open(f))
                except Exception:
                    continue
                if jj.get("so_sha256") == _lib or (_src and jj.get("src_sha256") == _src):
                    mc = (f, jj); break
            for blk, key in ((msm, "1048576"),):
                if mc and key in mc[1]:
                    pc = mc[1][key]["per_call"]
                    blk["roofline"]["traffic"] = pc["hbm_bytes_raw"]; blk["roofline"]["traffic_unit"] = "HBM-side bytes per call (FETCH_SIZE + WRITE_SIZE over every kernel of the call, incl. Infinity-Cache hits)"
                    blk["roofline"]["traffic_source"] = os.path.relpath(mc[0], ROOT)
                    if pc.get("int64_wave_instructions"):
                        blk["roofline"]["issued"] = {"int64_wave_instructions_per_call": pc["int64_wave_instructions"], "valu_wave_instructions_per_call": pc["valu_wave_instructions"],
                                                     "int64_lane_ops_per_s": pc["int64_wave_instructions"] * 64 / (ms * 1e-3), "frac_of_peak": pc["int64_wave_instructions"] * 64 / (ms * 1e-3) / MAD32_PEAK}
                else:
                    blk["roofline"]["traffic"] = None; blk["roofline"]["traffic_source"] = "no profiles/*msm_counters.json stamped with this library (tools/profile_msm.sh)"
            msm["_counters_file"] = mc[0] if mc else None
        msm["result_check"] = ("== (sum s_i*k_i mod n)*G by " + checker) if checker else ("unverified (oracle/_ref not present)" if rank == 0 else "rank 0")
        # CPU baseline for this half of the metric: the reference's secp256k1_ecmult_multi_var (what bench_ecmult times,
        # src/bench_ecmult.c:278-307 -- its largest size is 32768) on one host core, same box, same inputs
        if rank == 0 and ref is not None and not args.no_cpu_baseline:
            pts_h = pts.cpu().numpy()
            cb = {}
            for m in (1 << 15, 1 << 20):
                t = time.time(); rxy, rinf = ref.ecmult_multi(scs_h[:m], pts_h[:m]); t = time.time() - t
                cb["2^%d" % (m.bit_length() - 1)] = {"seconds": t, "mpoint_scalar_per_s": m / t / 1e6}
                if m == nm:
                    assert rinf == minf and rxy.tobytes() == xy, "MSM differs from the reference's ecmult_multi_var"
            msm["cpu_baseline"] = {"value": cb["2^20"]["mpoint_scalar_per_s"], "unit": "Mpoint-scalar/s", "cores": 1, "kind": "reference",
                                   "sample": "secp256k1_ecmult_multi_var (Pippenger) on 1 thread: 2^15 terms %.3f s = %.3f M/s, 2^20 terms %.2f s = %.3f M/s; the 2^20 result equals the GPU's"
                                             % (cb["2^15"]["seconds"], cb["2^15"]["mpoint_scalar_per_s"], cb["2^20"]["seconds"], cb["2^20"]["mpoint_scalar_per_s"])}
        del scs, pts
        # ---- 2^24 terms: the strong-scaling entry whose per-rank slice stays >= 2^21 terms up to 8 ranks (where one GPU is past its latency floor)
        if not args.no_msm_big:
            nb = 1 << 24
            ks_h, scs_h, scs, pts = msm_inputs(nb, 199)
            ms, xy, minf = msm_time(scs, pts)
            big = block(nb, ms, xy)
            if rank == 0 and msm.get("_counters_file"):
                jj = json.load(open(msm["_counters_file"]))
                if "16777216" in jj:
                    pc = jj["16777216"]["per_call"]
                    big["roofline"]["traffic"] = pc["hbm_bytes_raw"]; big["roofline"]["traffic_source"] = os.path.relpath(msm["_counters_file"], ROOT)
                    if pc.get("int64_wave_instructions"):
                        big["roofline"]["issued"] = {"int64_wave_instructions_per_call": pc["int64_wave_instructions"], "valu_wave_instructions_per_call": pc["valu_wave_instructions"],
                                                     "int64_lane_ops_per_s": pc["int64_wave_instructions"] * 64 / (ms * 1e-3), "frac_of_peak": pc["int64_wave_instructions"] * 64 / (ms * 1e-3) / MAD32_PEAK}
            if rank == 0:
                # the same identity, evaluated in 64 slices so that the Python integers stay short-lived
                tot = 0
                for a in range(0, nb, 1 << 18):
                    kb = ks_h[a:a + (1 << 18)]; sb = scs_h[a:a + (1 << 18)]
                    tot += sum(int.from_bytes(kb[i].tobytes(), "big") * int.from_bytes(sb[i].tobytes(), "big") for i in range(kb.shape[0]))
                tot %= ORDER
                if ref is not None:
                    exp_xy, exp_inf = ref.ecmult_batch(np.frombuffer(G_XY, np.uint8), np.zeros(32, np.uint8), ng=np.frombuffer(tot.to_bytes(32, "big"), np.uint8), a_inf=np.ones(1, np.uint8))
                    assert minf == int(exp_inf[0]) and xy == exp_xy[0].tobytes(), "2^24-term MSM differs from (sum s_i k_i)*G"
                    big["verified"] = True; big["result_check"] = "== (sum s_i*k_i mod n)*G by reference secp256k1_ecmult"
                else:
                    big["verified"] = False
            msm["strong_2p24"] = big
            del scs, pts

    # BASELINE configs 2 and 4 (and config 1's GPU side): same conventions as the headline -- inputs made by the reference, resident in HBM,
    # K stream-ordered `_dev` calls queued back to back and waited for once, every verdict checked in-run against the reference's.
    secondary = None
    if rank == 0 and not args.no_secondary and ref is not None:
        torch.cuda.synchronize(); time.sleep(1.0)
        secondary = measure_secondary(eng, ref, dev, max(3, args.steps), with_cpu=not args.no_cpu_baseline)
        if not args.no_next:
            try:
                secondary.update(measure_next_rows(eng, ref, dev, max(3, args.steps), with_cpu=not args.no_cpu_baseline))
            except Exception as ex:      # noqa: BLE001  (the headline must not be lost to a failure of these extra rows)
                secondary["next_rows_error"] = repr(ex)

    # The C-ABI engine group, from rank 0 alone (the other ranks wait at the barrier below, their GPUs idle): one process drives every GPU of the
    # run through s2k_group_* -- the path a C caller without torch.distributed takes.  With one GPU: two engines on it (they share the
    # device's tables), which shows what a second submitting thread buys on a single device.
    group = None
    if not args.no_group:
        if world > 1:
            dist.barrier()
        if rank == 0:
            ndev = torch.cuda.device_count()
            devs = [i % ndev for i in range(world)] if world > 1 else [local, local]      # devices 0..N-1 of the run (ranks sharing a device under gloo: the same sharing)
            try:
                group = measure_group(devs, commits, proofs, gens, ref, steps=max(2, min(args.steps, 4)))
            except Exception as ex:      # noqa: BLE001  (the headline must not be lost to a failure of this extra block)
                group = {"error": repr(ex)}
        if world > 1:
            dist.barrier()

    # The headline at narrower fixed-base tables (S2K_OPT_GTAB_BITS: 24 bits = 5.9 GB per table, 20 bits = 0.44 GB, against 21.5 GB at the default
    # 26): the same timed loop after the device's tables have been given back and rebuilt at that width -- what an engine on a partitioned or
    # shared GPU gets.  Verdicts are checked again; the default width is restored afterwards.
    width_values = None
    if rank == 0 and world == 1 and not args.no_widths:
        width_values = {}
        try:
            for bits in (24, 20):
                eng.set_option(Engine.OPT_GTAB_BITS, bits)
                d_res.zero_()
                dtw, kw = timed(args.steps, pipeline)
                assert bool(d_res.all().item()) and int(eng._lib.s2k_engine_gtable_bits(eng._h)) == bits, "verdicts or table width wrong at %d-bit tables" % bits
                width_values[bits] = {"value": n * args.steps / dtw, "ms_per_step": dtw / args.steps * 1e3, "ring_kernel_ms": float(np.mean(kw)),
                                      "table_gb_each": ((((256 + bits - 1) // bits) << (bits - 1)) + 1) * 64 / 1e9, "verified": True}
        except Exception as ex:      # noqa: BLE001  (the headline must not be lost to a failure of this extra block)
            width_values["error"] = repr(ex)
        finally:
            eng.set_option(Engine.OPT_GTAB_BITS, 26); eng.set_option(Engine.OPT_RP_INPUTS_READY, 0)

    if rank == 0:
        value = world * n * args.steps / dt
        kms = float(np.mean(kern_ms))
        achieved = PROOF_BYTES_ALGO * n / (kms * 1e-3) / 1e9
        # memory-side traffic and issued-instruction counters of the dominant kernel: PMC counters cannot be read from inside this process,
        # so the figures are committed rocprofv3 passes of this same command (profiles/, tools/profile_round.sh) -- used ONLY when the file
        # carries the sha256 of the library this process has loaded (the counters then belong to this binary); otherwise null + the reason
        import hashlib
        from secp256k1_zkp_amd import _native
        lib_sha = hashlib.sha256(open(_native.LIB_PATH, "rb").read()).hexdigest()
        # (hipcc's output is not reproducible byte for byte -- two builds of one source tree differ in the offload bundle's ids -- so a file
        #  also counts when it carries the sha256 of the library's SOURCES, csrc/* + the public header, as they are in this tree)
        src_sha = _native.sources_sha256()
        traffic, traffic_src, issued = None, None, None
        def _stamped(pattern):
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
                try:
                    j = json.load(open(f))
                except Exception:
                    continue
                if j.get("so_sha256") == lib_sha or (src_sha and j.get("src_sha256") == src_sha):
                    return f, j
            return None, None
        f_pmc, pj = _stamped("*pmc_rp_rings.json")
        if pj:
            traffic = pj["hbm_bytes_per_launch_raw"] * n / 16384.0
            traffic_src = os.path.relpath(f_pmc, ROOT) + (" (same library binary)" if pj.get("so_sha256") == lib_sha else " (same library sources, another build)")
        else:
            traffic_src = "no profiles/*pmc_rp_rings.json stamped with this library's sha256 (%s...): run tools/profile_round.sh on this build" % lib_sha[:12]
        # issued (not algorithmic) integer-MAC rate: SQ counter passes of this same command; SQ_INSTS_VALU_INT64 counts wave-level 64-bit
        # integer instructions (v_mad_u64_u32 and the 64-bit shifts)
        f_sq, sj = _stamped("*sq_counters.json")
        if sj:
            kname = "k_rp_rings_shared" if "k_rp_rings_shared" in sj else "k_rp_rings"
            cj = sj.get(kname, {})
            if "SQ_INSTS_VALU_INT64" in cj:
                int64 = cj["SQ_INSTS_VALU_INT64"]["mean_per_launch"] * n / 16384.0; valu = cj["SQ_INSTS_VALU"]["mean_per_launch"] * n / 16384.0
                issued = {"source": os.path.relpath(f_sq, ROOT), "kernel": kname, "int64_wave_instructions_per_launch": int64, "valu_wave_instructions_per_launch": valu,
                          "int64_lane_ops_per_s": int64 * 64 / (kms * 1e-3), "frac_of_peak": int64 * 64 / (kms * 1e-3) / MAD32_PEAK,
                          "note": "counter-backed issue rate of 64-bit integer VALU instructions (v_mad_u64_u32 + 64-bit shifts) in the rings kernel, this run's kernel time"}
        mad_rate = 4 * MAC64_PER_PROOF * n / (kms * 1e-3)
        out = {
            "metric": "64-bit Borromean rangeproof verifies/sec", "value": value, "unit": "verifies/s", "n_gpus": world,
            # the exchange library's own view of the job: world size after a checked all-reduce ("rccl" when the backend is nccl), the device of every rank
            "collective": {"backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend) if world > 1 else None,
                           "world_size": collective_world, "rank_devices": rank_devices},
            "rccl_world_size": collective_world if (world > 1 and backend == "nccl") else (1 if world == 1 else None),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (9x29-bit), 32x32->64 integer MAC", "data": data_desc,
            # verified: every timed proof was signed by the reference (oracle/_ref) and accepted; false when the run fell back to a tiled golden proof
            "verified": ref is not None,
            "value_serialized_calls": world * n * ser_steps / dt_ser, "ms_per_step_serialized_calls": dt_ser / ser_steps * 1e3,
            "value_distinct_generators": distinct["value"] if distinct else None,
            "value_gtab_bits_24": width_values.get(24, {}).get("value") if width_values else None,
            "value_gtab_bits_20": width_values.get(20, {}).get("value") if width_values else None,
            "table_widths": width_values,
            "ms_per_step_distinct_generators": distinct["ms_per_step"] if distinct else None,
            "generators": "value: secp256k1_generator_h for every proof (src/bench_rangeproof.c), fixed-base table of that generator cached by the engine; "
                          "value_distinct_generators: the same batch size, every proof its own random generator (no table applies: general form of the rings kernel)",
            "config": {"workload": "secp256k1_rangeproof_verify, batch of %d 64-bit proofs per GPU (exp=0, min_value=0, 32 rings x 4)" % n,
                       "batch_per_gpu": n, "sharding": "replicas (independent proofs, no collective)",
                       "calls_in_flight": ("K steps queued back to back, first stage of step k+1 on side streams under step k's ring kernel "
                                           "(S2K_OPT_RP_INPUTS_READY: inputs resident and untouched)") if pipeline
                                          else "same queueing, engine default: the first stage of a call waits for the call before it"},
            # the binding roofline of this path is the integer VALU: exact 256-bit modular arithmetic, no MFMA, ~0.05 % of HBM
            "roofline": {"bound": "valu", "kernel": "k_rp_rings_shared (+ k_rp_rings for wavefronts without a generator table)", "achieved": mad_rate / 1e12, "peak": MAD32_PEAK / 1e12,
                         "unit": "T lane-MAC/s (v_mad_u64_u32, 32x32+64)", "frac": mad_rate / MAD32_PEAK, "frac_of_architectural_peak": mad_rate / MAD32_PEAK_ARCH,
                         "traffic": traffic, "traffic_unit": "HBM-side bytes per launch (FETCH_SIZE + WRITE_SIZE, incl. Infinity-Cache hits)", "traffic_source": traffic_src,
                         "kernel_ms": kms, "kernel_ms_with_calls_in_flight": float(np.mean(kern_ms_pipe)), "issued": issued, "library_sha256": lib_sha, "library_sources_sha256": src_sha,
                         "note": "achieved = algorithmic 6.6e6 MAC64/proof (reference schedule, SURVEY 8d) x 4 v_mad_u64_u32 x proofs / kernel time (HIP events on the launch stream); "
                                 "peak measured with >= 10 ms launches (tools/ubench/issue_model.hip, profiles/r02a_issue_model.txt)"},
            "hbm_roofline": {"bound": "hbm", "kernel": "k_rp_rings_shared + k_rp_rings", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                             "algorithmic_bytes": PROOF_BYTES_ALGO * n, "note": "reported because the contract asks; not the binding bound"},
        }
        # where the wall clock of this process goes: the headline's timed region is a fraction of a second inside a run of tens of seconds
        # (input signing by the reference, table builds, the CPU baselines) -- a GPU-busy sampler around the whole process sees mostly idle
        out["timing"] = {"timed_region_s": dt, "timed_region_serialized_calls_s": dt_ser, "timed_steps": args.steps,
                         "process_wall_s_until_headline": time.time() - t_process}
        out["tables"] = tables
        if msm:
            msm.pop("_counters_file", None)
            out["msm"] = msm
        if dropin:
            out["dropin"] = dropin
        if secondary:
            out["secondary"] = secondary
        if group:
            out["group"] = group
        if not args.no_cpu_baseline:
            # (1) the reference's own bench programs (src/bench_rangeproof.c with min_bits = 64, src/bench_ecmult.c; timer of src/bench.h),
            #     one process and one taskset-pinned process per usable core; (2) the same functions through the oracle/_ref shim with
            #     OpenMP, on THIS run's proofs, as a cross-check of (1)
            shim = cpu_baseline(ref, commits, proofs, gens)
            rb = cpu_baseline_ref_benches()
            if rb:
                ac = rb["bench_rangeproof_64_all_cores"]
                out["cpu_baseline"] = {"value": ac["verifies_per_s"], "unit": "verifies/s", "cores": ac["processes"], "kind": "reference",
                                       "cpu_model": rb["cpu_model"], "hardware_threads": rb["hardware_threads"],
                                       "per_core": rb["bench_rangeproof_64"]["verifies_per_s_one_process"],
                                       "sample": "the reference's bench_rangeproof (src/bench_rangeproof.c, min_bits = 64, SECP256K1_BENCH_ITERS=36 x 10 repetitions), "
                                                 "%d processes pinned with taskset to the usable cores, avg column of src/bench.h: %.0f verifies/s in all; one process: %.1f /s "
                                                 "(%.1f us per bit); as shipped (min_bits = 32): %.1f us per bit"
                                                 % (ac["processes"], ac["verifies_per_s"], rb["bench_rangeproof_64"]["verifies_per_s_one_process"],
                                                    rb["bench_rangeproof_64"]["us_per_bit_min_avg_max"][1], rb["bench_rangeproof"]["us_per_bit_min_avg_max"][1]),
                                       "bench_programs": rb, "shim_cross_check": shim}
                if msm and "bench_ecmult_pippenger_32767p_g" in rb:
                    msm.setdefault("cpu_baseline", {})["bench_ecmult"] = rb["bench_ecmult_pippenger_32767p_g"]
                if secondary:
                    if "bench_schnorrsig_verify" in rb:
                        sv = rb["bench_schnorrsig_verify"]
                        secondary["bip340_2p16"]["cpu_baseline"] = {"value": sv["verifies_per_s_one_process"], "unit": "verifies/s", "cores": 1, "kind": "reference",
                                                                    "sample": "the reference's src/bench.c schnorrsig_verify, SECP256K1_BENCH_ITERS=4000 x 10 repetitions, avg column: %.1f us per verification" % sv["us_per_verify_min_avg_max"][1]}
                    if "bench_ecmult_pippenger_1023p_g" in rb:
                        k1 = rb["bench_ecmult_pippenger_1023p_g"]
                        secondary["bench_ecmult_1023p_g"]["cpu_baseline"] = {"value": k1["mpoint_scalar_per_s"], "unit": "Mpoint-scalar/s", "cores": 1, "kind": "reference",
                                                                             "sample": "the reference's bench_ecmult pippenger_wnaf, row ecmult_multi_1023p_g (src/bench_ecmult.c:278-307): %.2f us per point" % k1["us_per_point_min_avg_max"][1]}
            elif shim:
                if ref is None:
                    shim["kind"] = "port"
                out["cpu_baseline"] = shim
        out["timing"]["process_wall_s"] = time.time() - t_process
        print(json.dumps(out))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
