"""ctypes view of oracle/_ref/libsecp256k1_ref.so (the unmodified reference + oracle/ref_shim.c). Test-only."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libsecp256k1_ref.so")

from secp256k1_zkp_amd.constants import P, N, G_XY, GENERATOR_H  # noqa: F401  (one definition: the package's)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Ref:
    def __init__(self):
        self.lib = ctypes.CDLL(REF_PATH)
        L = self.lib
        L.ref_ecmult_multi.restype = ctypes.c_int
        L.ref_sha256_state_size.restype = ctypes.c_size_t

    # --- points / scalars helpers
    def call(self, name, nout, *args):
        outs = [ctypes.create_string_buffer(n) for n in nout]
        r = getattr(self.lib, name)(*outs, *args)
        return r, [o.raw for o in outs]

    def rand_point(self, rng):
        while True:
            x = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
            r, o = self.call("ref_ge_set_xquad", [64], x)
            if r:
                pt = o[0]
                if rng.integers(0, 2):
                    pt = pt[:32] + ((P - int.from_bytes(pt[32:], "big")) % P).to_bytes(32, "big")
                return pt

    def ecmult_batch(self, a_xy, na, ng=None, a_inf=None):
        a_xy = np.ascontiguousarray(a_xy, np.uint8); n = a_xy.size // 64
        na = np.ascontiguousarray(na, np.uint8)
        ng = None if ng is None else np.ascontiguousarray(ng, np.uint8)
        a_inf = None if a_inf is None else np.ascontiguousarray(a_inf, np.uint8)
        r = np.zeros((n, 64), np.uint8); inf = np.zeros(n, np.int32)
        self.lib.ref_ecmult_batch(_p(r), _p(inf), _p(a_xy), _p(a_inf), _p(na), _p(ng), ctypes.c_size_t(n))
        return r, inf

    def ecmult_multi(self, sc, pt_xy, g_sc=None, pt_inf=None, algo=0):
        sc = np.ascontiguousarray(sc, np.uint8); pt_xy = np.ascontiguousarray(pt_xy, np.uint8); n = sc.size // 32
        g = None if g_sc is None else (np.frombuffer(g_sc, np.uint8).copy() if isinstance(g_sc, (bytes, bytearray)) else np.ascontiguousarray(g_sc, np.uint8))
        pi = None if pt_inf is None else np.ascontiguousarray(pt_inf, np.uint8)
        r = np.zeros(64, np.uint8)
        inf = self.lib.ref_ecmult_multi(_p(r), _p(g), _p(sc), _p(pt_xy), _p(pi), ctypes.c_size_t(n), ctypes.c_int(algo))
        return r, inf

    # --- rangeproofs
    def make_rangeproofs(self, n, rng, min_bits=64, exp=0, min_value=0, gens64=None, values=None, threads=8):
        blinds = rng.integers(0, 256, (n, 32), dtype=np.uint8); blinds[:, 0] &= 0x7F
        if values is None:
            hi = 2**63 if min_bits >= 63 else 2**max(min_bits, 1)
            values = rng.integers(0, hi, n, dtype=np.uint64) + np.uint64(min_value)
        values = np.ascontiguousarray(values, np.uint64)
        if gens64 is None:
            gens64 = np.frombuffer(GENERATOR_H * n, np.uint8).reshape(n, 64).copy()
        stride = 5134
        commits = np.zeros((n, 33), np.uint8); proofs = np.zeros((n, stride), np.uint8); plens = np.zeros(n, np.uint64)
        ok = self.lib.ref_rangeproof_make_many(_p(commits), _p(proofs), ctypes.c_size_t(stride), _p(plens), _p(blinds), _p(values), _p(gens64),
                                               ctypes.c_uint64(min_value), ctypes.c_int(exp), ctypes.c_int(min_bits), ctypes.c_size_t(n),
                                               ctypes.c_int(threads))
        assert ok == 1
        plist = [proofs[i, :int(plens[i])].tobytes() for i in range(n)]
        return commits, plist, gens64, values

    @staticmethod
    def _pack_extra(extra, n):
        estride = max(max((len(e) for e in extra), default=1), 1)
        buf = np.zeros((n, estride), np.uint8)
        for i, e in enumerate(extra):
            buf[i, :len(e)] = np.frombuffer(e, np.uint8)
        return buf, estride, np.array([len(e) for e in extra], np.uint64)

    def make_rangeproofs_extra(self, n, rng, extra, min_bits=64, exp=0, min_value=0, gens64=None, threads=8):
        """like make_rangeproofs, every proof signed over its own extra_commit bytes (extra: list of n byte strings)"""
        blinds = rng.integers(0, 256, (n, 32), dtype=np.uint8); blinds[:, 0] &= 0x7F
        hi = 2**63 if min_bits >= 63 else 2**max(min_bits, 1)
        values = np.ascontiguousarray(rng.integers(0, hi, n, dtype=np.uint64) + np.uint64(min_value), np.uint64)
        if gens64 is None:
            gens64 = np.frombuffer(GENERATOR_H * n, np.uint8).reshape(n, 64).copy()
        gens64 = np.ascontiguousarray(gens64, np.uint8)
        ebuf, estride, elens = self._pack_extra(extra, n)
        stride = 5134
        commits = np.zeros((n, 33), np.uint8); proofs = np.zeros((n, stride), np.uint8); plens = np.zeros(n, np.uint64)
        ok = self.lib.ref_rangeproof_make_many_extra(_p(commits), _p(proofs), ctypes.c_size_t(stride), _p(plens), _p(blinds), _p(values), _p(gens64),
                                                     _p(ebuf), ctypes.c_size_t(estride), _p(elens), ctypes.c_uint64(min_value), ctypes.c_int(exp),
                                                     ctypes.c_int(min_bits), ctypes.c_size_t(n), ctypes.c_int(threads))
        assert ok == 1
        return commits, [proofs[i, :int(plens[i])].tobytes() for i in range(n)], gens64

    def rangeproof_verify_many_extra(self, commits33, plist, gens64, extra, threads=1):
        n = len(plist)
        stride = max(max((len(p) for p in plist), default=1), 1)
        proofs = np.zeros((n, stride), np.uint8)
        for i, p in enumerate(plist):
            proofs[i, :len(p)] = np.frombuffer(p, np.uint8)
        plens = np.array([len(p) for p in plist], np.uint64)
        ebuf, estride, elens = self._pack_extra(extra, n)
        res = np.zeros(n, np.int32); mn = np.zeros(n, np.uint64); mx = np.zeros(n, np.uint64)
        commits33 = np.ascontiguousarray(commits33, np.uint8); gens64 = np.ascontiguousarray(gens64, np.uint8)
        self.lib.ref_rangeproof_verify_many_extra(_p(res), _p(mn), _p(mx), _p(commits33), _p(proofs), ctypes.c_size_t(stride), _p(plens), _p(ebuf),
                                                  ctypes.c_size_t(estride), _p(elens), _p(gens64), ctypes.c_size_t(n), ctypes.c_int(threads))
        return res, mn, mx

    def make_rangeproofs_msg(self, n, rng, msg_len=0, min_bits=64, exp=0, min_value=0, values=None, threads=8):
        """proofs with random nonces and embedded random messages; returns (commits, proofs, gens, values, blinds, nonces, msgs)"""
        blinds = rng.integers(0, 256, (n, 32), dtype=np.uint8); blinds[:, 0] &= 0x7F
        if values is None:
            hi = 2**63 if min_bits >= 63 else 2**max(min_bits, 1)
            values = rng.integers(0, hi, n, dtype=np.uint64) + np.uint64(min_value)
        values = np.ascontiguousarray(values, np.uint64)
        gens64 = np.frombuffer(GENERATOR_H * n, np.uint8).reshape(n, 64).copy()
        nonces = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        msgs = rng.integers(0, 256, (n, max(msg_len, 1)), dtype=np.uint8)
        stride = 5134
        commits = np.zeros((n, 33), np.uint8); proofs = np.zeros((n, stride), np.uint8); plens = np.zeros(n, np.uint64)
        ok = self.lib.ref_rangeproof_make_many_msg(_p(commits), _p(proofs), ctypes.c_size_t(stride), _p(plens), _p(blinds), _p(values), _p(gens64), _p(nonces),
                                                   _p(msgs), ctypes.c_size_t(msg_len), ctypes.c_uint64(min_value), ctypes.c_int(exp), ctypes.c_int(min_bits),
                                                   ctypes.c_size_t(n), ctypes.c_int(threads))
        assert ok == 1
        plist = [proofs[i, :int(plens[i])].tobytes() for i in range(n)]
        return commits, plist, gens64, values, blinds, nonces, msgs[:, :msg_len]

    def rangeproof_rewind_many(self, commits33, plist, gens64, nonces, msg_capacity=4096, threads=1):
        """secp256k1_rangeproof_rewind per item -> (results, blinds (n,32), values, messages list, min, max)"""
        n = len(plist)
        stride = max(max((len(p) for p in plist), default=1), 1)
        proofs = np.zeros((n, stride), np.uint8)
        for i, p in enumerate(plist):
            proofs[i, :len(p)] = np.frombuffer(p, np.uint8)
        plens = np.array([len(p) for p in plist], np.uint64)
        res = np.zeros(n, np.int32); mn = np.zeros(n, np.uint64); mx = np.zeros(n, np.uint64)
        blind = np.zeros((n, 32), np.uint8); val = np.zeros(n, np.uint64)
        ms = max(msg_capacity, 1); msg = np.zeros((n, ms), np.uint8); ol = np.full(n, msg_capacity, np.uint64)
        commits33 = np.ascontiguousarray(commits33, np.uint8); gens64 = np.ascontiguousarray(gens64, np.uint8); nonces = np.ascontiguousarray(nonces, np.uint8)
        self.lib.ref_rangeproof_rewind_many(_p(res), _p(blind), _p(val), _p(msg) if msg_capacity else None, _p(ol), ctypes.c_size_t(ms), _p(nonces), _p(mn), _p(mx),
                                            _p(commits33), _p(proofs), ctypes.c_size_t(stride), _p(plens), _p(gens64), ctypes.c_size_t(n), ctypes.c_int(threads))
        msgs = [msg[i, :int(ol[i])].tobytes() if (res[i] and msg_capacity) else b"" for i in range(n)]
        return res, blind, val, msgs, mn, mx

    def rangeproof_verify_many(self, commits33, plist, gens64, threads=1):
        n = len(plist)
        stride = max(max((len(p) for p in plist), default=1), 1)
        proofs = np.zeros((n, stride), np.uint8)
        for i, p in enumerate(plist):
            proofs[i, :len(p)] = np.frombuffer(p, np.uint8)
        plens = np.array([len(p) for p in plist], np.uint64)
        res = np.zeros(n, np.int32); mn = np.zeros(n, np.uint64); mx = np.zeros(n, np.uint64)
        commits33 = np.ascontiguousarray(commits33, np.uint8); gens64 = np.ascontiguousarray(gens64, np.uint8)
        self.lib.ref_rangeproof_verify_many(_p(res), _p(mn), _p(mx), _p(commits33), _p(proofs), ctypes.c_size_t(stride), _p(plens), _p(gens64),
                                            ctypes.c_size_t(n), ctypes.c_int(threads))
        return res, mn, mx

    # --- schnorr
    def make_schnorr(self, n, rng, threads=8):
        sk = rng.integers(0, 256, (n, 32), dtype=np.uint8); sk[:, 0] &= 0x7F; sk[:, 31] |= 1
        msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        sigs = np.zeros((n, 64), np.uint8); pks = np.zeros((n, 32), np.uint8)
        ok = self.lib.ref_schnorrsig_make_many(_p(sigs), _p(pks), _p(sk), _p(msgs), ctypes.c_size_t(n), ctypes.c_int(threads))
        assert ok == 1
        return sigs, msgs, pks

    def schnorr_verify_many(self, sigs, msgs, pks, msglen=32, threads=1):
        n = sigs.size // 64
        res = np.zeros(n, np.int32)
        self.lib.ref_schnorrsig_verify_many(_p(res), _p(np.ascontiguousarray(sigs)), _p(np.ascontiguousarray(msgs)), ctypes.c_size_t(msglen),
                                            _p(np.ascontiguousarray(pks)), ctypes.c_size_t(n), ctypes.c_int(threads))
        return res

    # --- bppp norm argument (test-side flow of modules/bppp/tests_impl.h:385-435)
    def bppp_generators(self, n):
        out = np.zeros((n, 33), np.uint8)
        assert self.lib.ref_bppp_generators(_p(out), ctypes.c_size_t(n)) == 1
        return out

    def pedersen_commit(self, blinds, values, gen64=GENERATOR_H):
        """secp256k1_pedersen_commit per item -> (n,33) serialised commitments"""
        blinds = np.ascontiguousarray(blinds, np.uint8); values = np.ascontiguousarray(values, np.uint64); n = values.size
        out = np.zeros((n, 33), np.uint8); g = np.frombuffer(bytes(gen64), np.uint8).copy()
        assert self.lib.ref_pedersen_commit_many(_p(out), _p(blinds), _p(values), _p(g), ctypes.c_size_t(n)) == 1
        return out

    def pedersen_blind_sum(self, blinds, npositive):
        blinds = np.ascontiguousarray(blinds, np.uint8); n = blinds.size // 32
        out = np.zeros(32, np.uint8)
        assert self.lib.ref_pedersen_blind_sum(_p(out), _p(blinds), ctypes.c_size_t(n), ctypes.c_size_t(npositive)) == 1
        return out

    def make_balanced_tally(self, rng, n_in, n_out, gen64=GENERATOR_H):
        """(inputs (n_in,33), outputs (n_out,33)) with sum(inputs) == sum(outputs), as src/modules/generator/tests_impl.h:239-300 builds them"""
        vin = rng.integers(0, 2**40, n_in, dtype=np.uint64); tot = int(vin.sum())
        cuts = np.sort(rng.integers(0, tot + 1, n_out - 1)) if n_out > 1 else np.array([], np.int64)
        vout = np.diff(np.concatenate([[0], cuts, [tot]])).astype(np.uint64)
        blinds = rng.integers(0, 256, (n_in + n_out, 32), dtype=np.uint8); blinds[:, 0] &= 0x7F
        blinds[-1] = self.pedersen_blind_sum(blinds[:-1], n_in)          # sum(in) - sum(other outs)
        c = self.pedersen_commit(blinds, np.concatenate([vin, vout]), gen64)
        return c[:n_in].copy(), c[n_in:].copy()

    def pedersen_verify_tally_many(self, tallies):
        """list of (pos (k,33), neg (m,33)) -> int array; -1 where a commitment does not parse"""
        parts, off, npos = [], [0], []
        for pos, neg in tallies:
            pos = np.ascontiguousarray(pos, np.uint8).reshape(-1, 33); neg = np.ascontiguousarray(neg, np.uint8).reshape(-1, 33)
            parts += [pos, neg]; npos.append(pos.shape[0]); off.append(off[-1] + pos.shape[0] + neg.shape[0])
        data = np.ascontiguousarray(np.concatenate(parts)) if off[-1] else np.zeros((1, 33), np.uint8)
        off = np.array(off, np.uint64); npos = np.array(npos + [0], np.uint64); res = np.zeros(len(tallies), np.int32)
        self.lib.ref_pedersen_verify_tally_many(_p(res), _p(data), _p(off), _p(npos), ctypes.c_size_t(len(tallies)))
        return res

    def xonly_objects(self, pks):
        """(n,32) serialised x-only keys -> (n,64) secp256k1_xonly_pubkey objects as the reference holds them in memory"""
        pks = np.ascontiguousarray(pks, np.uint8); n = pks.size // 32
        out = np.zeros((n, 64), np.uint8)
        assert self.lib.ref_xonly_objects(_p(out), _p(pks), ctypes.c_size_t(n)) == 1
        return out

    def xonly_valid(self, pks):
        """which of the (n,32) serialised keys secp256k1_xonly_pubkey_parse accepts"""
        pks = np.ascontiguousarray(pks, np.uint8).reshape(-1, 32); tmp = np.zeros(64, np.uint8)
        return np.array([self.lib.ref_xonly_objects(_p(tmp), _p(pks[i].copy()), ctypes.c_size_t(1)) == 1 for i in range(pks.shape[0])])

    def halfagg_aggregate(self, pks, msgs, sigs):
        """secp256k1_schnorrsig_aggregate on serialised x-only keys; returns the aggregate bytes (32*(n+1))."""
        pks = np.ascontiguousarray(pks, np.uint8); msgs = np.ascontiguousarray(msgs, np.uint8); sigs = np.ascontiguousarray(sigs, np.uint8)
        n = sigs.size // 64
        out = np.zeros(32 * (n + 1), np.uint8); ln = ctypes.c_size_t(out.size)
        r = self.lib.ref_halfagg_aggregate(_p(out), ctypes.byref(ln), _p(pks), _p(msgs), _p(sigs), ctypes.c_size_t(n))
        assert r == 1 and ln.value == out.size
        return out.tobytes()

    def halfagg_verify(self, pks, msgs, aggsig, n=None):
        """secp256k1_schnorrsig_aggverify; -1 if a key does not parse"""
        pks = np.frombuffer(bytes(pks), np.uint8) if isinstance(pks, (bytes, bytearray)) else np.ascontiguousarray(pks, np.uint8)
        msgs = np.frombuffer(bytes(msgs), np.uint8) if isinstance(msgs, (bytes, bytearray)) else np.ascontiguousarray(msgs, np.uint8)
        agg = np.frombuffer(bytes(aggsig), np.uint8)
        if n is None:
            n = msgs.size // 32
        return int(self.lib.ref_halfagg_verify(_p(pks) if pks.size else None, _p(msgs) if msgs.size else None, ctypes.c_size_t(n), _p(agg), ctypes.c_size_t(agg.size)))

    def make_bppp(self, n, rng, g_len, h_len):
        """n norm-argument proofs over the deterministic generator set; returns the batch-API argument tuple."""
        gens = self.bppp_generators(g_len + h_len)
        sc = lambda k: (rng.integers(0, 256, (k, 32), dtype=np.uint8) & np.array([0x7F] + [0xFF] * 31, np.uint8))
        nr = max(int(np.log2(g_len)), int(np.log2(h_len)))
        plen = 65 * nr + 64
        proofs = np.zeros((n, plen), np.uint8); trs = np.zeros((n, 104), np.uint8); rhos = sc(n)
        cvs = np.zeros((n, h_len, 32), np.uint8); commits = np.zeros((n, 33), np.uint8)
        for i in range(n):
            nv, lv, cv = sc(g_len), sc(h_len), sc(h_len)
            mu = np.zeros(32, np.uint8)
            self.lib.ref_scalar_mul(_p(mu), _p(rhos[i]), _p(rhos[i]))
            cm = np.zeros(33, np.uint8)
            assert self.lib.ref_bppp_commit(_p(cm), _p(gens), ctypes.c_size_t(g_len + h_len), _p(nv), ctypes.c_size_t(g_len), _p(lv), _p(cv),
                                            ctypes.c_size_t(h_len), _p(mu)) == 1
            tr = np.zeros(104, np.uint8)
            self.lib.ref_bppp_transcript_init(_p(tr), _p(rhos[i]), _p(gens), ctypes.c_size_t(g_len + h_len), ctypes.c_size_t(g_len), _p(cv),
                                              ctypes.c_size_t(h_len), _p(cm))
            pl = ctypes.c_size_t(plen); pr = np.zeros(plen, np.uint8)
            assert self.lib.ref_bppp_norm_prove(_p(pr), ctypes.byref(pl), _p(tr), _p(rhos[i]), _p(gens), ctypes.c_size_t(g_len + h_len), _p(nv),
                                                ctypes.c_size_t(g_len), _p(lv), _p(cv), ctypes.c_size_t(h_len)) == 1
            assert pl.value == plen
            proofs[i] = pr; trs[i] = tr; cvs[i] = cv; commits[i] = cm
        return proofs, trs, rhos, gens, g_len, cvs, commits

    def bppp_verify_many(self, proofs, trs, rhos, gens, g_len, cvs, commits):
        n = rhos.shape[0]; h_len = cvs.shape[1]; res = np.zeros(n, np.int32)
        for i in range(n):
            res[i] = self.lib.ref_bppp_norm_verify(_p(np.ascontiguousarray(proofs[i])), ctypes.c_size_t(proofs.shape[1]), _p(np.ascontiguousarray(trs[i])),
                                                   _p(np.ascontiguousarray(rhos[i])), _p(gens), ctypes.c_size_t(gens.shape[0]), ctypes.c_size_t(g_len),
                                                   _p(np.ascontiguousarray(cvs[i])), ctypes.c_size_t(h_len), _p(np.ascontiguousarray(commits[i])))
        return res

    # --- surjection proofs
    def make_surjection(self, rng, n_inputs, n_used, which=None):
        """one valid proof (reference prover): returns (serialised proof bytes, in_tags (n_inputs,64) uint8, out_tag (64,) uint8)"""
        which = int(rng.integers(0, n_inputs)) if which is None else which
        seed = rng.integers(0, 256, 32, dtype=np.uint8)
        buf = np.zeros(2 + 32 + 32 * 257 + 8, np.uint8); tags = np.zeros((n_inputs, 64), np.uint8); out = np.zeros(64, np.uint8)
        self.lib.ref_surjection_make.restype = ctypes.c_size_t
        ln = self.lib.ref_surjection_make(_p(buf), ctypes.c_size_t(buf.size), _p(tags), _p(out), _p(seed), ctypes.c_size_t(n_inputs), ctypes.c_size_t(n_used),
                                          ctypes.c_size_t(which))
        assert ln > 0
        return buf[:ln].tobytes(), tags, out

    def surjection_verify(self, proof, tags, out):
        tags = np.ascontiguousarray(tags, np.uint8)
        return self.lib.ref_surjectionproof_verify_ser(proof, ctypes.c_size_t(len(proof)), _p(tags), ctypes.c_size_t(tags.size // 64), _p(np.ascontiguousarray(out, np.uint8)))


def lift_generator33(ser33):
    """33-byte serialised generator (0x0a/0x0b || x) -> 64-byte x||y object (secp256k1_generator_parse, generator/main_impl.h:62-80)"""
    x = int.from_bytes(ser33[1:], "big")
    y = pow((x * x * x + 7) % P, (P + 1) // 4, P)
    assert y * y % P == (x * x * x + 7) % P and (ser33[0] & 0xFE) == 10
    if ser33[0] & 1:
        y = (P - y) % P
    return x.to_bytes(32, "big") + y.to_bytes(32, "big")
