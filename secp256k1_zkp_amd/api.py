"""Host-side mirror of the reference's verify interfaces for the hot path, batch-shaped.

Names and argument meaning follow the reference (``secp256k1_ecmult``, ``secp256k1_ecmult_multi_var``,
``secp256k1_schnorrsig_verify``, ``secp256k1_rangeproof_verify``, ``secp256k1_bppp_rangeproof_norm_product_verify``);
each method returns per-item results equal to the reference's single-item call.  Inputs are numpy ``uint8`` arrays
(host path: staged through the engine's HBM workspace) or torch CUDA tensors (``*_dev`` path: already resident).
"""
import ctypes

import numpy as np

from . import _native


class S2KError(RuntimeError):
    pass


def _u8(a, shape=None):
    if isinstance(a, (bytes, bytearray, memoryview)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _need(what, arr, nbytes):
    """host arrays reach hipMemcpy with sizes derived from n: a wrong-shaped argument must be an exception, not an out-of-bounds read"""
    if arr is None:
        raise ValueError(f"{what}: missing array")
    if arr.size != nbytes:
        raise ValueError(f"{what}: expected {nbytes} bytes, got {arr.size}")


def _offsets_ok(what, off, nbytes):
    off = np.asarray(off)
    if off.size == 0 or int(off[0]) != 0 or (np.diff(off.astype(np.int64)) < 0).any() or int(off[-1]) > nbytes:
        raise ValueError(f"{what}: offsets must start at 0, be non-decreasing and end inside the data ({nbytes} bytes)")


def _dp(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Engine:
    """One engine per GPU/process (owns a stream, the generator table and an HBM workspace)."""

    def __init__(self, device=0):
        self._lib = _native.load()
        self._h = self._lib.s2k_engine_create(int(device))
        if not self._h:
            raise S2KError("s2k_engine_create failed: " + _native.last_error())
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.s2k_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, ok, what):
        if not ok:
            raise S2KError(f"{what} failed: {_native.last_error()}")

    def sync(self):
        self._check(self._lib.s2k_engine_sync(self._h), "s2k_engine_sync")

    OPT_RP_INPUTS_READY = 1      # include/secp256k1_zkp_amd.h: S2K_OPT_*
    OPT_RP_SPLIT = 2
    OPT_GEN_CACHE_SLOTS = 3
    OPT_GEN_CACHE_MIN = 4
    OPT_MSM_PIPELINE = 5
    OPT_MAX_LANES = 6
    OPT_STAGE_THREADS = 7
    OPT_HALFAGG_HOST_CHAIN = 8
    OPT_SYNC_SPLIT = 9
    OPT_MSM_MAX_TERMS = 10
    OPT_GTAB_BITS = 11

    def cache_generator(self, gen64):
        """Build the fixed-base table of one rangeproof generator now (s2k_engine_cache_generator)."""
        g = bytes(gen64)
        if len(g) != 64:
            raise ValueError("cache_generator: a generator is 64 bytes")
        self._check(self._lib.s2k_engine_cache_generator(self._h, g), "s2k_engine_cache_generator")

    def generator_cached(self, gen64):
        return bool(self._lib.s2k_engine_generator_cached(self._h, bytes(gen64)))

    def set_option(self, option, value):
        self._check(self._lib.s2k_engine_set_option(self._h, int(option), int(value)), "s2k_engine_set_option")

    def last_ms(self, which=0):
        return float(self._lib.s2k_engine_last_ms(self._h, which))

    def last_msm_fallback(self):
        """True if the most recent bucket MSM re-sorted exactly after a bucket region overflowed."""
        return bool(self._lib.s2k_engine_last_msm_fallback(self._h))

    def rp_handback(self):
        """(groups given to the shared-generator form, rings through the general form, rings handed back as suspect, rings handed back
        after an exceptional addition) of the most recent rangeproof call (s2k_engine_rp_handback; synchronises)."""
        out = np.zeros(4, np.uint32)
        self._check(self._lib.s2k_engine_rp_handback(self._h, _p(out)), "s2k_engine_rp_handback")
        return tuple(int(x) for x in out)

    # ---- secp256k1_ecmult (src/ecmult.h:47), batched ------------------------------------------------------
    def ecmult_batch(self, a_xy, na, ng=None, a_inf=None):
        a_xy = _u8(a_xy); n = a_xy.size // 64
        na = _u8(na); ng = None if ng is None else _u8(ng); a_inf = None if a_inf is None else _u8(a_inf)
        _need("ecmult_batch a_xy", a_xy, 64 * n); _need("ecmult_batch na", na, 32 * n)
        if ng is not None: _need("ecmult_batch ng", ng, 32 * n)
        if a_inf is not None: _need("ecmult_batch a_inf", a_inf, n)
        r = np.zeros((n, 64), np.uint8); inf = np.zeros(n, np.int32)
        self._check(self._lib.s2k_ecmult_batch(self._h, _p(r), _p(inf), _p(a_xy), _p(a_inf), _p(na), _p(ng), n), "s2k_ecmult_batch")
        return r, inf

    def ecmult_batch_dev(self, r_xy, r_inf, a_xy, na, ng=None, a_inf=None, stream=None):
        n = a_xy.numel() // 64
        self._check(self._lib.s2k_ecmult_batch_dev(self._h, stream, _dp(r_xy), _dp(r_inf), _dp(a_xy), _dp(a_inf), _dp(na), _dp(ng), n),
                    "s2k_ecmult_batch_dev")

    # ---- secp256k1_ecmult_multi_var (src/ecmult.h:62) ----------------------------------------------------------
    def ecmult_multi(self, sc, pt_xy, g_sc=None, pt_inf=None):
        sc = _u8(sc); pt_xy = _u8(pt_xy); n = sc.size // 32
        g_sc = None if g_sc is None else _u8(g_sc); pt_inf = None if pt_inf is None else _u8(pt_inf)
        _need("ecmult_multi sc", sc, 32 * n); _need("ecmult_multi pt_xy", pt_xy, 64 * n)
        if g_sc is not None: _need("ecmult_multi g_sc", g_sc, 32)
        if pt_inf is not None: _need("ecmult_multi pt_inf", pt_inf, n)
        r = np.zeros(64, np.uint8); inf = np.zeros(1, np.int32)
        self._check(self._lib.s2k_ecmult_multi(self._h, _p(r), _p(inf), _p(g_sc), _p(sc), _p(pt_xy), _p(pt_inf), n), "s2k_ecmult_multi")
        return r, int(inf[0])

    def ecmult_multi_dev(self, r_xy, r_inf, sc, pt_xy, g_sc=None, pt_inf=None, stream=None):
        n = sc.numel() // 32
        self._check(self._lib.s2k_ecmult_multi_dev(self._h, stream, _dp(r_xy), _dp(r_inf), _dp(g_sc), _dp(sc), _dp(pt_xy), _dp(pt_inf), n),
                    "s2k_ecmult_multi_dev")

    def ecmult_multi_many(self, sc, pt_xy, offsets, g_sc=None, pt_inf=None):
        """K independent sums in one launch chain (s2k_ecmult_multi_many): terms back to back, offsets[K + 1]; returns (K x 64 bytes, K flags)."""
        sc = _u8(sc); pt_xy = _u8(pt_xy); off = np.ascontiguousarray(offsets, dtype=np.uint64); k = off.size - 1
        n = int(off[-1]) if off.size else 0
        g_sc = None if g_sc is None else _u8(g_sc); pt_inf = None if pt_inf is None else _u8(pt_inf)
        _need("ecmult_multi_many sc", sc, 32 * n); _need("ecmult_multi_many pt_xy", pt_xy, 64 * n)
        if g_sc is not None: _need("ecmult_multi_many g_sc", g_sc, 32 * k)
        if pt_inf is not None: _need("ecmult_multi_many pt_inf", pt_inf, n)
        r = np.zeros((max(k, 0), 64), np.uint8); inf = np.zeros(max(k, 0), np.int32)
        self._check(self._lib.s2k_ecmult_multi_many(self._h, _p(r), _p(inf), _p(g_sc), _p(sc), _p(pt_xy), _p(pt_inf), _p(off), k), "s2k_ecmult_multi_many")
        return r, inf

    def ecmult_multi_many_dev(self, r_xy, r_inf, sc, pt_xy, offsets_host, g_sc=None, pt_inf=None, stream=None):
        off = np.ascontiguousarray(offsets_host, dtype=np.uint64)
        self._check(self._lib.s2k_ecmult_multi_many_dev(self._h, stream, _dp(r_xy), _dp(r_inf), _dp(g_sc), _dp(sc), _dp(pt_xy), _dp(pt_inf), _p(off), off.size - 1),
                    "s2k_ecmult_multi_many_dev")

    def ecmult_multi_partial_dev(self, r_gej28, sc, pt_xy, g_sc=None, pt_inf=None, stream=None):
        n = sc.numel() // 32
        self._check(self._lib.s2k_ecmult_multi_partial_dev(self._h, stream, _dp(r_gej28), _dp(g_sc), _dp(sc), _dp(pt_xy), _dp(pt_inf), n),
                    "s2k_ecmult_multi_partial_dev")

    def ecmult_multi_window_partial_dev(self, r_gej28, sc, pt_xy, part, parts, g_sc=None, pt_inf=None, stream=None):
        n = sc.numel() // 32
        self._check(self._lib.s2k_ecmult_multi_window_partial_dev(self._h, stream, _dp(r_gej28), _dp(g_sc), _dp(sc), _dp(pt_xy), _dp(pt_inf), n, part, parts),
                    "s2k_ecmult_multi_window_partial_dev")

    def gej_sum_dev(self, r_xy, r_inf, gej28, count, stream=None):
        self._check(self._lib.s2k_gej_sum_dev(self._h, stream, _dp(r_xy), _dp(r_inf), _dp(gej28), count), "s2k_gej_sum_dev")

    # ---- secp256k1_schnorrsig_verify (modules/schnorrsig/main_impl.h:215-261), batched -------------------------
    def schnorrsig_verify_batch(self, sigs, msgs, pubkeys, msglen=32, pk_format=0):
        sigs = _u8(sigs); msgs = _u8(msgs); pubkeys = _u8(pubkeys); n = sigs.size // 64
        _need("schnorrsig_verify_batch sigs", sigs, 64 * n); _need("schnorrsig_verify_batch msgs", msgs, msglen * n)
        _need("schnorrsig_verify_batch pubkeys", pubkeys, (64 if pk_format else 32) * n)
        res = np.zeros(n, np.int32)
        self._check(self._lib.secp256k1_schnorrsig_verify_batch(self._h, _p(res), _p(sigs), _p(msgs), msglen, _p(pubkeys), pk_format, n),
                    "secp256k1_schnorrsig_verify_batch")
        return res

    def schnorrsig_verify_batch_dev(self, results, sigs, msgs, pubkeys, msglen=32, pk_format=0, stream=None):
        n = sigs.numel() // 64
        self._check(self._lib.secp256k1_schnorrsig_verify_batch_dev(self._h, stream, _dp(results), _dp(sigs), _dp(msgs), msglen, _dp(pubkeys),
                                                                    pk_format, n), "secp256k1_schnorrsig_verify_batch_dev")

    def bppp_norm_product_verify_batch_dev(self, results, proofs, proof_len, transcripts, rho, gens33_dev, gens33_host, g_len, c_vec, c_vec_len, commits33, n,
                                           stream=None):
        """every array in HBM (torch uint8 tensors); gens33_host: the same generator set as a numpy array (cache key of the fixed-base table)"""
        gh = _u8(gens33_host)
        self._check(self._lib.secp256k1_bppp_norm_product_verify_batch_dev(self._h, stream, _dp(results), _dp(proofs), proof_len, _dp(transcripts), _dp(rho), _dp(gens33_dev),
                                                                            _p(gh), gh.size // 33, g_len, _dp(c_vec), c_vec_len, _dp(commits33), n),
                    "secp256k1_bppp_norm_product_verify_batch_dev")

    # ---- secp256k1_bppp_commit (modules/bppp/bppp_norm_product_impl.h:105-151), batched on the fixed-base tables -----
    def bppp_commit_batch(self, gens33, g_len, n_vec, l_vec, c_vec, mu):
        """gens33 (n_gens,33); n_vec (n,g_len,32); l_vec, c_vec (n,h_len,32); mu (n,32) -> (commits (n,33), set_ok (n,))"""
        gens33 = _u8(gens33); n_gens = gens33.size // 33
        mu = _u8(mu); n = mu.size // 32
        h_len = n_gens - g_len
        n_vec = _u8(n_vec); l_vec = _u8(l_vec); c_vec = _u8(c_vec)
        if n_vec.size != n * g_len * 32 or l_vec.size != n * h_len * 32 or c_vec.size != n * h_len * 32:
            raise ValueError("bppp_commit_batch: vector shapes do not match (n, g_len, h_len)")
        out = np.zeros((n, 33), np.uint8); res = np.zeros(n, np.int32)
        self._check(self._lib.secp256k1_bppp_commit_batch(self._h, _p(out), _p(res), _p(gens33), n_gens, g_len, _p(n_vec), _p(l_vec), _p(c_vec), h_len, _p(mu), n),
                    "secp256k1_bppp_commit_batch")
        return out, res

    # ---- secp256k1_schnorrsig_aggverify (modules/schnorrsig_halfagg/main_impl.h:108-198) as one MSM -------------
    def schnorrsig_aggverify(self, pubkeys, msgs32, aggsig, pk_format=0, n=None):
        """pubkeys: (n,32) serialised x-only keys (or (n,64) objects with pk_format=1); msgs32: (n,32); aggsig: bytes r_0..r_{n-1}|s.
        n defaults to the number of messages.  Returns the reference's verdict (0/1)."""
        pubkeys = _u8(pubkeys); msgs32 = _u8(msgs32); aggsig = _u8(aggsig)
        if n is None:
            n = msgs32.size // 32
        if msgs32.size < 32 * n or pubkeys.size < (64 if pk_format else 32) * n:
            raise ValueError("schnorrsig_aggverify: fewer keys / messages than n")
        res = np.zeros(1, np.int32)
        self._check(self._lib.secp256k1_schnorrsig_aggverify_amd(self._h, _p(res), _p(pubkeys) if n else None, pk_format, _p(msgs32) if n else None, n,
                                                                 _p(aggsig), aggsig.size), "secp256k1_schnorrsig_aggverify_amd")
        return int(res[0])

    # ---- secp256k1_pedersen_verify_tally (modules/generator/main_impl.h:371-396), batched -------------------------
    def pedersen_verify_tally_batch(self, tallies):
        """tallies: list of (positive, negative), each a (k,33) uint8 array (or list of 33-byte strings) of serialised commitments.
        Returns int32[n]."""
        parts, off, npos = [], [0], []
        for pos, neg in tallies:
            pos = _u8(b"".join(bytes(x) for x in pos) if isinstance(pos, (list, tuple)) else pos).reshape(-1, 33)
            neg = _u8(b"".join(bytes(x) for x in neg) if isinstance(neg, (list, tuple)) else neg).reshape(-1, 33)
            parts += [pos, neg]; npos.append(pos.shape[0]); off.append(off[-1] + pos.shape[0] + neg.shape[0])
        n = len(tallies)
        data = np.ascontiguousarray(np.concatenate(parts)) if parts and off[-1] else np.zeros((1, 33), np.uint8)
        off = np.array(off, np.uint64); npos = np.array(npos + [0], np.uint64)
        res = np.zeros(max(n, 1), np.int32)
        self._check(self._lib.secp256k1_pedersen_verify_tally_batch(self._h, _p(res), _p(data), _p(off), _p(npos), n), "secp256k1_pedersen_verify_tally_batch")
        return res[:n]

    # ---- secp256k1_rangeproof_verify (modules/rangeproof/main_impl.h:54-71), batched ---------------------------
    @staticmethod
    def pack(items):
        """list of bytes -> (concatenated uint8 array, uint64 offsets[n+1])"""
        off = np.zeros(len(items) + 1, np.uint64)
        if items:
            off[1:] = np.cumsum([len(x) for x in items], dtype=np.uint64)
        data = np.frombuffer(b"".join(items) or b"\0", dtype=np.uint8).copy()
        return data, off

    def rangeproof_verify_batch(self, commits33, proofs, gens64, extra=None):
        """commits33: (n,33) uint8; proofs: list of bytes or (data, offsets); gens64: (n,64); extra: optional list of bytes.
        Returns (results int32[n], min_value uint64[n], max_value uint64[n])."""
        data, off = proofs if isinstance(proofs, tuple) else self.pack(list(proofs))
        n = off.size - 1
        commits33 = _u8(commits33); gens64 = _u8(gens64)
        edata = eoff = None
        if extra is not None:
            edata, eoff = extra if isinstance(extra, tuple) else self.pack(list(extra))
        self._check_rp_shapes("rangeproof_verify_batch", n, commits33, gens64, data, off, edata, eoff)
        res = np.zeros(n, np.int32); mn = np.zeros(n, np.uint64); mx = np.zeros(n, np.uint64)
        self._check(self._lib.secp256k1_rangeproof_verify_batch(self._h, _p(res), _p(mn), _p(mx), _p(commits33), _p(data), _p(off),
                                                                 _p(edata), _p(eoff), _p(gens64), n), "secp256k1_rangeproof_verify_batch")
        return res, mn, mx

    def rangeproof_verify_batch_submit(self, commits33, proofs, gens64, extra=None):
        """Asynchronous form: gathers and queues the batch, returns a ticket object; `rangeproof_verify_batch_wait(ticket)` returns
        (results, min_value, max_value).  At most two tickets may be outstanding (include/secp256k1_zkp_amd.h)."""
        data, off = proofs if isinstance(proofs, tuple) else self.pack(list(proofs))
        n = off.size - 1
        commits33 = _u8(commits33); gens64 = _u8(gens64)
        edata = eoff = None
        if extra is not None:
            edata, eoff = extra if isinstance(extra, tuple) else self.pack(list(extra))
        self._check_rp_shapes("rangeproof_verify_batch_submit", n, commits33, gens64, data, off, edata, eoff)
        res = np.zeros(n, np.int32); mn = np.zeros(n, np.uint64); mx = np.zeros(n, np.uint64)
        t = ctypes.c_uint64(0)
        self._check(self._lib.secp256k1_rangeproof_verify_batch_submit(self._h, ctypes.byref(t), _p(res), _p(mn), _p(mx), _p(commits33), _p(data), _p(off),
                                                                        _p(edata), _p(eoff), _p(gens64), n), "secp256k1_rangeproof_verify_batch_submit")
        return (t.value, res, mn, mx)                  # the output arrays live in the ticket until it is waited for

    def rangeproof_verify_batch_wait(self, ticket):
        t, res, mn, mx = ticket
        self._check(self._lib.secp256k1_rangeproof_verify_batch_wait(self._h, ctypes.c_uint64(t)), "secp256k1_rangeproof_verify_batch_wait")
        return res, mn, mx

    @staticmethod
    def _check_rp_shapes(what, n, commits33, gens64, data, off, edata, eoff):
        _need(what + " commits33", commits33, 33 * n); _need(what + " gens64", gens64, 64 * n)
        if np.asarray(off).size != n + 1:
            raise ValueError(what + ": proof offsets must have n + 1 entries")
        _offsets_ok(what + " proofs", off, data.size)
        if eoff is not None:
            if np.asarray(eoff).size != n + 1:
                raise ValueError(what + ": extra_commit offsets must have n + 1 entries")
            _offsets_ok(what + " extra", eoff, edata.size)

    def rangeproof_rewind_batch(self, commits33, proofs, gens64, nonces, msg_capacity=4096, extra=None):
        """secp256k1_rangeproof_rewind per item.  Returns (results, blinds (n,32), values uint64[n], messages list[bytes], min, max)."""
        data, off = proofs if isinstance(proofs, tuple) else self.pack(list(proofs))
        n = off.size - 1
        commits33 = _u8(commits33); gens64 = _u8(gens64); nonces = _u8(nonces)
        edata = eoff = None
        if extra is not None:
            edata, eoff = extra if isinstance(extra, tuple) else self.pack(list(extra))
        self._check_rp_shapes("rangeproof_rewind_batch", n, commits33, gens64, data, off, edata, eoff)
        _need("rangeproof_rewind_batch nonces", nonces, 32 * n)
        res = np.zeros(n, np.int32); mn = np.zeros(n, np.uint64); mx = np.zeros(n, np.uint64)
        blind = np.zeros((n, 32), np.uint8); val = np.zeros(n, np.uint64)
        msg = np.zeros((n, max(msg_capacity, 1)), np.uint8) if msg_capacity else None
        ol = np.full(n, msg_capacity, np.uint64)
        self._check(self._lib.secp256k1_rangeproof_rewind_batch(self._h, _p(res), _p(blind), _p(val), _p(msg), _p(ol) if msg_capacity else None, msg_capacity,
                                                                 _p(nonces), _p(mn), _p(mx), _p(commits33), _p(data), _p(off), _p(edata), _p(eoff), _p(gens64), n),
                    "secp256k1_rangeproof_rewind_batch")
        msgs = [msg[i, :int(ol[i])].tobytes() if (msg_capacity and res[i]) else b"" for i in range(n)]
        return res, blind, val, msgs, mn, mx

    def rangeproof_verify_batch_dev(self, results, min_value, max_value, commits33, proofs, proof_off, gens64, n, extra=None, extra_off=None,
                                    stream=None):
        self._check(self._lib.secp256k1_rangeproof_verify_batch_dev(self._h, stream, _dp(results), _dp(min_value), _dp(max_value), _dp(commits33),
                                                                     _dp(proofs), _dp(proof_off), _dp(extra), _dp(extra_off), _dp(gens64), n),
                    "secp256k1_rangeproof_verify_batch_dev")

    # ---- secp256k1_surjectionproof_verify (modules/surjection/main_impl.h:360-402), batched ---------------------
    def surjectionproof_verify_batch(self, proofs, input_tags, output_tags64):
        """proofs: list of serialised proofs; input_tags: list of (k_i,64) uint8 arrays; output_tags64: (n,64)."""
        data, off = self.pack(list(proofs))
        n = off.size - 1
        toff = np.zeros(n + 1, np.uint64)
        toff[1:] = np.cumsum([np.asarray(t).size // 64 for t in input_tags], dtype=np.uint64)
        tags = np.concatenate([_u8(t).reshape(-1) for t in input_tags] + [np.zeros(64, np.uint8)])
        if len(input_tags) != n:
            raise ValueError("surjectionproof_verify_batch: one input-tag list per proof")
        _need("surjectionproof_verify_batch output_tags64", _u8(output_tags64), 64 * n)
        res = np.zeros(n, np.int32)
        self._check(self._lib.secp256k1_surjectionproof_verify_batch(self._h, _p(res), _p(data), _p(off), _p(tags), _p(toff), _p(_u8(output_tags64)), n),
                    "secp256k1_surjectionproof_verify_batch")
        return res

    # ---- secp256k1_bppp_rangeproof_norm_product_verify (modules/bppp/bppp_norm_product_impl.h:425-552), batched --
    def bppp_norm_product_verify_batch(self, proofs, transcripts, rho, gens33, g_len, c_vec, commits33):
        proofs = _u8(proofs); n = _u8(rho).size // 32
        proof_len = proofs.size // max(n, 1)
        gens33 = _u8(gens33); n_gens = gens33.size // 33
        c_vec = _u8(c_vec); c_len = c_vec.size // 32 // max(n, 1)
        _need("bppp_norm_product_verify_batch proofs", proofs, proof_len * n); _need("bppp_norm_product_verify_batch transcripts", _u8(transcripts), 104 * n)
        _need("bppp_norm_product_verify_batch c_vec", c_vec, 32 * c_len * n); _need("bppp_norm_product_verify_batch commits33", _u8(commits33), 33 * n)
        _need("bppp_norm_product_verify_batch gens33", gens33, 33 * n_gens)
        res = np.zeros(n, np.int32)
        self._check(self._lib.secp256k1_bppp_norm_product_verify_batch(self._h, _p(res), _p(proofs), proof_len, _p(_u8(transcripts)), _p(_u8(rho)),
                                                                        _p(gens33), n_gens, g_len, _p(c_vec), c_len, _p(_u8(commits33)), n),
                    "secp256k1_bppp_norm_product_verify_batch")
        return res


class Group:
    """The GPUs of one node behind one handle (s2k_group, include/secp256k1_zkp_amd.h): one engine and one host thread per entry of
    `devices`; batches of independent items are cut into contiguous ranges, one large multi-scalar multiplication is sharded by terms."""

    def __init__(self, devices):
        self._lib = _native.load()
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        self._h = self._lib.s2k_group_create(devs, len(devices))
        if not self._h:
            raise S2KError("s2k_group_create failed: " + _native.last_error())
        self.devices = [int(d) for d in devices]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.s2k_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self._lib.s2k_group_size(self._h))

    def _check(self, ok, what):
        if not ok:
            raise S2KError(f"{what} failed: {_native.last_error()}")

    def engine(self, i):
        """a non-owning Engine view of member i (options, generator cache)"""
        h = self._lib.s2k_group_engine(self._h, int(i))
        if not h:
            raise IndexError(i)
        e = Engine.__new__(Engine)
        e._lib = self._lib; e._h = h; e.device = self.devices[i]
        e.close = lambda: None                   # the group owns it
        return e

    def rangeproof_verify_batch(self, commits33, proofs, gens64, extra=None):
        data, off = proofs if isinstance(proofs, tuple) else Engine.pack(list(proofs))
        n = off.size - 1
        commits33 = _u8(commits33); gens64 = _u8(gens64)
        edata = eoff = None
        if extra is not None:
            edata, eoff = extra if isinstance(extra, tuple) else Engine.pack(list(extra))
        Engine._check_rp_shapes("rangeproof_verify_batch_group", n, commits33, gens64, data, off, edata, eoff)
        res = np.zeros(n, np.int32); mn = np.zeros(n, np.uint64); mx = np.zeros(n, np.uint64)
        self._check(self._lib.secp256k1_rangeproof_verify_batch_group(self._h, _p(res), _p(mn), _p(mx), _p(commits33), _p(data), _p(off), _p(edata), _p(eoff),
                                                                       _p(gens64), n), "secp256k1_rangeproof_verify_batch_group")
        return res, mn, mx

    def schnorrsig_verify_batch(self, sigs, msgs, pubkeys, msglen=32, pk_format=0):
        sigs = _u8(sigs); msgs = _u8(msgs); pubkeys = _u8(pubkeys); n = sigs.size // 64
        _need("schnorrsig_verify_batch_group sigs", sigs, 64 * n); _need("schnorrsig_verify_batch_group msgs", msgs, msglen * n)
        _need("schnorrsig_verify_batch_group pubkeys", pubkeys, (64 if pk_format else 32) * n)
        res = np.zeros(n, np.int32)
        self._check(self._lib.secp256k1_schnorrsig_verify_batch_group(self._h, _p(res), _p(sigs), _p(msgs), msglen, _p(pubkeys), pk_format, n),
                    "secp256k1_schnorrsig_verify_batch_group")
        return res

    def ecmult_multi(self, sc, pt_xy, g_sc=None, pt_inf=None):
        sc = _u8(sc); pt_xy = _u8(pt_xy); n = sc.size // 32
        g_sc = None if g_sc is None else _u8(g_sc); pt_inf = None if pt_inf is None else _u8(pt_inf)
        _need("ecmult_multi_group sc", sc, 32 * n); _need("ecmult_multi_group pt_xy", pt_xy, 64 * n)
        if g_sc is not None: _need("ecmult_multi_group g_sc", g_sc, 32)
        if pt_inf is not None: _need("ecmult_multi_group pt_inf", pt_inf, n)
        r = np.zeros(64, np.uint8); inf = np.zeros(1, np.int32)
        self._check(self._lib.s2k_ecmult_multi_group(self._h, _p(r), _p(inf), _p(g_sc), _p(sc), _p(pt_xy), _p(pt_inf), n), "s2k_ecmult_multi_group")
        return r, int(inf[0])

    def ecmult_multi_many(self, sc, pt_xy, offsets, g_sc=None, pt_inf=None):
        """K independent sums over the members of the group (s2k_ecmult_multi_many per member): the sums are independent objects, so they are
        cut into contiguous ranges of about equal numbers of TERMS, one range per member, one host thread per member (the call releases
        the GIL) -- no exchange between the GPUs, results in the caller's order.  A C caller does the same split over the engines of its
        s2k_group (s2k_group_engine); the reference has no counterpart (one sum per secp256k1_ecmult_multi_var call, ecmult_impl.h:822-867)."""
        import threading
        sc = _u8(sc); pt_xy = _u8(pt_xy); off = np.ascontiguousarray(offsets, dtype=np.uint64); k = off.size - 1
        n = int(off[-1]) if off.size else 0
        g_sc = None if g_sc is None else _u8(g_sc); pt_inf = None if pt_inf is None else _u8(pt_inf)
        _need("ecmult_multi_many_group sc", sc, 32 * n); _need("ecmult_multi_many_group pt_xy", pt_xy, 64 * n)
        if g_sc is not None: _need("ecmult_multi_many_group g_sc", g_sc, 32 * k)
        if pt_inf is not None: _need("ecmult_multi_many_group pt_inf", pt_inf, n)
        if k > 0 and (int(off[0]) != 0 or np.any(off[1:] < off[:-1])):
            raise S2KError("ecmult_multi_many_group failed: offsets must start at 0 and not decrease")
        from .parallel import shard_sums
        m = len(self)
        cut = shard_sums(off, m)                 # member i takes sums [cut[i], cut[i + 1])
        r = np.zeros((max(k, 0), 64), np.uint8); inf = np.zeros(max(k, 0), np.int32)
        errs = [None] * m

        def run(i):
            a, b = cut[i], cut[i + 1]
            if b <= a: return
            t0, t1 = int(off[a]), int(off[b])
            try:
                xy, fl = self.engine(i).ecmult_multi_many(sc.reshape(-1)[32 * t0:32 * t1], pt_xy.reshape(-1)[64 * t0:64 * t1], off[a:b + 1] - off[a],
                                                          None if g_sc is None else g_sc.reshape(-1)[32 * a:32 * b], None if pt_inf is None else pt_inf.reshape(-1)[t0:t1])
                r[a:b] = xy; inf[a:b] = fl
            except Exception as ex:                  # (reported by the calling thread)
                errs[i] = ex
        ts = [threading.Thread(target=run, args=(i,)) for i in range(m)]
        for t in ts: t.start()
        for t in ts: t.join()
        for ex in errs:
            if ex is not None: raise ex
        return r, inf

    def ecmult_multi_dev(self, sc_list, pt_list, g_sc_dev0=None, inf_list=None):
        """per engine: torch uint8 tensors resident on that engine's GPU (sc (n_i,32), pt (n_i,64)); returns (xy bytes, inf)"""
        k = len(self)
        assert len(sc_list) == k and len(pt_list) == k
        vp = ctypes.c_void_p * k
        a = vp(*[t.data_ptr() for t in sc_list]); b = vp(*[t.data_ptr() for t in pt_list])
        c = None if inf_list is None else vp(*[t.data_ptr() for t in inf_list])
        cnt = (ctypes.c_size_t * k)(*[t.numel() // 32 for t in sc_list])
        r = np.zeros(64, np.uint8); inf = np.zeros(1, np.int32)
        self._check(self._lib.s2k_ecmult_multi_group_dev(self._h, _p(r), _p(inf), _dp(g_sc_dev0), a, b, c, cnt), "s2k_ecmult_multi_group_dev")
        return r, int(inf[0])
