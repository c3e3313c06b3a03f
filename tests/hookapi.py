"""ctypes view of oracle/_ref/libsecp256k1_hooked.so: the unmodified reference + integration/secp256k1_amd_hook.c
(the reference-side adapter of the drop-in boundary).  Test-only."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOOKED_PATH = os.path.join(ROOT, "oracle", "_ref", "libsecp256k1_hooked.so")

_vp, _sz, _int = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
RP_FN = ctypes.CFUNCTYPE(_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz)
MSM_FN = ctypes.CFUNCTYPE(_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz)
SCH_FN = ctypes.CFUNCTYPE(_int, _vp, _vp, _vp, _vp, _sz, _vp, _int, _sz)
SJ_FN = ctypes.CFUNCTYPE(_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz)
TALLY_FN = ctypes.CFUNCTYPE(_int, _vp, _vp, _vp, _vp, _vp, _sz)
AGG_FN = ctypes.CFUNCTYPE(_int, _vp, _vp, _vp, _int, _vp, _sz, _vp, _sz)
REWIND_FN = ctypes.CFUNCTYPE(_int, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz)


class Backend(ctypes.Structure):
    """struct secp256k1_amd_backend (integration/secp256k1_amd_hook.h)"""
    _fields_ = [("engine", _vp), ("rangeproof_verify_batch", _vp), ("ecmult_multi", _vp), ("schnorrsig_verify_batch", _vp),
                ("surjectionproof_verify_batch", _vp), ("pedersen_verify_tally_batch", _vp), ("schnorrsig_aggverify", _vp), ("rangeproof_rewind_batch", _vp), ("rangeproof_verify_batch_ptrs", _vp), ("ecmult_batch", _vp),
                ("bppp_norm_product_verify_batch", _vp), ("rangeproof_verify_batch_ptrs_submit", _vp), ("rangeproof_verify_batch_wait", _vp)]


def fnptr(cfunc):
    return ctypes.cast(cfunc, _vp).value if cfunc is not None else None


def _ptr_array(bufs):
    """array of pointers to the given ctypes buffers / numpy rows (kept alive by the caller)"""
    arr = (_vp * len(bufs))()
    for i, b in enumerate(bufs):
        arr[i] = b.ctypes.data if isinstance(b, np.ndarray) else ctypes.addressof(b)
    return arr


class Hooked:
    SECP256K1_CONTEXT_NONE = 1

    def __init__(self):
        self.lib = ctypes.CDLL(HOOKED_PATH)
        L = self.lib
        L.secp256k1_context_create.restype = _vp; L.secp256k1_context_create.argtypes = [ctypes.c_uint]
        L.secp256k1_amd_set_backend.argtypes = [_vp]; L.secp256k1_amd_set_backend.restype = None
        L.secp256k1_amd_stats.argtypes = [ctypes.POINTER(_sz), ctypes.POINTER(_sz)]
        L.secp256k1_amd_rangeproof_verify_batch.argtypes = [_vp] * 10 + [_sz]
        L.secp256k1_amd_rangeproof_verify_batch_submit.argtypes = [_vp] * 11 + [_sz]
        L.secp256k1_amd_rangeproof_verify_batch_wait.argtypes = [_vp, ctypes.c_uint64]
        L.secp256k1_amd_schnorrsig_verify_batch.argtypes = [_vp, _vp, _vp, _vp, _sz, _vp, _sz]
        L.secp256k1_amd_surjectionproof_verify_batch.argtypes = [_vp] * 6 + [_sz]
        L.secp256k1_amd_pedersen_verify_tally_batch.argtypes = [_vp] * 6 + [_sz]
        L.secp256k1_amd_schnorrsig_aggverify.argtypes = [_vp, _vp, _vp, _sz, _vp, _sz]
        L.secp256k1_amd_rangeproof_rewind_batch.argtypes = [_vp] * 15 + [_sz]
        L.secp256k1_surjectionproof_parse.argtypes = [_vp, _vp, ctypes.c_char_p, _sz]
        L.hook_test_ecmult_multi.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, ctypes.c_long, ctypes.POINTER(_sz)]
        L.ref_bppp_norm_verify.argtypes = [ctypes.c_char_p, _sz, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, _sz, _sz, ctypes.c_char_p, _sz, ctypes.c_char_p]
        L.secp256k1_amd_set_msm_min_terms.argtypes = [_sz]; L.secp256k1_amd_set_msm_min_terms.restype = None
        L.hook_test_ecmult_batch.argtypes = [_vp] * 6 + [_sz]
        L.hook_test_bppp_batch.argtypes = [_vp, _vp, _sz, _vp, _vp, _vp, _sz, _sz, _vp, _sz, _vp, _sz]
        L.secp256k1_amd_set_msm_min_terms(0)       # the tests drive the MSM seam at every size; the default threshold has its own test
        self.ctx = L.secp256k1_context_create(self.SECP256K1_CONTEXT_NONE)
        self._keep = None

    def set_backend(self, engine=None, rangeproof=None, msm=None, schnorr=None, surjection=None, tally=None, aggverify=None, rewind=None, rangeproof_ptrs=None, ecmult_batch=None, bppp_batch=None,
                    rangeproof_submit=None, rangeproof_wait=None):
        """install function pointers (ctypes callbacks or raw addresses); all None -> CPU library"""
        def addr(f):
            return f if isinstance(f, int) or f is None else fnptr(f)
        if all(f is None for f in (rangeproof, msm, schnorr, surjection, tally, aggverify, rewind, rangeproof_ptrs, ecmult_batch, bppp_batch, rangeproof_submit, rangeproof_wait)):
            self.lib.secp256k1_amd_set_backend(None); self._keep = None
            return
        b = Backend(engine, addr(rangeproof), addr(msm), addr(schnorr), addr(surjection), addr(tally), addr(aggverify), addr(rewind), addr(rangeproof_ptrs), addr(ecmult_batch), addr(bppp_batch),
                    addr(rangeproof_submit), addr(rangeproof_wait))
        self._keep = (b, rangeproof, msm, schnorr, surjection, tally, aggverify, rewind, rangeproof_ptrs, ecmult_batch, bppp_batch, rangeproof_submit, rangeproof_wait)
        self.lib.secp256k1_amd_set_backend(ctypes.byref(b))

    def stats(self):
        a, b = _sz(0), _sz(0)
        self.lib.secp256k1_amd_stats(ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    # ---- batch entry points with the reference's own types (arrays of pointers to opaque objects) ----
    def rangeproof_verify_batch(self, commits33, plist, gens64, extra=None):
        n = len(plist)
        cobj = np.zeros((n, 64), np.uint8); cobj[:, :33] = np.ascontiguousarray(commits33, np.uint8).reshape(n, 33)      # 64-byte opaque objects
        gobj = np.ascontiguousarray(gens64, np.uint8).reshape(n, 64).copy()
        pbufs = [np.frombuffer(p if len(p) else b"\0", np.uint8).copy() for p in plist]
        plens = (_sz * n)(*[len(p) for p in plist])
        cp = _ptr_array([cobj[i] for i in range(n)]); gp = _ptr_array([gobj[i] for i in range(n)]); pp = _ptr_array(pbufs)
        res = (_int * n)(); mn = np.zeros(n, np.uint64); mx = np.zeros(n, np.uint64)
        ep = el = None
        if extra is not None:
            ebufs = [np.frombuffer(e if len(e) else b"\0", np.uint8).copy() for e in extra]
            ep = _ptr_array(ebufs); el = (_sz * n)(*[len(e) for e in extra])
        r = self.lib.secp256k1_amd_rangeproof_verify_batch(self.ctx, res, mn.ctypes.data, mx.ctypes.data, cp, pp, plens, ep, el, gp, n)
        assert r == 1
        return np.array(list(res), np.int32), mn, mx

    def rangeproof_verify_batch_submit(self, commits33, plist, gens64):
        """asynchronous adapter: -> ticket object for rangeproof_verify_batch_wait (which returns results, min, max)"""
        n = len(plist)
        cobj = np.zeros((n, 64), np.uint8); cobj[:, :33] = np.ascontiguousarray(commits33, np.uint8).reshape(n, 33)
        gobj = np.ascontiguousarray(gens64, np.uint8).reshape(n, 64).copy()
        pbufs = [np.frombuffer(p if len(p) else b"\0", np.uint8).copy() for p in plist]
        plens = (_sz * n)(*[len(p) for p in plist])
        cp = _ptr_array([cobj[i] for i in range(n)]); gp = _ptr_array([gobj[i] for i in range(n)]); pp = _ptr_array(pbufs)
        res = (_int * n)(); mn = np.zeros(n, np.uint64); mx = np.zeros(n, np.uint64)
        tk = ctypes.c_uint64(0)
        r = self.lib.secp256k1_amd_rangeproof_verify_batch_submit(self.ctx, ctypes.byref(tk), res, mn.ctypes.data, mx.ctypes.data, cp, pp, plens, None, None, gp, n)
        assert r == 1
        return (tk.value, res, mn, mx)

    def rangeproof_verify_batch_wait(self, ticket):
        tk, res, mn, mx = ticket
        assert self.lib.secp256k1_amd_rangeproof_verify_batch_wait(self.ctx, ctypes.c_uint64(tk)) == 1
        return np.array(list(res), np.int32), mn, mx

    def rangeproof_rewind_batch(self, commits33, plist, gens64, nonces, msg_capacity=4096):
        """-> (results, blinds (n,32), values, messages list, min, max): per item what secp256k1_rangeproof_rewind returns and writes"""
        n = len(plist)
        cobj = np.zeros((n, 64), np.uint8); cobj[:, :33] = np.ascontiguousarray(commits33, np.uint8).reshape(n, 33)
        gobj = np.ascontiguousarray(gens64, np.uint8).reshape(n, 64).copy()
        nn = np.ascontiguousarray(nonces, np.uint8).reshape(n, 32).copy()
        pbufs = [np.frombuffer(p if len(p) else b"\0", np.uint8).copy() for p in plist]
        plens = (_sz * n)(*[len(p) for p in plist])
        res = (_int * n)(); mn = np.zeros(n, np.uint64); mx = np.zeros(n, np.uint64)
        blind = np.full((n, 32), 0xAA, np.uint8); val = np.full(n, 77, np.uint64)
        mbufs = [np.zeros(max(msg_capacity, 1), np.uint8) for _ in range(n)]
        ol = (_sz * n)(*([msg_capacity] * n))
        r = self.lib.secp256k1_amd_rangeproof_rewind_batch(self.ctx, res, blind.ctypes.data, val.ctypes.data, _ptr_array(mbufs) if msg_capacity else None,
                                                           ol if msg_capacity else None, _ptr_array([nn[i] for i in range(n)]), mn.ctypes.data, mx.ctypes.data,
                                                           _ptr_array([cobj[i] for i in range(n)]), _ptr_array(pbufs), plens, None, None, _ptr_array([gobj[i] for i in range(n)]), n)
        assert r == 1
        res = np.array(list(res), np.int32)
        msgs = [mbufs[i][:ol[i]].tobytes() if (res[i] and msg_capacity) else b"" for i in range(n)]
        if msg_capacity:
            assert all(ol[i] == 0 for i in range(n) if not res[i])
        return res, blind, val, msgs, mn, mx

    def schnorrsig_verify_batch(self, sigs, msgs, pk_objs, msglen=32):
        sigs = np.ascontiguousarray(sigs, np.uint8).reshape(-1, 64); n = sigs.shape[0]
        msgs = np.ascontiguousarray(msgs, np.uint8).reshape(n, max(msglen, 1)); pk_objs = np.ascontiguousarray(pk_objs, np.uint8).reshape(n, 64)
        res = (_int * n)()
        r = self.lib.secp256k1_amd_schnorrsig_verify_batch(self.ctx, res, _ptr_array([sigs[i] for i in range(n)]), _ptr_array([msgs[i] for i in range(n)]), msglen,
                                                           _ptr_array([pk_objs[i] for i in range(n)]), n)
        assert r == 1
        return np.array(list(res), np.int32)

    def schnorrsig_aggverify(self, pk_objs, msgs32, aggsig):
        """secp256k1_schnorrsig_aggverify's own argument list: an array of xonly_pubkey objects, n*32 message bytes, the aggregate"""
        pk_objs = np.ascontiguousarray(pk_objs, np.uint8).reshape(-1, 64); n = pk_objs.shape[0]
        msgs32 = np.ascontiguousarray(msgs32, np.uint8).reshape(n, 32)
        agg = np.frombuffer(bytes(aggsig), np.uint8).copy()
        return int(self.lib.secp256k1_amd_schnorrsig_aggverify(self.ctx, pk_objs.ctypes.data, msgs32.ctypes.data, n, agg.ctypes.data, agg.size))

    def surjectionproof_verify_batch(self, items):
        """items: list of (serialised proof, input tags (k,64), output tag (64,)); proofs that do not parse are the caller's problem"""
        n = len(items)
        objs = [ctypes.create_string_buffer(8 + 32 + 32 * 257 + 64) for _ in range(n)]
        for o, (ser, _, _) in zip(objs, items):
            assert self.lib.secp256k1_surjectionproof_parse(self.ctx, o, ser, len(ser)) == 1
        tags = [np.ascontiguousarray(t, np.uint8).reshape(-1, 64).copy() for _, t, _ in items]
        outs = [np.ascontiguousarray(o, np.uint8).reshape(64).copy() for _, _, o in items]
        nt = (_sz * n)(*[t.shape[0] for t in tags])
        res = (_int * n)()
        r = self.lib.secp256k1_amd_surjectionproof_verify_batch(self.ctx, res, _ptr_array(objs), _ptr_array(tags), nt, _ptr_array(outs), n)
        assert r == 1
        return np.array(list(res), np.int32)

    def pedersen_verify_tally_batch(self, tallies):
        """tallies: list of (pos (k,33), neg (m,33)) of parseable commitments"""
        n = len(tallies); keep = []; pos_arrs = (_vp * n)(); neg_arrs = (_vp * n)(); pc = (_sz * n)(); nc = (_sz * n)()
        for t, (pos, neg) in enumerate(tallies):
            for which, arrs, cnt in ((pos, pos_arrs, pc), (neg, neg_arrs, nc)):
                c = np.ascontiguousarray(which, np.uint8).reshape(-1, 33); k = c.shape[0]
                obj = np.zeros((max(k, 1), 64), np.uint8); obj[:k, :33] = c
                pa = _ptr_array([obj[i] for i in range(k)]) if k else (_vp * 1)()
                keep += [obj, pa]; arrs[t] = ctypes.addressof(pa); cnt[t] = k
        res = (_int * n)()
        r = self.lib.secp256k1_amd_pedersen_verify_tally_batch(self.ctx, res, pos_arrs, pc, neg_arrs, nc, n)
        assert r == 1
        return np.array(list(res), np.int32)

    MSM_MIN_TERMS_DEFAULT = 256       # SECP256K1_AMD_MSM_MIN_TERMS_DEFAULT (integration/secp256k1_amd_hook.c)

    def set_msm_min_terms(self, n):
        self.lib.secp256k1_amd_set_msm_min_terms(n)

    def ecmult_batch(self, a_xy, na, ng=None, a_inf=None):
        """r[i] = na[i]*a[i] + ng[i]*G through secp256k1_ecmult_batch_amd (reference types inside the library)"""
        a_xy = np.ascontiguousarray(a_xy, np.uint8); n = a_xy.size // 64
        na = np.ascontiguousarray(na, np.uint8); ng = None if ng is None else np.ascontiguousarray(ng, np.uint8)
        ai = None if a_inf is None else np.ascontiguousarray(a_inf, np.uint8)
        r = np.zeros((n, 64), np.uint8); inf = np.zeros(n, np.int32)
        p = lambda a: None if a is None else a.ctypes.data
        assert self.lib.hook_test_ecmult_batch(p(r), p(inf), p(a_xy), p(ai), p(na), p(ng), n) == 1
        return r, inf

    def bppp_verify_batch(self, proofs, trs, rhos, gens, g_len, cvs, commits):
        """n proofs through secp256k1_amd_bppp_norm_product_verify_batch; arrays as ref.make_bppp returns them"""
        proofs = np.ascontiguousarray(proofs, np.uint8); n = proofs.shape[0]
        trs = np.ascontiguousarray(trs, np.uint8); rhos = np.ascontiguousarray(rhos, np.uint8); gens = np.ascontiguousarray(gens, np.uint8)
        cvs = np.ascontiguousarray(cvs, np.uint8); commits = np.ascontiguousarray(commits, np.uint8)
        res = (_int * n)()
        ok = self.lib.hook_test_bppp_batch(res, proofs.ctypes.data, proofs.shape[1], trs.ctypes.data, rhos.ctypes.data, gens.ctypes.data, gens.shape[0], g_len,
                                           cvs.ctypes.data, cvs.shape[1], commits.ctypes.data, n)
        assert ok == 1
        return np.array(list(res), np.int32)

    def ecmult_multi(self, sc, pt_xy, g_sc=None, pt_inf=None, fail_at=-1):
        """the MSM seam through secp256k1_ecmult_multi_var_amd: returns (xy, inf or -1, callback invocations)"""
        sc = np.ascontiguousarray(sc, np.uint8); pt_xy = np.ascontiguousarray(pt_xy, np.uint8); n = sc.size // 32
        g = None if g_sc is None else np.frombuffer(bytes(g_sc), np.uint8).copy()
        pi = None if pt_inf is None else np.ascontiguousarray(pt_inf, np.uint8)
        r = np.zeros(64, np.uint8); calls = _sz(0)
        p = lambda a: None if a is None else a.ctypes.data
        inf = self.lib.hook_test_ecmult_multi(p(r), p(g), p(sc), p(pt_xy), p(pi), n, fail_at, ctypes.byref(calls))
        return r, inf, calls.value
