"""Crafted Borromean rangeproofs for the parity tests (TEST INFRASTRUCTURE; every group operation goes through oracle/_ref).

What an adversarial prover can choose that an honest one never produces, and what the reference does with it:

* a ring key at infinity.  The four keys of ring i are P_j = C_i + j*B_i with B_i = -(4^i 10^exp)*H
  (src/modules/rangeproof/rangeproof_impl.h:19-51), and C_i comes straight from the proof bytes: C_i = -j*B_i makes P_j the point at
  infinity.  R_j = s_j*G + e_j*P_j then does not depend on e_j, so the hash chain of that ring no longer depends on e0 and a forger
  can close the loop without knowing any secret: `forge_infinity_keys` builds such a proof with an infinite key in EVERY ring -- it
  satisfies every equation secp256k1_borromean_verify checks EXCEPT the explicit rejection of infinite keys
  (src/modules/rangeproof/borromean_impl.h:78).  A verifier without that check accepts it; the reference returns 0.
* R_j = infinity (borromean_impl.h:84-86): with P_j = k*G of known k the forger picks s_j = -e_j*k (`forge_r_infinity`).
* a VALID proof whose verification runs into an exceptional addition: `sign` is a plain Borromean signer over the reference's group
  operations (it doubles as the check that this file's hashing matches the reference: its proofs verify), and
  `grind_exceptional_doubling` chooses the nonce of a ring so that, when the verifier evaluates s*G + e*P with the generator part taken
  digit by digit from the low end (signed 26-bit digits, the engine's fixed-base table), the accumulator in front of the LAST window equals
  that window's table entry: P + P inside an addition chain, on a proof the reference accepts.
"""
import hashlib

import numpy as np

from tests.refapi import GENERATOR_H, G_XY, N, P


def _b(k):
    return int(k % N).to_bytes(32, "big")


def _sha(*parts):
    return hashlib.sha256(b"".join(parts)).digest()


def _is_square(y):
    return y == 0 or pow(y, (P - 1) // 2, P) == 1


class Crafter:
    def __init__(self, ref, gen64=GENERATOR_H):
        self.ref = ref
        self.gen = bytes(gen64)
        self.G = np.frombuffer(G_XY, np.uint8).reshape(1, 64)
        self.H = np.frombuffer(self.gen, np.uint8).reshape(1, 64)

    # ---- group operations through the reference ------------------------------------------------------------------
    def lin(self, kh, kg):
        """kh*H + kg*G -> 64-byte affine or None (infinity)"""
        r, inf = self.ref.ecmult_batch(self.H, np.frombuffer(_b(kh), np.uint8), np.frombuffer(_b(kg), np.uint8))
        return None if inf[0] else r[0].tobytes()

    def lin_many(self, pts, na, ng):
        """na[i]*pts[i] + ng[i]*G for arrays of 64-byte points / 32-byte scalars"""
        return self.ref.ecmult_batch(np.ascontiguousarray(pts), np.ascontiguousarray(na), np.ascontiguousarray(ng))

    def mul_add(self, pt64, e, s):
        """e*pt + s*G"""
        r, inf = self.ref.ecmult_batch(np.frombuffer(pt64, np.uint8).reshape(1, 64), np.frombuffer(_b(e), np.uint8), np.frombuffer(_b(s), np.uint8))
        return None if inf[0] else r[0].tobytes()

    # ---- serialisations ----------------------------------------------------------------------------------------------
    @staticmethod
    def ser33(pt64):                       # secp256k1_eckey_pubkey_serialize33
        return bytes([2 | (pt64[63] & 1)]) + pt64[:32]

    @staticmethod
    def ser_point(pt64):                   # secp256k1_rangeproof_serialize_point: [ !is_square(y) ] || x
        return bytes([0 if _is_square(int.from_bytes(pt64[32:], "big")) else 1]) + pt64[:32]

    @classmethod
    def commit33(cls, pt64):               # secp256k1_pedersen_commitment_serialize: 9 ^ is_square(y)
        return bytes([8 | cls.ser_point(pt64)[0]]) + pt64[:32]

    # ---- proof assembly (min_value = 0, exp = 0, mantissa = 2 * rings: every ring has four keys) --------------------------------
    def _assemble(self, ring_pts, e0, s):
        rings = len(s) // 4
        hdr = bytes([0x40, 2 * rings - 1])
        signs = bytearray((rings + 6) >> 3)
        xs = b""
        for i, c in enumerate(ring_pts):
            if self.ser_point(c)[0]:
                signs[i >> 3] |= 1 << (i & 7)
            xs += c[:32]
        return hdr + bytes(signs) + xs + e0 + b"".join(_b(x) for x in s)

    def _m(self, commit_pt, ring_pts, rings, extra=b""):
        hdr = bytes([0x40, 2 * rings - 1])
        return _sha(self.ser_point(commit_pt), self.ser_point(self.gen), hdr, b"".join(self.ser_point(c) for c in ring_pts), extra)

    @staticmethod
    def _hash_e(e, m, ring, pos):          # secp256k1_borromean_hash
        return _sha(e, m, ring.to_bytes(4, "big"), pos.to_bytes(4, "big"))

    # ---- an honest signer ------------------------------------------------------------------------------------------------
    def sign(self, rng, rings, value, nonces=None, s_override=None):
        """(commit33, proof) for `value` < 4^rings; nonces: optional {ring: k}; s_override: optional {(ring, pos): s} for forged positions"""
        rnd = lambda: int.from_bytes(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), "big") % (N - 1) + 1
        d = [(value >> (2 * i)) & 3 for i in range(rings)]
        x = [rnd() for _ in range(rings)]
        commit_pt = self.lin(value, sum(x))
        ring_pts = [self.lin(d[i] * 4**i, x[i]) for i in range(rings - 1)]
        key = lambda i, j: self.lin((d[i] - j) * 4**i, x[i])              # P_{i,j} = x_i G + (d_i - j) 4^i H
        m = self._m(commit_pt, ring_pts, rings)
        k = [(nonces or {}).get(i, rnd()) for i in range(rings)]
        s = [[(s_override or {}).get((i, j), rnd()) for j in range(4)] for i in range(rings)]
        outs = []
        for i in range(rings):
            R = self.lin(0, k[i])
            for j in range(d[i] + 1, 4):
                e = int.from_bytes(self._hash_e(self.ser33(R), m, i, j), "big")
                R = self.mul_add(key(i, j), e, s[i][j])
            outs.append(self.ser33(R))
        e0 = _sha(b"".join(outs), m)
        for i in range(rings):
            e = int.from_bytes(self._hash_e(e0, m, i, 0), "big")
            for j in range(d[i]):
                R = self.mul_add(key(i, j), e, s[i][j])
                e = int.from_bytes(self._hash_e(self.ser33(R), m, i, j + 1), "big")
            s[i][d[i]] = (k[i] - e * x[i]) % N
        return self.commit33(commit_pt), self._assemble(ring_pts, e0, [v for row in s for v in row])

    # ---- forgeries ---------------------------------------------------------------------------------------------------------
    def forge_infinity_keys(self, rng, rings, js=None, neg=False):
        """A proof with P_{i, js[i]} = infinity in every ring (js[i] in 1..3) that satisfies every other verification equation.
        neg: lift the ring commitments with the OTHER sign (C_i = +j*B_i: no key is infinite, but C_i still has the x of a multiple of
        the ring base -- the proof then simply fails its hash check, in the reference and here)."""
        rnd = lambda: int.from_bytes(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), "big") % (N - 1) + 1
        js = js or [int(rng.integers(1, 4)) for _ in range(rings)]
        sgn = -1 if neg else 1
        # C_i = -j_i B_i = j_i 4^i H; the last ring's first key is commit - sum C_i, so commit = sum_i j_i 4^i H (blinding factor 0)
        cs = [sgn * js[i] * 4**i for i in range(rings)]                         # first key of ring i = cs[i] * H
        ring_pts = [self.lin(cs[i], 0) for i in range(rings - 1)]
        commit_pt = self.lin(sum(cs), 0)
        m = self._m(commit_pt, ring_pts, rings)
        s = [[rnd() for _ in range(4)] for _ in range(rings)]
        if neg:
            return self.commit33(commit_pt), self._assemble(ring_pts, bytes(rng.integers(0, 256, 32, dtype=np.uint8)), [v for row in s for v in row])
        outs = []
        for i in range(rings):
            R = self.lin(0, s[i][js[i]])                                        # s*G + e*infinity
            for j in range(js[i] + 1, 4):
                e = int.from_bytes(self._hash_e(self.ser33(R), m, i, j), "big")
                R = self.mul_add(self.lin(cs[i] - j * 4**i, 0), e, s[i][j])
            outs.append(self.ser33(R))
        e0 = _sha(b"".join(outs), m)
        return self.commit33(commit_pt), self._assemble(ring_pts, e0, [v for row in s for v in row])

    def forge_r_infinity(self, rng, rings, ring, pos=0):
        """A proof in which R = s*G + e*P is the point at infinity at (ring, pos = 0): the ring's commitment is k*G with k known and
        s = -e*k.  (ring < rings - 1.)  The reference returns 0 (borromean_impl.h:84-86)."""
        assert pos == 0 and ring < rings - 1
        rnd = lambda: int.from_bytes(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), "big") % (N - 1) + 1
        ks = [rnd() for _ in range(rings - 1)]
        ring_pts = [self.lin(0, k) for k in ks]
        commit_pt = self.lin(0, rnd())
        m = self._m(commit_pt, ring_pts, rings)
        e0 = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
        s = [[rnd() for _ in range(4)] for _ in range(rings)]
        e = int.from_bytes(self._hash_e(e0, m, ring, 0), "big")
        s[ring][0] = (-e * ks[ring]) % N
        return self.commit33(commit_pt), self._assemble(ring_pts, e0, [v for row in s for v in row])

    @staticmethod
    def fixed_base_top(s, D):
        """(top digit, shift) of the engine's signed D-bit fixed-base recoding of the scalar s (csrc/ecmult.h: gtab_recode)"""
        W = (256 + D - 1) // D
        K = sum(1 << (D - 1 + D * w) for w in range(W - 1))
        return (s + K) >> (D * (W - 1)), D * (W - 1)

    def grind_exceptional_doubling(self, rng, D=26, w_lo=1, w_hi=None, chunk=1 << 15):
        """A VALID one-ring proof (mantissa 1: two keys; value 0: the real signature sits at position 0, whose key is C = x*G) with the nonce
        chosen so that, in a fixed-base evaluation of s*G by signed D-bit digits taken from the low end (the engine's table of G), the
        accumulator in front of the LAST window equals that window's table entry:  e*C + sum_{w < W-1} d_w 2^(D w) G == d_top 2^(D (W-1)) G,
        i.e. the nonce is k = 2 d_top 2^(D (W-1)) for the d_top that the resulting s = k - e*x really has.  e depends on k through the hash
        chain, so the candidates d_top = w_lo .. w_hi-1 are run through the ring (batched through the reference) until one fits: each
        fits with probability 2^-(256 - D (W-1)).  Returns (commit33, proof, d_top) or None."""
        W = (256 + D - 1) // D
        shift = D * (W - 1)
        w_hi = w_hi or (1 << (256 - shift))
        rnd = lambda: int.from_bytes(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), "big") % (N - 1) + 1
        x = rnd(); s1 = rnd()
        commit_pt = self.lin(0, x)
        hdr = bytes([0x40, 0])                                                  # exp 0, mantissa 1: one ring of two keys
        m = _sha(self.ser_point(commit_pt), self.ser_point(self.gen), hdr)
        key1 = np.frombuffer(self.lin(-1, x), np.uint8)                         # P_1 = C - H
        s1b = np.frombuffer(_b(s1), np.uint8)
        for lo in range(w_lo, w_hi, chunk):
            ws = range(lo, min(lo + chunk, w_hi)); cnt = len(ws)
            ks = [(2 * w << shift) % N for w in ws]
            R, _ = self.lin_many(np.tile(self.G, (cnt, 1)), np.frombuffer(b"".join(_b(k) for k in ks), np.uint8).reshape(cnt, 32), np.zeros((cnt, 32), np.uint8))
            es = b"".join(self._hash_e(self.ser33(R[t].tobytes()), m, 0, 1) for t in range(cnt))
            R, _ = self.lin_many(np.tile(key1, (cnt, 1)), np.frombuffer(es, np.uint8).reshape(cnt, 32), np.tile(s1b, (cnt, 1)))
            for t, w in enumerate(ws):
                e0 = _sha(self.ser33(R[t].tobytes()), m)
                e = int.from_bytes(self._hash_e(e0, m, 0, 0), "big") % N
                s0 = (ks[t] - e * x) % N
                if s0 != 0 and e != 0 and self.fixed_base_top(s0, D)[0] == w:
                    # the accumulator in front of the last window really is that window's entry
                    assert (e * x + s0 - (w << shift)) % N == (w << shift) % N
                    return self.commit33(commit_pt), hdr + e0 + _b(s0) + _b(s1), w
        return None

    # ---- a verifier WITHOUT the reference's two infinity rejections (what the forgeries are measured against) ---------------------------
    @staticmethod
    def _lift(x32, negate):
        x = int.from_bytes(x32, "big"); y = pow((x * x * x + 7) % P, (P + 1) // 4, P)
        assert y * y % P == (x * x * x + 7) % P
        if negate:
            y = (P - y) % P
        return x32 + y.to_bytes(32, "big")

    def _comb(self, pts, scalars):
        """sum scalars[i]*pts[i] -> 64 bytes or None"""
        n = len(pts)
        r, inf = self.ref.ecmult_multi(np.frombuffer(b"".join(_b(k) for k in scalars), np.uint8).reshape(n, 32), np.frombuffer(b"".join(pts), np.uint8).reshape(n, 64))
        return None if inf else r.tobytes()

    def unchecked_verify(self, commit33, proof):
        """The verification equations of secp256k1_rangeproof_verify for this file's proof shape (min_value 0, exp 0, even mantissa), with
        e*infinity = infinity evaluated instead of rejected: 1 when e0 closes the loop."""
        rings = (proof[1] + 1) // 2
        nsign = (rings + 6) >> 3
        signs = proof[2:2 + nsign]; off = 2 + nsign
        ring_pts = [self._lift(proof[off + 32 * i:off + 32 * i + 32], (signs[i >> 3] >> (i & 7)) & 1) for i in range(rings - 1)]
        off += 32 * (rings - 1)
        e0 = proof[off:off + 32]; off += 32
        s = [int.from_bytes(proof[off + 32 * t:off + 32 * t + 32], "big") for t in range(4 * rings)]
        commit_pt = self._lift(commit33[1:], commit33[0] & 1)
        m = self._m(commit_pt, ring_pts, rings)
        last = self._comb([commit_pt] + ring_pts, [1] + [N - 1] * (rings - 1))
        firsts = ring_pts + [last]
        outs = []
        for i in range(rings):
            e = self._hash_e(e0, m, i, 0)
            for j in range(4):
                key = firsts[i] if j == 0 else (self._comb([firsts[i], self.gen], [1, -j * 4**i]) if firsts[i] else self.lin(-j * 4**i, 0))
                ei = int.from_bytes(e, "big") % N
                R = self.mul_add(key, ei, s[4 * i + j]) if key else self.lin(0, s[4 * i + j])
                if R is None:
                    return 0
                if j < 3:
                    e = self._hash_e(self.ser33(R), m, i, j + 1)
            outs.append(self.ser33(R))
        return int(_sha(b"".join(outs), m) == e0)


class SurjectionCrafter:
    """Forged surjection proofs (one Borromean ring over the used inputs, keys T_out - T_in[j]:
    src/modules/surjection/surjection_impl.h:66-95).  An input tag EQUAL to the output tag makes its key the point at infinity:
    `forge_infinity` satisfies every equation of the verification except borromean_verify's rejection of such a key."""

    def __init__(self, ref):
        self.ref = ref
        self.c = Crafter(ref)

    @staticmethod
    def _msg(tags, out):                                   # secp256k1_surjection_genmessage (surjection_impl.h:19-37)
        return _sha(*[bytes([2 + (t[63] & 1)]) + t[:32] for t in list(tags) + [out]])

    def _sub(self, a64, b64):
        """a - b as 64 bytes or None"""
        nb = b64[:32] + ((P - int.from_bytes(b64[32:], "big")) % P).to_bytes(32, "big")
        r, inf = self.ref.ecmult_multi(np.frombuffer(_b(1) * 2, np.uint8).reshape(2, 32), np.frombuffer(a64 + nb, np.uint8).reshape(2, 64))
        return None if inf else r.tobytes()

    def _assemble(self, n_inputs, used, e0, s):
        bm = bytearray((n_inputs + 7) // 8)
        for u in used:
            bm[u >> 3] |= 1 << (u & 7)
        return bytes([n_inputs & 0xFF, n_inputs >> 8]) + bytes(bm) + e0 + b"".join(_b(v) for v in s)

    def forge_infinity(self, rng, n_inputs, used, inf_at):
        """used: sorted input indices of the ring; the tag of used[inf_at] equals the output tag.  Returns (proof, tags (n,64), out (64,))."""
        rnd = lambda: int.from_bytes(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), "big") % (N - 1) + 1
        out = self.ref.rand_point(rng)
        tags = [self.ref.rand_point(rng) for _ in range(n_inputs)]
        tags[used[inf_at]] = out
        m = self._msg(tags, out)
        s = [rnd() for _ in used]
        R = self.c.lin(0, s[inf_at])                                            # s*G + e*infinity
        for j in range(inf_at + 1, len(used)):
            e = int.from_bytes(Crafter._hash_e(Crafter.ser33(R), m, 0, j), "big")
            R = self.c.mul_add(self._sub(out, tags[used[j]]), e, s[j])
        e0 = _sha(Crafter.ser33(R), m)
        return self._assemble(n_inputs, used, e0, s), np.frombuffer(b"".join(tags), np.uint8).reshape(n_inputs, 64).copy(), np.frombuffer(out, np.uint8).copy()

    def forge_r_infinity(self, rng, n_inputs, used):
        """R = infinity at ring position 0: the first used input tag is T_out - k*G with k known and s_0 = -e*k."""
        rnd = lambda: int.from_bytes(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), "big") % (N - 1) + 1
        out = self.ref.rand_point(rng)
        tags = [self.ref.rand_point(rng) for _ in range(n_inputs)]
        k = rnd()
        tags[used[0]] = self._sub(out, self.c.lin(0, k))
        m = self._msg(tags, out)
        e0 = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
        e = int.from_bytes(Crafter._hash_e(e0, m, 0, 0), "big")
        s = [rnd() for _ in used]
        s[0] = (-e * k) % N
        return self._assemble(n_inputs, used, e0, s), np.frombuffer(b"".join(tags), np.uint8).reshape(n_inputs, 64).copy(), np.frombuffer(out, np.uint8).copy()

    def unchecked_verify(self, proof, tags, out):
        """secp256k1_surjectionproof_verify's equations with e*infinity = infinity evaluated instead of rejected"""
        tags = [bytes(t) for t in np.asarray(tags, np.uint8).reshape(-1, 64)]; out = bytes(out)
        n_inputs = proof[0] | (proof[1] << 8); bl = (n_inputs + 7) // 8
        used = [i for i in range(n_inputs) if (proof[2 + (i >> 3)] >> (i & 7)) & 1]
        data = proof[2 + bl:]
        e0 = data[:32]; s = [int.from_bytes(data[32 + 32 * j:64 + 32 * j], "big") for j in range(len(used))]
        m = self._msg(tags, out)
        e = Crafter._hash_e(e0, m, 0, 0)
        for j, u in enumerate(used):
            key = self._sub(out, tags[u])
            R = self.c.mul_add(key, int.from_bytes(e, "big") % N, s[j]) if key else self.c.lin(0, s[j])
            if R is None:
                return 0
            if j + 1 < len(used):
                e = Crafter._hash_e(Crafter.ser33(R), m, 0, j + 1)
        return int(_sha(Crafter.ser33(R), m) == e0)
