// sha256.h -- SHA-256 for the verification kernels (the hash sits *inside* the Borromean ring chain).
//
// Role of the reference's src/hash_impl.h (transform :51-138, write/finalize :145-194).  Two flavours:
//   * sha256_compress(): one 64-byte block held as 16 big-endian words in registers, fully unrolled, static
//     indexing only -- used on the hot path where the message layout is fixed (borromean_hash, BIP-340 challenge);
//   * sha256_stream: a byte-oriented streaming context for the cold per-proof kernels (prologue/epilogue),
//     where message boundaries are data dependent.
#pragma once
#include "s2k_common.h"

S2K_HD u32 sha_rotr(u32 x, int n) { return (x >> n) | (x << (32 - n)); }

S2K_HD void sha256_init(u32 s[8]) {
    s[0] = 0x6a09e667u; s[1] = 0xbb67ae85u; s[2] = 0x3c6ef372u; s[3] = 0xa54ff53au;
    s[4] = 0x510e527fu; s[5] = 0x9b05688cu; s[6] = 0x1f83d9abu; s[7] = 0x5be0cd19u;
}

#define S2K_SHA_K(i) sha256_k[i]
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__
#endif
static const u32 sha256_k[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};

// three-way XOR as ONE instruction on the device (v_bitop3_b32 with the parity truth table: the compiler forms it for Ch and Maj
// but leaves the sigma functions as two v_xor_b32 each: 224 instructions of a 1 709-instruction block)
#if defined(__HIP_DEVICE_COMPILE__)
#define S2K_XOR3(a, b, c) ((u32)__builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96))
#define S2K_CH(e, f, g) ((u32)__builtin_amdgcn_bitop3_b32((e), (f), (g), 0xCA))          /* e ? f : g, bitwise */
#define S2K_MAJ(a, b, c) ((u32)__builtin_amdgcn_bitop3_b32((a), (b), (c), 0xE8))
#else
#define S2K_XOR3(a, b, c) ((a) ^ (b) ^ (c))
#define S2K_CH(e, f, g) (((e) & (f)) ^ (~(e) & (g)))
#define S2K_MAJ(a, b, c) (((a) & (b)) ^ ((a) & (c)) ^ ((b) & (c)))
#endif
// s <- compress(s, w); w is consumed (used as the rolling message schedule).
S2K_HD void sha256_compress(u32 s[8], u32 w[16]) {
    u32 a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const u32 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            const u32 s0 = S2K_XOR3(sha_rotr(w15, 7), sha_rotr(w15, 18), w15 >> 3);
            const u32 s1 = S2K_XOR3(sha_rotr(w2, 17), sha_rotr(w2, 19), w2 >> 10);
            w[i & 15] += s0 + w[(i + 9) & 15] + s1;
        }
        const u32 S1 = S2K_XOR3(sha_rotr(e, 6), sha_rotr(e, 11), sha_rotr(e, 25));
        const u32 ch = S2K_CH(e, f, g);
        const u32 t1 = h + S1 + ch + S2K_SHA_K(i) + w[i & 15];
        const u32 S0 = S2K_XOR3(sha_rotr(a, 2), sha_rotr(a, 13), sha_rotr(a, 22));
        const u32 mj = S2K_MAJ(a, b, c);
        const u32 t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    s[0] += a; s[1] += b; s[2] += c; s[3] += d; s[4] += e; s[5] += f; s[6] += g; s[7] += h;
}

// ---- streaming context (cold paths) -----------------------------------------------------------------------
struct sha256_stream {
    u32 s[8];
    u32 buf[16];     // big-endian words of the partial block
    u64 bytes;
};
S2K_HD void sha256_stream_init(sha256_stream& c) {
    sha256_init(c.s);
    for (int i = 0; i < 16; i++) c.buf[i] = 0;
    c.bytes = 0;
}
S2K_HD void sha256_stream_put(sha256_stream& c, unsigned char v) {
    const u32 pos = (u32)(c.bytes & 63);
    const u32 wi = pos >> 2, sh = 24 - 8 * (pos & 3);
    c.buf[wi] = (c.buf[wi] & ~(0xFFu << sh)) | ((u32)v << sh);
    c.bytes++;
    if (pos == 63) {
        u32 w[16];
        for (int i = 0; i < 16; i++) w[i] = c.buf[i];
        sha256_compress(c.s, w);
    }
}
S2K_HD void sha256_stream_write(sha256_stream& c, const unsigned char* p, size_t n) {
    for (size_t i = 0; i < n; i++) sha256_stream_put(c, p[i]);
}
S2K_HD void sha256_stream_finalize(sha256_stream& c, unsigned char out[32]) {
    const u64 bits = c.bytes << 3;
    sha256_stream_put(c, 0x80);
    while ((c.bytes & 63) != 56) sha256_stream_put(c, 0);
    for (int i = 7; i >= 0; i--) sha256_stream_put(c, (unsigned char)(bits >> (8 * i)));
    for (int i = 0; i < 8; i++) s2k_store_be32(out + 4 * i, c.s[i]);
}
