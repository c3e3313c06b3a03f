// host_sha256.h -- SHA-256 compression on the HOST, for the one stage of this engine that is a serial hash chain over the whole input:
// the randomizer hash of half-aggregate verification (halfagg.h; src/modules/schnorrsig_halfagg/main_impl.h:153-163), where z_i hashes
// the prefix r_0|pk_0|m_0|...|r_i|pk_i|m_i.  A Merkle-Damgard chain cannot be spread over lanes: one wavefront of the GPU needs ~2.4 us
// per 64-byte block (118 ms for 2^15 signatures, the whole rest of the verification is ~1.5 ms), a host core with the SHA extensions
// ~40 ns.  The host-buffer entry point therefore walks the chain here -- underneath the point-lifting kernel it has already queued --
// and uploads the chain states; finalising every z_i, the challenges and the multi-scalar multiplication stay on the device, and the
// `_dev` form (inputs already in HBM) keeps the device chain.  This is not a CPU path for the verification: nothing is decided here.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

static const uint32_t k_host_sha256_k[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
    0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
    0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
    0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};

static inline uint32_t host_sha_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
// one block, portable
static inline void host_sha256_block(uint32_t s[8], const unsigned char* p) {
    uint32_t w[64];
    for (int t = 0; t < 16; t++) w[t] = ((uint32_t)p[4 * t] << 24) | ((uint32_t)p[4 * t + 1] << 16) | ((uint32_t)p[4 * t + 2] << 8) | (uint32_t)p[4 * t + 3];
    for (int t = 16; t < 64; t++) {
        const uint32_t s0 = host_sha_rotr(w[t - 15], 7) ^ host_sha_rotr(w[t - 15], 18) ^ (w[t - 15] >> 3);
        const uint32_t s1 = host_sha_rotr(w[t - 2], 17) ^ host_sha_rotr(w[t - 2], 19) ^ (w[t - 2] >> 10);
        w[t] = w[t - 16] + s0 + w[t - 7] + s1;
    }
    uint32_t a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
    for (int t = 0; t < 64; t++) {
        const uint32_t t1 = h + (host_sha_rotr(e, 6) ^ host_sha_rotr(e, 11) ^ host_sha_rotr(e, 25)) + ((e & f) ^ (~e & g)) + k_host_sha256_k[t] + w[t];
        const uint32_t t2 = (host_sha_rotr(a, 2) ^ host_sha_rotr(a, 13) ^ host_sha_rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    s[0] += a; s[1] += b; s[2] += c; s[3] += d; s[4] += e; s[5] += f; s[6] += g; s[7] += h;
}
#if defined(__x86_64__)
// one block with the SHA extensions (sha256rnds2 / sha256msg1 / sha256msg2); state kept in the caller's uint32[8] between blocks
__attribute__((target("sha,sse4.1,ssse3"))) static inline void host_sha256_block_ni(uint32_t s[8], const unsigned char* p) {
    const __m128i mask = _mm_set_epi64x(0x0c0d0e0f08090a0bLL, 0x0405060700010203LL);
    __m128i tmp = _mm_loadu_si128((const __m128i*)&s[0]);
    __m128i st1 = _mm_loadu_si128((const __m128i*)&s[4]);
    tmp = _mm_shuffle_epi32(tmp, 0xB1);                  // CDAB
    st1 = _mm_shuffle_epi32(st1, 0x1B);                  // EFGH
    __m128i st0 = _mm_alignr_epi8(tmp, st1, 8);          // ABEF
    st1 = _mm_blend_epi16(st1, tmp, 0xF0);               // CDGH
    const __m128i abef = st0, cdgh = st1;
    __m128i m0 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 0)), mask);
    __m128i m1 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16)), mask);
    __m128i m2 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 32)), mask);
    __m128i m3 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 48)), mask);
    __m128i msg;
#define HS_K(i) _mm_loadu_si128((const __m128i*)&k_host_sha256_k[4 * (i)])
#define HS_RND(m, i) msg = _mm_add_epi32(m, HS_K(i)); st1 = _mm_sha256rnds2_epu32(st1, st0, msg); msg = _mm_shuffle_epi32(msg, 0x0E); st0 = _mm_sha256rnds2_epu32(st0, st1, msg)
    HS_RND(m0, 0);
    HS_RND(m1, 1); m0 = _mm_sha256msg1_epu32(m0, m1);
    HS_RND(m2, 2); m1 = _mm_sha256msg1_epu32(m1, m2);
    // rounds 12..59: the message words are extended four at a time
#define HS_EXT(ma, mb, mc, md, i) \
    msg = _mm_add_epi32(md, HS_K(i)); st1 = _mm_sha256rnds2_epu32(st1, st0, msg); \
    tmp = _mm_alignr_epi8(md, mc, 4); ma = _mm_add_epi32(ma, tmp); ma = _mm_sha256msg2_epu32(ma, md); \
    msg = _mm_shuffle_epi32(msg, 0x0E); st0 = _mm_sha256rnds2_epu32(st0, st1, msg); mc = _mm_sha256msg1_epu32(mc, md)
    HS_EXT(m0, m1, m2, m3, 3);
    HS_EXT(m1, m2, m3, m0, 4);
    HS_EXT(m2, m3, m0, m1, 5);
    HS_EXT(m3, m0, m1, m2, 6);
    HS_EXT(m0, m1, m2, m3, 7);
    HS_EXT(m1, m2, m3, m0, 8);
    HS_EXT(m2, m3, m0, m1, 9);
    HS_EXT(m3, m0, m1, m2, 10);
    HS_EXT(m0, m1, m2, m3, 11);
    HS_EXT(m1, m2, m3, m0, 12);
    // rounds 52..55 and 56..59 still extend m2 and m3, without a further msg1
    msg = _mm_add_epi32(m1, HS_K(13)); st1 = _mm_sha256rnds2_epu32(st1, st0, msg);
    tmp = _mm_alignr_epi8(m1, m0, 4); m2 = _mm_add_epi32(m2, tmp); m2 = _mm_sha256msg2_epu32(m2, m1);
    msg = _mm_shuffle_epi32(msg, 0x0E); st0 = _mm_sha256rnds2_epu32(st0, st1, msg);
    msg = _mm_add_epi32(m2, HS_K(14)); st1 = _mm_sha256rnds2_epu32(st1, st0, msg);
    tmp = _mm_alignr_epi8(m2, m1, 4); m3 = _mm_add_epi32(m3, tmp); m3 = _mm_sha256msg2_epu32(m3, m2);
    msg = _mm_shuffle_epi32(msg, 0x0E); st0 = _mm_sha256rnds2_epu32(st0, st1, msg);
    HS_RND(m3, 15);
#undef HS_EXT
#undef HS_RND
#undef HS_K
    st0 = _mm_add_epi32(st0, abef);
    st1 = _mm_add_epi32(st1, cdgh);
    tmp = _mm_shuffle_epi32(st0, 0x1B);                  // FEBA
    st1 = _mm_shuffle_epi32(st1, 0xB1);                  // DCHG
    st0 = _mm_blend_epi16(tmp, st1, 0xF0);               // DCBA
    st1 = _mm_alignr_epi8(st1, tmp, 8);                  // ABEF -> HGFE
    _mm_storeu_si128((__m128i*)&s[0], st0);
    _mm_storeu_si128((__m128i*)&s[4], st1);
}
static inline int host_sha256_have_ni() { static const int have = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3"); return have; }
#else
static inline int host_sha256_have_ni() { return 0; }
#endif
static inline void host_sha256_compress(uint32_t s[8], const unsigned char* block64) {
#if defined(__x86_64__)
    if (host_sha256_have_ni()) { host_sha256_block_ni(s, block64); return; }
#endif
    host_sha256_block(s, block64);
}
