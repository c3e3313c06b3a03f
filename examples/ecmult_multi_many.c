/* Plain C caller of the engine's C ABI: K independent multi-scalar multiplications in ONE call (s2k_ecmult_multi_many), each compared with the
 * single call (s2k_ecmult_multi) on the same terms -- what a caller of the reference gets from K calls of secp256k1_ecmult_multi_var
 * (src/ecmult.h:62; bench_ecmult's sums, src/bench_ecmult.c:262-276).
 *
 *   gcc -std=c99 -Iinclude examples/ecmult_multi_many.c -o mm secp256k1_zkp_amd/libsecp256k1_zkp_amd.so -Wl,-rpath,$PWD/secp256k1_zkp_amd
 *   ./mm [K [terms per sum]]          prints "OK K sums of n terms" or the first difference */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "secp256k1_zkp_amd.h"

static const unsigned char G_XY[64] = {
    0x79, 0xBE, 0x66, 0x7E, 0xF9, 0xDC, 0xBB, 0xAC, 0x55, 0xA0, 0x62, 0x95, 0xCE, 0x87, 0x0B, 0x07, 0x02, 0x9B, 0xFC, 0xDB, 0x2D, 0xCE, 0x28, 0xD9, 0x59, 0xF2, 0x81, 0x5B, 0x16, 0xF8, 0x17, 0x98,
    0x48, 0x3A, 0xDA, 0x77, 0x26, 0xA3, 0xC4, 0x65, 0x5D, 0xA4, 0xFB, 0xFC, 0x0E, 0x11, 0x08, 0xA8, 0xFD, 0x17, 0xB4, 0x48, 0xA6, 0x85, 0x54, 0x19, 0x9C, 0x47, 0xD0, 0x8F, 0xFB, 0x10, 0xD4, 0xB8};

static unsigned int lcg = 12345u;
static void fill(unsigned char *p, size_t n) { size_t i; for (i = 0; i < n; i++) { lcg = lcg * 1664525u + 1013904223u; p[i] = (unsigned char)(lcg >> 24); } }

int main(int argc, char **argv) {
    const size_t K = argc > 1 ? (size_t)atol(argv[1]) : 8, n = argc > 2 ? (size_t)atol(argv[2]) : 200, N = K * n;
    unsigned char *sc = malloc(32 * N + 32), *pts = malloc(64 * N + 64), *ks = malloc(32 * N + 32), *gs = malloc(32 * K), *gpts = malloc(64 * N + 64), *zero = calloc(32 * N + 32, 1);
    unsigned char *r_many = malloc(64 * K), r_one[64];
    int32_t *inf_many = malloc(sizeof(int32_t) * K), *pinf = malloc(sizeof(int32_t) * (N + 1)), inf_one;
    uint64_t *off = malloc(sizeof(uint64_t) * (K + 1));
    size_t i, s;
    s2k_engine *e = s2k_engine_create(0);
    if (!e) { fprintf(stderr, "engine: %s\n", s2k_last_error()); return 1; }
    /* points: k_i * G through the batched double multiplication (na = 0, ng = k_i) */
    for (i = 0; i < N; i++) memcpy(gpts + 64 * i, G_XY, 64);
    fill(ks, 32 * N); fill(sc, 32 * N); fill(gs, 32 * K);
    if (!s2k_ecmult_batch(e, pts, pinf, gpts, NULL, zero, ks, N)) { fprintf(stderr, "engine: %s\n", s2k_last_error()); return 1; }
    for (s = 0; s <= K; s++) off[s] = (uint64_t)(s * n);
    if (K > 2) off[1] = off[2];                      /* one empty sum and one of double length */
    if (!s2k_ecmult_multi_many(e, r_many, inf_many, gs, sc, pts, NULL, off, K)) { fprintf(stderr, "engine: %s\n", s2k_last_error()); return 1; }
    for (s = 0; s < K; s++) {
        const size_t lo = (size_t)off[s], cnt = (size_t)(off[s + 1] - off[s]);
        if (!s2k_ecmult_multi(e, r_one, &inf_one, gs + 32 * s, sc + 32 * lo, pts + 64 * lo, NULL, cnt)) { fprintf(stderr, "engine: %s\n", s2k_last_error()); return 1; }
        if (inf_one != inf_many[s] || memcmp(r_one, r_many + 64 * s, 64)) { printf("sum %lu differs\n", (unsigned long)s); return 1; }
    }
    printf("OK %lu sums of %lu terms\n", (unsigned long)K, (unsigned long)n);
    s2k_engine_destroy(e);
    return 0;
}
