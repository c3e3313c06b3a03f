/* Plain C caller of the engine's C ABI: verify one Borromean rangeproof with the reference's own argument list
 * (secp256k1_rangeproof_verify, include/secp256k1_rangeproof.h:70-80) and a small batch through the batch entry point.
 *
 *   gcc -std=c99 -Iinclude examples/rangeproof_verify.c -o rp_verify secp256k1_zkp_amd/libsecp256k1_zkp_amd.so -Wl,-rpath,$PWD/secp256k1_zkp_amd
 *   ./rp_verify commit33.bin proof.bin generator64.bin
 * prints "<result> <min_value> <max_value>" for the single call and for a batch holding the proof and a corrupted copy. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "secp256k1_zkp_amd.h"

static size_t slurp(const char *path, unsigned char *buf, size_t cap) {
    FILE *f = fopen(path, "rb"); size_t n;
    if (!f) { perror(path); exit(2); }
    n = fread(buf, 1, cap, f); fclose(f);
    return n;
}

int main(int argc, char **argv) {
    unsigned char commit[64], proof[2 * 5200], gen[2 * 64], commits[2 * 33];
    uint64_t mn = 0, mx = 0, mins[2], maxs[2], off[3];
    int32_t res[2];
    size_t plen;
    int r;
    s2k_engine *e;
    if (argc != 4) { fprintf(stderr, "usage: %s commit33 proof generator64\n", argv[0]); return 2; }
    memset(commit, 0, sizeof(commit));
    if (slurp(argv[1], commit, 33) != 33) return 2;              /* the first 33 bytes of a secp256k1_pedersen_commitment object */
    plen = slurp(argv[2], proof, 5200);
    if (slurp(argv[3], gen, 64) != 64) return 2;

    r = secp256k1_rangeproof_verify_amd(NULL, &mn, &mx, commit, proof, plen, NULL, 0, gen);
    if (!r && s2k_last_error()[0]) fprintf(stderr, "engine: %s\n", s2k_last_error());
    printf("%d %llu %llu\n", r, (unsigned long long)mn, (unsigned long long)mx);

    e = s2k_engine_create(0);
    if (!e) { fprintf(stderr, "engine: %s\n", s2k_last_error()); return 1; }
    memcpy(proof + plen, proof, plen); proof[plen + plen / 2] ^= 1;      /* item 1: one flipped bit */
    memcpy(commits, commit, 33); memcpy(commits + 33, commit, 33);
    memcpy(gen + 64, gen, 64);
    off[0] = 0; off[1] = plen; off[2] = 2 * plen;
    if (!secp256k1_rangeproof_verify_batch(e, res, mins, maxs, commits, proof, off, NULL, NULL, gen, 2)) { fprintf(stderr, "engine: %s\n", s2k_last_error()); return 1; }
    printf("%d %llu %llu\n%d\n", (int)res[0], (unsigned long long)mins[0], (unsigned long long)maxs[0], (int)res[1]);
    s2k_engine_destroy(e);
    return 0;
}
