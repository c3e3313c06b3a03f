// s2k_common.h -- shared qualifiers / small helpers for the gfx950 secp256k1 engine.
//
// All arithmetic headers in this directory are written for the CDNA4 vector ALU
// (one curve point per lane, everything in VGPRs).  They are plain `__device__`
// code; the `__host__` half of S2K_HD exists only so that tests/host_emul can run the
// *same* per-lane arithmetic on the CPU of a GPU-less CI container and compare it with
// the oracle.  The shipped library never executes these functions on the host.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define S2K_HD __host__ __device__ __forceinline__
#define S2K_HD_NOINLINE __host__ __device__ __noinline__
#define S2K_D __device__ __forceinline__
#else
#define S2K_HD static inline
#define S2K_HD_NOINLINE static
#define S2K_D static inline
#endif

typedef uint32_t u32;
typedef uint64_t u64;

S2K_HD u32 s2k_load_be32(const unsigned char* p) {
    return ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3];
}
S2K_HD void s2k_store_be32(unsigned char* p, u32 v) {
    p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v;
}
