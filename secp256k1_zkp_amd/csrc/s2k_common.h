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

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "secp256k1_zkp_amd is written for gfx950 (MI355X) only: v_bitop3_b32, the 9x29 limb schedule and the wave-cooperative DPP code assume it; build with --offload-arch=gfx950"
#endif
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

// S2K_CHAIN(acc): keeps a 64-bit accumulator a single linear chain of v_mad_u64_u32 (product + running sum in one
// instruction) and pins the order in which fe_mul / fe_sqr interleave their two chains.  Without it LLVM's reassociation
// splits every column into sub-chains and joins them with v_lshl_add_u64 (~29 extra 64-bit adds per field multiplication
// inside the big kernels).  A dependent v_mad_u64_u32 pair needs one wait state (s_nop 0), which is why the two chains are
// issued alternately.  Measured on MI355X: rangeproofs +2.4 %, MSM +3 %, bare double multiplications +4 %, BP++ +5 %.
// -DS2K_NO_LINEAR_CHAINS restores the compiler's own schedule.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(S2K_NO_LINEAR_CHAINS)
#define S2K_CHAIN(x) asm volatile("" : "+v"(x))
#else
#define S2K_CHAIN(x) ((void)0)
#endif

// S2K_OPAQUE(x): make the value of a 32-bit register opaque to the optimiser (no instruction is emitted).
// Needed around every 32x32->64 product: ROCm 7.2's AMDGPU backend, when it can prove both operands of a 64-bit
// product fit in 24 bits (e.g. the top limb after `& 0xFFFFFF`), first narrows the product to mul24 -- which lets
// it DROP the mask as "bits not demanded" -- and then re-widens it to a full v_mad_u64_u32 on the unmasked
// register.  Found on MI355X as wrong top/bottom limbs in chained fe_sqr; see tests/test_gpu_prims.py::test_chained.
#if defined(__HIP_DEVICE_COMPILE__)
#define S2K_OPAQUE(x) asm("" : "+v"(x))
#else
#define S2K_OPAQUE(x) ((void)0)
#endif

typedef uint32_t u32;
typedef uint64_t u64;

// S2K_PROF (diagnostic builds only, -DS2K_PROF): per-region shader-clock attribution.  S2K_PROF_MARK(i) adds the cycles since
// the previous mark of this wave to slot i of a device-global table (one atomic per wave); tools/prof_regions.py reads it.
#if defined(S2K_PROF) && (defined(__HIPCC__) || defined(__HIP__))
__device__ unsigned long long s2k_prof_slots[16];
#endif
#if defined(S2K_PROF) && defined(__HIP_DEVICE_COMPILE__)
struct s2k_prof_clock { unsigned long long t; };
#define S2K_PROF_DECL s2k_prof_clock _pc; _pc.t = __builtin_readcyclecounter()
#define S2K_PROF_RESET _pc.t = __builtin_readcyclecounter()
#define S2K_PROF_MARK(i) do { const unsigned long long _n = __builtin_readcyclecounter(); \
        if ((threadIdx.x & 63) == 0) atomicAdd(&s2k_prof_slots[i], _n - _pc.t); _pc.t = __builtin_readcyclecounter(); } while (0)
#else
#define S2K_PROF_DECL
#define S2K_PROF_RESET
#define S2K_PROF_MARK(i) do { } while (0)
#endif

S2K_HD u32 s2k_load_be32(const unsigned char* p) {
    return ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3];
}
S2K_HD void s2k_store_be32(unsigned char* p, u32 v) {
    p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v;
}
