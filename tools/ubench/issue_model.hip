// Micro-benchmark: VALU issue model of gfx950 for the instruction streams of 256-bit modular arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../secp256k1_zkp_amd/csrc -o issue_model issue_model.hip && ./issue_model
// For k = 1..8 resident waves per SIMD (enforced with a dynamic-LDS request of 160 KiB / k per 256-lane workgroup,
// so that exactly k workgroups = k waves per SIMD are co-resident on every CU) it times, with >= 10 ms launches:
//   * v_mad_u64_u32 in 8 / 4 / 2 / 1 dependent chains (1 chain = every instruction waits on its predecessor, with the
//     s_nop 0 the assembler-level hazard rule wants between a v_mad_u64_u32 and a consumer of its result),
//   * full-rate v_add_u32, half-rate v_lshrrev_b64, and the 3 MAC : 1 AND mix of a product-scanning column,
//   * the product's own building blocks from csrc/fe.h / group.h (fe_mul, fe_sqr, lockstep pair, gej_double, gej_add_ge).
// Output: cycles per wave-instruction per SIMD at the nominal 2.4 GHz and chip-wide lane-ops/s; the largest
// v_mad_u64_u32 rate is the integer-MAC peak used as bench.py's VALU roofline denominator.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "group.h"

#define CHECK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){fprintf(stderr,"HIP error %s at %d\n",hipGetErrorString(e_),__LINE__); exit(1);} }while(0)

extern __shared__ unsigned char dyn_lds[];

#define KB(name) __global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed, int iters) { \
  uint32_t x0=threadIdx.x+seed, x1=x0*3+1, x2=x0*5+2, x3=x0*7+3; \
  uint64_t a0=x0,a1=x1,a2=x2,a3=x3,a4=x0+9,a5=x1+9,a6=x2+9,a7=x3+9; \
  uint32_t b0=x0,b1=x1,b2=x2,b3=x3,b4=x0+9,b5=x1+9,b6=x2+9,b7=x3+9; \
  if (seed == 0xdeadbeef) dyn_lds[threadIdx.x] = 1; \
  for (int it=0; it<iters; ++it) {
#define KE } \
  uint64_t s=a0^a1^a2^a3^a4^a5^a6^a7; uint32_t t=b0^b1^b2^b3^b4^b5^b6^b7; \
  if ((uint32_t)s + t == 0x12345) out[threadIdx.x]=1; }

KB(k_mac8)
  asm volatile(
   "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n"
   "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
   "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"
   "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x0),"v"(x1) : "vcc");
KE
KB(k_mac4)
  asm volatile(
   "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
   "v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
   "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
   "v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3) : "v"(x0),"v"(x1) : "vcc");
KE
KB(k_mac2)
  asm volatile(
   "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n"
   "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n"
   "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n"
   "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n"
   : "+v"(a0),"+v"(a1) : "v"(x0),"v"(x1) : "vcc");
KE
// one chain: every MAC consumes its predecessor; "s_nop 0" between them as the compiler emits it (counted as 8 MACs)
KB(k_mac1_nop)
  asm volatile(
   "v_mad_u64_u32 %0, vcc, %1, %2, %0\n s_nop 0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n s_nop 0\n"
   "v_mad_u64_u32 %0, vcc, %1, %2, %0\n s_nop 0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n s_nop 0\n"
   "v_mad_u64_u32 %0, vcc, %1, %2, %0\n s_nop 0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n s_nop 0\n"
   "v_mad_u64_u32 %0, vcc, %1, %2, %0\n s_nop 0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n s_nop 0\n"
   : "+v"(a0) : "v"(x0),"v"(x1) : "vcc");
KE
// one chain, the wait state filled with a full-rate instruction instead of s_nop (counted as 8 MACs)
KB(k_mac1_fill)
  asm volatile(
   "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_and_b32 %1, %2, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_and_b32 %1, %2, %1\n"
   "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_and_b32 %1, %2, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_and_b32 %1, %2, %1\n"
   "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_and_b32 %1, %2, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_and_b32 %1, %2, %1\n"
   "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_and_b32 %1, %2, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_and_b32 %1, %2, %1\n"
   : "+v"(a0),"+v"(b0) : "v"(x0),"v"(x1) : "vcc");
KE

// data dependence of the MAC rate (board power management): the same 8-chain stream with all-zero operands and with full-width
// pseudo-random operands (k_mac8 above multiplies ~10-bit values)
#define KBV(name, X0, X1) __global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed, int iters) { \
  uint32_t x0=(X0), x1=(X1); \
  uint64_t a0=x0,a1=x1,a2=x0^x1,a3=x0+x1,a4=(uint64_t)x0*x1,a5=a4^x0,a6=a4+x1,a7=a4*3; \
  if (seed == 0xdeadbeef) dyn_lds[threadIdx.x] = 1; \
  for (int it=0; it<iters; ++it) { \
  asm volatile( \
   "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n" \
   "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n" \
   "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n" \
   "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n" \
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x0),"v"(x1) : "vcc"); \
  } \
  uint64_t s=a0^a1^a2^a3^a4^a5^a6^a7; if ((uint32_t)s == 0x12345) out[threadIdx.x]=1; }
KBV(k_mac8_zero, seed >> 31, seed >> 30)
KBV(k_mac8_rand, (threadIdx.x + seed) * 2654435761u ^ 0x9e3779b9u, (threadIdx.x * 40503u + seed) * 2246822519u ^ 0x85ebca6bu)
KBV(k_mac8_29bit, ((threadIdx.x + seed) * 2654435761u ^ 0x9e3779b9u) & 0x1fffffffu, ((threadIdx.x * 40503u + seed) * 2246822519u ^ 0x85ebca6bu) & 0x1fffffffu)
KBV(k_mac8_x256, ((threadIdx.x + seed) * 2654435761u ^ 0x9e3779b9u) & 0x1fffffffu, 256u + (seed >> 31))

KB(k_add8)
  asm volatile(
   "v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n"
   "v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n"
   : "+v"(b0),"+v"(b1),"+v"(b2),"+v"(b3),"+v"(b4),"+v"(b5),"+v"(b6),"+v"(b7) : "v"(x0));
KE
KB(k_shr64_8)
  asm volatile(
   "v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3\n"
   "v_lshrrev_b64 %4, 1, %4\n v_lshrrev_b64 %5, 1, %5\n v_lshrrev_b64 %6, 1, %6\n v_lshrrev_b64 %7, 1, %7\n"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7));
KE
// the column step of a product scan: 6 MACs on two alternating chains, one v_and, one 64-bit shift (8 instructions)
KB(k_mix_col)
  asm volatile(
   "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
   "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
   "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
   "v_and_b32 %2, 0x1fffffff, %2\n v_lshrrev_b64 %3, 29, %3\n"
   : "+v"(a0),"+v"(a1),"+v"(b0),"+v"(a2) : "v"(x0),"v"(x1) : "vcc");
KE

// ---- the real building blocks ------------------------------------------------------------------------------
__device__ __forceinline__ void fe_seed(fe& r, uint32_t s) {
    for (int i = 0; i < 9; i++) r.n[i] = (s * (2654435761u + 2 * i) + i) & (i == 8 ? FE_TOPM : FE_M);
}
#define FB(name) __global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed, int iters) { \
  if (seed == 0xdeadbeef) dyn_lds[threadIdx.x] = 1;
#define FE_END(v) { uint32_t s = 0; for (int i = 0; i < 9; i++) s ^= (v).n[i]; if (s == 0x12345) out[threadIdx.x] = s; } }

FB(k_fe_mul) fe x, y; fe_seed(x, threadIdx.x + seed); fe_seed(y, threadIdx.x * 7 + 3);
  for (int it = 0; it < iters; ++it) fe_mul(x, x, y);
FE_END(x)
FB(k_fe_sqr) fe x; fe_seed(x, threadIdx.x + seed);
  for (int it = 0; it < iters; ++it) fe_sqr(x, x);
FE_END(x)
FB(k_fe_mul2) fe x, y, z; fe_seed(x, threadIdx.x + seed); fe_seed(y, threadIdx.x * 7 + 3); fe_seed(z, threadIdx.x * 11 + 5);
  for (int it = 0; it < iters; ++it) fe_mul2(x, x, y, z, z, y);          // counted as 2 products
  fe_add(x, z);
FE_END(x)
FB(k_fe_sqr2) fe x, z; fe_seed(x, threadIdx.x + seed); fe_seed(z, threadIdx.x * 11 + 5);
  for (int it = 0; it < iters; ++it) fe_sqr2(x, x, z, z);
  fe_add(x, z);
FE_END(x)
FB(k_gej_double) gej p; fe_seed(p.x, threadIdx.x + seed); fe_seed(p.y, threadIdx.x * 7 + 3); fe_seed(p.z, threadIdx.x * 11 + 5); p.inf = 0;
  for (int it = 0; it < iters; ++it) { gej t; gej_double(t, p); p = t; }
  fe_norm_weak(p.x); fe_norm_weak(p.y); fe_add(p.x, p.y); fe_add(p.x, p.z);
FE_END(p.x)
FB(k_gej_add_ge) gej p; ge q; fe_seed(p.x, threadIdx.x + seed); fe_seed(p.y, threadIdx.x * 7 + 3); fe_seed(p.z, threadIdx.x * 11 + 5); p.inf = 0;
  fe_seed(q.x, threadIdx.x * 13 + 1); fe_seed(q.y, threadIdx.x * 17 + 2);
  for (int it = 0; it < iters; ++it) { gej t; gej_add_ge(t, p, q); p = t; }
  fe_norm_weak(p.x); fe_norm_weak(p.y); fe_add(p.x, p.y); fe_add(p.x, p.z);
FE_END(p.x)

FB(k_fe_muladd) fe x, y, z; fe_seed(x, threadIdx.x + seed); fe_seed(y, threadIdx.x * 7 + 3); fe_seed(z, threadIdx.x * 11 + 5);
  for (int it = 0; it < iters; ++it) fe_muladd<false, true>(x, x, y, z, z);           // one fused pair (product + square, one reduction)
FE_END(x)
FB(k_gej_double_lean) gej p; fe_seed(p.x, threadIdx.x + seed); fe_seed(p.y, threadIdx.x * 7 + 3); fe_seed(p.z, threadIdx.x * 11 + 5); p.inf = 0;
  for (int it = 0; it < iters; ++it) gej_double_lean(p, p);
  fe_add(p.x, p.y); fe_add(p.x, p.z);
FE_END(p.x)
FB(k_gej_add_ge_lean) gej p; ge q; fe_seed(p.x, threadIdx.x + seed); fe_seed(p.y, threadIdx.x * 7 + 3); fe_seed(p.z, threadIdx.x * 11 + 5); p.inf = 0;
  fe_seed(q.x, threadIdx.x * 13 + 1); fe_seed(q.y, threadIdx.x * 17 + 2); int acc = 0;
  for (int it = 0; it < iters; ++it) acc += gej_add_ge_lean(p, p, q);
  fe_add(p.x, p.y); fe_add(p.x, p.z); p.x.n[0] += acc;
FE_END(p.x)


// fe_mul with the fold's second multiply-accumulate (256 * u_{k-1}) treated three ways: MODE 0 as shipped, 1 dropped (what a free
// fold term would buy; wrong results), 2 replaced by one v_lshl_add_u64 (what a shift-and-add fold of a mixed-radix layout would cost)
template <int MODE>
__device__ __forceinline__ void fe_mul_variant(fe& r, const fe& a_in, const fe& b_in) {
    u32 a[FE_LIMBS], b[FE_LIMBS];
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) { a[i] = a_in.n[i]; b[i] = b_in.n[i]; }
    S2K_OPAQUE(a[8]); S2K_OPAQUE(b[8]);
    u32 k256 = 256u; S2K_OPAQUE(k256);
    u64 c = 0, d = 0; u32 u = 0; u64 uprev = 0;
#pragma unroll
    for (int k = 0; k < FE_LIMBS; k++) {
#pragma unroll
        for (int t = 0; t < FE_LIMBS; t++) {
            if (k < 8 && k + 1 + t < FE_LIMBS) { const int i = k + 1 + t, j = 9 + k - i; d += (u64)a[i] * b[j]; S2K_CHAIN(d); }
            if (t <= k) { const int i = t, j = k - t; c += (u64)a[i] * b[j]; S2K_CHAIN(c); }
        }
        if (k < 8) { u = (u32)d & FE_M; d >>= FE_BITS; } else u = (u32)d;
        c += (u64)u * 31264u; S2K_CHAIN(c);
        if (k > 0) {
            if (MODE == 0) { c += (u64)(u32)uprev * k256; S2K_CHAIN(c); }
            if (MODE == 2) { asm volatile("v_lshl_add_u64 %0, %1, 4, %0" : "+v"(c) : "v"(uprev)); }
        }
        uprev = u;
        r.n[k] = (u32)c & FE_M; c >>= FE_BITS;
    }
    fe_mul_tail(r, c, u);
}
FB(k_fe_mul_v0) fe x, y; fe_seed(x, threadIdx.x + seed); fe_seed(y, threadIdx.x * 7 + 3);
  for (int it = 0; it < iters; ++it) fe_mul_variant<0>(x, x, y);
FE_END(x)
FB(k_fe_mul_v1) fe x, y; fe_seed(x, threadIdx.x + seed); fe_seed(y, threadIdx.x * 7 + 3);
  for (int it = 0; it < iters; ++it) fe_mul_variant<1>(x, x, y);
FE_END(x)
FB(k_fe_mul_v2) fe x, y; fe_seed(x, threadIdx.x + seed); fe_seed(y, threadIdx.x * 7 + 3);
  for (int it = 0; it < iters; ++it) fe_mul_variant<2>(x, x, y);
FE_END(x)

typedef void (*kern_t)(uint32_t*, uint32_t, int);
struct entry { const char* name; kern_t k; double ops_per_iter; double approx_cyc_per_iter; };

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3;
    printf("device %s CUs=%d nominal clock=%.0f MHz\n", prop.gcnArchName, ncu, clk / 1e6);
    uint32_t* out; CHECK(hipMalloc(&out, 8192));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    entry es[] = {
        {"v_mad_u64_u32 x8 chains", k_mac8, 8, 36}, {"v_mad_u64_u32 x4 chains", k_mac4, 8, 36}, {"v_mad_u64_u32 x2 chains", k_mac2, 8, 40},
        {"v_mad_u64_u32 1 chain + s_nop 0", k_mac1_nop, 8, 60}, {"v_mad_u64_u32 1 chain + v_and fill", k_mac1_fill, 8, 60},
        {"mac8 zero operands", k_mac8_zero, 8, 36}, {"mac8 random 32-bit operands", k_mac8_rand, 8, 36}, {"mac8 random 29-bit operands", k_mac8_29bit, 8, 36}, {"mac8 29-bit x 256", k_mac8_x256, 8, 36},
        {"v_add_u32 x8", k_add8, 8, 20}, {"v_lshrrev_b64 x8", k_shr64_8, 8, 36}, {"column mix 6 mac+and+shr64", k_mix_col, 8, 36},
        {"fe_mul (fe.h)", k_fe_mul, 1, 800}, {"fe_sqr (fe.h)", k_fe_sqr, 1, 600}, {"fe_mul2 lockstep (per product)", k_fe_mul2, 2, 1500},
        {"fe_sqr2 lockstep (per product)", k_fe_sqr2, 2, 1100}, {"gej_double", k_gej_double, 1, 5000}, {"gej_add_ge", k_gej_add_ge, 1, 8000},
        {"variant0 fe_mul as shipped", k_fe_mul_v0, 1, 800}, {"variant1 fe_mul without the x256 MACs", k_fe_mul_v1, 1, 800}, {"variant2 fe_mul x256 as v_lshl_add_u64", k_fe_mul_v2, 1, 800},
        {"fe_muladd (product + square, one reduction)", k_fe_muladd, 1, 1100}, {"gej_double_lean", k_gej_double_lean, 1, 4500}, {"gej_add_ge_lean", k_gej_add_ge_lean, 1, 7500}};
    const char* only = (argc > 2) ? argv[2] : nullptr;
    const double target_ms = (argc > 1) ? atof(argv[1]) : 12.0;
    for (auto& e : es) {
        if (only && !strstr(e.name, only)) continue;
        hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, (const void*)e.k));
        CHECK(hipFuncSetAttribute((const void*)e.k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        printf("== %-36s VGPRs=%d\n", e.name, fa.numRegs);
        for (int k : {1, 2, 4, 8}) {
            const int alloc = ((fa.numRegs + 7) / 8) * 8;
            const int kmax = alloc ? 512 / alloc : 8;
            if (k > kmax) continue;
            const size_t lds = (size_t)(160 * 1024 / k) & ~(size_t)255;
            const int rounds = 2, blocks = ncu * k * rounds;
            // iters so that one launch lasts ~target_ms: rounds * k * iters * cyc_per_iter / clk
            int iters = (int)(target_ms * 1e-3 * clk / (rounds * k * e.approx_cyc_per_iter));
            if (iters < 16) iters = 16;
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), lds, 0, out, 1u, iters / 8 + 1); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), lds, 0, out, 2u, iters);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double wave_ops_per_simd = (double)rounds * k * iters * e.ops_per_iter;     // every SIMD runs rounds*k waves in sequence/parallel
            const double cyc = ms * 1e-3 * clk / wave_ops_per_simd;
            const double lane_ops = (double)blocks * 256 * iters * e.ops_per_iter / (ms * 1e-3);
            printf("   waves/SIMD=%d  %8.3f ms  %8.2f cyc/wave-op/SIMD  %.4e lane-ops/s\n", k, ms, cyc, lane_ops);
        }
    }
    return 0;
}
