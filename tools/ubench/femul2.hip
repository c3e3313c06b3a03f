// Micro-benchmark: the production fe_mul / fe_sqr (csrc/fe.h) against inline-asm accumulate-chain variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o femul2 femul2.hip && ./femul2
#include "../../secp256k1_zkp_amd/csrc/fe.h"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
#define ITERS 512
#define MAD(c, x, y) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(x), "v"(y) : "vcc")

__device__ __forceinline__ void fe_mul_asm1(fe& r, const fe& A, const fe& B) {     // one asm statement per multiply-accumulate
    u32 a[9], b[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { a[i] = A.n[i]; b[i] = B.n[i]; }
    u32 k256 = 256u, k31264 = 31264u;
    u64 c = 0, d = 0; u32 u = 0, uprev = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        if (k < 8) {
#pragma unroll
            for (int i = 0; i < 9; i++) { const int j = 9 + k - i; if (j < 0 || j >= 9) continue; MAD(d, a[i], b[j]); }
            u = (u32)d & FE_M; d >>= 29;
        } else u = (u32)d;
#pragma unroll
        for (int i = 0; i < 9; i++) { const int j = k - i; if (j < 0 || j >= 9) continue; MAD(c, a[i], b[j]); }
        MAD(c, u, k31264);
        if (k > 0) MAD(c, uprev, k256);
        uprev = u;
        r.n[k] = (u32)c & FE_M; c >>= 29;
    }
    fe_mul_tail(r, c, u);
}
// one asm statement per column: products of column k into acc (operands listed explicitly)
#define M1(c,x0,y0) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(x0),"v"(y0) : "vcc")
#define M2(c,x0,y0,x1,y1) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %3, %4, %0" : "+v"(c) : "v"(x0),"v"(y0),"v"(x1),"v"(y1) : "vcc")
#define M3(c,x0,y0,x1,y1,x2,y2) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_mad_u64_u32 %0, vcc, %5, %6, %0" : "+v"(c) : "v"(x0),"v"(y0),"v"(x1),"v"(y1),"v"(x2),"v"(y2) : "vcc")
template <int K> __device__ __forceinline__ void col(u64& acc, const u32* a, const u32* b) {     // column K of a 9x9 product, 3 products per asm
    constexpr int lo = K < 9 ? 0 : K - 8, hi = K < 9 ? K : 8;      // i in [lo, hi]
    constexpr int n = hi - lo + 1;
    if constexpr (n >= 1 && n < 2) M1(acc, a[lo], b[K - lo]);
    if constexpr (n == 2) M2(acc, a[lo], b[K - lo], a[lo + 1], b[K - lo - 1]);
    if constexpr (n >= 3) {
        M3(acc, a[lo], b[K - lo], a[lo + 1], b[K - lo - 1], a[lo + 2], b[K - lo - 2]);
        if constexpr (n == 4) M1(acc, a[lo + 3], b[K - lo - 3]);
        if constexpr (n == 5) M2(acc, a[lo + 3], b[K - lo - 3], a[lo + 4], b[K - lo - 4]);
        if constexpr (n >= 6) {
            M3(acc, a[lo + 3], b[K - lo - 3], a[lo + 4], b[K - lo - 4], a[lo + 5], b[K - lo - 5]);
            if constexpr (n == 7) M1(acc, a[lo + 6], b[K - lo - 6]);
            if constexpr (n == 8) M2(acc, a[lo + 6], b[K - lo - 6], a[lo + 7], b[K - lo - 7]);
            if constexpr (n == 9) M3(acc, a[lo + 6], b[K - lo - 6], a[lo + 7], b[K - lo - 7], a[lo + 8], b[K - lo - 8]);
        }
    }
}
template <int K> __device__ __forceinline__ void mul_step(fe& r, u64& c, u64& d, u32& u, u32& uprev, const u32* a, const u32* b, u32 k256, u32 k31264) {
    if constexpr (K < 8) { col<9 + K>(d, a, b); u = (u32)d & FE_M; d >>= 29; } else u = (u32)d;
    col<K>(c, a, b);
    if constexpr (K > 0) M2(c, u, k31264, uprev, k256); else M1(c, u, k31264);
    uprev = u;
    r.n[K] = (u32)c & FE_M; c >>= 29;
}
__device__ __forceinline__ void fe_mul_asm3(fe& r, const fe& A, const fe& B) {
    u32 a[9], b[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { a[i] = A.n[i]; b[i] = B.n[i]; }
    u32 k256 = 256u, k31264 = 31264u;
    u64 c = 0, d = 0; u32 u = 0, uprev = 0;
    mul_step<0>(r, c, d, u, uprev, a, b, k256, k31264); mul_step<1>(r, c, d, u, uprev, a, b, k256, k31264); mul_step<2>(r, c, d, u, uprev, a, b, k256, k31264);
    mul_step<3>(r, c, d, u, uprev, a, b, k256, k31264); mul_step<4>(r, c, d, u, uprev, a, b, k256, k31264); mul_step<5>(r, c, d, u, uprev, a, b, k256, k31264);
    mul_step<6>(r, c, d, u, uprev, a, b, k256, k31264); mul_step<7>(r, c, d, u, uprev, a, b, k256, k31264); mul_step<8>(r, c, d, u, uprev, a, b, k256, k31264);
    fe_mul_tail(r, c, u);
}

template <int V> __global__ void __launch_bounds__(256) k_mul(u32* out, u32 seed) {
    fe x, y;
    for (int i = 0; i < 9; i++) { x.n[i] = (threadIdx.x * 77 + seed + i * 1234567u) & FE_M; y.n[i] = (0x1234567u * (threadIdx.x + i + 1)) & FE_M; }
    x.n[8] &= FE_TOPM; y.n[8] &= FE_TOPM;
    for (int it = 0; it < ITERS; it++) {
        if (V == 0) fe_mul(x, x, y); else if (V == 1) fe_mul_asm1(x, x, y); else if (V == 3) fe_mul_asm3(x, x, y); else fe_sqr(x, x);
    }
    u32 s = 0; for (int i = 0; i < 9; i++) s ^= x.n[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
    u32* out; CHECK(hipMalloc(&out, 2048 * 256 * 4));
    u32* h0 = (u32*)malloc(2048 * 256 * 4); u32* h1 = (u32*)malloc(2048 * 256 * 4);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct E { const char* name; void (*k)(u32*, u32); } es[] = {{"fe_mul (compiler)", k_mul<0>}, {"fe_mul asm per-mad", k_mul<1>}, {"fe_mul asm per-3", k_mul<3>}, {"fe_sqr (compiler)", k_mul<2>}};
    for (int wpc : {8, 16, 32}) {
        int blocks = ncu * wpc / 4; printf("--- waves/CU=%d\n", wpc);
        for (auto& e : es) {
            e.k<<<blocks, 256>>>(out, 1); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0)); for (int r = 0; r < 3; r++) e.k<<<blocks, 256>>>(out, 1); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
            double cyc = (ms * 1e-3 * clk) / ((double)blocks * 4 * ITERS / (ncu * 4.0));
            printf("%-22s %8.3f ms  %.3e fe-ops/s  %.0f cyc/wave-op/SIMD\n", e.name, ms, (double)blocks * 256 * ITERS / (ms * 1e-3), cyc);
        }
    }
    // correctness of the asm variants against the compiler version (same inputs)
    k_mul<0><<<64, 256>>>(out, 7); CHECK(hipMemcpy(h0, out, 64 * 256 * 4, hipMemcpyDeviceToHost));
    for (int v = 0; v < 2; v++) {
        if (v == 0) k_mul<1><<<64, 256>>>(out, 7); else k_mul<3><<<64, 256>>>(out, 7);
        CHECK(hipMemcpy(h1, out, 64 * 256 * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 64 * 256; i++) bad += h0[i] != h1[i];
        printf("asm variant %d mismatches vs compiler version: %d\n", v, bad);
    }
    return 0;
}
