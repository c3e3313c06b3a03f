// Micro-benchmark: issue rate of the integer / fp64 VALU instructions that bound
// secp256k1 field arithmetic on gfx950 (MI355X).  Standalone (no torch):
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
// Prints ops/s chip-wide and cycles per wave-instruction per SIMD (at the measured clock).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

#define ITERS 2048
#define UNROLL 8   // independent chains

#define KERNEL_BEGIN(name) \
__global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) { \
  uint32_t x0=threadIdx.x+seed, x1=x0*3+1, x2=x0*5+2, x3=x0*7+3; \
  uint64_t a0=x0,a1=x1,a2=x2,a3=x3,a4=x0+9,a5=x1+9,a6=x2+9,a7=x3+9; \
  uint32_t b0=x0,b1=x1,b2=x2,b3=x3,b4=x0+9,b5=x1+9,b6=x2+9,b7=x3+9; \
  double d0=x0,d1=x1,d2=x2,d3=x3,d4=x0+9,d5=x1+9,d6=x2+9,d7=x3+9; \
  for (int it=0; it<ITERS; ++it) {
#define KERNEL_END \
  } \
  uint64_t s=a0^a1^a2^a3^a4^a5^a6^a7; uint32_t t=b0^b1^b2^b3^b4^b5^b6^b7; double ds=d0+d1+d2+d3+d4+d5+d6+d7; \
  if ((uint32_t)s + t + (uint32_t)ds == 0x12345) out[threadIdx.x]=1; }

KERNEL_BEGIN(k_mad_u64_u32)
  asm volatile(
   "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n"
   "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
   "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"
   "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x0),"v"(x1) : "vcc");
KERNEL_END

KERNEL_BEGIN(k_mul_lo_u32)
  asm volatile(
   "v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %1, %8, %1\n v_mul_lo_u32 %2, %8, %2\n v_mul_lo_u32 %3, %8, %3\n"
   "v_mul_lo_u32 %4, %8, %4\n v_mul_lo_u32 %5, %8, %5\n v_mul_lo_u32 %6, %8, %6\n v_mul_lo_u32 %7, %8, %7\n"
   : "+v"(b0),"+v"(b1),"+v"(b2),"+v"(b3),"+v"(b4),"+v"(b5),"+v"(b6),"+v"(b7) : "v"(x0));
KERNEL_END

KERNEL_BEGIN(k_mul_hi_u32)
  asm volatile(
   "v_mul_hi_u32 %0, %8, %0\n v_mul_hi_u32 %1, %8, %1\n v_mul_hi_u32 %2, %8, %2\n v_mul_hi_u32 %3, %8, %3\n"
   "v_mul_hi_u32 %4, %8, %4\n v_mul_hi_u32 %5, %8, %5\n v_mul_hi_u32 %6, %8, %6\n v_mul_hi_u32 %7, %8, %7\n"
   : "+v"(b0),"+v"(b1),"+v"(b2),"+v"(b3),"+v"(b4),"+v"(b5),"+v"(b6),"+v"(b7) : "v"(x0));
KERNEL_END

KERNEL_BEGIN(k_mad_u32_u24)
  asm volatile(
   "v_mad_u32_u24 %0, %8, %0, %0\n v_mad_u32_u24 %1, %8, %1, %1\n v_mad_u32_u24 %2, %8, %2, %2\n v_mad_u32_u24 %3, %8, %3, %3\n"
   "v_mad_u32_u24 %4, %8, %4, %4\n v_mad_u32_u24 %5, %8, %5, %5\n v_mad_u32_u24 %6, %8, %6, %6\n v_mad_u32_u24 %7, %8, %7, %7\n"
   : "+v"(b0),"+v"(b1),"+v"(b2),"+v"(b3),"+v"(b4),"+v"(b5),"+v"(b6),"+v"(b7) : "v"(x0));
KERNEL_END

KERNEL_BEGIN(k_add_u32)
  asm volatile(
   "v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n"
   "v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n"
   : "+v"(b0),"+v"(b1),"+v"(b2),"+v"(b3),"+v"(b4),"+v"(b5),"+v"(b6),"+v"(b7) : "v"(x0));
KERNEL_END

KERNEL_BEGIN(k_and_b32)
  asm volatile(
   "v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n"
   "v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7\n"
   : "+v"(b0),"+v"(b1),"+v"(b2),"+v"(b3),"+v"(b4),"+v"(b5),"+v"(b6),"+v"(b7) : "v"(x0));
KERNEL_END

KERNEL_BEGIN(k_lshl_add_u64)
  asm volatile(
   "v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
   "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(d0));
KERNEL_END

KERNEL_BEGIN(k_lshrrev_b64)
  asm volatile(
   "v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3\n"
   "v_lshrrev_b64 %4, 1, %4\n v_lshrrev_b64 %5, 1, %5\n v_lshrrev_b64 %6, 1, %6\n v_lshrrev_b64 %7, 1, %7\n"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7));
KERNEL_END

KERNEL_BEGIN(k_alignbit_b32)
  asm volatile(
   "v_alignbit_b32 %0, %8, %0, 26\n v_alignbit_b32 %1, %8, %1, 26\n v_alignbit_b32 %2, %8, %2, 26\n v_alignbit_b32 %3, %8, %3, 26\n"
   "v_alignbit_b32 %4, %8, %4, 26\n v_alignbit_b32 %5, %8, %5, 26\n v_alignbit_b32 %6, %8, %6, 26\n v_alignbit_b32 %7, %8, %7, 26\n"
   : "+v"(b0),"+v"(b1),"+v"(b2),"+v"(b3),"+v"(b4),"+v"(b5),"+v"(b6),"+v"(b7) : "v"(x0));
KERNEL_END

KERNEL_BEGIN(k_fma_f64)
  asm volatile(
   "v_fma_f64 %0, %8, %0, %0\n v_fma_f64 %1, %8, %1, %1\n v_fma_f64 %2, %8, %2, %2\n v_fma_f64 %3, %8, %3, %3\n"
   "v_fma_f64 %4, %8, %4, %4\n v_fma_f64 %5, %8, %5, %5\n v_fma_f64 %6, %8, %6, %6\n v_fma_f64 %7, %8, %7, %7\n"
   : "+v"(d0),"+v"(d1),"+v"(d2),"+v"(d3),"+v"(d4),"+v"(d5),"+v"(d6),"+v"(d7) : "v"(d0));
KERNEL_END

KERNEL_BEGIN(k_add_f64)
  asm volatile(
   "v_add_f64 %0, %8, %0\n v_add_f64 %1, %8, %1\n v_add_f64 %2, %8, %2\n v_add_f64 %3, %8, %3\n"
   "v_add_f64 %4, %8, %4\n v_add_f64 %5, %8, %5\n v_add_f64 %6, %8, %6\n v_add_f64 %7, %8, %7\n"
   : "+v"(d0),"+v"(d1),"+v"(d2),"+v"(d3),"+v"(d4),"+v"(d5),"+v"(d6),"+v"(d7) : "v"(d0));
KERNEL_END

KERNEL_BEGIN(k_addc_chain)   // add_co + 2 wait states + addc, the carry-chain cost
  asm volatile(
   "v_add_co_u32 %0, vcc, %8, %0\n s_nop 1\n v_addc_co_u32 %1, vcc, %8, %1, vcc\n s_nop 1\n v_addc_co_u32 %2, vcc, %8, %2, vcc\n s_nop 1\n v_addc_co_u32 %3, vcc, %8, %3, vcc\n"
   "s_nop 1\n v_addc_co_u32 %4, vcc, %8, %4, vcc\n s_nop 1\n v_addc_co_u32 %5, vcc, %8, %5, vcc\n s_nop 1\n v_addc_co_u32 %6, vcc, %8, %6, vcc\n s_nop 1\n v_addc_co_u32 %7, vcc, %8, %7, vcc\n"
   : "+v"(b0),"+v"(b1),"+v"(b2),"+v"(b3),"+v"(b4),"+v"(b5),"+v"(b6),"+v"(b7) : "v"(x0) : "vcc");
KERNEL_END

// mixed: 1 mad_u64_u32 + 1 cheap op interleaved: do they overlap or serialize?
KERNEL_BEGIN(k_mad_plus_and)
  asm volatile(
   "v_mad_u64_u32 %0, vcc, %16, %17, %0\n v_and_b32 %8, %16, %8\n v_mad_u64_u32 %1, vcc, %16, %17, %1\n v_and_b32 %9, %16, %9\n"
   "v_mad_u64_u32 %2, vcc, %16, %17, %2\n v_and_b32 %10, %16, %10\n v_mad_u64_u32 %3, vcc, %16, %17, %3\n v_and_b32 %11, %16, %11\n"
   "v_mad_u64_u32 %4, vcc, %16, %17, %4\n v_and_b32 %12, %16, %12\n v_mad_u64_u32 %5, vcc, %16, %17, %5\n v_and_b32 %13, %16, %13\n"
   "v_mad_u64_u32 %6, vcc, %16, %17, %6\n v_and_b32 %14, %16, %14\n v_mad_u64_u32 %7, vcc, %16, %17, %7\n v_and_b32 %15, %16, %15\n"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7),
     "+v"(b0),"+v"(b1),"+v"(b2),"+v"(b3),"+v"(b4),"+v"(b5),"+v"(b6),"+v"(b7) : "v"(x0),"v"(x1) : "vcc");
KERNEL_END

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; int ops_per_iter; };

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int ncu = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  printf("device %s CUs=%d clock=%.0f MHz\n", prop.gcnArchName, ncu, clk/1e6);
  uint32_t* out; CHECK(hipMalloc(&out, 4096));
  Entry es[] = {
    {"v_mad_u64_u32", k_mad_u64_u32, 8}, {"v_mul_lo_u32", k_mul_lo_u32, 8}, {"v_mul_hi_u32", k_mul_hi_u32, 8},
    {"v_mad_u32_u24", k_mad_u32_u24, 8}, {"v_add_u32", k_add_u32, 8}, {"v_and_b32", k_and_b32, 8},
    {"v_lshl_add_u64", k_lshl_add_u64, 8}, {"v_lshrrev_b64", k_lshrrev_b64, 8}, {"v_alignbit_b32", k_alignbit_b32, 8},
    {"v_fma_f64", k_fma_f64, 8}, {"v_add_f64", k_add_f64, 8}, {"addc_chain(8 links)", k_addc_chain, 8},
    {"mad_u64_u32+and pair", k_mad_plus_and, 8},
  };
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int wpc : {4, 8, 16, 32}) {       // waves per CU
    int blocks = ncu * wpc / 4;          // 256 threads = 4 waves
    printf("--- waves/CU=%d (blocks=%d x 256)\n", wpc, blocks);
    for (auto& e : es) {
      e.k<<<blocks,256>>>(out, 1); CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0)); for (int r=0;r<5;r++) e.k<<<blocks,256>>>(out, r); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms/=5;
      double wave_instr = (double)blocks*4*ITERS*e.ops_per_iter;   // wave-level instrs
      double per_simd = wave_instr/(ncu*4.0);
      double cyc = (ms*1e-3*clk)/per_simd;
      printf("%-24s %8.3f ms  %.3e lane-ops/s  %.2f cyc/wave-instr/SIMD\n", e.name, ms, wave_instr*64/(ms*1e-3), cyc);
    }
  }
  return 0;
}
