// Micro-benchmark: candidate secp256k1 field-multiplication layouts on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o femul_bench femul_bench.hip && ./femul_bench
// Each thread runs a dependent chain x = x*y (mod p-ish; only timing matters here, the
// real implementation is parity-tested elsewhere).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
typedef unsigned __int128 u128;
#define ITERS 512

struct fe52 { uint64_t n[5]; };
__device__ __forceinline__ void mul52(fe52& r, const fe52& A, const fe52& B) {
  uint64_t a0=A.n[0],a1=A.n[1],a2=A.n[2],a3=A.n[3],a4=A.n[4];
  uint64_t b0=B.n[0],b1=B.n[1],b2=B.n[2],b3=B.n[3],b4=B.n[4];
  const uint64_t M=0xFFFFFFFFFFFFFULL,R=0x1000003D10ULL;
  u128 c,d; uint64_t t3,t4,tx,u0;
  d=(u128)a0*b3+(u128)a1*b2+(u128)a2*b1+(u128)a3*b0;
  c=(u128)a4*b4;
  d+=(u128)R*(uint64_t)c; c>>=64;
  t3=(uint64_t)d&M; d>>=52;
  d+=(u128)a0*b4+(u128)a1*b3+(u128)a2*b2+(u128)a3*b1+(u128)a4*b0;
  d+=(u128)(R<<12)*(uint64_t)c;
  t4=(uint64_t)d&M; d>>=52; tx=t4>>48; t4&=(M>>4);
  c=(u128)a0*b0;
  d+=(u128)a1*b4+(u128)a2*b3+(u128)a3*b2+(u128)a4*b1;
  u0=(uint64_t)d&M; d>>=52; u0=(u0<<4)|tx;
  c+=(u128)u0*(R>>4);
  r.n[0]=(uint64_t)c&M; c>>=52;
  c+=(u128)a0*b1+(u128)a1*b0;
  d+=(u128)a2*b4+(u128)a3*b3+(u128)a4*b2;
  c+=(u128)((uint64_t)d&M)*R; d>>=52;
  r.n[1]=(uint64_t)c&M; c>>=52;
  c+=(u128)a0*b2+(u128)a1*b1+(u128)a2*b0;
  d+=(u128)a3*b4+(u128)a4*b3;
  c+=(u128)R*(uint64_t)d; d>>=64;
  r.n[2]=(uint64_t)c&M; c>>=52;
  c+=(u128)(R<<12)*(uint64_t)d; c+=t3;
  r.n[3]=(uint64_t)c&M; c>>=52; r.n[4]=(uint64_t)c+t4;
}

template<int L, int BITS, uint32_t FOLD, int SH> struct feL { uint32_t n[L]; };
template<int L, int BITS, uint32_t FOLD, int SH>
__device__ __forceinline__ void mulL(feL<L,BITS,FOLD,SH>& R, const feL<L,BITS,FOLD,SH>& A, const feL<L,BITS,FOLD,SH>& B) {
  const uint32_t M=(1u<<BITS)-1;
  uint32_t t[2*L]; uint64_t c=0;
  #pragma unroll
  for (int k=0;k<2*L-1;k++){
    #pragma unroll
    for (int i=0;i<L;i++){ int j=k-i; if(j<0||j>=L) continue; c+=(uint64_t)A.n[i]*B.n[j]; }
    t[k]=(uint32_t)c&M; c>>=BITS; }
  t[2*L-1]=(uint32_t)c;
  uint64_t d=0;
  #pragma unroll
  for(int k=0;k<L;k++){ d+=t[k]; d+=(uint64_t)t[k+L]*FOLD; if(k>0) d+=(uint64_t)t[k+L-1]<<SH; R.n[k]=(uint32_t)d&M; d>>=BITS; }
  d+=(uint64_t)t[2*L-1]<<SH;
  uint64_t e=(uint64_t)R.n[0]+d*FOLD; R.n[0]=(uint32_t)e&M; e>>=BITS; e+=R.n[1]+(d<<SH); R.n[1]=(uint32_t)e&M; e>>=BITS; R.n[2]+=(uint32_t)e;
}
template<int L, int BITS, uint32_t FOLD, int SH>
__device__ __forceinline__ void sqrL(feL<L,BITS,FOLD,SH>& R, const feL<L,BITS,FOLD,SH>& A) {
  const uint32_t M=(1u<<BITS)-1;
  uint32_t t[2*L]; uint64_t c=0;
  uint32_t A2[L];
  #pragma unroll
  for (int i=0;i<L;i++) A2[i]=A.n[i]*2;
  #pragma unroll
  for (int k=0;k<2*L-1;k++){
    #pragma unroll
    for (int i=0;i<L;i++){ int j=k-i; if(j<0||j>=L||i>j) continue; if(i==j) c+=(uint64_t)A.n[i]*A.n[j]; else c+=(uint64_t)A2[i]*A.n[j]; }
    t[k]=(uint32_t)c&M; c>>=BITS; }
  t[2*L-1]=(uint32_t)c;
  uint64_t d=0;
  #pragma unroll
  for(int k=0;k<L;k++){ d+=t[k]; d+=(uint64_t)t[k+L]*FOLD; if(k>0) d+=(uint64_t)t[k+L-1]<<SH; R.n[k]=(uint32_t)d&M; d>>=BITS; }
  d+=(uint64_t)t[2*L-1]<<SH;
  uint64_t e=(uint64_t)R.n[0]+d*FOLD; R.n[0]=(uint32_t)e&M; e>>=BITS; e+=R.n[1]+(d<<SH); R.n[1]=(uint32_t)e&M; e>>=BITS; R.n[2]+=(uint32_t)e;
}
typedef feL<10,26,15632u,10> fe26;
typedef feL<9,29,31264u,8> fe29;

__global__ void __launch_bounds__(256) k_mul52(uint64_t* out, uint32_t seed){
  fe52 x,y; for(int i=0;i<5;i++){x.n[i]=(threadIdx.x*77+seed+i)&0xFFFFFFFFFFFFFULL; y.n[i]=(0x123456789ABCDULL*(threadIdx.x+i+1))&0xFFFFFFFFFFFFFULL;}
  for(int it=0;it<ITERS;it++){ mul52(x,x,y); }
  uint64_t s=0; for(int i=0;i<5;i++) s^=x.n[i]; if(s==0x1234) out[threadIdx.x]=s;
}
template<class FE,int L> __global__ void __launch_bounds__(256) k_mulL(uint64_t* out, uint32_t seed){
  FE x,y; for(int i=0;i<L;i++){x.n[i]=(threadIdx.x*77+seed+i)&0x3FFFFFF; y.n[i]=(0x1234567u*(threadIdx.x+i+1))&0x3FFFFFF;}
  for(int it=0;it<ITERS;it++){ mulL(x,x,y); }
  uint64_t s=0; for(int i=0;i<L;i++) s^=x.n[i]; if(s==0x1234) out[threadIdx.x]=s;
}
template<class FE,int L> __global__ void __launch_bounds__(256) k_sqrL(uint64_t* out, uint32_t seed){
  FE x; for(int i=0;i<L;i++){x.n[i]=(threadIdx.x*77+seed+i)&0x3FFFFFF;}
  for(int it=0;it<ITERS;it++){ sqrL(x,x); }
  uint64_t s=0; for(int i=0;i<L;i++) s^=x.n[i]; if(s==0x1234) out[threadIdx.x]=s;
}
// two independent chains per thread (ILP)
template<class FE,int L> __global__ void __launch_bounds__(256) k_mulL2(uint64_t* out, uint32_t seed){
  FE x,y,z; for(int i=0;i<L;i++){x.n[i]=(threadIdx.x*77+seed+i)&0x3FFFFFF; y.n[i]=(0x1234567u*(threadIdx.x+i+1))&0x3FFFFFF; z.n[i]=x.n[i]^0x55;}
  for(int it=0;it<ITERS/2;it++){ mulL(x,x,y); mulL(z,z,y); }
  uint64_t s=0; for(int i=0;i<L;i++) s^=x.n[i]^z.n[i]; if(s==0x1234) out[threadIdx.x]=s;
}

int main(){
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop,0));
  int ncu=prop.multiProcessorCount; double clk=prop.clockRate*1e3;
  printf("device %s CUs=%d clock=%.0f MHz\n",prop.gcnArchName,ncu,clk/1e6);
  uint64_t* out; CHECK(hipMalloc(&out,8192));
  hipEvent_t e0,e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  struct E{const char* name; void(*k)(uint64_t*,uint32_t);} es[]={
    {"fe_mul 5x52 int128",k_mul52},{"fe_mul 10x26",k_mulL<fe26,10>},{"fe_mul 9x29",k_mulL<fe29,9>},
    {"fe_sqr 10x26",k_sqrL<fe26,10>},{"fe_sqr 9x29",k_sqrL<fe29,9>},{"fe_mul 10x26 x2 ILP",k_mulL2<fe26,10>},{"fe_mul 9x29 x2 ILP",k_mulL2<fe29,9>}};
  for(int wpc: {4,8,16,32}){
    int blocks=ncu*wpc/4; printf("--- waves/CU=%d\n",wpc);
    for(auto&e:es){
      e.k<<<blocks,256>>>(out,1); CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0)); for(int r=0;r<3;r++) e.k<<<blocks,256>>>(out,r); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms,e0,e1)); ms/=3;
      double n=(double)blocks*256*ITERS;
      double cyc_per_wave_op=(ms*1e-3*clk)/((double)blocks*4*ITERS/(ncu*4.0));
      printf("%-22s %8.3f ms  %.3e fe-ops/s  %.0f cyc/wave-op/SIMD\n",e.name,ms,n/(ms*1e-3),cyc_per_wave_op);
    }
  }
  return 0;
}
