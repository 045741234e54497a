// VALU issue-rate microbenchmark (development aid): cycles per wave-instruction for scalar and packed f32
// ops at 1, 2 and 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void k(float* out, int iters, float seed) {
  f2 a[8], b = {seed, seed * 0.5f}, c = {0.25f, 0.125f};
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f2{seed + i, seed - i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (KIND == 3) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x)); }
        if (KIND == 4) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x)); }
        if (KIND == 5) { asm volatile("v_mov_b32 %0, %1" : "=v"(a[i].x) : "v"(a[(i + 1) & 7].y)); }
        if (KIND == 6) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(b), "v"(c)); }
        if (KIND == 7) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x)); }
      }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  if (s == 1234.5f) out[threadIdx.x] = s;
}

template <int KIND>
int run(const char* name, float* d) {
  const int iters = 4000;
  for (int waves_per_simd : {1, 2, 4, 8}) {
    int threads = 64 * 4 * waves_per_simd;  // one block per CU
    if (threads > 1024) { threads = 1024; }
    int blocks = 256 * (waves_per_simd == 8 ? 2 : 1);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<KIND><<<blocks, threads>>>(d, 10, 1.0f);
    CHK(hipEventRecord(e0));
    k<KIND><<<blocks, threads>>>(d, iters, 1.0f);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double insts_per_wave = (double)iters * 32;
    double cyc = ms * 1e-3 * 2.4e9;  // at the nominal 2.4 GHz
    printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f cyc/inst/wave, %.2f cyc per inst per SIMD (nominal clock)\n", name, waves_per_simd, ms,
           cyc / insts_per_wave, cyc / insts_per_wave / waves_per_simd);
  }
  return 0;
}

int main() {
  float* d; CHK(hipMalloc(&d, 4096));
  run<0>("v_pk_fma_f32", d); run<1>("v_pk_add_f32", d); run<2>("v_pk_mul_f32", d);
  run<3>("v_fma_f32", d); run<4>("v_add_f32", d); run<5>("v_mov_b32", d); run<6>("pk_fma opsel", d); run<7>("v_cndmask", d);
  return 0;
}
