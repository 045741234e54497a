// VALU issue rate by operand pattern (development aid): what does a wave-instruction cost when its operands are DISTINCT, changing
// registers - as in the butterfly networks - rather than one accumulator and two constants (valu_rate.hip)?
//   hipcc --offload-arch=gfx950 -O3 -o valu_mix valu_mix.hip && ./valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

// 32 registers r[0..31]; instruction i writes r[i % 32] from r[(i + 5) % 32], r[(i + 11) % 32], r[(i + 17) % 32]: every operand
// distinct, every result consumed 15 ... 27 instructions later (no dependency stall at 2 or more waves per SIMD)
template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float seed) {
  float r[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) r[i] = seed + i * 0.001f + threadIdx.x * 1e-6f;
  const unsigned long long mask64 = __builtin_amdgcn_read_exec() ^ (0x5555555555555555ull * (unsigned)(seed != 7.f));
  const float mf = (threadIdx.x & 1) ? 1.f : 0.f;
  const unsigned mi = (threadIdx.x & 1) ? 0xffffffffu : 0u;
  f2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) p[i] = f2{seed + i, seed - i * 0.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      float& d = r[i % 32];
      const float a = r[(i + 5) % 32], b = r[(i + 11) % 32], c = r[(i + 17) % 32];
      if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));        // 3 distinct sources
      if (KIND == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));                   // d += a b
      if (KIND == 2) asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));                    // 2 sources
      if (KIND == 3) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
      if (KIND == 4) asm volatile("v_fmamk_f32 %0, %1, 0x3f3504f3, %2" : "=v"(d) : "v"(a), "v"(b));     // a * literal + b
      if (KIND == 5) asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(d) : "v"(a), "v"(b));                // 2 distinct sources
      if (KIND == 6) asm volatile("v_fma_f32 %0, %1, 2.0, %2" : "=v"(d) : "v"(a), "v"(b));               // inline constant
      if (KIND == 7) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
      if (KIND == 8) asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(d) : "v"(a), "v"(b), "v"(c));        // with a source modifier
      if (KIND == 30) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(d), "+v"(r[(i + 5) % 32]));   // gfx950: swaps d[32:63] with the other register's [0:31]
      if (KIND == 31) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(d), "+v"(r[(i + 5) % 32]));
      if (KIND == 32) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(a));
      if (KIND == 33) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(a));
      if (KIND == 34) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(d) : "v"(a), "v"(b));
      if (KIND == 35) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(mask64));  // mask in an SGPR pair, as the compiler emits it
      if (KIND == 36) {  // the same selection by arithmetic: d = a + m (b - a), m = 0 / 1 per lane
        float tdiff;
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(tdiff) : "v"(b), "v"(a));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(mf), "v"(tdiff), "v"(a));
      }
      if (KIND == 37) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mi), "v"(a), "v"(b));  // bitwise select with a per-lane all-ones / all-zeros mask
      if (KIND >= 10 && KIND < 30) {
        f2& pd = p[i % 16];
        const f2 pa = p[(i + 3) % 16], pb = p[(i + 7) % 16], pc = p[(i + 11) % 16];
        if (KIND == 10) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pd) : "v"(pa), "v"(pb), "v"(pc));
        if (KIND == 11) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pd) : "v"(pa), "v"(pb));
        if (KIND == 12) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pd) : "v"(pa), "v"(pb));
        if (KIND == 13) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(pd) : "v"(pa), "v"(pb), "v"(pc));
        if (KIND == 14) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(pd) : "v"(pa), "v"(pb));
        if (KIND == 15) asm volatile("v_pk_fma_f32 %0, %1, %2, %2" : "=v"(pd) : "v"(pa), "v"(pb));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += r[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += p[i].x + p[i].y;
  if (s == 1234.5f) out[threadIdx.x] = s;
}

// alternating mixes as the networks have them: K20 = fma,add,sub,mul round robin;  K21 = the same as packed ops
template <int KIND>
__global__ __launch_bounds__(1024) void kmix(float* out, int iters, float seed) {
  float r[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) r[i] = seed + i * 0.001f + threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      float& d = r[i % 32];
      const float a = r[(i + 5) % 32], b = r[(i + 11) % 32], c = r[(i + 17) % 32];
      if (KIND == 0) {
        if (i % 4 == 0) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        if (i % 4 == 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
        if (i % 4 == 2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
        if (i % 4 == 3) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
      } else {  // dependent chain of length 1: every instruction consumes the previous result
        float& e = r[(i + 31) % 32];
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(e), "v"(b), "v"(c));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += r[i];
  if (s == 1234.5f) out[threadIdx.x] = s;
}

template <class F>
int time_it(const char* name, F launch, double flops_per_inst) {
  const int iters = 2000;
  for (int wps : {1, 2, 4}) {
    const int threads = 256 * wps, blocks = 256;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    launch(blocks, threads, 10);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    launch(blocks, threads, iters);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double insts = (double)iters * 64, cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f cyc per inst per wave, %.2f per inst per SIMD (at 2.4 GHz)  %.1f TFLOP/s\n", name, wps, ms, cyc / insts,
           cyc / insts / wps, flops_per_inst * 64 * insts * wps * 4 * 256 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
#define RUN(KERNEL, NAME, FL) time_it(NAME, [&](int b, int t, int it) { KERNEL<<<b, t>>>(d, it, 1.0f); }, FL)

int main() {
  float* d; CHK(hipMalloc(&d, 8192));
  RUN(k<0>, "v_fma 3 distinct src", 2); RUN(k<8>, "v_fma 3 src, neg modifier", 2); RUN(k<1>, "v_fmac d += a b", 2); RUN(k<5>, "v_fma 2 distinct src", 2);
  RUN(k<6>, "v_fma inline const", 2); RUN(k<4>, "v_fmamk literal", 2);
  RUN(k<2>, "v_add", 1); RUN(k<7>, "v_sub", 1); RUN(k<3>, "v_mul", 1);
  RUN(k<10>, "v_pk_fma 3 distinct src", 4); RUN(k<15>, "v_pk_fma 2 distinct src", 4); RUN(k<13>, "v_pk_fma op_sel swizzle", 4);
  RUN(k<11>, "v_pk_add", 2); RUN(k<14>, "v_pk_add neg", 2); RUN(k<12>, "v_pk_mul", 2);
  RUN(kmix<0>, "mix fma/add/fmac/sub", 1.5); RUN(kmix<1>, "fma chain (dependent)", 2);
  RUN(k<30>, "v_permlane32_swap", 0); RUN(k<31>, "v_permlane16_swap", 0); RUN(k<32>, "v_mov_dpp quad_perm", 0); RUN(k<33>, "v_mov_dpp row_ror:8", 0);
  RUN(k<34>, "v_cndmask (vcc)", 0); RUN(k<35>, "v_cndmask_e64 (sgpr mask)", 0); RUN(k<36>, "select by sub + fma (2 inst)", 0); RUN(k<37>, "v_bfi select", 0);
  return 0;
}
