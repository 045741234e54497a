// Microbenchmark (development aid): LDS exchange throughput per CU for the access shapes of the patch kernel.
// One 512-thread workgroup per CU (8 waves), each wave owns a private region (like X1).  Shapes:
//   row  : lane-consecutive (address = J*STRIDE + lane)          - the "write" side of a register<->lane transpose
//   col  : lane-strided     (address = lane*STRIDE + J)          - the "read" side
// for 4-, 8- and 16-byte elements.  Reports bytes per clock per CU (peak 128 B/clk).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <class V, int NREG, int STRIDE, bool WRITE, bool COL>
__global__ __launch_bounds__(512) void k(float* out, int reps, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int EL = sizeof(V) / 4;
  V* lds = reinterpret_cast<V*>(smem) + (threadIdx.x >> 6) * (NREG * STRIDE);   // per-wave private region
  const int lane = threadIdx.x & 63;
  V v[NREG];
#pragma unroll
  for (int j = 0; j < NREG; ++j) for (int e = 0; e < EL; ++e) reinterpret_cast<float*>(&v[j])[e] = (float)(lane + j + e);
  // initialise the region
  for (int j = 0; j < NREG; ++j) lds[j * STRIDE + lane] = v[j];
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    if (WRITE) {
#pragma unroll
      for (int j = 0; j < NREG; ++j) lds[COL ? lane * STRIDE + j : j * STRIDE + lane] = v[j];
    } else {
#pragma unroll
      for (int j = 0; j < NREG; ++j) {
        V x = lds[COL ? lane * STRIDE + j : j * STRIDE + lane];
        for (int e = 0; e < EL; ++e) reinterpret_cast<float*>(&v[j])[e] += reinterpret_cast<float*>(&x)[e];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float acc = 0;
#pragma unroll
  for (int j = 0; j < NREG; ++j) for (int e = 0; e < EL; ++e) acc += reinterpret_cast<float*>(&v[j])[e];
  if (acc == 1.2345f) out[threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <class V, int NREG, int STRIDE, bool WRITE, bool COL>
int run(const char* name, float* out, unsigned long long* d_cyc) {
  const int reps = 200;
  size_t lds_bytes = (size_t)8 * NREG * STRIDE * sizeof(V);
  if (lds_bytes > 160 * 1024) { printf("%-40s needs %zu KiB LDS: skipped\n", name, lds_bytes >> 10); return 0; }
  CHK(hipFuncSetAttribute((const void*)k<V, NREG, STRIDE, WRITE, COL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CHK(hipEventRecord(e0));
    k<V, NREG, STRIDE, WRITE, COL><<<256, 512, lds_bytes>>>(out, reps, d_cyc);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    CHK(hipEventElapsedTime(&ms, e0, e1));
  }
  unsigned long long cyc; CHK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
  double bytes = (double)reps * 512 * NREG * sizeof(V);  // per CU
  printf("%-40s %6.1f B/clk/CU (shader clock), %6.1f GB/s/CU by wall time, %.2f us per 128 KiB\n", name, bytes / (double)cyc,
         bytes / (ms * 1e-3) / 1e9, 131072.0 / (bytes / (ms * 1e-3)) * 1e6);
  return 0;
}

int main() {
  float* out; unsigned long long* cyc; CHK(hipMalloc(&out, 4096)); CHK(hipMalloc(&cyc, 8));
  run<float, 64, 65, true, false>("b32 write row (J*65+lane)", out, cyc);
  run<float, 64, 65, false, true>("b32 read  col (lane*65+J)", out, cyc);
  run<float, 64, 65, true, true>("b32 write col (lane*65+J)", out, cyc);
  run<float, 64, 65, false, false>("b32 read  row (J*65+lane)", out, cyc);
  run<float, 64, 64, false, false>("b32 read  row stride 64", out, cyc);
  run<f2, 32, 65, true, false>("b64 write row (J*65+lane), 32 regs", out, cyc);
  run<f2, 32, 65, false, true>("b64 read  col (lane*65+J), 32 regs", out, cyc);
  run<f2, 32, 65, true, true>("b64 write col, 32 regs", out, cyc);
  run<f2, 32, 65, false, false>("b64 read  row, 32 regs", out, cyc);
  run<f2, 32, 64, false, false>("b64 read  row stride 64, 32 regs", out, cyc);
  run<f4, 16, 65, true, false>("b128 write row, 16 regs", out, cyc);
  run<f4, 16, 65, false, false>("b128 read  row, 16 regs", out, cyc);
  run<f4, 16, 65, false, true>("b128 read  col (lane*65+J), 16 regs", out, cyc);
  run<f4, 16, 65, true, true>("b128 write col, 16 regs", out, cyc);
  return 0;
}
