// Microbenchmark (development aid): per-CU load throughput with one 512-thread workgroup per CU and 32
// independent 16-byte (or 8-byte) loads in flight per thread, for HBM-streamed vs L2-resident data.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int W>  // W = 4 (16 B/lane) or 2 (8 B/lane)
__global__ __launch_bounds__(512) void k(const float* buf, size_t region_floats, int rounds, float* out) {
  extern __shared__ float pad[];  // force one workgroup per CU
  float acc = 0;
  const size_t chunk = 512 * 32 * W;  // floats per round per workgroup
  for (int r = 0; r < rounds; ++r) {
    size_t base = ((size_t)blockIdx.x * rounds + r) * chunk % region_floats;
    if (W == 4) {
      f4 v[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = *reinterpret_cast<const f4*>(buf + base + ((size_t)i * 512 + threadIdx.x) * 4);
#pragma unroll
      for (int i = 0; i < 32; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    } else {
      f2 v[64];
#pragma unroll
      for (int i = 0; i < 64; ++i) v[i] = *reinterpret_cast<const f2*>(buf + base + ((size_t)i * 512 + threadIdx.x) * 2);
#pragma unroll
      for (int i = 0; i < 64; ++i) acc += v[i].x + v[i].y;
    }
  }
  if (acc == 1.2345f) out[threadIdx.x] = acc + pad[0];
}

template <int W>
int run(const char* name, const float* d, size_t region_bytes, int blocks, float* out) {
  const int rounds = 16;
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  CHK(hipFuncSetAttribute((const void*)k<W>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  for (int rep = 0; rep < 2; ++rep) {
    CHK(hipEventRecord(e0));
    k<W><<<blocks, 512, 100 * 1024>>>(d, region_bytes / 4, rounds, out);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  }
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  double bytes = (double)blocks * rounds * 512 * 32 * 16;
  printf("%-34s blocks %3d: %.3f ms, %.2f TB/s total, %.1f GB/s per workgroup\n", name, blocks, ms, bytes / ms / 1e9, bytes / ms / 1e6 / blocks);
  return 0;
}

int main() {
  float *d, *out; size_t big = (size_t)1 << 30;
  CHK(hipMalloc(&d, big)); CHK(hipMemset(d, 0, big)); CHK(hipMalloc(&out, 4096));
  run<4>("16B loads, HBM stream (1 GB)", d, big, 256, out);
  run<4>("16B loads, 8 MB region (L2)", d, (size_t)8 << 20, 256, out);
  run<4>("16B loads, 64 MB region (MALL)", d, (size_t)64 << 20, 256, out);
  run<2>("8B loads, HBM stream (1 GB)", d, big, 256, out);
  run<2>("8B loads, 8 MB region (L2)", d, (size_t)8 << 20, 256, out);
  run<4>("16B loads, HBM, 32 CUs only", d, big, 32, out);
  run<4>("16B loads, L2 8MB, 32 CUs only", d, (size_t)8 << 20, 32, out);
  run<4>("16B loads, HBM, 8 CUs only", d, big, 8, out);
  return 0;
}
