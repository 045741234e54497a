// Microbenchmark (development aid): does the DRAM care how a patch's 256 KiB of plane stores are laid out?
// Every workgroup alternates a 256 KiB streaming read (stand-in for the K stream) with a 256 KiB store burst,
// either as 256 rows x 1 KiB with a 16 KiB row stride (row-major colour planes, what K1 does) or as one
// contiguous 256 KiB tile (tile-major planes).  8-byte non-temporal stores, 16-byte non-temporal loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool TILED, bool READ>
__global__ __launch_bounds__(512) void k(const float* src, float* dst, int rounds, float* sink) {
  extern __shared__ float pad[];
  const int t = threadIdx.x;
  float acc = 0;
  for (int r = 0; r < rounds; ++r) {
    unsigned id = (unsigned)blockIdx.x * 977u + (unsigned)r * 131u;
    if (READ) {
      const f4* s = reinterpret_cast<const f4*>(src) + (size_t)(id % 4096u) * 16384;  // 256 KiB chunks of a 1 GiB buffer
      f4 v[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = __builtin_nontemporal_load(s + i * 512 + t);
#pragma unroll
      for (int i = 0; i < 32; ++i) acc += v[i].x + v[i].w;
    }
    unsigned tile = id % 1024u;  // 4 planes x 16 x 16 tiles of 256 x 256
    if (TILED) {
      f2* base = reinterpret_cast<f2*>(dst + (size_t)tile * 65536);
#pragma unroll
      for (int i = 0; i < 64; ++i) __builtin_nontemporal_store(f2{acc, (float)i}, base + i * 512 + t);
    } else {
      int plane = tile >> 8, ty = (tile >> 4) & 15, tx = tile & 15;
      float* base = dst + ((size_t)plane * 4096 + (size_t)ty * 256) * 4096 + (size_t)tx * 256;
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        int row = i * 4 + (t >> 7), col = (t & 127) * 2;
        __builtin_nontemporal_store(f2{acc, (float)i}, reinterpret_cast<f2*>(base + (size_t)row * 4096 + col));
      }
    }
  }
  if (acc == 1.2345f) sink[t] = acc + pad[0];
}

template <bool TILED, bool READ>
int run(const char* name, const float* src, float* dst, float* sink) {
  const int rounds = 16;
  CHK(hipFuncSetAttribute((const void*)k<TILED, READ>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CHK(hipEventRecord(e0));
    k<TILED, READ><<<256, 512, 100 * 1024>>>(src, dst, rounds, sink);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    CHK(hipEventElapsedTime(&ms, e0, e1));
  }
  double bytes = 256.0 * rounds * 262144 * (READ ? 2 : 1);
  printf("%-44s %.3f ms  %.2f TB/s (%s)\n", name, ms, bytes / ms / 1e9, READ ? "read + write" : "write only");
  return 0;
}

int main() {
  float *src, *dst, *sink;
  CHK(hipMalloc(&src, (size_t)1 << 30)); CHK(hipMemset(src, 0, (size_t)1 << 30));
  CHK(hipMalloc(&dst, (size_t)4 * 4096 * 4096 * 4)); CHK(hipMalloc(&sink, 4096));
  run<false, false>("rows of 1 KiB, stride 16 KiB, stores only", src, dst, sink);
  run<true, false>("contiguous 256 KiB tiles, stores only", src, dst, sink);
  run<false, true>("rows of 1 KiB, stride 16 KiB, with reads", src, dst, sink);
  run<true, true>("contiguous 256 KiB tiles, with reads", src, dst, sink);
  return 0;
}
