// Microbenchmark (development aid): write W MB streaming, then read it back either in the same order
// (oldest data first) or in reverse order (newest first).  Does the 256 MB Infinity Cache serve the tail?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void wr(f4* p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = f4{1, 2, 3, 4};
}
__global__ void rd(const f4* p, size_t n4, int reverse, float* out) {
  float acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f4 a = p[reverse ? n4 - 1 - i : i];
    acc += a.x + a.w;
  }
  if (acc == 1.5f) out[0] = acc;
}
int main() {
  size_t cap = (size_t)1 << 30; f4* d; float* o;
  CHK(hipMalloc(&d, cap)); CHK(hipMalloc(&o, 64));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (size_t mb : {64, 128, 192, 268, 400, 700}) {
    size_t n4 = (mb << 20) / 16;
    for (int reverse = 0; reverse < 2; ++reverse) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        wr<<<8192, 256>>>(d, n4);
        CHK(hipEventRecord(e0));
        rd<<<8192, 256>>>(d, n4, reverse, o);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
      }
      printf("write %4zu MB then read %s: %.3f ms = %.2f TB/s\n", mb, reverse ? "newest-first" : "oldest-first", best, (double)(mb << 20) / best / 1e9);
    }
  }
  return 0;
}
