// Microbenchmark (development aid): chip-wide rate of no-return global atomics, dense 256-B wave shape,
// each address hit 4 times (like the 4x overlap-add), float vs u32 vs u64, on a 64 MB image-sized buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void k(void* buf, size_t n, int passes) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (int p = 0; p < passes; ++p)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      if (KIND == 0) unsafeAtomicAdd(reinterpret_cast<float*>(buf) + i, 1.0f);
      if (KIND == 1) atomicAdd(reinterpret_cast<unsigned*>(buf) + i, 1u);
      if (KIND == 2) atomicAdd(reinterpret_cast<unsigned long long*>(buf) + i, 1ull);
      if (KIND == 3) reinterpret_cast<float*>(buf)[i] += 1.0f;  // plain RMW for comparison
      if (KIND == 4) __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(buf) + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
template <int KIND>
int run(const char* name, void* d, size_t elems, size_t esize) {
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHK(hipEventRecord(e0));
    k<KIND><<<4096, 256>>>(d, elems, 4);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  }
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-26s %zu M elems x4: %.3f ms -> %.2f TB/s of added bytes, %.1f G adds/s\n", name, elems >> 20, ms,
         (double)elems * 4 * esize / ms / 1e9, (double)elems * 4 / ms / 1e6);
  return 0;
}
int main() {
  void* d; CHK(hipMalloc(&d, (size_t)256 << 20)); CHK(hipMemset(d, 0, (size_t)256 << 20));
  size_t px = (size_t)16 << 20;  // 4096^2 pixels
  run<0>("float atomic", d, px, 4);
  run<1>("u32 atomic (agent)", d, px, 4);
  run<4>("u32 atomic (workgroup)", d, px, 4);
  run<2>("u64 atomic", d, px, 8);
  run<3>("plain float RMW", d, px, 4);
  return 0;
}
