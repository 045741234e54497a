// Microbenchmark (development aid): how fast are repeated stores / read-modify-writes to an image-sized
// buffer (67 MB, fits the 256 MB Infinity Cache) compared with streaming stores to a 4x larger one?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// mode 0: plain store, 1: nt store, 2: plain RMW, 3: sc1 (agent-scope relaxed atomic) load+store RMW on dwords, 4: read only
template <int MODE>
__global__ void k(float* buf, size_t n4, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  f4* p = reinterpret_cast<f4*>(buf);
  float acc = 0;
  for (; i < n4; i += stride) {
    if (MODE == 0) p[i] = f4{v, v, v, v};
    if (MODE == 1) __builtin_nontemporal_store(f4{v, v, v, v}, p + i);
    if (MODE == 2) { f4 a = p[i]; a += v; p[i] = a; }
    if (MODE == 3) {
      float* q = buf + 4 * i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = __hip_atomic_load(q + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + j, a + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (MODE == 4) { f4 a = p[i]; acc += a.x + a.y + a.z + a.w; }
  }
  if (MODE == 4 && acc == 1.2345f) buf[0] = acc;
}

template <int MODE>
int run(const char* name, float* d, size_t bytes_region, int passes) {
  size_t n4 = bytes_region / 16;
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHK(hipEventRecord(e0));
    for (int p = 0; p < passes; ++p) k<MODE><<<2048, 256>>>(d, n4, 1.0f);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  }
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  double moved = (double)bytes_region * passes * ((MODE == 2 || MODE == 3) ? 2 : 1);
  printf("%-28s region %4zu MB x %d passes: %.3f ms, %.2f TB/s of load+store bytes\n", name, bytes_region >> 20, passes, ms, moved / ms / 1e9);
  return 0;
}

int main() {
  float* d; size_t big = (size_t)272 << 20;
  CHK(hipMalloc(&d, big)); CHK(hipMemset(d, 0, big));
  size_t img = (size_t)64 << 20;
  run<0>("plain store, streaming", d, big, 1);
  run<1>("nt store, streaming", d, big, 1);
  run<0>("plain store, 64MB x4", d, img, 4);
  run<1>("nt store, 64MB x4", d, img, 4);
  run<2>("plain RMW, 64MB x4", d, img, 4);
  run<3>("sc1 dword RMW, 64MB x4", d, img, 4);
  run<2>("plain RMW, streaming 272MB", d, big, 1);
  run<4>("read, streaming 272MB", d, big, 1);
  run<4>("read, 64MB x4", d, img, 4);
  run<4>("read, 16MB x16", d, (size_t)16 << 20, 16);
  return 0;
}
