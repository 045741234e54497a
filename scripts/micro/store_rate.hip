// Microbenchmark (development aid): how fast can one CU emit a patch-shaped store burst (256 rows x 1 KiB,
// row stride 16 KiB, 8 or 16 bytes per lane), as a function of how many CUs store at once and of the duty
// cycle (idle time between bursts, as in the patch kernel where a store phase is ~1/6 of a patch's life).
// Reports the time to *issue* the burst (the wave is blocked while the store queue is full) and to drain it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int W, bool NT>
__global__ __launch_bounds__(512) void k(float* buf, int tiles_x, int tiles_y, int rounds, int idle_ticks,
                                         unsigned long long* stamps) {
  extern __shared__ float pad[];
  const int t = threadIdx.x;
  unsigned long long issue = 0, drain = 0;
  for (int r = 0; r < rounds; ++r) {
    unsigned tile = ((unsigned)blockIdx.x * 2654435761u + (unsigned)r * 40503u) % (unsigned)(tiles_x * tiles_y * 4);
    int plane = tile / (tiles_x * tiles_y), ty = (tile / tiles_x) % tiles_y, tx = tile % tiles_x;
    float* base = buf + ((size_t)plane * tiles_y * 256 + (size_t)ty * 256) * (tiles_x * 256) + (size_t)tx * 256;
    const size_t ld = (size_t)tiles_x * 256;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (W == 2) {  // lane -> (row parity, 32 column pairs) like store_patch: a wave covers 2 rows x 256 B... here 1 KiB rows: 4 waves per row pair
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        int row = i * 4 + (t >> 7), col = (t & 127) * 2;
        f2 v = {(float)i, (float)t};
        f2* p = reinterpret_cast<f2*>(base + (size_t)row * ld + col);
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        int row = i * 8 + (t >> 6), col = (t & 63) * 4;
        f4 v = {(float)i, (float)t, 0.f, 1.f};
        f4* p = reinterpret_cast<f4*>(base + (size_t)row * ld + col);
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
      }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    issue += t1 - t0; drain += t2 - t0;
    if (idle_ticks > 0) { while (__builtin_amdgcn_s_memrealtime() - t2 < (unsigned long long)idle_ticks) __builtin_amdgcn_s_sleep(16); }
  }
  if (t == 0) { stamps[blockIdx.x * 2] = issue; stamps[blockIdx.x * 2 + 1] = drain; }
  if (pad[0] == 1.2345f) buf[0] = 1;
}

template <int W, bool NT>
int run(const char* name, float* d, int blocks, int idle_us, unsigned long long* d_st) {
  const int rounds = 16;
  CHK(hipFuncSetAttribute((const void*)k<W, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHK(hipEventRecord(e0));
    k<W, NT><<<blocks, 512, 100 * 1024>>>(d, 16, 16, rounds, idle_us * 100, d_st);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  }
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> st(blocks * 2);
  CHK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
  double is = 0, dr = 0;
  for (int b = 0; b < blocks; ++b) { is += st[2 * b]; dr += st[2 * b + 1]; }
  is = is / blocks / rounds * 0.01; dr = dr / blocks / rounds * 0.01;  // us per burst (100 MHz clock)
  printf("%-22s blocks %3d idle %3d us: issue %6.2f us, issue+drain %6.2f us per 256 KiB burst (%.0f GB/s per CU while storing), kernel %.3f ms\n",
         name, blocks, idle_us, is, dr, 262144.0 / dr / 1e3, ms);
  return 0;
}

int main() {
  float* d; unsigned long long* st;
  CHK(hipMalloc(&d, (size_t)4 * 4096 * 4096 * 4)); CHK(hipMemset(d, 0, (size_t)4 * 4096 * 4096 * 4)); CHK(hipMalloc(&st, 256 * 16));
  for (int blocks : {256, 64, 8}) {
    for (int idle : {0, 45}) {
      run<2, true>("8B nt", d, blocks, idle, st);
      run<2, false>("8B plain", d, blocks, idle, st);
      run<4, true>("16B nt", d, blocks, idle, st);
      run<4, false>("16B plain", d, blocks, idle, st);
    }
  }
  return 0;
}
