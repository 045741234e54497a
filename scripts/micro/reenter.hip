#include <hip/hip_runtime.h>
#include <cstdio>
struct P { int* out; int n; unsigned* q; };
// jump back to the first instruction of kernel `SYM` with the state the hardware hands a fresh workgroup:
// s[0:1] = kernarg segment, s2 = workgroup id x, v0 = workitem id x, exec = all lanes
#if defined(__HIP_DEVICE_COMPILE__)
#define REENTER(SYM, BLK_, TID_)                                                                  \
  do {                                                                                            \
    const void* ka_ = __builtin_amdgcn_kernarg_segment_ptr();                                     \
    asm volatile(                                                                                 \
        "s_mov_b64 s[92:93], %[ka]\n\t"                                                           \
        "s_mov_b32 s94, %[blk]\n\t"                                                              \
        "v_mov_b32 v0, %[tid]\n\t"                                                                \
        "s_getpc_b64 s[90:91]\n"                                                                  \
        "1:\n\t"                                                                                  \
        "s_add_u32 s90, s90, " #SYM "-1b\n\t"                                                     \
        "s_addc_u32 s91, s91, -1\n\t"                                                             \
        "s_mov_b64 s[0:1], s[92:93]\n\t"                                                          \
        "s_mov_b32 s2, s94\n\t"                                                                  \
        "s_mov_b64 exec, -1\n\t"                                                                  \
        "s_setpc_b64 s[90:91]\n\t" ::[ka] "s"(ka_),                                               \
        [blk] "s"(BLK_), [tid] "v"(TID_)                                                          \
        : "s90", "s91", "s92", "s93", "s94", "s0", "s1", "s2", "v0", "memory", "scc");           \
    __builtin_unreachable();                                                                      \
  } while (0)
#else
#define REENTER(SYM, BLK_, TID_) ((void)0)
#endif

template <int K>
__device__ __forceinline__ void body(P p) {
  __shared__ unsigned nxt;
  const unsigned b = blockIdx.x, t = threadIdx.x;
  atomicAdd(p.out + b, 1 + (t == 3 ? K : 0));
  if (t == 0) nxt = atomicAdd(p.q, 1u) + gridDim.x;
  __syncthreads();
  const unsigned n = __builtin_amdgcn_readfirstlane(nxt);
  if (n < (unsigned)p.n) REENTER(rpsf_test_kern, n, t);
}
extern "C" __global__ __launch_bounds__(256) void rpsf_test_kern(P p) { body<0>(p); }
int main() {
  int n = 100000, *d; unsigned* q;
  (void)hipMalloc(&d, n * 4); (void)hipMemset(d, 0, n * 4); (void)hipMalloc(&q, 4); (void)hipMemset(q, 0, 4);
  rpsf_test_kern<<<512, 256>>>(P{d, n, q});
  int* h = new int[n]; (void)hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < n; ++i) bad += h[i] != 256;
  printf("bad=%d err=%s\n", bad, hipGetErrorString(hipGetLastError()));
}
