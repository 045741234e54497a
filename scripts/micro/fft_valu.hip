// VALU lab (development aid): the arithmetic of one 256-pixel patch - window, the three forward stages, the frequency step's pair words,
// the inverse - exactly as patch_body2 calls the per-thread phase functions, in a loop on registers: no global memory, no LDS exchanges, no
// barriers.  Time per pass and the static instruction mix of the loop body say what the instruction stream itself costs on a SIMD with two
// waves (the product's occupancy) - the quantity the patch kernel's period is mostly made of (profiles/r04a: arithmetic alone 0.136 of 0.189 ms).
//   hipcc --offload-arch=gfx950 -std=c++20 -O3 -fno-slp-vectorize [-DLAB_...] -o fft_valu fft_valu.hip && ./fft_valu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../../regularizepsf_amd/csrc/rpsf_core.hpp"
#include "../../regularizepsf_amd/csrc/rpsf_core2.hpp"
using namespace rpsf;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using C = Cfg256v2;

extern "C" __global__ __launch_bounds__(512, 2) void lab(float* out, const cf* tw_g, const float* win_g, const uint16_t* tab, int iters) {
  __shared__ cf tw[C::N];
  __shared__ float win[C::N];
  const int t = threadIdx.x;
  if (t < C::N) tw[t] = tw_g[t], win[t] = win_g[t];
  __syncthreads();
  GroupIds<C> gids;
  gids.load(tab, t);
  cf v[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) v[j] = cf{1.0f + 0.001f * (float)(t + j), 0.5f - 0.002f * (float)(t - j)};
  cf k[2 * C::KCH];
#pragma unroll
  for (int j = 0; j < 2 * C::KCH; ++j) k[j] = cf{0.5f + 0.01f * j, 0.25f - 0.001f * t};
  for (int it = 0; it < iters; ++it) {
    window_patch2<C>(t, v, win);
    stage1h<C, 0, false>(t, v, tw);
    stage1h<C, 1, false>(t, v, tw);
    stage2h<C, 0, false>(t, v, tw);
    stage2h<C, 1, false>(t, v, tw);
    stage3_rows<C, false, 0, 0>(t, gids, v);
    stage3_rows<C, false, 1, 0>(t, gids, v);
    stage3_cols<C, false, 0>(v);
    StaticFor<0, C::NCHUNK>::run([&]<int CI>() RPSF_AI {
      pair_words<C, 0, CI * C::KCH, C::KCH>(gids, v, k, tw);
      asm volatile("" : "+v"(k[0].x), "+v"(k[1].y), "+v"(k[5].x), "+v"(k[9].y));  // (the next chunk's words are other values)
    });
    stage3_cols<C, true, 0>(v);
    stage3_rows<C, true, 0, 0>(t, gids, v);
    stage3_rows<C, true, 1, 0>(t, gids, v);
    stage2h<C, 0, true>(t, v, tw);
    stage2h<C, 1, true>(t, v, tw);
    stage1h<C, 0, true>(t, v, tw);
    stage1h<C, 1, true>(t, v, tw);
    window_patch2<C>(t, v, win);
    // keep the magnitudes bounded from pass to pass (one multiply per value: 128 instructions of ~7000)
#pragma unroll
    for (int j = 0; j < 64; ++j) v[j] = v[j] * 1e-3f;
    asm volatile("" ::: "memory");
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 64; ++j) s += v[j].x + v[j].y;
  if (s == 1234.5f) out[t] = s;
}

int main() {
  std::vector<cf> tw(C::N);
  std::vector<float> win(C::N);
  for (int i = 0; i < C::N; ++i) {
    const double a = -2.0 * M_PI * i / C::N;
    tw[i] = cf{(float)cos(a), (float)sin(a)};
    win[i] = (float)sin((i + 0.5) * (M_PI / C::N));
  }
  std::vector<uint16_t> tab((size_t)C::T * C::P);
  build_slot_table2<C>(tab.data());
  cf* d_tw; float *d_win, *d_out; uint16_t* d_tab;
  CHK(hipMalloc(&d_tw, tw.size() * sizeof(cf))); CHK(hipMalloc(&d_win, win.size() * 4)); CHK(hipMalloc(&d_out, 4096)); CHK(hipMalloc(&d_tab, tab.size() * 2));
  CHK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_win, win.data(), win.size() * 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_tab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
  const int iters = 200;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    lab<<<256, 512>>>(d_out, d_tw, d_win, d_tab, 20);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    lab<<<256, 512>>>(d_out, d_tw, d_win, d_tab, iters);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%.3f ms for %d passes on every CU: %.2f us per patch-equivalent of arithmetic (2 waves per SIMD)\n", ms, iters, ms * 1e3 / iters);
  }
  return 0;
}
