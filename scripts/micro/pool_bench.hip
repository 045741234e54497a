// Development micro-benchmark of the host worker pool (csrc/rpsf_hostpipe.hpp) doing the streamed path's staging job - one 2048^2 frame in
// (pageable -> pinned) and one out (pinned -> pageable, float32 and widened to float64) per job - with the caller's frames first-touched on a chosen
// NUMA node and both PCIe directions kept busy, as in rpsf_apply_frames_host.
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -I../../regularizepsf_amd/csrc -o bin/pool_bench pool_bench.hip -lpthread
#include "rpsf_hostpipe.hpp"

#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <fstream>
#include <string>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int node_of(void* p) {
  int status = -1;
  void* page = (void*)((uintptr_t)p & ~(uintptr_t)4095);
  syscall(SYS_move_pages, 0, 1UL, &page, nullptr, &status, 0);
  return status;
}
static int first_cpu_of(int node) {
  std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
  int a = 0;
  f >> a;
  return a;
}

int main() {
  using namespace rpsf_host;
  HostPool& pool = HostPool::get(0);
  std::printf("pool width %d, workers on node %d\n", pool.width(), pool.numa_node());
  const size_t count = (size_t)2048 * 2048, bytes = count * 4;
  const int RING = 24;
  float *pin_in = nullptr, *pin_out = nullptr, *pin_dma0 = nullptr, *pin_dma1 = nullptr;
  hipHostMalloc((void**)&pin_in, bytes, hipHostMallocDefault), hipHostMalloc((void**)&pin_out, bytes, hipHostMallocDefault);
  hipHostMalloc((void**)&pin_dma0, bytes, hipHostMallocDefault), hipHostMalloc((void**)&pin_dma1, bytes, hipHostMallocDefault);
  std::memset(pin_in, 0, bytes), std::memset(pin_out, 0, bytes), std::memset(pin_dma0, 0, bytes), std::memset(pin_dma1, 0, bytes);
  void *d0 = nullptr, *d1 = nullptr;
  hipMalloc(&d0, bytes), hipMalloc(&d1, bytes);
  hipStream_t s0, s1;
  hipStreamCreateWithFlags(&s0, hipStreamNonBlocking), hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  for (int parts : {16, 64}) {
    pool.run(parts, [&](int) {});
    double t0 = now_ms();
    for (int i = 0; i < 2000; ++i) pool.run(parts, [&](int) {});
    std::printf("empty job, %d parts: %.2f us per run\n", parts, (now_ms() - t0) / 2000 * 1e3);
  }
  for (int dma_on = 0; dma_on < 2; ++dma_on) {
    std::atomic<bool> stop{false};
    std::thread dma;
    if (dma_on) dma = std::thread([&] {
        while (!stop.load()) {
          hipMemcpyAsync(d0, pin_dma0, bytes, hipMemcpyHostToDevice, s0);
          hipMemcpyAsync(pin_dma1, d1, bytes, hipMemcpyDeviceToHost, s1);
          hipStreamSynchronize(s0), hipStreamSynchronize(s1);
        }
      });
    for (int U = 0; U < 2; ++U) {
      std::vector<float*> ring(RING);
      std::vector<double*> ring64(RING);
      std::thread toucher([&] {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(first_cpu_of(U) + 5, &set);
        sched_setaffinity(0, sizeof(set), &set);
        for (auto& p : ring) p = (float*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0), std::memset(p, 1, bytes);
        for (auto& p : ring64) p = (double*)mmap(nullptr, 2 * bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0), std::memset(p, 1, 2 * bytes);
      });
      toucher.join();
      for (int parts : {16, 64}) {
        for (int wide = 0; wide < 2; ++wide) {
          double best = 1e30, sum = 0;
          const int iters = 48;
          for (int it = 0; it < iters; ++it) {
            const float* src = ring[it % RING];
            void* dst = wide ? (void*)ring64[(it + 7) % RING] : (void*)ring[(it + 7) % RING];
            double t0 = now_ms();
            pool.run(parts, [&](int t) {
              size_t a, b;
              split_range(0, count, t, parts, a, b);
              narrow_or_copy(pin_in, src, false, a, b);
              widen_or_copy(dst, wide != 0, pin_out, a, b);
            });
            const double ms = now_ms() - t0;
            best = std::min(best, ms), sum += ms;
          }
          std::printf("dma %d, user frames on node %d (checked %d), %2d parts, out %s: staging job of one 2048^2 frame each way: mean %.3f ms, best %.3f ms\n", dma_on, U,
                      node_of(ring[0]), parts, wide ? "float64" : "float32", sum / iters, best);
          std::fflush(stdout);
        }
      }
      for (auto p : ring) munmap(p, bytes);
      for (auto p : ring64) munmap(p, 2 * bytes);
    }
    stop.store(true);
    if (dma_on) dma.join();
  }
  return 0;
}
