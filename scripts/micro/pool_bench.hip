// Development micro-benchmark of the host worker pool (csrc/rpsf_hostpipe.hpp): cost of an empty job, and the staging copy of a
// 67 MB frame as one job against 16 jobs of 4 MiB (what rpsf_apply_host does so that the copies overlap the PCIe transfers).
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -I../../regularizepsf_amd/csrc -o bin/pool_bench pool_bench.hip -lpthread
#include "rpsf_hostpipe.hpp"

#include <algorithm>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  using namespace rpsf_host;
  HostPool& pool = HostPool::get(0);
  std::printf("pool width %d, node %d\n", pool.width(), pool.numa_node());
  const size_t count = (size_t)4096 * 4096, bytes = count * 4;
  const int RING = 12;
  float* pin = nullptr;
  hipHostMalloc((void**)&pin, bytes, hipHostMallocDefault);
  std::memset(pin, 0, bytes);
  std::vector<float*> ring(RING);
  for (auto& p : ring) {
    p = (float*)std::aligned_alloc(4096, bytes);
    std::memset(p, 1, bytes);
  }
  for (int parts : {16, 32, 64}) {
    std::atomic<int> sink{0};
    pool.run(parts, [&](int) { sink.fetch_add(1, std::memory_order_relaxed); });
    double t0 = now_ms();
    for (int i = 0; i < 2000; ++i) pool.run(parts, [&](int) {});
    std::printf("empty job, %d parts: %.2f us per run\n", parts, (now_ms() - t0) / 2000 * 1e3);
  }
  for (int rep = 0; rep < 2; ++rep)
    for (int n_chunks : {1, 4, 16}) {
      for (int per_thread : {1, 2, 4}) {
        const int parts = pool.width() * per_thread;
        const size_t per_chunk = count / n_chunks;
        double best = 1e30;
        for (int it = 0; it < 12; ++it) {
          const float* src = ring[it % RING];
          double t0 = now_ms();
          for (int c = 0; c < n_chunks; ++c)
            pool.run(parts, [&](int t) {
              size_t a, b;
              split_range(c * per_chunk, (c + 1) * per_chunk, t, parts, a, b);
              narrow_or_copy(pin, src, false, a, b);
            });
          best = std::min(best, now_ms() - t0);
        }
        std::printf("67 MB frame in %2d chunk job(s) of %3d parts: %.3f ms (%.0f GB/s)\n", n_chunks, parts, best, bytes / best / 1e6);
      }
    }
  return 0;
}
