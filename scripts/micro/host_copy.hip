// Development micro-benchmark: how fast can a frame get between the caller's pageable arrays and the GPU on this box?
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -o bin/host_copy host_copy.hip -lpthread && bin/host_copy
// (a) staging copies by host threads, working set far larger than the L3 (a ring of distinct pageable frames), thread counts and
//     placements (unpinned / spread over the CCDs of one NUMA node), with and without PCIe DMA running beside them;
// (b) the zero-copy alternative: hipHostRegister of the caller's frame, DMA straight from / to it, hipHostUnregister -
//     fresh buffers every time, from one and from several threads.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static std::vector<int> node_cpus(int node) {
  std::vector<int> cpus;
  std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
  std::string s;
  std::getline(f, s);
  size_t i = 0;
  while (i < s.size()) {
    size_t j = s.find(',', i);
    if (j == std::string::npos) j = s.size();
    std::string part = s.substr(i, j - i);
    size_t d = part.find('-');
    int a = std::atoi(part.c_str()), b = d == std::string::npos ? a : std::atoi(part.c_str() + d + 1);
    for (int c = a; c <= b; ++c) cpus.push_back(c);
    i = j + 1;
  }
  return cpus;
}

// placement: -1 unpinned; node n: thread t on physical core (t * stride) % cores of that node, stride 8 = one per CCD first
template <class F>
static double run_threads(int T, int node, int iters, F&& fn) {
  std::vector<int> cpus = node >= 0 ? node_cpus(node) : std::vector<int>();
  if (!cpus.empty()) cpus.resize(cpus.size() / 2);  // first half of the list: one hardware thread per core
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  std::vector<std::thread> pool;
  for (int t = 0; t < T; ++t)
    pool.emplace_back([&, t] {
      if (!cpus.empty()) {
        const int cores = (int)cpus.size(), ccds = cores / 8;
        const int core = (t % ccds) * 8 + (t / ccds) % 8;
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(cpus[core % cores], &set);
        sched_setaffinity(0, sizeof(set), &set);
      }
      ready.fetch_add(1);
      while (!go.load()) {}
      for (int it = 0; it < iters; ++it) fn(t, T, it);
    });
  while (ready.load() < T) {}
  const double t0 = now_ms();
  go.store(true);
  for (auto& th : pool) th.join();
  return (now_ms() - t0) / iters;
}

int main() {
  const size_t count = (size_t)4096 * 4096, bytes = count * 4;  // one 4096^2 float32 frame
  const int RING = 12;                                           // 805 MB of pageable frames: nothing stays in the L3
  float *pin_in = nullptr, *pin_out = nullptr, *pin_dma0 = nullptr, *pin_dma1 = nullptr;
  hipHostMalloc((void**)&pin_in, bytes, hipHostMallocDefault);
  hipHostMalloc((void**)&pin_out, bytes, hipHostMallocDefault);
  hipHostMalloc((void**)&pin_dma0, bytes, hipHostMallocDefault);
  hipHostMalloc((void**)&pin_dma1, bytes, hipHostMallocDefault);
  void *d0 = nullptr, *d1 = nullptr;
  hipMalloc(&d0, bytes);
  hipMalloc(&d1, bytes);
  std::vector<float*> ring(RING);
  for (auto& p : ring) {
    p = (float*)std::aligned_alloc(4096, bytes);
    for (size_t i = 0; i < count; i += 1024) p[i] = 1.f;
    std::memset(p, 0, bytes);
  }
  std::memset(pin_in, 0, bytes), std::memset(pin_out, 0, bytes), std::memset(pin_dma0, 0, bytes), std::memset(pin_dma1, 0, bytes);
  int gpu_node = -1;
  {
    char bdf[64] = {};
    hipDeviceGetPCIBusId(bdf, sizeof(bdf), 0);
    std::string lower(bdf);
    for (auto& c : lower) c = (char)std::tolower(c);
    std::ifstream f("/sys/bus/pci/devices/" + lower + "/numa_node");
    if (f) f >> gpu_node;
    std::printf("GPU %s numa_node %d\n", bdf, gpu_node);
  }
  hipStream_t s0, s1;
  hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  auto split = [&](int t, int T, size_t& a, size_t& b) {
    size_t span = ((count + T - 1) / T + 15) & ~(size_t)15;
    a = std::min(count, t * span), b = std::min(count, a + span);
  };
  for (int dma = 0; dma < 2; ++dma) {
    std::atomic<bool> stop{false};
    std::thread dma_thread;
    if (dma) dma_thread = std::thread([&] {  // keeps both PCIe directions busy from other pinned buffers
        while (!stop.load()) {
          hipMemcpyAsync(d0, pin_dma0, bytes, hipMemcpyHostToDevice, s0);
          hipMemcpyAsync(pin_dma1, d1, bytes, hipMemcpyDeviceToHost, s1);
          hipStreamSynchronize(s0);
          hipStreamSynchronize(s1);
        }
      });
    for (int node : {-1, 0, 1})
      for (int T : {8, 16, 32, 64}) {
        const int iters = 24;
        double in_ms = run_threads(T, node, iters, [&](int t, int TT, int it) {
          size_t a, b; split(t, TT, a, b);
          const float* src = ring[it % RING];
          for (size_t i = a; i < b; ++i) __builtin_nontemporal_store(src[i], pin_in + i);
        });
        double in_mc = run_threads(T, node, iters, [&](int t, int TT, int it) {
          size_t a, b; split(t, TT, a, b);
          std::memcpy(pin_in + a, ring[it % RING] + a, (b - a) * 4);
        });
        double out_ms = run_threads(T, node, iters, [&](int t, int TT, int it) {
          size_t a, b; split(t, TT, a, b);
          float* dst = ring[it % RING];
          for (size_t i = a; i < b; ++i) __builtin_nontemporal_store(pin_out[i], dst + i);
        });
        std::printf("dma %d node %2d threads %2d | 67 MB frame: in nt %.3f ms (%.0f GB/s), in memcpy %.3f, out nt %.3f (threads are not barriered per frame)\n",
                    dma, node, T, in_ms, bytes / in_ms / 1e6, in_mc, out_ms);
        std::fflush(stdout);
      }
    stop.store(true);
    if (dma) dma_thread.join();
  }
  // ---- zero copy: register a fresh frame, DMA from it, unregister
  for (size_t nb : {bytes / 4, bytes}) {
    for (int T : {1, 2, 4}) {
      std::vector<double> reg(T, 0.0), unreg(T, 0.0);
      const int iters = 6;
      double wall = run_threads(T, -1, iters, [&](int t, int TT, int it) {
        float* p = ring[(it * TT + t) % RING];
        double t0 = now_ms();
        hipError_t e = hipHostRegister(p, nb, hipHostRegisterDefault);
        double t1 = now_ms();
        if (e != hipSuccess) std::printf("register failed: %s\n", hipGetErrorString(e));
        hipHostUnregister(p);
        double t2 = now_ms();
        reg[t] += (t1 - t0) / iters, unreg[t] += (t2 - t1) / iters;
      });
      std::printf("hipHostRegister %.1f MB from %d thread(s): register %.3f ms, unregister %.3f ms per call, wall %.3f ms per round\n", nb / 1e6, T, reg[0], unreg[0], wall);
    }
    float* p = ring[0];
    hipHostRegister(p, nb, hipHostRegisterDefault);
    for (int r = 0; r < 2; ++r) {
      hipDeviceSynchronize();
      double t0 = now_ms();
      hipMemcpyAsync(d0, p, nb, hipMemcpyHostToDevice, s0);
      hipStreamSynchronize(s0);
      double t1 = now_ms();
      hipMemcpyAsync(p, d1, nb, hipMemcpyDeviceToHost, s1);
      hipStreamSynchronize(s1);
      double t2 = now_ms();
      std::printf("DMA on registered memory %.1f MB: h2d %.3f ms (%.1f GB/s), d2h %.3f ms (%.1f GB/s)\n", nb / 1e6, t1 - t0, nb / (t1 - t0) / 1e6, t2 - t1, nb / (t2 - t1) / 1e6);
    }
    hipHostUnregister(p);
  }
  return 0;
}
