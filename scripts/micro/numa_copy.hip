// Development micro-benchmark: staging copies on a two-socket host - where should the source, the destination and the copying threads sit?
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -o bin/numa_copy numa_copy.hip -lpthread && bin/numa_copy
// A pageable "user" frame ring is first-touched on node U, the copying threads run on node T (spread over its cores), the pinned staging buffer is
// where hipHostMalloc puts it (reported through move_pages).  Directions: in = user -> pinned (streaming stores), out = pinned -> user.
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

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
static int node_of(void* p) {
  int status = -1;
  void* page = (void*)((uintptr_t)p & ~(uintptr_t)4095);
  syscall(SYS_move_pages, 0, 1UL, &page, nullptr, &status, 0);
  return status;
}
static std::vector<int> node_cores(int node) {
  std::vector<int> cpus;
  std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
  std::string s;
  std::getline(f, s);
  int a = 0, b = 0;
  if (std::sscanf(s.c_str(), "%d-%d", &a, &b) == 2)
    for (int c = a; c <= b; ++c) cpus.push_back(c);
  return cpus;
}
static void pin_to(int cpu) {
  cpu_set_t set;
  CPU_ZERO(&set);
  CPU_SET(cpu, &set);
  sched_setaffinity(0, sizeof(set), &set);
}
template <class F>
static double run_threads(int T, int node, int iters, F&& fn) {
  std::vector<int> cores = node_cores(node);
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  std::vector<std::thread> pool;
  for (int t = 0; t < T; ++t)
    pool.emplace_back([&, t] {
      pin_to(cores[(size_t)t * cores.size() / T]);
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
  const size_t count = (size_t)4096 * 4096, bytes = count * 4;
  const int RING = 10;
  float *pin_in = nullptr, *pin_out = nullptr, *pin_dma0 = nullptr, *pin_dma1 = nullptr;
  hipHostMalloc((void**)&pin_in, bytes, hipHostMallocDefault);
  hipHostMalloc((void**)&pin_out, bytes, hipHostMallocDefault);
  hipHostMalloc((void**)&pin_dma0, bytes, hipHostMallocDefault);
  hipHostMalloc((void**)&pin_dma1, bytes, hipHostMallocDefault);
  std::memset(pin_in, 0, bytes), std::memset(pin_out, 0, bytes), std::memset(pin_dma0, 0, bytes), std::memset(pin_dma1, 0, bytes);
  void *d0 = nullptr, *d1 = nullptr;
  hipMalloc(&d0, bytes), hipMalloc(&d1, bytes);
  int gpu_node = -1;
  {
    char bdf[64] = {};
    hipDeviceGetPCIBusId(bdf, sizeof(bdf), 0);
    for (char* c = bdf; *c; ++c) *c = (char)std::tolower(*c);
    std::ifstream f(std::string("/sys/bus/pci/devices/") + bdf + "/numa_node");
    if (f) f >> gpu_node;
  }
  std::printf("GPU numa node %d; hipHostMalloc'd staging sits on node %d / %d (in / out)\n", gpu_node, node_of(pin_in), node_of(pin_out));
  hipStream_t s0, s1;
  hipStreamCreateWithFlags(&s0, hipStreamNonBlocking), hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  std::atomic<bool> stop{false};
  std::thread dma([&] {  // both PCIe directions busy from other pinned buffers, as in the streamed path
    while (!stop.load()) {
      hipMemcpyAsync(d0, pin_dma0, bytes, hipMemcpyHostToDevice, s0);
      hipMemcpyAsync(pin_dma1, d1, bytes, hipMemcpyDeviceToHost, s1);
      hipStreamSynchronize(s0), hipStreamSynchronize(s1);
    }
  });
  for (int U = 0; U < 2; ++U) {
    std::vector<float*> ring(RING);
    std::vector<double*> ring64(RING / 2);
    std::thread toucher([&] {  // first touch on node U
      pin_to(node_cores(U)[3]);
      for (auto& p : ring) {
        p = (float*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        std::memset(p, 1, bytes);
      }
      for (auto& p : ring64) {
        p = (double*)mmap(nullptr, 2 * bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        std::memset(p, 1, 2 * bytes);
      }
    });
    toucher.join();
    std::printf("user frames first-touched on node %d (checked: %d)\n", U, node_of(ring[0]));
    auto split = [&](int t, int T, size_t& a, size_t& b) {
      size_t span = ((count + T - 1) / T + 15) & ~(size_t)15;
      a = std::min(count, t * span), b = std::min(count, a + span);
    };
    for (int Tn = 0; Tn < 2; ++Tn)
      for (int T : {8, 16, 32}) {
        const int iters = 20;
        double in_ms = run_threads(T, Tn, iters, [&](int t, int TT, int it) {
          size_t a, b; split(t, TT, a, b);
          const float* src = ring[it % RING];
          for (size_t i = a; i < b; ++i) __builtin_nontemporal_store(src[i], pin_in + i);
        });
        double out_ms = run_threads(T, Tn, iters, [&](int t, int TT, int it) {
          size_t a, b; split(t, TT, a, b);
          float* dst = ring[it % RING];
          for (size_t i = a; i < b; ++i) __builtin_nontemporal_store(pin_out[i], dst + i);
        });
        double wide_ms = run_threads(T, Tn, iters, [&](int t, int TT, int it) {
          size_t a, b; split(t, TT, a, b);
          double* dst = ring64[it % (RING / 2)];
          for (size_t i = a; i < b; ++i) __builtin_nontemporal_store((double)pin_out[i], dst + i);
        });
        std::printf("user on node %d, threads on node %d x %2d | 67 MB frame: in %.3f ms (%.0f GB/s), out %.3f ms, out widened to f64 %.3f ms   [DMA busy both ways]\n",
                    U, Tn, T, in_ms, bytes / in_ms / 1e6, out_ms, wide_ms);
        std::fflush(stdout);
      }
    for (auto p : ring) munmap(p, bytes);
    for (auto p : ring64) munmap(p, 2 * bytes);
  }
  stop.store(true);
  dma.join();
  return 0;
}
