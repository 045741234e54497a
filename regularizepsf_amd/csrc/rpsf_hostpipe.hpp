// rpsf_hostpipe.hpp - host side of the host-array entry points (rpsf_apply, rpsf_apply_host, rpsf_apply_frames_host ...):
// a persistent pool of worker threads for the dtype conversions and staging copies, and the pinned / device staging slots
// and streams of the three-stream pipeline (H2D of group f + 1 || patch launch of group f || D2H + widening of group f - 1).
// What a caller of the reference writes is `[transform.apply(image) for image in images]` (regularizepsf/transform.py:85-177
// called in a loop): every frame crosses PCIe twice, which costs several times the kernel, so the copies of neighbouring
// frames have to overlap each other and the kernel.  Host-only code, included by rpsf.hip.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>

#include <hip/hip_runtime.h>

namespace rpsf_host {

// ------------------------------------------------------------------------------------------------
// Worker pool: created at the first host-array call of the process, never destroyed (no entry point creates a thread per call).
// run(parts, fn) executes fn(0) ... fn(parts - 1) on the workers and on the calling thread and returns when all are done.
//   * One job at a time (callers on several threads - distinct plans - take turns per job: a job is a few hundred microseconds).
//   * No mutex on the path of a job: it is published through one atomic pointer, parts are claimed with fetch_add, workers that
//     idle poll an epoch counter for 400 us before they go to sleep on a condition variable, and the caller polls for the last
//     part.  (A first version handed jobs out under a std::mutex: sixteen threads convoyed on it, 60 us per call, 1.35 ms to stage a
//     67 MB frame in 16 chunks against 0.25 ms now - profiles/r05d, r05e.)
//   * Placement: a staging copy streams between the caller's pageable array and pinned memory near the GPU; measured on the
//     2 x EPYC 9575F hosts of the MI355X boxes (scripts/micro/host_copy.hip, profiles/r05c_host_copy.log): 16 threads spread over the
//     CCDs of the GPU's NUMA node move a 67 MB frame in 0.20-0.25 ms (270-330 GB/s), the same threads left to the scheduler in
//     0.33-0.52 ms, on the other socket in 0.65 ms.  The workers are therefore pinned, evenly spaced over the physical cores of the
//     node of the first device a host-array call is made for (RPSF_HOST_AFFINITY=0: leave them to the scheduler), inside the
//     affinity mask the process was started with.
// RPSF_HOST_THREADS overrides the width (default: min(16, hardware threads)).
// ------------------------------------------------------------------------------------------------
class HostPool {
 public:
  static HostPool& get(int device = 0) {
    static HostPool* pool = new HostPool(device);  // leaked on purpose: the workers must not be joined from a static destructor
    return *pool;
  }
  int width() const { return std::max(1, n_workers_); }
  int numa_node() const { return node_; }

  template <class F>
  void run(int parts, F&& fn) {
    run(parts, fn, [] {});
  }
  // ... with something for the calling thread to do while the workers are at it (the streamed path enqueues the previous group's copies and launch)
  // `every`: only every n-th worker takes parts (the workers are spaced evenly over the node's cores, two per CCD at the default width: every = 2
  // is one per CCD - the same set of cores every time, where "the first 8 of 16 to arrive" would be a different set per call)
  template <class F, class M>
  void run(int parts, F&& fn, M&& meanwhile, int every = 1) {
    if (parts <= 0) {
      meanwhile();
      return;
    }
    if (parts == 1 || n_workers_ == 0) {
      meanwhile();
      for (int i = 0; i < parts; ++i) fn(i);
      return;
    }
    std::lock_guard<std::mutex> one_job(submit_);
    Job job;
    job.call = [](void* ctx, int i) { (*static_cast<std::remove_reference_t<F>*>(ctx))(i); };
    job.ctx = &fn, job.parts = parts, job.every = std::max(1, std::min(every, std::max(1, n_workers_)));
    current_.store(&job);
    epoch_.fetch_add(1);
    if (sleepers_.load() > 0) {
      std::lock_guard<std::mutex> lock(sleep_);
      wake_.notify_all();
    }
    meanwhile();
    // The caller does not take parts itself: it is wherever the application's thread happens to run - on the two-socket hosts often
    // the socket away from the GPU, where one part takes several times as long as on a worker and the whole job waits for it
    // (profiles/r05e: 53 us per 4 MiB chunk with the caller working, against 20 us without).
    const auto t0 = std::chrono::steady_clock::now();
    for (int spin = 0; job.done.load(std::memory_order_acquire) != parts; ++spin) {
      __builtin_ia32_pause();
      if ((spin & 4095) == 4095 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) std::this_thread::yield();
    }
    current_.store(nullptr);
    while (inside_.load() != 0) __builtin_ia32_pause();  // nobody still looks at `job` (it lives on this stack frame)
  }

 private:
  struct Job {
    void (*call)(void*, int) = nullptr;
    void* ctx = nullptr;
    int parts = 0, every = 1;
    alignas(64) std::atomic<int> next{0};
    alignas(64) std::atomic<int> done{0};
  };

  static std::vector<int> node_cores(int node) {  // physical cores of a NUMA node: the first range of its cpulist ("0-63,128-191")
    std::vector<int> cpus;
    if (node < 0) return cpus;
    char path[96], text[256] = {};
    std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    if (FILE* f = std::fopen(path, "r")) {
      if (std::fgets(text, sizeof(text), f)) {
        int a = -1, b = -1;
        const int got = std::sscanf(text, "%d-%d", &a, &b);
        if (got == 1) b = a;
        for (int c = a; got >= 1 && c <= b; ++c) cpus.push_back(c);
      }
      std::fclose(f);
    }
    return cpus;
  }

  explicit HostPool(int device) {
    unsigned n = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = std::getenv("RPSF_HOST_THREADS")) n = (unsigned)std::max(1, std::min(256, std::atoi(e)));
    n_workers_ = n > 1 ? (int)n : 0;  // (one thread: the caller does the work itself)
    std::vector<int> cores;
    const char* aff = std::getenv("RPSF_HOST_AFFINITY");
    if (!(aff && std::atoi(aff) == 0)) {
      char bdf[64] = {};
      if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), device) == hipSuccess) {
        for (char* c = bdf; *c; ++c) *c = (char)std::tolower(*c);
        char path[128];
        std::snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
        if (FILE* f = std::fopen(path, "r")) {
          if (std::fscanf(f, "%d", &node_) != 1) node_ = -1;
          std::fclose(f);
        }
      }
      cpu_set_t allowed;
      CPU_ZERO(&allowed);
      if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
        for (int c : node_cores(node_))
          if (CPU_ISSET(c, &allowed)) cores.push_back(c);
      if ((int)cores.size() < n_workers_) cores.clear();  // a narrower cpuset than the pool: leave it to the scheduler
    }
    for (int i = 0; i < n_workers_; ++i) {
      const int cpu = cores.empty() ? -1 : cores[(size_t)i * cores.size() / n_workers_];
      std::thread([this, cpu, i] {
        if (cpu >= 0) {
          cpu_set_t set;
          CPU_ZERO(&set);
          CPU_SET(cpu, &set);
          (void)sched_setaffinity(0, sizeof(set), &set);
        }
        worker(i);
      }).detach();
    }
  }

  void worker(int index) {
    unsigned seen = 0;
    for (;;) {
      inside_.fetch_add(1);
      if (Job* job = current_.load())
        if (index % job->every == 0) work_on(*job);
      inside_.fetch_sub(1);
      // wait for the next job: it usually follows within a few hundred microseconds (chunk after chunk of a frame, group after group of a batch)
      const auto t0 = std::chrono::steady_clock::now();
      bool idle = false;
      for (int spin = 0; epoch_.load(std::memory_order_acquire) == seen && !idle; ++spin) {
        __builtin_ia32_pause();
        if ((spin & 255) == 255) idle = std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400);
      }
      if (idle) {
        std::unique_lock<std::mutex> lock(sleep_);
        sleepers_.fetch_add(1);
        wake_.wait(lock, [&] { return epoch_.load() != seen; });
        sleepers_.fetch_sub(1);
      }
      seen = epoch_.load();
    }
  }

  static void work_on(Job& job) {
    for (;;) {
      const int i = job.next.fetch_add(1, std::memory_order_relaxed);
      if (i >= job.parts) return;
      job.call(job.ctx, i);
      job.done.fetch_add(1, std::memory_order_acq_rel);
    }
  }

  std::mutex submit_, sleep_;
  std::condition_variable wake_;
  alignas(64) std::atomic<Job*> current_{nullptr};
  alignas(64) std::atomic<unsigned> epoch_{0};
  alignas(64) std::atomic<int> inside_{0};
  std::atomic<int> sleepers_{0};
  int n_workers_ = 0, node_ = -1;
};

// ------------------------------------------------------------------------------------------------
// Conversions between the caller's arrays and the float32 staging (transform.py:117 `astype(float)` on the way in - the
// kernels compute in float32, so a float64 image is narrowed before it crosses PCIe - and :174-177, a float64 result, on
// the way out).  Streaming stores: neither side is read again by the core that wrote it.
// ------------------------------------------------------------------------------------------------
inline void narrow_or_copy(float* dst, const void* src, bool src_f64, size_t a, size_t b) {
  if (b <= a) return;
  // (streaming stores on this side too: ordinary stores - hoping the copy engine would find the staged lines in the L3 - measured 5-15 %
  // slower end to end, profiles/r05k_stage_nt_ab.log; and they beat memcpy by 1.2-1.4x in isolation, scripts/micro/host_copy.hip)
  if (!src_f64) {
    const float* s = static_cast<const float*>(src);
    for (size_t i = a; i < b; ++i) __builtin_nontemporal_store(s[i], dst + i);
  } else {
    const double* s = static_cast<const double*>(src);
    for (size_t i = a; i < b; ++i) __builtin_nontemporal_store((float)s[i], dst + i);
  }
  __builtin_ia32_sfence();  // the DMA engine reads the staging next
}
inline void widen_or_copy(void* dst, bool dst_f64, const float* src, size_t a, size_t b) {
  if (b <= a) return;
  if (!dst_f64) {
    float* d = static_cast<float*>(dst);
    for (size_t i = a; i < b; ++i) __builtin_nontemporal_store(src[i], d + i);
  } else {
    double* d = static_cast<double*>(dst);
    for (size_t i = a; i < b; ++i) __builtin_nontemporal_store((double)src[i], d + i);
  }
  __builtin_ia32_sfence();
}
// thread t of T takes this part of [lo, hi) (multiples of 16 elements: whole cache lines of the float32 side)
inline void split_range(size_t lo, size_t hi, int t, int T, size_t& a, size_t& b) {
  const size_t span = ((hi - lo + T - 1) / T + 15) & ~(size_t)15;
  a = std::min(hi, lo + (size_t)t * span), b = std::min(hi, a + span);
}

// ------------------------------------------------------------------------------------------------
// Staging slots of one plan: `depth` groups of frames in flight, each with pinned host and device buffers for both
// directions, one stream per direction beside the plan's own (compute) stream, and the events that chain them.
// Owned by the plan, grown on demand, freed with it; no entry point allocates per call once the sizes have been seen.
// ------------------------------------------------------------------------------------------------
struct HostPipe {
  static constexpr int MAX_DEPTH = 4, MAX_CHUNKS = 16, MAX_BANDS = 16, MAX_PIECES = 2 * MAX_BANDS + 8;
  int depth = 0;
  size_t slot_floats = 0;
  hipStream_t st_in = nullptr, st_out = nullptr;
  float* h_in[MAX_DEPTH] = {};
  float* h_out[MAX_DEPTH] = {};
  float* d_in[MAX_DEPTH] = {};
  float* d_out[MAX_DEPTH] = {};
  hipEvent_t ev_in[MAX_DEPTH] = {}, ev_k[MAX_DEPTH] = {}, ev_out[MAX_DEPTH] = {};
  hipEvent_t ev_chunk[MAX_PIECES] = {};  // download pieces of one frame (>= MAX_CHUNKS)
  hipEvent_t ev_band_in[MAX_BANDS] = {}, ev_band_k[MAX_BANDS] = {};  // a single frame cut into row bands (host_one_frame)

  void release_buffers() {
    for (int s = 0; s < MAX_DEPTH; ++s) {
      (void)hipHostFree(h_in[s]);
      (void)hipHostFree(h_out[s]);
      (void)hipFree(d_in[s]);
      (void)hipFree(d_out[s]);
      h_in[s] = h_out[s] = d_in[s] = d_out[s] = nullptr;
    }
    depth = 0, slot_floats = 0;
  }
  void destroy() {
    if (st_in) (void)hipStreamSynchronize(st_in);
    if (st_out) (void)hipStreamSynchronize(st_out);
    release_buffers();
    for (auto* evs : {ev_in, ev_k, ev_out})
      for (int s = 0; s < MAX_DEPTH; ++s)
        if (evs[s]) (void)hipEventDestroy(evs[s]);
    for (auto& e : ev_chunk)
      if (e) (void)hipEventDestroy(e);
    for (auto* evs : {ev_band_in, ev_band_k})
      for (int s = 0; s < MAX_BANDS; ++s)
        if (evs[s]) (void)hipEventDestroy(evs[s]);
    if (st_in) (void)hipStreamDestroy(st_in);
    if (st_out) (void)hipStreamDestroy(st_out);
  }
};

}  // namespace rpsf_host
