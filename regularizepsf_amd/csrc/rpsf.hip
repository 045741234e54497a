// rpsf.hip - host side of librpsf_hip.so: plans, launches and the C ABI of include/rpsf.h; the RCCL and
// hipFFT loaders.  The device code is in rpsf_kernels.hpp (kernels) and rpsf_core.hpp (per-thread phases).
//
// Kernels (rpsf_kernels.hpp):
//   patch_kernel<C>        K1: fused gather+pad+window -> 2-D DFT -> x folded K -> inverse DFT -> window ->
//                          overlap-add; one workgroup per patch (or several small patches per workgroup),
//                          patch resident in registers, LDS used only for the inter-stage transposes.
//                          Stands in for regularizepsf/transform.py:151-169.
//   pack_kernel<C>         folds K to K_h = (K(k)+conj K(-k))/2 and re-orders it into the per-thread
//                          streaming layout of K1 (one-time, at set_transfer).
//   build_transfer_kernel  K2: transform.py:78-82 element-wise.
//   psf_fft_kernel<C>      K3: psf.py:216-219, forward half of K1 written out as a full spectrum.
//   add_rows_kernel        K4: seam accumulate for the multi-GPU row-band split.
//   sum_planes_kernel      K5: output = sum of the four colour planes the patches of one parity class write.
//   generic_*_kernel       gather / multiply / scatter around hipFFT for patch sizes without a plan.
// Development-only build switches (never set by regularizepsf_amd/build.py): RPSF_STAMPS / RPSF_WAVE_STAMPS (per-phase
// timestamps, scripts/stamps.py) and the RPSF2_ABL_* timing ablations of rpsf_kernels2.hpp (results are wrong, timing only).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define RPSF_HOST_TU 1
#include "rpsf_device.hpp"
#include "rpsf_hostpipe.hpp"

RPSF_PLANS_V1(RPSF_DECL_V1)
RPSF_PLANS_V2(RPSF_DECL_V2)
RPSF_PLANS_V3(RPSF_DECL_V3)

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return fail(e_ == hipErrorOutOfMemory ? RPSF_E_NOMEM : RPSF_E_HIP,                                \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                                   \
  } while (0)

extern "C" const char* rpsf_last_error(void) { return g_err.c_str(); }

// Development sweeps read their knobs from the environment ONLY in builds made with -DRPSF_DEV_ENV (scripts/, never the product): a stray variable
// in a production environment must not change which kernel runs.  The shipped library reads two variables, both about the host-side thread pool
// (rpsf_hostpipe.hpp: RPSF_HOST_THREADS, RPSF_HOST_AFFINITY, documented in include/rpsf.h); everything a caller or a test may want to
// choose is a plan option (rpsf_plan_set_option).
static const char* dev_env(const char* name) {
#if defined(RPSF_DEV_ENV)
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

template <class F>
static int dispatch_v3(int N, F&& f) {
  switch (N) {
    case 64: return f.template operator()<Cfg3_64>();
    case 32: return f.template operator()<Cfg3_32>();
    case 16: return f.template operator()<Cfg3_16>();
    default: return fail(RPSF_E_UNSUPPORTED, "no third-generation plan for this patch size");
  }
}
static bool has_v3(int N) { return N == 64 || N == 32 || N == 16; }

// Python's slice arithmetic for [start, stop) over a length-n axis (a negative bound wraps once): the window rule of the saturation fill
static void py_slice(long start, long stop, long n, long* lo, long* hi) {
  if (start < 0) start = std::max(start + n, 0L);
  if (stop < 0) stop = std::max(stop + n, 0L);
  *lo = std::min(start, n);
  *hi = std::min(stop, n);
}

// Device allocation that is freed on every exit path of the entry points that make temporaries
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
  template <class T>
  T* as() const { return static_cast<T*>(p); }
  void* release() {
    void* q = p;
    p = nullptr;
    return q;
  }
};


// Host scratch of the saturation branch (rpsf_apply_host_saturated), per staging slot, kept between calls
struct SatScratch {
  std::vector<double> padded;
  std::vector<uint8_t> mask;
  struct Raw {
    int64_t idx;
    double value;
  };
  std::vector<Raw> raw;
};

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
struct rpsf_plan {
  int device = 0, N = 0, n_patches = 0;
  int32_t* d_coords = nullptr;
  uint16_t* d_tab = nullptr;
  uint32_t* d_pairtab = nullptr;
  cf* d_tw = nullptr;
  float* d_win = nullptr;
  cf* d_g = nullptr;
  cf* d_gs = nullptr;
  rpsf_host::HostPipe* pipe = nullptr;  // host-array entry points: staging slots, copy streams (rpsf_hostpipe.hpp)
  // Views: a plan over a subset of another plan's patches that shares its tables, its packed K (desc.z = the patch's index in the
  // parent) and its stream - the row bands a single large host frame is cut into so that its upload, its patches and its download
  // overlap (host_one_frame).  Owned by the parent, built for one frame shape.
  std::vector<SatScratch> sat;  // rpsf_apply_host_saturated: per staging slot, the 2N-padded float64 frame, its mask and the raw values (host scratch, kept between calls)
  rpsf_plan* parent = nullptr;
  std::vector<int32_t> k_index;           // view: patch i of this plan is patch k_index[i] of the parent
  std::vector<rpsf_plan*> bands;          // parent: its row-band views
  std::vector<int> band_rows;             // bands.size() + 1 output row boundaries
  std::vector<int> band_in_rows;          // per band: image rows [0, band_in_rows[b]) must be resident before it runs
  int bands_h = 0, bands_w = 0, bands_mode = -1, bands_want = -1;
  bool have_k = false;
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t g_elems = 0, gs_elems = 0;
  std::vector<int32_t> h_coords;
  // overlap-add strategy: on regular half-overlap lattices direct accumulation through the XCD's L2 (three-stage
  // plans) or colour planes + plane sum (the small-patch plans); float atomics for any other corner list
  int overlap_mode = 0;  // 0 auto, 1 atomics, 2 planes, 3 direct
  int stagger_us = -1, cu_count = 256;  // -1: automatic (12 us for persistent launches of 256 patches and more, else none)
  int round_capacity = 0;  // patches the chip holds at once (CUs x resident workgroups x patches per workgroup)
  bool lattice = false;
  bool direct_ok = false;  // lattice and one patch per workgroup
  bool v2 = false;         // second-generation kernel (rpsf_core2.hpp): N = 128, 256
  // Fused plane sum (second-generation kernels, one frame, aligned geometry): the workgroups behind the patches in the
  // grid sum a lattice tile as soon as all its contributors have published their (write-through) plane stores, so
  // the sum runs on the CUs the partial last round leaves idle and no second kernel is needed.
  uint32_t* d_tile_done = nullptr;   // per tile: contributors done, never reset (epoch * contributors after every apply)
  uint32_t* d_sum_order = nullptr;   // all tiles, the ones whose contributors run first first
  uint32_t done_epoch = 0;
  size_t done_frames = 1;            // frames the tile counters are sized for (batches: one set per frame)
  int done_last_frames = 1;          // frames of the last fused launch (the counters restart when the number changes)
  uint32_t* d_sum_queue = nullptr;   // position in d_sum_order, never reset
  uint32_t sum_queue_base = 0;
  bool no_fuse = false;
  int sum_first = -1;                // RPSF_SUM_FIRST override of sum_first_for(), -1 = none
  bool prefetch = false;             // persistent launches: head summing workgroups touch the image first (RPSF_PREFETCH)
  uint32_t* d_prefetch_tiles = nullptr;  // per chunk: lattice tiles in the order the chunk's patches first need them
  uint32_t prefetch_first[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int head_patches = 0;              // (measured neutral, profiles/r03i: off) persistent launches: patches a head summing workgroup computes before it sums (RPSF_HEAD_PATCHES)
  bool fuse_pays = false;            // the second-generation plans (N = 128, 256)
  // Persistent patch workgroups (patch_kernel2_256p; fused launches of the 256-pixel plan): per-XCD slot queues, never reset
  bool persist = false;
  bool k_cached = false;             // 128-pixel persistent launches take patch_kernel2_128pc (plain loads of the pair words): see rpsf_plan_create
  int reserved_cus = 0;              // persistent launches leave this many CUs without a patch workgroup (rpsf_plan_set_reserved_cus)
  uint32_t* d_xq = nullptr;          // 8 counters, one per 128-byte line
  uint32_t xq_base[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint4* d_quads = nullptr;        // per processing-order slot: quadrant words (rpsf_core.hpp, store_patch_direct)
  uint8_t* d_tile_info = nullptr;  // per lattice tile: static side mask | 16 if any patch covers it
  uint32_t* d_flags = nullptr;     // per (frame, tile)
  uint32_t* d_dyn = nullptr;       // per (frame, tile)
  uint32_t* d_chunk_xcc = nullptr; // 8 words
  size_t flag_frames = 0;
  uint32_t epoch = 0;
  int orphan_mod = 0;              // testing aid (RPSF_OPT_DEBUG_ORPHAN)
  int plane_nt_opt = -1, host_bands_opt = -1, stream_group_opt = 0, stream_depth_opt = 0;  // rpsf_plan_set_option; -1 / 0: the library decides
  int last_host_bands = 0;  // row bands the last single host frame was cut into (rpsf_plan_host_bands)
  hipEvent_t ev_busy = nullptr;    // end of the last apply (applies on different streams are serialised)
  hipStream_t last_stream = nullptr;
  bool busy_valid = false;
  int lat_r0 = 0, lat_c0 = 0, nti = 0, ntj = 0;
  uint8_t* d_cover = nullptr;
  int4* d_desc = nullptr;
  std::vector<int32_t> h_order;
  unsigned long long* d_stamps = nullptr;
  float* d_sink = nullptr;  // write-only scratch for the out-of-image pixels of rim patches
  // generic (hipFFT) plans: any patch size without a compiled kernel
  int corner_min[2] = {0, 0}, corner_max[2] = {0, 0};  // extreme patch corners (row, col)
  bool generic = false;
  cf* d_kfull = nullptr;       // the caller's K, (n, N, N) complex64, unfolded
  cf* d_fft_buf = nullptr;     // fft_chunk x N x N complex work buffer
  float* d_win_generic = nullptr;
  uint8_t* d_colour_generic = nullptr;  // colour class per patch when the corners allow one (generic_colours), else null: atomic adds
  void* fft_plan = nullptr;    // hipfftHandle for fft_chunk patches
  int fft_chunk = 0;
  // third generation (rpsf_kernels3.hpp, N <= 64 on a complete lattice of at least 2 x 2 patches): regions, job lists, packed K
  bool sweep_ok = false;
  Job3* d_jobs3 = nullptr;
  Region3* d_regions3 = nullptr;
  int n_regions3 = 0, ks3 = 0, par_j3 = 0;
  std::vector<int32_t> slot3;  // lattice cell -> transfer-kernel slot
  long slabs3 = 0, patch_slots3 = 0;
  float* d_k3 = nullptr;      // n_patches x Cfg3::K_FLOATS (a view shares its parent's)
  float* d_zero3 = nullptr;   // 16 bytes of zeros
  uint32_t* h_err3 = nullptr;  // page-locked word the sweep kernel sets when a job waited for its predecessors beyond the bound (a view uses its parent's)
  uint32_t* d_err3 = nullptr;  // ... as the device sees it
  bool owns_err3 = false;
  unsigned long long* d_stamps3 = nullptr;  // development builds (-DRPSF3_STAMPS)
  size_t k3_floats = 0;
  float* d_planes = nullptr;
  size_t planes_floats = 0;  // per plane
  size_t planes_frames = 0;  // frames the allocation holds (4 planes each)
};

enum OverlapKind { OV_ATOMIC = 0, OV_PLANES = 1, OV_DIRECT = 2, OV_SWEEP = 3 };
static OverlapKind overlap_kind(const rpsf_plan* p);

static uint64_t morton2(uint32_t a, uint32_t b) {
  auto spread = [](uint64_t x) {
    x &= 0xffffffffull;
    x = (x | (x << 16)) & 0x0000ffff0000ffffull;
    x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
    x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
  };
  return (spread(a) << 1) | spread(b);
}

// Regions and job lists of the sweep kernel for `target_regions` workgroups (rpsf_plan3.hpp); replaces the plan's previous lists
static int build_sweep_lists(rpsf_plan* p, int target_regions) {
  const int nli = p->nti - 1, nlj = p->ntj - 1;
  Plan3 plan;
  const bool built = dispatch_v3(p->N, [&]<class C>() -> int {
    return plan3_build(C::N, C::KSMAX, C::WAVES, nli, nlj, p->slot3.data(), p->par_j3, target_regions, plan) ? RPSF_OK : RPSF_E_STATE;
  }) == RPSF_OK;
  if (!built) return RPSF_OK;  // (the plan keeps its other paths)
  (void)hipFree(p->d_jobs3), (void)hipFree(p->d_regions3);
  p->d_jobs3 = nullptr, p->d_regions3 = nullptr, p->sweep_ok = false;
  HIP_TRY(hipMalloc(&p->d_jobs3, plan.jobs.size() * sizeof(Job3)));
  HIP_TRY(hipMemcpy(p->d_jobs3, plan.jobs.data(), plan.jobs.size() * sizeof(Job3), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&p->d_regions3, plan.regions.size() * sizeof(Region3)));
  HIP_TRY(hipMemcpy(p->d_regions3, plan.regions.data(), plan.regions.size() * sizeof(Region3), hipMemcpyHostToDevice));
  if (!p->d_zero3) {
    HIP_TRY(hipMalloc(&p->d_zero3, 64));
    HIP_TRY(hipMemset(p->d_zero3, 0, 64));
  }
  if (!p->h_err3) {
    if (p->parent && p->parent->h_err3) {
      p->h_err3 = p->parent->h_err3, p->d_err3 = p->parent->d_err3;
    } else {
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->h_err3), 64, hipHostMallocMapped));
      std::memset(p->h_err3, 0, 64);
      HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&p->d_err3), p->h_err3, 0));
      p->owns_err3 = true;
    }
  }
  p->n_regions3 = (int)plan.regions.size(), p->ks3 = plan.ks, p->slabs3 = plan.slabs, p->patch_slots3 = plan.patch_slots;
  p->sweep_ok = true;
#if defined(RPSF3_STAMPS) || defined(RPSF3_DUMP)
  (void)hipFree(p->d_stamps3);
  HIP_TRY(hipMalloc(&p->d_stamps3, (size_t)std::max(512, p->n_regions3) * 8 * 8 * 16 * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(p->d_stamps3, 0, (size_t)std::max(512, p->n_regions3) * 8 * 8 * 16 * sizeof(unsigned long long)));
#endif
  return RPSF_OK;
}

// Regular lattice test + colour classes + tile tables + processing order (host, at plan creation)
static int setup_lattice(rpsf_plan* p) {
  const int n = p->n_patches, half = p->N / 2;
  int r0 = p->h_coords[0], c0 = p->h_coords[1], r1 = r0, c1 = c0;
  for (int i = 0; i < n; ++i) {
    r0 = std::min(r0, p->h_coords[2 * i]), r1 = std::max(r1, p->h_coords[2 * i]);
    c0 = std::min(c0, p->h_coords[2 * i + 1]), c1 = std::max(c1, p->h_coords[2 * i + 1]);
  }
  bool ok = true;
  for (int i = 0; i < n && ok; ++i)
    ok = (p->h_coords[2 * i] - r0) % half == 0 && (p->h_coords[2 * i + 1] - c0) % half == 0;
  int nti = 0, ntj = 0;
  if (ok) {
    nti = (r1 - r0) / half + 2, ntj = (c1 - c0) / half + 2;
    if ((size_t)nti * ntj >= ((size_t)1 << 24)) ok = false;
  }
  std::vector<uint8_t> cls(n, 0);
  std::vector<int32_t> cell;  // lattice cell -> patch (or -1)
  const int nli = nti - 1, nlj = ntj - 1;
  const int par_i = ok && p->parent && p->parent->lattice ? ((r0 - p->parent->lat_r0) / half) & 1 : 0;
  const int par_j = ok && p->parent && p->parent->lattice ? ((c0 - p->parent->lat_c0) / half) & 1 : 0;
  if (ok) {
    cell.assign((size_t)nli * nlj, -1);
    for (int i = 0; i < n && ok; ++i) {
      const int li = (p->h_coords[2 * i] - r0) / half, lj = (p->h_coords[2 * i + 1] - c0) / half;
      if (cell[(size_t)li * nlj + lj] >= 0) ok = false;  // duplicate corner: two patches in one plane cell
      cell[(size_t)li * nlj + lj] = i;
      // (a view takes its colours from the parent's lattice: the planes are summed in colour order, so a band's pixels then come out
      // bit-identical to the whole-frame apply's)
      cls[i] = (uint8_t)((((li + par_i) & 1) << 1) | ((lj + par_j) & 1));
    }
  }
  p->lattice = ok;
  const int t = p->N * p->N / 2 / 64, teams = t >= 64 ? 1 : 64 / t;
  const int chunk = ((n + 7) / 8 + teams - 1) / teams * teams;  // as launch_patches cuts the order
  p->direct_ok = ok && teams == 1;
  // ---- processing order: 8 chunks, one per XCD (workgroups b and b + 8 share one) ----
  p->h_order.resize(n);
  if (ok) {
    // Column strips walked boustrophedon, cut into 8 equal runs: compact regions, so that the four patches over a
    // tile mostly run on one XCD (they read the same pixels through one L2, and the tile can be accumulated there).
    int strips = nlj >= 8 ? std::max(4, nlj / 8) : 1;  // strips about 8 patches wide (4096^2: 4 as before; 8192^2: 8, -1.2 % against 4)
    if (const char* e = dev_env("RPSF_STRIPS")) strips = std::max(1, std::min(nlj, std::atoi(e)));  // development sweeps
    int k = 0;
    bool meet = true;  // (4096^2 / 256: 0.1877 vs 0.1900 ms with alternating strip directions; RPSF_ORDER_MEET=0 selects those)
    if (const char* e = dev_env("RPSF_ORDER_MEET")) meet = std::atoi(e) != 0;
    for (int s2 = 0; s2 < strips; ++s2) {
      const int ja = (int)((long)nlj * s2 / strips), jb = (int)((long)nlj * (s2 + 1) / strips);
      if (meet) {
        // the upper half of every strip top-down, the lower half bottom-up: the two XCDs of a strip meet in the middle at the end, and
        // neighbouring strips walk the same rows at the same time, so the tiles on region borders do not wait a whole launch for
        // their last contributor (their planes would long have left the Infinity Cache)
        const int mid = (nli + 1) / 2;
        for (int li = 0; li < mid; ++li)
          for (int lj = ja; lj < jb; ++lj)
            if (cell[(size_t)li * nlj + lj] >= 0) p->h_order[k++] = cell[(size_t)li * nlj + lj];
        for (int li = nli - 1; li >= mid; --li)
          for (int lj = ja; lj < jb; ++lj)
            if (cell[(size_t)li * nlj + lj] >= 0) p->h_order[k++] = cell[(size_t)li * nlj + lj];
        continue;
      }
      for (int step = 0; step < nli; ++step) {
        const int li = (s2 & 1) ? nli - 1 - step : step;
        for (int lj = ja; lj < jb; ++lj)
          if (cell[(size_t)li * nlj + lj] >= 0) p->h_order[k++] = cell[(size_t)li * nlj + lj];
      }
    }
    // Inside a chunk the patches on the rim of the lattice go first.  They hang over the image edge and take the slower
    // padded gather / cropped store path (+50 % per patch at N = 256); dispatched first, they are the long jobs of a
    // longest-job-first list schedule: a CU that drew one simply takes one patch fewer later on, instead of a late rim
    // patch stretching the last round.  (Round 1: N = 256 215 -> 207 us, 2048^2 / N = 128 67 -> 58 us.)
    if (!dev_env("RPSF_NO_RIM_FIRST")) {
      auto rim = [&](int32_t i) {
        const int r = p->h_coords[2 * i], c = p->h_coords[2 * i + 1];
        return r == r0 || r == r1 || c == c0 || c == c1;
      };
      // ... until the rim patches got their 16-byte paths: they are now the cheaper ones (half or a quarter of the stores).  With one
      // patch per CU (N = 256) they go LAST, so that the patches of the partial last round are the short ones (4096^2: -1 %,
      // profiles/r02av); with four workgroups per CU (N = 128) first is still the better order (2048^2: 0.0685 vs 0.0705 ms).
      bool rim_last = p->N == 256;
      if (const char* e = dev_env("RPSF_RIM_LAST")) rim_last = std::atoi(e) != 0;
      for (int x = 0; x < 8; ++x) {
        const int lo = std::min(n, x * chunk), hi = std::min(n, lo + chunk);
        if (rim_last)
          std::stable_partition(p->h_order.begin() + lo, p->h_order.begin() + hi, [&](int32_t i) { return !rim(i); });
        else
          std::stable_partition(p->h_order.begin() + lo, p->h_order.begin() + hi, rim);
      }
    }
  } else {
    std::vector<std::pair<uint64_t, int32_t>> keyed(n);
    for (int i = 0; i < n; ++i)
      keyed[i] = {morton2((uint32_t)((p->h_coords[2 * i] - r0) / half), (uint32_t)((p->h_coords[2 * i + 1] - c0) / half)), i};
    std::sort(keyed.begin(), keyed.end());
    for (int i = 0; i < n; ++i) p->h_order[i] = keyed[i].second;
  }
  {
    std::vector<int4> desc(n);
    for (int s2 = 0; s2 < n; ++s2) {
      int i = p->h_order[s2];
      desc[s2] = make_int4(p->h_coords[2 * i], p->h_coords[2 * i + 1], p->k_index.empty() ? i : p->k_index[i], ok ? cls[i] : 0);
    }
    HIP_TRY(hipMalloc(&p->d_desc, sizeof(int4) * n));
    HIP_TRY(hipMemcpy(p->d_desc, desc.data(), sizeof(int4) * n, hipMemcpyHostToDevice));
  }
  if (!ok) return RPSF_OK;
  p->lat_r0 = r0, p->lat_c0 = c0, p->nti = nti, p->ntj = ntj;
  // ---- third generation: regions and job lists (rpsf_plan3.hpp) when every lattice cell has its patch ----
  if (has_v3(p->N) && nli >= 2 && nlj >= 2 && (size_t)nli * nlj == (size_t)n) {
    p->slot3.resize((size_t)nli * nlj);
    for (size_t c = 0; c < p->slot3.size(); ++c) p->slot3[c] = p->k_index.empty() ? cell[c] : p->k_index[cell[c]];
    p->par_j3 = par_j;
    const int rc3 = build_sweep_lists(p, std::max(8, p->cu_count));
    if (rc3 != RPSF_OK) return rc3;
  }
  // ---- tiles: coverage, owner chunk, ranks ----
  std::vector<int> chunk_of(n), seq_of(n);
  for (int s2 = 0; s2 < n; ++s2) chunk_of[p->h_order[s2]] = s2 / chunk, seq_of[p->h_order[s2]] = s2;
  std::vector<uint8_t> cover((size_t)nti * ntj, 0), tile_info((size_t)nti * ntj, 0);
  std::vector<uint32_t> quad_of((size_t)n * 4, quad_word(QUAD_NONE, 0, 0));
  for (int ti = 0; ti < nti; ++ti)
    for (int tj = 0; tj < ntj; ++tj) {
      int who[4], nwho = 0;  // contributors by colour
      for (int a2 = 0; a2 < 2; ++a2)
        for (int b2 = 0; b2 < 2; ++b2) {
          const int li = ti - a2, lj = tj - b2;
          if (li < 0 || lj < 0 || li >= nli || lj >= nlj) continue;
          const int i = cell[(size_t)li * nlj + lj];
          if (i >= 0) who[nwho++] = i;
        }
      std::sort(who, who + nwho, [&](int a2, int b2) { return seq_of[a2] < seq_of[b2]; });  // accumulation order = processing order
      const size_t tile = (size_t)ti * ntj + tj;
      int owner = -1, best = 0;
      for (int k = 0; k < nwho; ++k) {
        cover[tile] |= (uint8_t)(1u << cls[who[k]]);
        int cnt = 0;
        for (int m = 0; m < nwho; ++m) cnt += chunk_of[who[m]] == chunk_of[who[k]];
        if (cnt > best) best = cnt, owner = chunk_of[who[k]];  // ties: the chunk of the earliest contributor
      }
      int rank = 0;
      uint8_t side = 0;
      for (int k = 0; k < nwho; ++k) {
        const int i = who[k];
        const int li = (p->h_coords[2 * i] - r0) / half, lj = (p->h_coords[2 * i + 1] - c0) / half;
        const int q = 2 * (ti - li) + (tj - lj);
        if (p->direct_ok && chunk_of[i] == owner) {
          quad_of[(size_t)i * 4 + q] = quad_word(QUAD_DIRECT, (uint32_t)rank++, (uint32_t)tile);
        } else {
          quad_of[(size_t)i * 4 + q] = quad_word(QUAD_SIDE, 0, (uint32_t)tile);
          side |= (uint8_t)(1u << cls[i]);
        }
      }
      tile_info[tile] = (uint8_t)(side | (nwho ? 16 : 0));
    }
  HIP_TRY(hipMalloc(&p->d_cover, cover.size()));
  HIP_TRY(hipMemcpy(p->d_cover, cover.data(), cover.size(), hipMemcpyHostToDevice));
  if (p->v2) {  // fused plane sum: tile order (by the slot of the last contributor: the dispatch order inside a chunk) and counters
    std::vector<std::pair<int, uint32_t>> keyed;
    for (int ti = 0; ti < nti; ++ti)
      for (int tj = 0; tj < ntj; ++tj) {
        int last = -1;
        for (int a2 = 0; a2 < 2; ++a2)
          for (int b2 = 0; b2 < 2; ++b2) {
            const int li = ti - a2, lj = tj - b2;
            if (li < 0 || lj < 0 || li >= nli || lj >= nlj) continue;
            const int i = cell[(size_t)li * nlj + lj];
            if (i >= 0) last = std::max(last, seq_of[i] % chunk);
          }
        keyed.push_back({last, (uint32_t)(ti * ntj + tj)});
      }
    std::stable_sort(keyed.begin(), keyed.end(), [](const auto& a2, const auto& b2) { return a2.first < b2.first; });
    std::vector<uint32_t> order(keyed.size());
    for (size_t i = 0; i < keyed.size(); ++i) order[i] = keyed[i].second;
    HIP_TRY(hipMalloc(&p->d_sum_order, order.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(p->d_sum_order, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&p->d_tile_done, order.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemset(p->d_tile_done, 0, order.size() * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&p->d_sum_queue, sizeof(uint32_t)));
    HIP_TRY(hipMemset(p->d_sum_queue, 0, sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&p->d_xq, 8 * 32 * sizeof(uint32_t)));
    HIP_TRY(hipMemset(p->d_xq, 0, 8 * 32 * sizeof(uint32_t)));
    {  // image prefetch lists: for every chunk, each lattice tile once, in the order the chunk's slots first touch it
      std::vector<uint32_t> tiles;
      for (int x = 0; x < 8; ++x) {
        p->prefetch_first[x] = (uint32_t)tiles.size();
        std::vector<char> seen((size_t)nti * ntj, 0);
        for (int s2 = std::min(n, x * chunk); s2 < std::min(n, (x + 1) * chunk); ++s2) {
          const int i = p->h_order[s2];
          const int li = (p->h_coords[2 * i] - r0) / half, lj = (p->h_coords[2 * i + 1] - c0) / half;
          for (int q = 0; q < 4; ++q) {
            const size_t tile = (size_t)(li + (q >> 1)) * ntj + (lj + (q & 1));
            if (!seen[tile]) seen[tile] = 1, tiles.push_back((uint32_t)tile);
          }
        }
      }

      p->prefetch_first[8] = (uint32_t)tiles.size();
      HIP_TRY(hipMalloc(&p->d_prefetch_tiles, std::max<size_t>(1, tiles.size()) * sizeof(uint32_t)));
      HIP_TRY(hipMemcpy(p->d_prefetch_tiles, tiles.data(), tiles.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
  }
  if (p->direct_ok) {
    std::vector<uint4> quads(n);
    for (int s2 = 0; s2 < n; ++s2) {
      const uint32_t* q = &quad_of[(size_t)p->h_order[s2] * 4];
      quads[s2] = make_uint4(q[0], q[1], q[2], q[3]);
    }
    HIP_TRY(hipMalloc(&p->d_quads, sizeof(uint4) * n));
    HIP_TRY(hipMemcpy(p->d_quads, quads.data(), sizeof(uint4) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&p->d_tile_info, tile_info.size()));
    HIP_TRY(hipMemcpy(p->d_tile_info, tile_info.data(), tile_info.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&p->d_chunk_xcc, 8 * sizeof(uint32_t)));
    HIP_TRY(hipMemset(p->d_chunk_xcc, 0, 8 * sizeof(uint32_t)));
  }
  return RPSF_OK;
}

template <class F>
static int dispatch_n(int N, F&& f) {
  switch (N) {
    case 256: return f.template operator()<Cfg256>();
    case 128: return f.template operator()<Cfg128>();
    case 64: return f.template operator()<Cfg64>();
    case 32: return f.template operator()<Cfg32>();
    case 16: return f.template operator()<Cfg16>();
    default: return fail(RPSF_E_UNSUPPORTED, "patch size " + std::to_string(N) + " has no compiled plan (16..256, powers of two)");
  }
}

template <class F>
static int dispatch_v2(int N, F&& f) {
  switch (N) {
    case 256: return f.template operator()<Cfg256v2>();
    case 128: return f.template operator()<Cfg128v2>();
    default: return fail(RPSF_E_UNSUPPORTED, "no second-generation plan for this patch size");
  }
}
static bool has_v2(int N) { return (N == 256 || N == 128) && !dev_env("RPSF_V1"); }

static void host_tables(int N, std::vector<cf>& tw, std::vector<float>& win) {
  tw.resize(N);
  win.resize(N);
  for (int k = 0; k < N; ++k) {
    double a = -2.0 * M_PI * k / N;
    tw[k] = cf{(float)std::cos(a), (float)std::sin(a)};
    win[k] = (float)std::sin((k + 0.5) * (M_PI / N));  // transform.py:151-154
  }
}

template <class C>
static int upload_tables(int device, uint16_t** d_tab, cf** d_tw, float** d_win, uint32_t** d_pairtab = nullptr) {
  std::vector<uint16_t> tab((size_t)C::T * C::NSLOT * 2);
  build_slot_table<C>(tab.data());
  std::vector<uint32_t> pt((size_t)C::PT_WORDS + 1);
  if (d_pairtab && build_pair_table<C>(tab.data(), pt.data()) > C::NP) return fail(RPSF_E_STATE, "pair table overflow (internal)");
  std::vector<cf> tw;
  std::vector<float> win;
  host_tables(C::N, tw, win);
  HIP_TRY(hipSetDevice(device));
  DevBuf b_pt, b_tab, b_tw, b_win;  // handed over only when everything has been uploaded: nothing leaks on an error
  if (d_pairtab) {
    HIP_TRY(b_pt.alloc(pt.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(b_pt.p, pt.data(), pt.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  HIP_TRY(b_tab.alloc(tab.size() * sizeof(uint16_t)));
  HIP_TRY(hipMemcpy(b_tab.p, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  HIP_TRY(b_tw.alloc(tw.size() * sizeof(cf)));
  HIP_TRY(hipMemcpy(b_tw.p, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice));
  if (d_win) {
    HIP_TRY(b_win.alloc(win.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(b_win.p, win.data(), win.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if (d_pairtab) *d_pairtab = static_cast<uint32_t*>(b_pt.release());
  *d_tab = static_cast<uint16_t*>(b_tab.release());
  *d_tw = static_cast<cf*>(b_tw.release());
  if (d_win) *d_win = static_cast<float*>(b_win.release());
  return RPSF_OK;
}

template <class C>
static int upload_tables2(int device, uint16_t** d_tab, cf** d_tw, float** d_win, uint32_t** d_ot) {
  std::vector<uint16_t> tab((size_t)C::T * C::NSLOT * 2);
  build_slot_table2<C>(tab.data());
  std::vector<uint32_t> ot((size_t)C::ORBIT_ROUNDS * 64);
  if (special_slots2<C>() > 64 || build_orbit_table2<C>(tab.data(), ot.data()) != C::NORBIT)
    return fail(RPSF_E_STATE, "slot / orbit table inconsistent (internal)");
  std::vector<cf> tw;
  std::vector<float> win;
  host_tables(C::N, tw, win);
  HIP_TRY(hipSetDevice(device));
  DevBuf b_ot, b_tab, b_tw, b_win;
  HIP_TRY(b_ot.alloc(ot.size() * sizeof(uint32_t)));
  HIP_TRY(hipMemcpy(b_ot.p, ot.data(), ot.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(b_tab.alloc(tab.size() * sizeof(uint16_t)));
  HIP_TRY(hipMemcpy(b_tab.p, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  HIP_TRY(b_tw.alloc(tw.size() * sizeof(cf)));
  HIP_TRY(hipMemcpy(b_tw.p, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice));
  HIP_TRY(b_win.alloc(win.size() * sizeof(float)));
  HIP_TRY(hipMemcpy(b_win.p, win.data(), win.size() * sizeof(float), hipMemcpyHostToDevice));
  *d_ot = static_cast<uint32_t*>(b_ot.release());
  *d_tab = static_cast<uint16_t*>(b_tab.release());
  *d_tw = static_cast<cf*>(b_tw.release());
  *d_win = static_cast<float*>(b_win.release());
  return RPSF_OK;
}

extern "C" int rpsf_device_count(int* count) {
  if (!count) return fail(RPSF_E_BADARG, "count is null");
  HIP_TRY(hipGetDeviceCount(count));
  return RPSF_OK;
}

extern "C" int rpsf_device_info(int device, int* compute_units, char* name, size_t name_len) {
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (name && name_len) {
    std::snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  return RPSF_OK;
}


// ------------------------------------------------------------------------------------------------
// Colour classes for the fallback's overlap-add (any N, odd ones too): calculate_covering (regularizepsf/util.py:10-53) lays four sub-grids of
// pitch N - corner rows k N and k N - ceil(N / 2), the same for columns - so the corner rows of a covering take at most two residues modulo
// N, and so do the columns.  Colour = (row residue class, column residue class): two distinct patches of one colour differ by a multiple of N
// in a coordinate, i.e. they do not overlap.  False (atomic adds) for corner lists without that structure or with a corner listed twice.
static bool generic_colours(int N, const int32_t* coords_rc, int n, std::vector<uint8_t>& colours) {
  int res[2][2], nres[2] = {0, 0};
  colours.assign((size_t)n, 0);
  std::vector<std::pair<int32_t, int32_t>> seen((size_t)n);
  for (int i = 0; i < n; ++i) {
    seen[(size_t)i] = {coords_rc[2 * i], coords_rc[2 * i + 1]};
    for (int axis = 0; axis < 2; ++axis) {
      const int r = ((coords_rc[2 * i + axis] % N) + N) % N;
      int cls = -1;
      for (int k = 0; k < nres[axis]; ++k)
        if (res[axis][k] == r) cls = k;
      if (cls < 0) {
        if (nres[axis] == 2) return false;
        cls = nres[axis], res[axis][nres[axis]++] = r;
      }
      colours[(size_t)i] |= (uint8_t)(cls << (1 - axis));
    }
  }
  std::sort(seen.begin(), seen.end());
  return std::adjacent_find(seen.begin(), seen.end()) == seen.end();
}

// Any other patch size: fallback through hipFFT (loaded with dlopen, like RCCL).
// The reference accepts every square patch size (odd ones included, transform.py:151-164); the
// hand-written plans cover 16..256.  For the rest: gather+pad+window into a complex batch, batched
// 2-D C2C forward, times the caller's full K (no folding: Re(ifft2(.)) is taken literally), C2C
// inverse, window again, float atomic overlap-add.  Correctness path, not a tuned one.
// ------------------------------------------------------------------------------------------------
struct HipfftApi {
  void* lib = nullptr;
  int (*plan_many)(void**, int, int*, int*, int, int, int*, int, int, int, int) = nullptr;
  int (*set_stream)(void*, hipStream_t) = nullptr;
  int (*exec_c2c)(void*, void*, void*, int) = nullptr;
  int (*destroy)(void*) = nullptr;
  std::string error;
  std::mutex guard;  // plans may be created from several threads
  bool load() {
    std::lock_guard<std::mutex> lock(guard);
    if (lib) return true;
    if (!error.empty()) return false;
    for (const char* n : {"libhipfft.so", "libhipfft.so.0", "/opt/rocm/lib/libhipfft.so"}) {
      lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) {
      error = "patch sizes other than 16, 32, 64, 128, 256 need libhipfft.so, which could not be loaded";
      return false;
    }
    plan_many = reinterpret_cast<decltype(plan_many)>(dlsym(lib, "hipfftPlanMany"));
    set_stream = reinterpret_cast<decltype(set_stream)>(dlsym(lib, "hipfftSetStream"));
    exec_c2c = reinterpret_cast<decltype(exec_c2c)>(dlsym(lib, "hipfftExecC2C"));
    destroy = reinterpret_cast<decltype(destroy)>(dlsym(lib, "hipfftDestroy"));
    if (!plan_many || !set_stream || !exec_c2c || !destroy) {
      error = "libhipfft.so lacks hipfftPlanMany / hipfftSetStream / hipfftExecC2C / hipfftDestroy";
      lib = nullptr;
      return false;
    }
    return true;
  }
};
static HipfftApi g_hipfft;

static void set_corner_extremes(rpsf_plan* p) {
  for (int d = 0; d < 2; ++d) p->corner_min[d] = p->corner_max[d] = p->h_coords[d];
  for (int i = 1; i < p->n_patches; ++i)
    for (int d = 0; d < 2; ++d) {
      p->corner_min[d] = std::min(p->corner_min[d], p->h_coords[2 * i + d]);
      p->corner_max[d] = std::max(p->corner_max[d], p->h_coords[2 * i + d]);
    }
}

// parent / k_index: a view of `parent` (shares its tables, packed K and stream; k_index[i] = the parent's index of patch i)
static int plan_create_impl(rpsf_plan** out, int device, int patch_size, int n_patches, const int32_t* coords_rc, rpsf_plan* parent,
                            const int32_t* k_index) {
  if (!out || !coords_rc) return fail(RPSF_E_BADARG, "null argument");
  if (n_patches <= 0) return fail(RPSF_E_BADARG, "n_patches must be positive");
  const int N = patch_size;
  const bool compiled = dispatch_n(N, []<class C>() { return RPSF_OK; }) == RPSF_OK;
  if (!compiled) {
    if (N < 2 || N > 4096) return fail(RPSF_E_UNSUPPORTED, "patch size " + std::to_string(N) + " is outside 2..4096");
    if (!g_hipfft.load()) return fail(RPSF_E_UNSUPPORTED, g_hipfft.error);
  }
  auto* p = new rpsf_plan;
  p->device = device, p->N = N, p->n_patches = n_patches;
  p->generic = !compiled;
  p->parent = parent;
  if (parent) p->k_index.assign(k_index, k_index + n_patches);
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(device));
    if (parent) {
      if (p->generic) return fail(RPSF_E_UNSUPPORTED, "views need a compiled plan");
      p->stream = parent->stream;
    } else {
      HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    }
    for (auto& e : p->ev) HIP_TRY(hipEventCreate(&e));
    if (p->generic) {
      HIP_TRY(hipMalloc(&p->d_coords, sizeof(int32_t) * 2 * n_patches));
      HIP_TRY(hipMemcpy(p->d_coords, coords_rc, sizeof(int32_t) * 2 * n_patches, hipMemcpyHostToDevice));
      p->h_coords.assign(coords_rc, coords_rc + 2 * (size_t)n_patches);
      set_corner_extremes(p);
      std::vector<float> win(N);
      for (int i = 0; i < N; ++i) win[i] = (float)std::sin((i + 0.5) * M_PI / N);  // transform.py:151-155
      HIP_TRY(hipMalloc(&p->d_win_generic, sizeof(float) * N));
      HIP_TRY(hipMemcpy(p->d_win_generic, win.data(), sizeof(float) * N, hipMemcpyHostToDevice));
      {
        std::vector<uint8_t> colours;
        if (generic_colours(N, p->h_coords.data(), n_patches, colours)) {
          HIP_TRY(hipMalloc(&p->d_colour_generic, colours.size()));
          HIP_TRY(hipMemcpy(p->d_colour_generic, colours.data(), colours.size(), hipMemcpyHostToDevice));
          p->lattice = true;  // (what rpsf_plan_set_overlap_mode(2) and the automatic mode ask for)
        }
      }
      const size_t per = (size_t)N * N;
      HIP_TRY(hipMalloc(&p->d_kfull, per * n_patches * sizeof(cf)));
      p->g_elems = per * n_patches;
      p->fft_chunk = (int)std::min<size_t>((size_t)n_patches, std::max<size_t>(1, ((size_t)256 << 20) / (per * sizeof(cf))));
      HIP_TRY(hipMalloc(&p->d_fft_buf, per * p->fft_chunk * sizeof(cf)));
      int dims[2] = {N, N};
      if (g_hipfft.plan_many(&p->fft_plan, 2, dims, nullptr, 1, (int)per, nullptr, 1, (int)per, /*HIPFFT_C2C*/ 0x29, p->fft_chunk) != 0)
        return fail(RPSF_E_HIP, "hipfftPlanMany failed");
      if (g_hipfft.set_stream(p->fft_plan, p->stream) != 0) return fail(RPSF_E_HIP, "hipfftSetStream failed");
      return RPSF_OK;
    }
    {
      hipDeviceProp_t prop;
      HIP_TRY(hipGetDeviceProperties(&prop, device));
      p->cu_count = prop.multiProcessorCount;
    }
    HIP_TRY(hipMalloc(&p->d_coords, sizeof(int32_t) * 2 * n_patches));
    HIP_TRY(hipMemcpy(p->d_coords, coords_rc, sizeof(int32_t) * 2 * n_patches, hipMemcpyHostToDevice));
    p->h_coords.assign(coords_rc, coords_rc + 2 * (size_t)n_patches);
    set_corner_extremes(p);
    HIP_TRY(hipMalloc(&p->d_sink, 128 * sizeof(float)));
#if defined(RPSF_STAMPS)
#if defined(RPSF_WAVE_STAMPS)
    constexpr size_t STAMP_WAVES = 8;
#else
    constexpr size_t STAMP_WAVES = 1;
#endif
    HIP_TRY(hipMalloc(&p->d_stamps, sizeof(unsigned long long) * 16 * STAMP_WAVES * (size_t)n_patches));
    HIP_TRY(hipMemset(p->d_stamps, 0, sizeof(unsigned long long) * 16 * STAMP_WAVES * (size_t)n_patches));
#endif
    p->v2 = has_v2(N);
    p->no_fuse = false;  // (rpsf_plan_set_option RPSF_OPT_FUSE)
    // (measured, profiles/r02u, r02v: 32 of them are worth -1 % at 4096^2 and -2.5 % at 8192^2; 48 cost more patch time than they hide)
    p->sum_first = -1;  // decided per launch (sum_first_for) unless the environment pins it
    if (const char* e = dev_env("RPSF_SUM_FIRST")) p->sum_first = std::max(0, std::atoi(e)) / 8 * 8;
    if (const char* e = dev_env("RPSF_HEAD_PATCHES")) p->head_patches = std::atoi(e) > 0 ? 1 : 0;
    if (const char* e = dev_env("RPSF_PREFETCH")) p->prefetch = std::atoi(e) != 0;
    if (const char* e = dev_env("RPSF_STAGGER_US")) p->stagger_us = std::max(0, std::atoi(e));  // development sweeps
    if (const char* e = dev_env("RPSF_RESERVED_CUS")) p->reserved_cus = std::min(128, std::max(0, std::atoi(e)));
    // (until the plane stores were kept in the Infinity Cache the fused sum cost the 128-pixel plan 3 %; now it gains 3 ... 6 %)
    p->fuse_pays = N >= 128 || dev_env("RPSF_FUSE_ALWAYS") != nullptr;
    // (256-pixel plan: profiles/r02ag, -3.7 % per apply at 4096^2 from the re-entry alone; 128-pixel plan, whose four workgroups per CU
    // hide one another's dispatch: 8 x 2048^2 0.334 vs 0.343 ms, single frames unchanged)
    p->persist = N == 256 || N == 128;  // (rpsf_plan_set_option RPSF_OPT_PERSIST)
    if (p->persist) {
      if (N == 256)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_kernel2_256p), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)Launch2<Cfg256v2>::LDS_BYTES));
      else {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_kernel2_128p), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)Launch2<Cfg128v2>::LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_kernel2_128pc), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)Launch2<Cfg128v2>::LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_kernel2_128pcs), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)Launch2<Cfg128v2>::LDS_BYTES));
        // Plain instead of streaming loads of the pair words where the plan's K (66,560 B per patch) fits the 256 MiB Infinity Cache beside the
        // planes: measured (profiles/r04av) -6 % per apply at 72 MB of K, +6.6 % at 160 MB; RPSF_K_CACHED=0/1 overrides (tests run both forms).
        p->k_cached = (size_t)(parent ? parent->n_patches : n_patches) * Cfg128v2::G_PER_PATCH * sizeof(cf) <= ((size_t)96 << 20);
      }
    }
    int rl = p->v2 ? dispatch_v2(N, [&]<class C>() -> int {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_kernel2<C>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)Launch2<C>::LDS_BYTES));
      int per_cu = 0;
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, patch_kernel2<C>, Launch2<C>::WG, Launch2<C>::LDS_BYTES));
      p->round_capacity = p->cu_count * std::max(1, per_cu);
      return RPSF_OK;
    }) : dispatch_n(N, [&]<class C>() -> int {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_kernel<C>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)Launch<C>::LDS_BYTES));
      int per_cu = 0;
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, patch_kernel<C>, Launch<C>::WG, Launch<C>::LDS_BYTES));
      p->round_capacity = p->cu_count * std::max(1, per_cu) * Launch<C>::TEAMS;
      return RPSF_OK;
    });
    if (rl != RPSF_OK) return rl;
    HIP_TRY(hipEventCreateWithFlags(&p->ev_busy, hipEventDisableTiming));
    rl = setup_lattice(p);
    if (rl != RPSF_OK) return rl;
    if (p->sweep_ok) {
      rl = dispatch_v3(N, [&]<class C>() -> int {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_kernel_kc<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        if (!parent) {
          p->k3_floats = (size_t)C::K_FLOATS * n_patches;
          HIP_TRY(hipMalloc(&p->d_k3, p->k3_floats * sizeof(float)));
        }
        return RPSF_OK;
      });
      if (rl != RPSF_OK) return rl;
      if (parent) {
        if (parent->d_k3) p->d_k3 = parent->d_k3, p->k3_floats = parent->k3_floats;
        else p->sweep_ok = false;  // (the parent's lattice was not complete: it has no third-generation K)
      }
    }
    if (parent) {  // tables and packed K are the parent's
      p->d_tab = parent->d_tab, p->d_pairtab = parent->d_pairtab, p->d_tw = parent->d_tw, p->d_win = parent->d_win;
      p->d_g = parent->d_g, p->d_gs = parent->d_gs, p->g_elems = parent->g_elems, p->gs_elems = parent->gs_elems;
      p->have_k = parent->have_k;
      p->overlap_mode = parent->overlap_mode, p->stagger_us = parent->stagger_us;
      return RPSF_OK;
    }
    if (p->v2)
      return dispatch_v2(N, [&]<class C>() -> int {
        int r2 = upload_tables2<C>(device, &p->d_tab, &p->d_tw, &p->d_win, &p->d_pairtab);
        if (r2 != RPSF_OK) return r2;
        p->g_elems = (size_t)C::G_PER_PATCH * n_patches;
        p->gs_elems = (size_t)C::GS_PER_PATCH * n_patches;
        HIP_TRY(hipMalloc(&p->d_g, p->g_elems * sizeof(cf)));
        HIP_TRY(hipMalloc(&p->d_gs, (p->gs_elems + 1) * sizeof(cf)));
        return RPSF_OK;
      });
    return dispatch_n(N, [&]<class C>() -> int {
      int r2 = upload_tables<C>(device, &p->d_tab, &p->d_tw, &p->d_win, &p->d_pairtab);
      if (r2 != RPSF_OK) return r2;
      p->g_elems = (size_t)C::G_PER_PATCH * n_patches;
      p->gs_elems = (size_t)C::GS_PER_PATCH * n_patches;
      HIP_TRY(hipMalloc(&p->d_g, p->g_elems * sizeof(cf)));
      HIP_TRY(hipMalloc(&p->d_gs, (p->gs_elems + 1) * sizeof(cf)));
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_kernel<C>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)Launch<C>::LDS_BYTES));
      return RPSF_OK;
    });
  };
  const int rc = body();
  if (rc != RPSF_OK) {
    std::string keep = g_err;
    rpsf_plan_destroy(p);
    g_err = keep;
    return rc;
  }
  *out = p;
  return RPSF_OK;
}

extern "C" int rpsf_plan_create(rpsf_plan** out, int device, int patch_size, int n_patches, const int32_t* coords_rc) {
  return plan_create_impl(out, device, patch_size, n_patches, coords_rc, nullptr, nullptr);
}

extern "C" void* rpsf_plan_stream(rpsf_plan* p) { return p ? (void*)p->stream : nullptr; }

extern "C" void rpsf_plan_destroy(rpsf_plan* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  for (rpsf_plan* band : p->bands) rpsf_plan_destroy(band);
  p->bands.clear();
  (void)hipFree(p->d_coords);
  if (!p->parent) {
    (void)hipFree(p->d_tab);
    (void)hipFree(p->d_pairtab);
    (void)hipFree(p->d_tw);
    (void)hipFree(p->d_win);
    (void)hipFree(p->d_g);
    (void)hipFree(p->d_gs);
  }
  if (p->pipe) {
    p->pipe->destroy();
    delete p->pipe;
  }
  (void)hipFree(p->d_jobs3);
  (void)hipFree(p->d_regions3);
  (void)hipFree(p->d_zero3);
  if (p->owns_err3) (void)hipHostFree(p->h_err3);
  (void)hipFree(p->d_stamps3);
  if (!p->parent) (void)hipFree(p->d_k3);
  (void)hipFree(p->d_cover);
  (void)hipFree(p->d_desc);
  (void)hipFree(p->d_stamps);
  (void)hipFree(p->d_sink);
  (void)hipFree(p->d_kfull);
  (void)hipFree(p->d_fft_buf);
  (void)hipFree(p->d_win_generic);
  (void)hipFree(p->d_colour_generic);
  if (p->fft_plan && g_hipfft.destroy) (void)g_hipfft.destroy(p->fft_plan);
  (void)hipFree(p->d_planes);
  (void)hipFree(p->d_prefetch_tiles);
  (void)hipFree(p->d_quads);
  (void)hipFree(p->d_tile_info);
  (void)hipFree(p->d_flags);
  (void)hipFree(p->d_dyn);
  (void)hipFree(p->d_chunk_xcc);
  (void)hipFree(p->d_tile_done);
  (void)hipFree(p->d_sum_order);
  (void)hipFree(p->d_sum_queue);
  (void)hipFree(p->d_xq);
  if (p->ev_busy) (void)hipEventDestroy(p->ev_busy);
  for (auto& e : p->ev)
    if (e) (void)hipEventDestroy(e);
  if (p->stream && !p->parent) (void)hipStreamDestroy(p->stream);
  delete p;
}

static void drop_bands(rpsf_plan* p) {
  for (rpsf_plan* band : p->bands) rpsf_plan_destroy(band);
  p->bands.clear(), p->band_rows.clear(), p->band_in_rows.clear();
  p->bands_h = p->bands_w = 0, p->bands_mode = -1, p->bands_want = -1;
}

static int pack_range(rpsf_plan* p, const cf* d_kfull, int first_patch, int count) {
  if (p->d_k3 && !p->parent) {
    const int rc3 = dispatch_v3(p->N, [&]<class C>() -> int {
      const size_t total = (size_t)(C::K_FLOATS / 2) * count;
      pack_kernel3<C><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, p->stream>>>(
          d_kfull, count, reinterpret_cast<cf*>(p->d_k3 + (size_t)first_patch * C::K_FLOATS));
      HIP_TRY(hipGetLastError());
      return RPSF_OK;
    });
    if (rc3 != RPSF_OK) return rc3;
  }
  if (p->v2)
    return dispatch_v2(p->N, [&]<class C>() -> int {
      const size_t total = ((size_t)C::G_PER_PATCH + C::GS_PER_PATCH) * count;
      pack_kernel2<C><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, p->stream>>>(
          d_kfull, count, p->d_tab, p->d_pairtab, p->d_g + (size_t)first_patch * C::G_PER_PATCH,
          p->d_gs + (size_t)first_patch * C::GS_PER_PATCH);
      HIP_TRY(hipGetLastError());
      return RPSF_OK;
    });
  return dispatch_n(p->N, [&]<class C>() -> int {
    size_t total = (size_t)C::G_PER_PATCH * count;
    int block = 256;
    size_t grid = (total + block - 1) / block;
    pack_kernel<C><<<dim3((unsigned)grid), dim3(block), 0, p->stream>>>(
        d_kfull, count, p->d_tab, p->d_pairtab, p->d_g + (size_t)first_patch * C::G_PER_PATCH,
        p->d_gs + (size_t)first_patch * C::GS_PER_PATCH);
    HIP_TRY(hipGetLastError());
    return RPSF_OK;
  });
}

extern "C" int rpsf_plan_set_transfer(rpsf_plan* p, const float* k_host) {
  if (!p || !k_host) return fail(RPSF_E_BADARG, "null argument");
  HIP_TRY(hipSetDevice(p->device));
  const size_t per = (size_t)p->N * p->N;
  if (p->generic) {
    HIP_TRY(hipMemcpy(p->d_kfull, k_host, per * p->n_patches * sizeof(cf), hipMemcpyHostToDevice));
    p->have_k = true;
    return RPSF_OK;
  }
  int chunk = (int)std::max<size_t>(1, (size_t)(64u << 20) / (per * sizeof(cf)));
  if (chunk > p->n_patches) chunk = p->n_patches;
  DevBuf b_tmp;
  HIP_TRY(b_tmp.alloc(per * sizeof(cf) * chunk));
  cf* d_tmp = b_tmp.as<cf>();
  int rc = RPSF_OK;
  for (int first = 0; first < p->n_patches && rc == RPSF_OK; first += chunk) {
    int cnt = std::min(chunk, p->n_patches - first);
    hipError_t e = hipMemcpyAsync(d_tmp, reinterpret_cast<const cf*>(k_host) + (size_t)first * per,
                                  per * sizeof(cf) * cnt, hipMemcpyHostToDevice, p->stream);
    if (e != hipSuccess) { rc = fail(RPSF_E_HIP, hipGetErrorString(e)); break; }
    rc = pack_range(p, d_tmp, first, cnt);
    if (rc == RPSF_OK) {
      e = hipStreamSynchronize(p->stream);
      if (e != hipSuccess) rc = fail(RPSF_E_HIP, hipGetErrorString(e));
    }
  }
  if (rc == RPSF_OK) p->have_k = true;
  for (rpsf_plan* band : p->bands) band->have_k = p->have_k;
  return rc;
}

extern "C" int rpsf_plan_set_transfer_device(rpsf_plan* p, const void* k_dev) {
  if (!p || !k_dev) return fail(RPSF_E_BADARG, "null argument");
  HIP_TRY(hipSetDevice(p->device));
  if (p->generic) {
    HIP_TRY(hipMemcpy(p->d_kfull, k_dev, (size_t)p->N * p->N * p->n_patches * sizeof(cf), hipMemcpyDeviceToDevice));
    p->have_k = true;
    return RPSF_OK;
  }
  int rc = pack_range(p, reinterpret_cast<const cf*>(k_dev), 0, p->n_patches);
  if (rc != RPSF_OK) return rc;
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->have_k = true;
  return RPSF_OK;
}

// construct -> pack in one pass (transform.py:78-82 evaluated where the packer reads K): complex64 spectra of the plan's patches on its device
extern "C" int rpsf_plan_set_transfer_spectra_device(rpsf_plan* p, const void* s_c64_dev, const void* t_c64_dev, double alpha, double epsilon) {
  if (!p || !s_c64_dev || !t_c64_dev) return fail(RPSF_E_BADARG, "null argument");
  if (p->parent) return fail(RPSF_E_STATE, "a view shares its parent's transfer kernel");
  HIP_TRY(hipSetDevice(p->device));
  const cf* s = reinterpret_cast<const cf*>(s_c64_dev);
  const cf* t = reinterpret_cast<const cf*>(t_c64_dev);
  int rc = RPSF_OK;
  if (p->generic) {  // the fallback multiplies by the caller's unfolded K: K2 straight into the plan's copy
    rc = rpsf_build_transfer_device(p->device, (size_t)p->N * p->N * p->n_patches, s, t, 0, alpha, epsilon, p->d_kfull, p->stream);
  } else if (p->v2) {
    rc = dispatch_v2(p->N, [&]<class C>() -> int {
      const size_t total = ((size_t)C::G_PER_PATCH + C::GS_PER_PATCH) * p->n_patches;
      pack_spectra_kernel2<C><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, p->stream>>>(s, t, (float)alpha, (float)epsilon, p->n_patches,
                                                                                                 p->d_tab, p->d_pairtab, p->d_g, p->d_gs);
      HIP_TRY(hipGetLastError());
      return RPSF_OK;
    });
  } else {
    if (p->d_k3) {
      rc = dispatch_v3(p->N, [&]<class C>() -> int {
        const size_t total = (size_t)(C::K_FLOATS / 2) * p->n_patches;
        pack_spectra_kernel3<C><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, p->stream>>>(s, t, (float)alpha, (float)epsilon, p->n_patches,
                                                                                                   reinterpret_cast<cf*>(p->d_k3));
        HIP_TRY(hipGetLastError());
        return RPSF_OK;
      });
      if (rc != RPSF_OK) return rc;
    }
    rc = dispatch_n(p->N, [&]<class C>() -> int {
      const size_t total = (size_t)C::G_PER_PATCH * p->n_patches;
      pack_spectra_kernel<C><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, p->stream>>>(s, t, (float)alpha, (float)epsilon, p->n_patches,
                                                                                                p->d_tab, p->d_pairtab, p->d_g, p->d_gs);
      HIP_TRY(hipGetLastError());
      return RPSF_OK;
    });
  }
  if (rc != RPSF_OK) return rc;
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->have_k = true;
  for (rpsf_plan* band : p->bands) band->have_k = true;
  return RPSF_OK;
}

extern "C" int rpsf_plan_transfer_bytes(const rpsf_plan* p, size_t* bytes) {
  if (!p || !bytes) return fail(RPSF_E_BADARG, "null argument");
  // (the representation the plan's applies read: the third generation's when the sweep kernel runs them)
  *bytes = p->sweep_ok && overlap_kind(p) == OV_SWEEP ? p->k3_floats * sizeof(float) : (p->g_elems + p->gs_elems) * sizeof(cf);
  return RPSF_OK;
}

static int check_geometry(const rpsf_plan* p, const rpsf_geometry* g) {
  if (!g) return fail(RPSF_E_BADARG, "geometry is null");
  if (g->height <= 0 || g->width <= 0) return fail(RPSF_E_BADARG, "image shape must be positive");
  if (g->pad_mode < 0 || g->pad_mode > RPSF_PAD_WRAP) return fail(RPSF_E_BADARG, "unknown pad mode");
  if (g->ld_image < g->width || g->ld_out < g->width) return fail(RPSF_E_BADARG, "row stride smaller than the image width");
  if (g->image_row0 < 0 || g->image_rows <= 0 || g->image_row0 + g->image_rows > g->height || g->out_row0 < 0 ||
      g->out_rows <= 0 || g->out_row0 + g->out_rows > g->height)
    return fail(RPSF_E_BADARG, "resident row window lies outside the image");
  const int N = p->N;
  // same reach as the reference's 2N padding (transform.py:119-123,141-149); the extreme corners decide
  const long rlo = (long)p->corner_min[0] + g->origin_row, rhi = (long)p->corner_max[0] + g->origin_row;
  const long clo = (long)p->corner_min[1] + g->origin_col, chi = (long)p->corner_max[1] + g->origin_col;
  if (rlo < -2L * N || rhi > (long)g->height + N || clo < -2L * N || chi > (long)g->width + N) {
    for (int i = 0; i < p->n_patches; ++i) {  // name the first offender
      long r = (long)p->h_coords[2 * i] + g->origin_row, c = (long)p->h_coords[2 * i + 1] + g->origin_col;
      if (r < -2L * N || r > (long)g->height + N || c < -2L * N || c > (long)g->width + N)
        return fail(RPSF_E_BADARG, "patch corner (" + std::to_string(r) + ", " + std::to_string(c) +
                                       ") lies outside the 2N-padded image");
    }
  }
  return RPSF_OK;
}

static SumParams make_sum_params(const rpsf_plan* p, float* d_out, const rpsf_geometry& g, int row_begin, int row_end) {
  SumParams sp;
  sp.planes = p->d_planes, sp.plane_stride = p->planes_floats, sp.out = d_out;
  sp.rows = row_end - row_begin, sp.row_begin = row_begin;
  sp.W = g.width, sp.ld_planes = g.width, sp.ld_out = g.ld_out, sp.row0 = g.out_row0;
  sp.lat_r0 = p->lat_r0 + g.origin_row, sp.lat_c0 = p->lat_c0 + g.origin_col;
  sp.half_shift = 0;
  while ((1 << (sp.half_shift + 1)) < p->N) ++sp.half_shift;
  sp.nti = p->nti, sp.ntj = p->ntj, sp.cover = p->d_cover;
  sp.planes_frame_floats = 4 * p->planes_floats, sp.out_frame_floats = 0;
  return sp;
}

// A batch of frames that share the plan's transfer kernel: frame f at image + f*im_stride, out + f*out_stride (floats)
struct Batch {
  int frames = 1;
  size_t im_stride = 0, out_stride = 0;
};


// Summing workgroups that run beside the patches from the start of a fused launch (multiple of 8: one per XCD), by the
// amount of work in the launch.  256-pixel plan (512-thread workgroups, one per CU): r02y, a band of 520 patches 130-138 us
// with 8, 131-146 with 32; r02av, plane stores kept in the Infinity Cache: 4096^2 0.193 / 0.190 / 0.196 ms with 8 / 16 / 24,
// 8192^2 0.78 / 0.77 / 0.74 / 0.765 ms with 8 / 16 / 32 / 48 - the sooner a tile is summed, the likelier its planes are still
// cached.  128-pixel plan (128-thread workgroups, four per CU): 4096^2 0.217 / 0.214 / 0.212 / 0.206 ms with 16 / 64 / 96 / 128
// against 0.218 ms with the separate sum kernel; 2048^2 0.067 ms with 0 ... 24 against 0.069; 8 x 2048^2 0.372 / 0.360 / 0.352 /
// 0.343 / 0.340 ms with 8 / 32 / 64 / 128 / 192 against 0.362.
static int sum_first_for(const rpsf_plan* p, int frames) {
  if (p->sum_first >= 0) return p->sum_first;
  const long work = (long)p->n_patches * frames;
  if (p->N == 128) return work >= 4096 ? 160 : work >= 2048 ? 128 : work >= 512 ? 8 : 0;  // (160 from 4096 patch-frames on: -2 ... -4 %, profiles/r04bf)
  return work >= 2048 ? 32 : work >= 1024 ? 16 : work >= 512 ? 8 : 0;
}

// What the persistent kernels are compiled for (patch_body2's HOT instantiation has no pixel-by-pixel rim paths): every 16-byte unit
// of a patch - four pixels of one row starting at a column that is a multiple of 4 - maps under np.pad's index map to four consecutive
// image columns (ascending or descending) or to the fill.  True for 'constant', 'symmetric' and 'wrap' when the width is a multiple of
// 4 (no unit straddles an image edge or a reflection); 'reflect' and 'edge' tear units apart.  Other launches take patch_kernel2.
// (self-contained: the patch columns themselves - lattice origin + origin_col - and the plane stride are checked here too, not left to the
// `fused` predicate of launch_apply, so that relaxing that one can never hand the HOT kernels a unit they have no path for)
static bool hot_geometry(const rpsf_plan* p, const float* d_img, const rpsf_geometry& g, size_t im_stride) {
  return (g.pad_mode == RPSF_PAD_CONSTANT || g.pad_mode == RPSF_PAD_SYMMETRIC || g.pad_mode == RPSF_PAD_WRAP) && g.width % 4 == 0 &&
         g.ld_image % 4 == 0 && g.origin_col % 4 == 0 && im_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(d_img) & 15) == 0 &&
         p->lattice && ((long)p->lat_c0 + g.origin_col) % 4 == 0 && p->planes_floats % 4 == 0;
}

// fused: the plane sum runs in this launch (see rpsf_plan::d_tile_done)
static int launch_patches(rpsf_plan* p, const float* d_img, float* d_out, const rpsf_geometry& g, OverlapKind kind,
                          hipStream_t st, Batch b = Batch(), bool fused = false) {
  auto fill = [&](PatchParams& pp, int teams) {
    const int count = p->n_patches;
    pp.im = ImageView{d_img, g.height, g.width, g.ld_image, g.pad_mode, g.pad_value, g.image_row0, g.image_rows};
    if (kind != OV_ATOMIC)
      pp.ov = OutView{p->d_planes, g.height, g.width, g.width, g.out_row0, g.out_rows, p->planes_floats, p->d_sink};
    else
      pp.ov = OutView{d_out, g.height, g.width, g.ld_out, g.out_row0, g.out_rows, 0, p->d_sink};
    pp.origin_row = g.origin_row, pp.origin_col = g.origin_col;
    pp.desc = p->d_desc, pp.n_patches = count, pp.seq_base = 0;
    pp.tab = p->d_tab, pp.pairtab = p->d_pairtab, pp.tw = p->d_tw, pp.win = p->d_win, pp.g = p->d_g, pp.gs = p->d_gs;
    pp.stamps = p->d_stamps;
    pp.chunk = ((count + 7) / 8 + teams - 1) / teams * teams;  // patches per XCD, whole workgroups
    pp.stagger_ticks = std::max(0, p->stagger_us) * 100;
    pp.n_frames = b.frames, pp.im_frame_floats = b.im_stride;
    pp.ov_frame_floats = kind != OV_ATOMIC ? 4 * p->planes_floats : b.out_stride;
    pp.dv = OutView{nullptr, 0, 0, 0, 0, 0, 0, nullptr};
    if (kind == OV_DIRECT) {
      pp.dv = OutView{d_out, g.height, g.width, g.ld_out, g.out_row0, g.out_rows, 0, p->d_sink};
      pp.dv_frame_floats = b.out_stride;
      pp.quads = p->d_quads, pp.flags = p->d_flags, pp.dyn_side = p->d_dyn, pp.chunk_xcc = p->d_chunk_xcc;
      pp.flag_epoch = p->epoch, pp.n_tiles = (uint32_t)(p->nti * p->ntj), pp.orphan_mod = p->orphan_mod;
    }
  };
  if (p->v2)
    return dispatch_v2(p->N, [&]<class C>() -> int {
      PatchParams pp{};
      fill(pp, 1);
      pp.stagger_blocks = p->round_capacity;
      size_t blocks = (size_t)8 * pp.chunk * b.frames;
      if (blocks > 0x7fffffffu) return fail(RPSF_E_BADARG, "batch too large for one launch");
      pp.slot0 = 0, pp.patch_blocks = (int)blocks;
      if (fused) {
        const int n_tiles = p->nti * p->ntj;
        pp.tile_done = p->d_tile_done, pp.quads = p->d_quads, pp.n_tiles = (uint32_t)n_tiles;
        TileSum& ts = pp.ts;
        ts.planes = p->d_planes, ts.plane_stride = p->planes_floats, ts.ld_planes = g.width;
        ts.out = d_out, ts.ld_out = g.ld_out;
        ts.rows = g.out_rows, ts.W = g.width, ts.row0 = g.out_row0;
        ts.lat_r0 = p->lat_r0 + g.origin_row, ts.lat_c0 = p->lat_c0 + g.origin_col, ts.half = p->N / 2, ts.ntj = p->ntj;
        ts.cover = p->d_cover, ts.tiles = p->d_sum_order, ts.count = n_tiles * b.frames;
        ts.done = p->d_tile_done, ts.epoch = p->done_epoch;
        ts.n_frames = b.frames, ts.n_tiles = (uint32_t)n_tiles, ts.n_tiles_listed = (uint32_t)n_tiles;
        // Persistent batches whose planes would not fit the Infinity Cache side by side run frame after frame in one launch: every
        // frame keeps the cache behaviour of a single apply, and the next frame's patches take the CUs the previous one's last
        // patches leave idle (RPSF_FRAME_MAJOR=0/1 overrides).
        bool frame_major = std::is_same_v<C, Cfg256v2> && p->persist && b.frames > 1 && p->n_patches >= 1024 &&
                           16.0 * (double)p->planes_floats * b.frames > 256.0 * 1048576.0;  // (8 x 2048^2: 0.056 ms per frame side by side, 0.061 in turn)
        if (const char* e = dev_env("RPSF_FRAME_MAJOR"))  // development sweeps: 0 = never, 2 = any persistent batch
          frame_major = std::atoi(e) == 2 ? (p->persist && b.frames > 1) : (frame_major && std::atoi(e) != 0);
        pp.frame_major = ts.frame_major = frame_major ? 1 : 0;
        // Frames of a batch side by side keep the planes of ALL of them live at once (8 x 2048^2: 537 MB against a 256 MiB Infinity Cache): the 128-pixel
        // kernels then store them with the streaming hint - 0.3156 -> 0.3009 ms (-4.7 %); a single frame loses 8 % that way, and so does the 256-pixel
        // plan at either size (profiles/r04bd).  RPSF_PLANE_NT=0/1 overrides (development sweeps).
        // (needs the plain-load form of K - a batch shares it - and planes well beyond the Infinity Cache: 4 x 2048^2, 268 MB, still loses 3 %; 6 x, 403 MB, gains 5 %)
        pp.plane_nt = std::is_same_v<C, Cfg128v2> && p->k_cached && b.frames > 1 && !frame_major && 16.0 * (double)p->planes_floats * b.frames > 300.0 * 1048576.0;
        if (p->plane_nt_opt >= 0) pp.plane_nt = p->plane_nt_opt != 0 && std::is_same_v<C, Cfg128v2> && p->k_cached;  // (RPSF_OPT_PLANE_NT)
        const int tune_frames = b.frames;  // (the settings of a single apply measured worse here: 0.203 vs 0.186 ms per frame at 8 x 4096^2)
        pp.sum_first = sum_first_for(p, tune_frames);
        ts.planes_frame_floats = 4 * p->planes_floats, ts.out_frame_floats = b.out_stride;
        int nsum = std::max(8, std::min(ts.count, p->round_capacity));  // at the tail: as many summing workgroups as the chip holds
        ts.queue = p->d_sum_queue, ts.queue_base = p->sum_queue_base;
        {
          // persistent form: as many patch workgroups as the chip holds beside the summing ones; each works through the slots
          // of its XCD's chunk and ends as a summing workgroup itself (no workgroups behind the patches).
          // FORWARD PROGRESS.  HIP promises neither that a grid is resident as a whole nor an order of dispatch, and other launches
          // (a second plan on another stream, RCCL's kernels) may hold CUs.  What this launch needs: (1) every patch slot is drawn
          // from its chunk's queue - none is tied to a particular workgroup - so the slots of chunk x are worked off by whichever
          // workgroups with blockIdx % 8 == x are resident, and a workgroup turns to summing only when its chunk's queue is empty,
          // i.e. when every remaining patch of the chunk is in the hands of a resident workgroup that does not wait for anything;
          // (2) summing workgroups wait (poll + s_sleep) only for tiles whose patches are drawn or will be drawn by (1).  So the
          // launch completes as soon as, for every chunk, ONE patch workgroup gets a CU: in dispatch order the first sum_first + 8
          // workgroups.  The only workgroups that hold a CU without progress of their own are the sum_first head summing ones (<= 32 of
          // 512 threads for the 256-pixel plan; up to 160 of 128 threads - four to a CU, 40 CUs' worth - for the 128-pixel plan from 4096
          // patch-frames on, sum_first_for); concurrent persistent launches therefore cannot starve one another unless their head summing
          // workgroups alone fill the chip (tests: test_two_persistent_plans_on_two_streams, 256- and 128-pixel plans).
          // (queue positions of an XCD: chunk slots x frames, the frames of a slot side by side)
          const int rows = std::min(pp.chunk * b.frames, std::max(1, (p->round_capacity - pp.sum_first - p->reserved_cus) / 8));
          if (p->persist && rows > 0 && hot_geometry(p, d_img, g, b.im_stride)) {
            pp.persist = rows, pp.xq = p->d_xq;
            // a head summing workgroup would idle through the first patch period (no tile is complete before that): it computes one patch
            // of its XCD's chunk first (RPSF_HEAD_PATCHES=0: off)
            pp.head_patches = p->head_patches;
            pp.prefetch = p->prefetch && p->d_prefetch_tiles ? 1 : 0;
            pp.prefetch_tiles = p->d_prefetch_tiles;
            for (int x = 0; x < 9; ++x) pp.prefetch_first[x] = p->prefetch_first[x];
            // persistent workgroups keep the phase they start with: holding the resident ones back by up to 10 us spreads the
            // store bursts of the chip over the patch period (profiles/r02ai, r02ak: -2..3 % from four rounds of patches on; with the
            // plane stores kept in the Infinity Cache, r02av: 0.210 / 0.208 / 0.193 / 0.190 / 0.189 / 0.191 / 0.195 ms at 0 / 5 / 8 / 10 / 12 / 15 / 20 us)
            // (and smaller launches too: 2048^2 0.0811 -> 0.0787 ms, 3072^2 0.138 -> 0.131 ms, a band of 520 patches 128 -> 121 us)
            // (long launches take more: 8192^2 0.736 / 0.726 / 0.714 / 0.703 / 0.698 / 0.719 ms at 6 / 12 / 18 / 24 / 30 / 36 us)
            // (round 4, seven-barrier pass, profiles/r04az: 4096^2 at 12 / 16 / 20 us - the repeated frame 0.1835 / 0.1843 / 0.1880 ms, a NEW frame every step
            // 0.2019 / 0.1962 / 0.1933: 16 us from 1024 patches of the 256-pixel plan on; 8192^2 0.714 / 0.688 / 0.691 / 0.690 at 12 / 16 / 20 / 24)
            if (p->stagger_us < 0 && (long)p->n_patches * tune_frames >= 256) {
              const long work = (long)p->n_patches * tune_frames;
              pp.stagger_ticks = work >= 2048 ? 2400 : (work >= 1024 && std::is_same_v<C, Cfg256v2>) ? 1600 : 1200;
              // The 128-pixel plan - four workgroups per CU, out of step with one another anyway - is better off WITHOUT it since round 4 (profiles/r04ba):
              // 2048^2 0.0655 -> 0.0603 ms (-8 %), 3072^2 -6 %, 8 x 2048^2 0.3252 -> 0.3198; only single frames of 4096^2 and more still take a short one
              // (6 us: -1 ... -3 %).
              if constexpr (std::is_same_v<C, Cfg128v2>) pp.stagger_ticks = tune_frames == 1 && p->n_patches >= 4096 ? 600 : 0;
            }
            for (int x = 0; x < 8; ++x) {
              pp.xq_base[x] = p->xq_base[x];
              // draws of this launch: one per slot and frame, plus the one past the end that tells each of the chunk's workgroups to stop
              // (the first sum_first / 8 positions of a chunk go to the head summing workgroups without a draw when those compute a patch first)
              const int slots_x = std::min(pp.chunk, std::max(0, p->n_patches - x * pp.chunk)) * b.frames;
              p->xq_base[x] += (uint32_t)(std::max(0, slots_x - (pp.head_patches ? pp.sum_first / 8 : 0)) + rows);
            }
            const int wgs = pp.sum_first + 8 * rows;
            p->sum_queue_base += (uint32_t)(ts.count + wgs);  // every workgroup draws one position past the end
            if (p->k_cached && pp.plane_nt)
              PersistentKernel2<C>::fn_k_cached_planes_nt<<<dim3((unsigned)wgs), dim3(Launch2<C>::WG), Launch2<C>::LDS_BYTES, st>>>(pp);
            else if (p->k_cached)
              PersistentKernel2<C>::fn_k_cached<<<dim3((unsigned)wgs), dim3(Launch2<C>::WG), Launch2<C>::LDS_BYTES, st>>>(pp);
            else
              PersistentKernel2<C>::fn<<<dim3((unsigned)wgs), dim3(Launch2<C>::WG), Launch2<C>::LDS_BYTES, st>>>(pp);
            HIP_TRY(hipGetLastError());
            return RPSF_OK;
          }
        }
        nsum += pp.sum_first;
        p->sum_queue_base += (uint32_t)(ts.count + nsum);  // every workgroup draws one position past the end
        blocks += (size_t)nsum;
      }
      patch_kernel2<C><<<dim3((unsigned)blocks), dim3(Launch2<C>::WG), Launch2<C>::LDS_BYTES, st>>>(pp);
      HIP_TRY(hipGetLastError());
      return RPSF_OK;
    });
  return dispatch_n(p->N, [&]<class C>() -> int {
    PatchParams pp{};
    constexpr int TEAMS = Launch<C>::TEAMS;
    fill(pp, TEAMS);
    pp.stagger_blocks = p->cu_count * std::max(1, 512 / Launch<C>::WG);
    const size_t blocks = (size_t)8 * (pp.chunk / TEAMS) * b.frames;
    if (blocks > 0x7fffffffu) return fail(RPSF_E_BADARG, "batch too large for one launch");
    patch_kernel<C><<<dim3((unsigned)blocks), dim3(Launch<C>::WG), Launch<C>::LDS_BYTES, st>>>(pp);
    HIP_TRY(hipGetLastError());
    return RPSF_OK;
  });
}

static int launch_fixup(rpsf_plan* p, float* d_out, const rpsf_geometry& g, hipStream_t st, Batch b) {
  FixParams fp{};
  fp.planes = p->d_planes, fp.plane_stride = p->planes_floats, fp.planes_frame_floats = 4 * p->planes_floats, fp.ld_planes = g.width;
  fp.out = d_out, fp.ld_out = g.ld_out, fp.out_frame_floats = b.out_stride;
  fp.rows = g.out_rows, fp.W = g.width, fp.row0 = g.out_row0;
  fp.lat_r0 = p->lat_r0 + g.origin_row, fp.lat_c0 = p->lat_c0 + g.origin_col, fp.half = p->N / 2, fp.ntj = p->ntj;
  fp.tile_info = p->d_tile_info, fp.flags = p->d_flags, fp.dyn_side = p->d_dyn;
  fp.epoch = p->epoch, fp.n_tiles = (uint32_t)(p->nti * p->ntj);
  for (int f0 = 0; f0 < b.frames; f0 += 65535) {  // grid.y limit
    FixParams q = fp;
    q.planes += (size_t)f0 * q.planes_frame_floats, q.out += (size_t)f0 * q.out_frame_floats;
    q.flags += (size_t)f0 * q.n_tiles, q.dyn_side += (size_t)f0 * q.n_tiles;
    fixup_kernel<<<dim3(fp.n_tiles * FIX_SUB, (unsigned)std::min(65535, b.frames - f0)), dim3(256), 0, st>>>(q);
  }
  HIP_TRY(hipGetLastError());
  return RPSF_OK;
}

static int launch_sum(rpsf_plan* p, float* d_out, const rpsf_geometry& g, int row_begin, int row_end, hipStream_t st,
                      Batch b = Batch()) {
  if (row_end <= row_begin) return RPSF_OK;
  SumParams sp = make_sum_params(p, d_out, g, row_begin, row_end);
  sp.out_frame_floats = b.out_stride;
  size_t total = (size_t)((g.width + 3) / 4) * sp.rows;
  for (int f0 = 0; f0 < b.frames; f0 += 65535) {  // grid.y limit
    SumParams q = sp;
    q.planes += (size_t)f0 * q.planes_frame_floats, q.out += (size_t)f0 * q.out_frame_floats;
      sum_planes_kernel<<<dim3((unsigned)((total + 255) / 256), (unsigned)std::min(65535, b.frames - f0)), dim3(256), 0, st>>>(q);
  }
  HIP_TRY(hipGetLastError());
  return RPSF_OK;
}

// One apply.  ev_k0 / ev_k1 (optional) bracket the patch-kernel launches for timing.
static OverlapKind overlap_kind(const rpsf_plan* p) {
  switch (p->overlap_mode) {
    case 1: return OV_ATOMIC;
    case 2: return OV_PLANES;
    case 3: return OV_DIRECT;
    case 4: return OV_SWEEP;
    default: return p->sweep_ok ? OV_SWEEP : p->lattice ? OV_PLANES : OV_ATOMIC;  // direct stays opt-in until it beats the planes
  }
}
static size_t plane_floats_needed(const rpsf_geometry& g) { return ((size_t)g.out_rows * g.width + 3) & ~(size_t)3; }

static int launch_apply_generic(rpsf_plan* p, const float* d_img, float* d_out, const rpsf_geometry& g, hipStream_t st,
                                hipEvent_t ev_k0, hipEvent_t ev_k1, Batch b) {
  if (g.image_row0 != 0 || g.image_rows != g.height || g.out_row0 != 0 || g.out_rows != g.height)
    return fail(RPSF_E_UNSUPPORTED, "row-band windows need a patch size with a compiled plan (16, 32, 64, 128, 256)");
  if (st != p->stream) {
    HIP_TRY(hipStreamSynchronize(p->stream));  // the FFT plan is bound to one stream at a time
    if (g_hipfft.set_stream(p->fft_plan, st) != 0) return fail(RPSF_E_HIP, "hipfftSetStream failed");
  }
  const size_t per = (size_t)p->N * p->N;
  const float scale = 1.0f / (float)per;  // hipFFT's inverse is unnormalised
  if (ev_k0) HIP_TRY(hipEventRecord(ev_k0, st));
  for (int f = 0; f < b.frames; ++f) {
    const float* img = d_img + (size_t)f * b.im_stride;
    float* out = d_out + (size_t)f * b.out_stride;
    HIP_TRY(hipMemset2DAsync(out, (size_t)g.ld_out * sizeof(float), 0, (size_t)g.width * sizeof(float), g.out_rows, st));
    for (int first = 0; first < p->n_patches; first += p->fft_chunk) {
      GenericGeom gg;
      gg.N = p->N, gg.first = first, gg.count = std::min(p->fft_chunk, p->n_patches - first);
      gg.origin_row = g.origin_row, gg.origin_col = g.origin_col;
      gg.im = ImageView{img, g.height, g.width, g.ld_image, g.pad_mode, g.pad_value, 0, g.height};
      gg.ov = OutView{out, g.height, g.width, g.ld_out, 0, g.height, 0, nullptr};
      const size_t total = per * gg.count;
      const unsigned grid = (unsigned)((total + 255) / 256);
      generic_gather_kernel<<<dim3(grid), dim3(256), 0, st>>>(gg, p->d_coords, p->d_win_generic, p->d_fft_buf);
      // a short last chunk still runs the full-size plan: the surplus patches hold stale data and are never scattered
      if (g_hipfft.exec_c2c(p->fft_plan, p->d_fft_buf, p->d_fft_buf, /*HIPFFT_FORWARD*/ -1) != 0)
        return fail(RPSF_E_HIP, "hipfftExecC2C (forward) failed");
      generic_multiply_kernel<<<dim3(grid), dim3(256), 0, st>>>(p->d_fft_buf, p->d_kfull + (size_t)first * per, total, scale);
      if (g_hipfft.exec_c2c(p->fft_plan, p->d_fft_buf, p->d_fft_buf, /*HIPFFT_BACKWARD*/ 1) != 0)
        return fail(RPSF_E_HIP, "hipfftExecC2C (inverse) failed");
      if (p->d_colour_generic && overlap_kind(p) != OV_ATOMIC) {
        for (int colour = 0; colour < 4; ++colour)
          generic_scatter_kernel<<<dim3(grid), dim3(256), 0, st>>>(gg, p->d_coords, p->d_win_generic, p->d_fft_buf, p->d_colour_generic, colour);
      } else {
        generic_scatter_kernel<<<dim3(grid), dim3(256), 0, st>>>(gg, p->d_coords, p->d_win_generic, p->d_fft_buf, nullptr, -1);
      }
      HIP_TRY(hipGetLastError());
    }
  }
  if (ev_k1) HIP_TRY(hipEventRecord(ev_k1, st));
  if (st != p->stream) {
    HIP_TRY(hipStreamSynchronize(st));
    if (g_hipfft.set_stream(p->fft_plan, p->stream) != 0) return fail(RPSF_E_HIP, "hipfftSetStream failed");
  }
  return RPSF_OK;
}

// Third generation (N <= 64): the whole apply is this one launch (rpsf_kernels3.hpp); frames of a batch along grid.y
static int launch_sweep(rpsf_plan* p, const float* d_img, float* d_out, const rpsf_geometry& g, hipStream_t st, hipEvent_t ev_k0, hipEvent_t ev_k1,
                        Batch b) {
  const int half = p->N / 2;
  const long r0 = (long)p->lat_r0 + g.origin_row, c0 = (long)p->lat_c0 + g.origin_col;
  // pixels of the resident window that the lattice does not cover are written by nobody: the reference leaves them zero
  if (r0 > g.out_row0 || r0 + (long)p->nti * half < (long)g.out_row0 + g.out_rows || c0 > 0 || c0 + (long)p->ntj * half < g.width)
    for (int f = 0; f < b.frames; ++f)
      HIP_TRY(hipMemset2DAsync(d_out + (size_t)f * b.out_stride, (size_t)g.ld_out * sizeof(float), 0, (size_t)g.width * sizeof(float), g.out_rows, st));
  SweepParams sp{};
  sp.im = ImageView{d_img, g.height, g.width, g.ld_image, g.pad_mode, g.pad_value, g.image_row0, g.image_rows};
  const bool col_aligned = (c0 & 3) == 0;
  sp.aligned_in = col_aligned && g.ld_image % 4 == 0 && b.im_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(d_img) & 15) == 0;
  const int aligned_out = col_aligned && g.ld_out % 4 == 0 && b.out_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0;
  sp.fl = Flush3{d_out, g.ld_out, g.out_row0, g.out_rows, g.height, g.width, aligned_out};
  sp.lat_r0 = (int)r0, sp.lat_c0 = (int)c0;
  sp.jobs = p->d_jobs3, sp.regions = p->d_regions3, sp.n_regions = p->n_regions3, sp.group = (p->n_regions3 + 7) / 8;
  sp.k3 = p->d_k3, sp.win = p->d_win, sp.zeros = p->d_zero3, sp.err = p->d_err3, sp.stamps = p->d_stamps3;
  sp.im_frame_floats = b.im_stride, sp.out_frame_floats = b.out_stride;
  // K by plain loads when frames share it or it is small enough to stay in the Infinity Cache from one apply to the next, else streamed
  const bool k_plain = b.frames > 1 || p->k3_floats * sizeof(float) <= ((size_t)96 << 20);
  if (ev_k0) HIP_TRY(hipEventRecord(ev_k0, st));
  for (int f0 = 0; f0 < b.frames; f0 += 65535) {  // grid.y limit
    SweepParams q = sp;
    q.im.img += (size_t)f0 * b.im_stride, q.fl.out += (size_t)f0 * b.out_stride;
    const dim3 grid((unsigned)(8 * sp.group), (unsigned)std::min(65535, b.frames - f0));
    const int rc = dispatch_v3(p->N, [&]<class C>() -> int {
      if (k_plain) sweep_kernel_kc<C><<<grid, dim3(C::WG), C::LDS_BYTES, st>>>(q);
      else sweep_kernel<C><<<grid, dim3(C::WG), C::LDS_BYTES, st>>>(q);
      HIP_TRY(hipGetLastError());
      return RPSF_OK;
    });
    if (rc != RPSF_OK) return rc;
  }
  if (ev_k1) HIP_TRY(hipEventRecord(ev_k1, st));
  return RPSF_OK;
}

static int launch_apply(rpsf_plan* p, const float* d_img, float* d_out, const rpsf_geometry& g, hipStream_t st,
                        hipEvent_t ev_k0, hipEvent_t ev_k1 = nullptr, Batch b = Batch()) {
  if (p->generic) return launch_apply_generic(p, d_img, d_out, g, st, ev_k0, ev_k1, b);
  const OverlapKind kind = overlap_kind(p);
  if (kind == OV_SWEEP) {
    if (!p->sweep_ok) return fail(RPSF_E_STATE, "the sweep kernel needs a 16-, 32- or 64-pixel patch on a complete lattice of at least 2 x 2 patches");
    return launch_sweep(p, d_img, d_out, g, st, ev_k0, ev_k1, b);
  }
  if (kind != OV_ATOMIC && !p->lattice) return fail(RPSF_E_STATE, "colour planes need a regular half-overlap lattice of patch corners");
  if (kind == OV_DIRECT && !p->direct_ok) return fail(RPSF_E_STATE, "direct overlap-add needs a lattice and a 128- or 256-pixel patch");
  if (kind == OV_DIRECT && (size_t)g.out_rows * g.ld_out * sizeof(float) >= ((size_t)1 << 32))
    return fail(RPSF_E_UNSUPPORTED, "direct overlap-add addresses the output through a 32-bit buffer offset: frame too large");
  // The plan's scratch (planes, flags) serves one apply at a time: an apply on another stream waits for the last one.
  if (p->busy_valid && st != p->last_stream) HIP_TRY(hipStreamWaitEvent(st, p->ev_busy, 0));
  if (kind != OV_ATOMIC) {
    const size_t need = plane_floats_needed(g);
    if (need > p->planes_floats || (size_t)b.frames > p->planes_frames) {  // four planes per frame in flight
      const size_t per = std::max(need, p->planes_floats), frames = std::max((size_t)b.frames, p->planes_frames);
      HIP_TRY(hipDeviceSynchronize());  // earlier applies, on whatever stream, may still use the old planes
      (void)hipFree(p->d_planes);
      p->d_planes = nullptr, p->planes_floats = 0, p->planes_frames = 0;
      HIP_TRY(hipMalloc(&p->d_planes, 4 * per * frames * sizeof(float)));
      p->planes_floats = per, p->planes_frames = frames;
    }
  }
  bool clear = kind == OV_ATOMIC;
  if (kind == OV_DIRECT) {
    const size_t n_tiles = (size_t)p->nti * p->ntj;
    if ((size_t)b.frames > p->flag_frames) {
      HIP_TRY(hipDeviceSynchronize());
      (void)hipFree(p->d_flags);
      (void)hipFree(p->d_dyn);
      p->d_flags = p->d_dyn = nullptr, p->flag_frames = 0;
      HIP_TRY(hipMalloc(&p->d_flags, n_tiles * b.frames * sizeof(uint32_t)));
      HIP_TRY(hipMalloc(&p->d_dyn, n_tiles * b.frames * sizeof(uint32_t)));
      HIP_TRY(hipMemset(p->d_flags, 0, n_tiles * b.frames * sizeof(uint32_t)));
      HIP_TRY(hipMemset(p->d_dyn, 0, n_tiles * b.frames * sizeof(uint32_t)));
      p->flag_frames = (size_t)b.frames;
    }
    if (++p->epoch >= (1u << 24) - 1) {  // the epoch field of the flag words is 24 bits wide
      HIP_TRY(hipMemsetAsync(p->d_flags, 0, n_tiles * p->flag_frames * sizeof(uint32_t), st));
      HIP_TRY(hipMemsetAsync(p->d_chunk_xcc, 0, 8 * sizeof(uint32_t), st));
      p->epoch = 1;
    }
    // pixels of the resident window that no lattice tile covers are never written by the kernels below
    const int half = p->N / 2;
    const long tr0 = (long)p->lat_r0 + g.origin_row, tc0 = (long)p->lat_c0 + g.origin_col;
    clear = tr0 > g.out_row0 || tr0 + (long)p->nti * half < (long)g.out_row0 + g.out_rows || tc0 > 0 ||
            tc0 + (long)p->ntj * half < g.width;
  }
  if (clear)
    for (int f = 0; f < b.frames; ++f)
      HIP_TRY(hipMemset2DAsync(d_out + (size_t)f * b.out_stride, (size_t)g.ld_out * sizeof(float), 0,
                               (size_t)g.width * sizeof(float), g.out_rows, st));
  // Fused plane sum: one frame, every plane line written whole by one store instruction (see sum_tile)
  const long tile_r0 = (long)p->lat_r0 + g.origin_row, tile_c0 = (long)p->lat_c0 + g.origin_col;
  const bool fused = kind == OV_PLANES && p->v2 && p->fuse_pays && p->d_tile_done && !p->no_fuse && b.frames <= 255 && g.width % 32 == 0 &&
                     g.ld_out % 4 == 0 && tile_c0 % 32 == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0 &&
                     16 * plane_floats_needed(g) < ((size_t)1 << 32);  // the planes are addressed through one 32-bit buffer offset
  if (fused) {
    const size_t n_tiles = (size_t)p->nti * p->ntj;
    if ((size_t)b.frames > p->done_frames) {  // one set of tile counters per frame of a batch
      HIP_TRY(hipDeviceSynchronize());
      (void)hipFree(p->d_tile_done);
      p->d_tile_done = nullptr;
      HIP_TRY(hipMalloc(&p->d_tile_done, n_tiles * b.frames * sizeof(uint32_t)));
      HIP_TRY(hipMemset(p->d_tile_done, 0, n_tiles * b.frames * sizeof(uint32_t)));
      p->done_frames = (size_t)b.frames, p->done_epoch = 0;
    }
    // A launch advances the counters of frames [0, b.frames) only, and a tile is complete at epoch x contributors: when the frame
    // count differs from the previous fused launch's (batch -> single apply -> batch, or the short last group of a batch) the
    // counters of the higher frames would lag behind the epoch for ever - and the summing workgroups would wait for ever.
    // Start a new count whenever the frame count changes (and when the epoch would overflow the counters).
    if (b.frames != p->done_last_frames || p->done_epoch + 1 >= (1u << 29)) {  // the counters hold epoch * contributors
      HIP_TRY(hipMemsetAsync(p->d_tile_done, 0, n_tiles * p->done_frames * sizeof(uint32_t), st));
      p->done_epoch = 0;
    }
    p->done_last_frames = b.frames;
    ++p->done_epoch;
    // the tile sums write lattice tiles only: pixels of the window outside the tile grid are cleared here
    const int half = p->N / 2;
    if (tile_r0 > g.out_row0 || tile_r0 + (long)p->nti * half < (long)g.out_row0 + g.out_rows || tile_c0 > 0 ||
        tile_c0 + (long)p->ntj * half < g.width)
      for (int f = 0; f < b.frames; ++f)
        HIP_TRY(hipMemset2DAsync(d_out + (size_t)f * b.out_stride, (size_t)g.ld_out * sizeof(float), 0, (size_t)g.width * sizeof(float),
                                 g.out_rows, st));
  }
  if (ev_k0) HIP_TRY(hipEventRecord(ev_k0, st));
  int rc = launch_patches(p, d_img, d_out, g, kind, st, b, fused);
  if (rc != RPSF_OK) return rc;
  if (ev_k1) HIP_TRY(hipEventRecord(ev_k1, st));
  if (kind == OV_PLANES && !fused) rc = launch_sum(p, d_out, g, 0, g.out_rows, st, b);
  if (kind == OV_DIRECT) rc = launch_fixup(p, d_out, g, st, b);
  if (rc != RPSF_OK) return rc;
  HIP_TRY(hipEventRecord(p->ev_busy, st));
  p->last_stream = st, p->busy_valid = true;
  return RPSF_OK;
}

extern "C" int rpsf_plan_set_overlap_mode(rpsf_plan* p, int mode) {
  if (!p || mode < 0 || mode > 4) return fail(RPSF_E_BADARG, "mode must be 0 (auto), 1 (atomics), 2 (colour planes), 3 (direct) or 4 (sweep)");
  if (mode == 4 && !p->sweep_ok) return fail(RPSF_E_STATE, "the sweep kernel needs a 16-, 32- or 64-pixel patch on a complete lattice of at least 2 x 2 patches");
  if (mode == 2 && !p->lattice) return fail(RPSF_E_STATE, "colour planes need a regular half-overlap lattice of patch corners");
  if (mode == 3 && !p->direct_ok) return fail(RPSF_E_STATE, "direct overlap-add needs a regular half-overlap lattice and a 128- or 256-pixel patch");
  p->overlap_mode = mode;
  drop_bands(p);
  return RPSF_OK;
}

// Diagnostic builds only (-DRPSF_STAMPS): copy out the 16 per-patch phase timestamps (10 ns ticks).
extern "C" int rpsf_plan_set_option(rpsf_plan* p, int option, int value) {
  if (!p) return fail(RPSF_E_BADARG, "null plan");
  switch (option) {
    case RPSF_OPT_PERSIST:
      if (value != 0 && value != 1) return fail(RPSF_E_BADARG, "RPSF_OPT_PERSIST takes 0 or 1");
      p->persist = value != 0 && (p->N == 256 || p->N == 128);
      break;
    case RPSF_OPT_FUSE:
      if (value != 0 && value != 1) return fail(RPSF_E_BADARG, "RPSF_OPT_FUSE takes 0 or 1");
      p->no_fuse = value == 0;
      break;
    case RPSF_OPT_K_CACHED:
      if (value != 0 && value != 1) return fail(RPSF_E_BADARG, "RPSF_OPT_K_CACHED takes 0 or 1");
      p->k_cached = value != 0;
      break;
    case RPSF_OPT_PLANE_NT:
      if (value < -1 || value > 1) return fail(RPSF_E_BADARG, "RPSF_OPT_PLANE_NT takes -1 (automatic), 0 or 1");
      p->plane_nt_opt = value;
      break;
    case RPSF_OPT_HOST_BANDS:
      if (value < -1 || value > 64) return fail(RPSF_E_BADARG, "RPSF_OPT_HOST_BANDS takes -1 (automatic) or 0..64");
      p->host_bands_opt = value;
      break;
    case RPSF_OPT_STREAM_GROUP:
      if (value < 0 || value > 64) return fail(RPSF_E_BADARG, "RPSF_OPT_STREAM_GROUP takes 0 (automatic) or 1..64");
      p->stream_group_opt = value;
      break;
    case RPSF_OPT_STREAM_DEPTH:
      if (value < 0 || value > 16) return fail(RPSF_E_BADARG, "RPSF_OPT_STREAM_DEPTH takes 0 (automatic) or 1..16");
      p->stream_depth_opt = value;
      break;
    case RPSF_OPT_DEBUG_ORPHAN:
      if (value < 0) return fail(RPSF_E_BADARG, "RPSF_OPT_DEBUG_ORPHAN takes 0 (off) or a positive modulus");
      p->orphan_mod = value;
      break;
    default: return fail(RPSF_E_BADARG, "unknown plan option");
  }
  drop_bands(p);  // (views inherit nothing: they are rebuilt with the parent's settings at the next host frame)
  return RPSF_OK;
}

extern "C" int rpsf_plan_set_sweep_regions(rpsf_plan* p, int target_regions) {
  if (!p || target_regions < 1 || target_regions > (1 << 20)) return fail(RPSF_E_BADARG, "target_regions must be 1..2^20");
  if (!p->sweep_ok) return fail(RPSF_E_STATE, "the sweep kernel needs a 16-, 32- or 64-pixel patch on a complete lattice of at least 2 x 2 patches");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipStreamSynchronize(p->stream));
  HIP_TRY(hipDeviceSynchronize());  // (applies on other streams may still read the old lists)
  drop_bands(p);
  return build_sweep_lists(p, target_regions);
}
extern "C" int rpsf_plan_host_bands(const rpsf_plan* p, int* bands) {
  if (!p || !bands) return fail(RPSF_E_BADARG, "null argument");
  *bands = p->last_host_bands;
  return RPSF_OK;
}
extern "C" int rpsf_plan_sweep_info(const rpsf_plan* p, int* regions, long* jobs, long* patch_slots, int* slabs_per_phase) {
  if (!p) return fail(RPSF_E_BADARG, "null plan");
  if (regions) *regions = p->sweep_ok ? p->n_regions3 : 0;
  if (jobs) *jobs = p->sweep_ok ? p->slabs3 : 0;
  if (patch_slots) *patch_slots = p->sweep_ok ? p->patch_slots3 : 0;
  if (slabs_per_phase) *slabs_per_phase = p->sweep_ok ? p->ks3 : 0;
  return RPSF_OK;
}

extern "C" int rpsf_plan_debug_stamps(rpsf_plan* p, unsigned long long* host, size_t count) {
  if (!p || !host) return fail(RPSF_E_BADARG, "null argument");
  if (p->d_stamps3) {  // third generation: [region][wave][job slot < 8][16]
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());
    count = std::min(count, (size_t)std::max(512, p->n_regions3) * 8 * 8 * 16);
    HIP_TRY(hipMemcpy(host, p->d_stamps3, count * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RPSF_OK;
  }
  if (!p->d_stamps) return fail(RPSF_E_STATE, "phase timestamps exist only in builds with -DRPSF_STAMPS");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipDeviceSynchronize());
#if defined(RPSF_WAVE_STAMPS)
  count = std::min(count, (size_t)16 * 8 * p->n_patches);
#else
  count = std::min(count, (size_t)16 * p->n_patches);
#endif
  HIP_TRY(hipMemcpy(host, p->d_stamps, count * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return RPSF_OK;
}


extern "C" int rpsf_plan_set_reserved_cus(rpsf_plan* p, int cus) {
  if (!p || cus < 0 || cus > 128) return fail(RPSF_E_BADARG, "reserved CUs must be 0..128");
  p->reserved_cus = cus;
  return RPSF_OK;
}

// Confine the plan's own stream - and with it every launch the plan makes on it - to the compute units of `mask` (bit i of word i / 32:
// hipExtStreamCreateWithCUMask's numbering), and size its persistent launches for that many.  With a second stream masked to the complement
// (rpsf_stream_create) a caller gets two partitions of the chip that cannot take each other's CUs: the persistent patch launch needs whole
// CUs, and small kernels dispatched beside it on an unmasked stream land on CUs its workgroups are waiting for (profiles/r05m).
extern "C" int rpsf_plan_set_cu_mask(rpsf_plan* p, const uint32_t* mask, int words) {
  if (!p || !mask || words <= 0) return fail(RPSF_E_BADARG, "bad argument");
  if (p->parent) return fail(RPSF_E_STATE, "a view runs on its parent's stream");
  int cus = 0;
  for (int i = 0; i < words; ++i) cus += __builtin_popcount(mask[i]);
  if (cus < 8) return fail(RPSF_E_BADARG, "the mask must leave the plan at least 8 compute units");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipStreamSynchronize(p->stream));
  drop_bands(p);
  hipStream_t masked = nullptr;
  HIP_TRY(hipExtStreamCreateWithCUMask(&masked, (uint32_t)words, mask));
  if (p->generic && g_hipfft.set_stream(p->fft_plan, masked) != 0) {
    (void)hipStreamDestroy(masked);
    return fail(RPSF_E_HIP, "hipfftSetStream failed");
  }
  (void)hipStreamDestroy(p->stream);
  p->stream = masked;
  p->busy_valid = false;
  if (p->cu_count > 0) p->round_capacity = p->round_capacity / p->cu_count * cus;
  p->cu_count = cus;
  return RPSF_OK;
}

extern "C" int rpsf_stream_create(int device, const uint32_t* mask, int words, void** stream) {
  if (!stream) return fail(RPSF_E_BADARG, "null argument");
  HIP_TRY(hipSetDevice(device));
  hipStream_t st = nullptr;
  if (mask && words > 0)
    HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask));
  else
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *stream = st;
  return RPSF_OK;
}
extern "C" int rpsf_stream_destroy(void* stream) {
  if (stream) HIP_TRY(hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)));
  return RPSF_OK;
}

extern "C" int rpsf_plan_set_image_prefetch(rpsf_plan* p, int on) {
  if (!p) return fail(RPSF_E_BADARG, "null plan");
  p->prefetch = on != 0;
  return RPSF_OK;
}

extern "C" int rpsf_plan_set_stagger(rpsf_plan* p, int microseconds) {
  if (!p || microseconds < 0 || microseconds > 1000) return fail(RPSF_E_BADARG, "stagger must be 0..1000 us");
  p->stagger_us = microseconds;
  drop_bands(p);
  return RPSF_OK;
}


extern "C" int rpsf_apply_device(rpsf_plan* p, const void* image_dev, void* out_dev, const rpsf_geometry* geom,
                                 void* stream) {
  if (!p || !image_dev || !out_dev) return fail(RPSF_E_BADARG, "null argument");
  if (!p->have_k) return fail(RPSF_E_STATE, "no transfer kernel installed (call rpsf_plan_set_transfer first)");
  int rc = check_geometry(p, geom);
  if (rc != RPSF_OK) return rc;
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : p->stream;
  return launch_apply(p, reinterpret_cast<const float*>(image_dev), reinterpret_cast<float*>(out_dev), *geom, st, nullptr);
}


// Frames per launch group: the colour planes take 16 bytes per output pixel per frame in flight; keep
// them under a quarter of the device memory.
static int batch_group_frames(const rpsf_plan* p, const rpsf_geometry& g, int n_frames) {
  if (p->generic || overlap_kind(p) == OV_ATOMIC || overlap_kind(p) == OV_SWEEP) return n_frames;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)64 << 30;
  const size_t per_frame = 16 * plane_floats_needed(g);
  return (int)std::max<size_t>(1, std::min<size_t>((size_t)n_frames, (total_b / 4) / per_frame));
}

static int check_batch(rpsf_plan* p, const void* a, const void* b, int n_frames, size_t im_stride, size_t out_stride,
                       const rpsf_geometry* geom) {
  if (!p || !a || !b) return fail(RPSF_E_BADARG, "null argument");
  if (n_frames <= 0) return fail(RPSF_E_BADARG, "n_frames must be positive");
  if (!p->have_k) return fail(RPSF_E_STATE, "no transfer kernel installed (call rpsf_plan_set_transfer first)");
  int rc = check_geometry(p, geom);
  if (rc != RPSF_OK) return rc;
  if (n_frames > 1 && (im_stride < (size_t)(geom->image_rows - 1) * geom->ld_image + geom->width ||
                       out_stride < (size_t)(geom->out_rows - 1) * geom->ld_out + geom->width))
    return fail(RPSF_E_BADARG, "frame stride smaller than one frame");
  return RPSF_OK;
}

static int launch_batch(rpsf_plan* p, const float* d_imgs, float* d_outs, int n_frames, size_t im_stride, size_t out_stride,
                        const rpsf_geometry& g, hipStream_t st, hipEvent_t ev_k0 = nullptr, hipEvent_t ev_k1 = nullptr) {
  const int group = batch_group_frames(p, g, n_frames);
  for (int f0 = 0; f0 < n_frames; f0 += group) {
    Batch b;
    b.frames = std::min(group, n_frames - f0), b.im_stride = im_stride, b.out_stride = out_stride;
    const bool first = f0 == 0, last = f0 + group >= n_frames;
    // (with several groups the kernel-time bracket covers everything between the first group's patch launch
    //  and the last group's, plane sums of the earlier groups included)
    int rc = launch_apply(p, d_imgs + (size_t)f0 * im_stride, d_outs + (size_t)f0 * out_stride, g, st,
                          first ? ev_k0 : nullptr, last ? ev_k1 : nullptr, b);
    if (rc != RPSF_OK) return rc;
  }
  return RPSF_OK;
}

extern "C" int rpsf_apply_batch_device(rpsf_plan* p, const void* images_dev, void* outs_dev, int n_frames,
                                       size_t image_stride, size_t out_stride, const rpsf_geometry* geom, void* stream) {
  int rc = check_batch(p, images_dev, outs_dev, n_frames, image_stride, out_stride, geom);
  if (rc != RPSF_OK) return rc;
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : p->stream;
  return launch_batch(p, reinterpret_cast<const float*>(images_dev), reinterpret_cast<float*>(outs_dev), n_frames,
                      image_stride, out_stride, *geom, st);
}

// `iters` applies BACK TO BACK on the plan's stream, each bracketed by its own events (four per iteration, created here and destroyed on
// return), one synchronisation at the end: the launches queue up behind one another exactly as in a caller's loop, so the event times are
// those of the loop's kernels (with a host synchronisation after every apply, as until round 4, every launch started on an idle device
// and read 2 % longer than the wall-clock step that contains it).  total_ms[i]: the whole apply i, kernel_ms[i]: its patch-kernel launches.
template <class Launch>
static int timed_loop(rpsf_plan* p, int iters, float* total_ms, float* kernel_ms, Launch&& launch) {
  std::vector<hipEvent_t> ev((size_t)4 * iters, nullptr);
  auto cleanup = [&] {
    for (auto e : ev)
      if (e) (void)hipEventDestroy(e);
  };
  int rc = RPSF_OK;
  hipError_t err = hipSuccess;
  for (auto& e : ev)
    if (err == hipSuccess) err = hipEventCreate(&e);
  for (int i = 0; i < iters && err == hipSuccess && rc == RPSF_OK; ++i) {
    err = hipEventRecord(ev[4 * i], p->stream);
    if (err == hipSuccess) rc = launch(ev[4 * i + 1], ev[4 * i + 2]);
    if (err == hipSuccess && rc == RPSF_OK) err = hipEventRecord(ev[4 * i + 3], p->stream);
  }
  if (err == hipSuccess && rc == RPSF_OK) err = hipStreamSynchronize(p->stream);
  for (int i = 0; i < iters && err == hipSuccess && rc == RPSF_OK; ++i) {
    float ms = 0.f;
    if (total_ms) {
      err = hipEventElapsedTime(&ms, ev[4 * i], ev[4 * i + 3]);
      total_ms[i] = ms;
    }
    if (kernel_ms && err == hipSuccess) {
      err = hipEventElapsedTime(&ms, ev[4 * i + 1], ev[4 * i + 2]);
      kernel_ms[i] = ms;
    }
  }
  if (err != hipSuccess || rc != RPSF_OK) (void)hipStreamSynchronize(p->stream);
  cleanup();
  if (rc != RPSF_OK) return rc;
  if (err != hipSuccess) return fail(RPSF_E_HIP, std::string("timed applies: ") + hipGetErrorString(err));
  return RPSF_OK;
}

extern "C" int rpsf_apply_batch_device_timed(rpsf_plan* p, const void* images_dev, void* outs_dev, int n_frames,
                                             size_t image_stride, size_t out_stride, const rpsf_geometry* geom, int iters,
                                             float* total_ms, float* kernel_ms) {
  if (iters <= 0) return fail(RPSF_E_BADARG, "bad argument");
  int rc = check_batch(p, images_dev, outs_dev, n_frames, image_stride, out_stride, geom);
  if (rc != RPSF_OK) return rc;
  HIP_TRY(hipSetDevice(p->device));
  return timed_loop(p, iters, total_ms, kernel_ms, [&](hipEvent_t k0, hipEvent_t k1) {
    return launch_batch(p, reinterpret_cast<const float*>(images_dev), reinterpret_cast<float*>(outs_dev), n_frames, image_stride,
                        out_stride, *geom, p->stream, k0, k1);
  });
}

// ------------------------------------------------------------------------------------------------
// Host arrays in, host arrays out (what ArrayPSFTransform.apply hands over and gets back: transform.py:117 astype on the
// way in, :174-177 a float64 result on the way out).  Nothing below creates a thread or allocates per call once a plan
// has seen its frame size: conversions run on the persistent pool, staging lives in the plan's HostPipe.
// ------------------------------------------------------------------------------------------------
using rpsf_host::HostPipe;
using rpsf_host::HostPool;

static int pipe_ensure(rpsf_plan* p, size_t slot_floats, int depth) {
  if (!p->pipe) p->pipe = new HostPipe;
  HostPipe& q = *p->pipe;
  if (!q.st_in) {
    HIP_TRY(hipStreamCreateWithFlags(&q.st_in, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&q.st_out, hipStreamNonBlocking));
    for (auto* evs : {q.ev_in, q.ev_k, q.ev_out})
      for (int s = 0; s < HostPipe::MAX_DEPTH; ++s) HIP_TRY(hipEventCreateWithFlags(&evs[s], hipEventDisableTiming));
    for (auto& e : q.ev_chunk) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto* evs : {q.ev_band_in, q.ev_band_k})
      for (int s = 0; s < HostPipe::MAX_BANDS; ++s) HIP_TRY(hipEventCreateWithFlags(&evs[s], hipEventDisableTiming));
  }
  if (slot_floats <= q.slot_floats && depth <= q.depth) return RPSF_OK;
  // grow (every host entry point drains its own work before it returns: nothing is in flight on these buffers)
  const size_t want = std::max(slot_floats, q.slot_floats);
  const int d = std::max(depth, q.depth);
  HIP_TRY(hipStreamSynchronize(p->stream));
  q.release_buffers();
  for (int s = 0; s < d; ++s) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&q.h_in[s]), want * sizeof(float), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&q.h_out[s]), want * sizeof(float), hipHostMallocDefault));
    HIP_TRY(hipMalloc(&q.d_in[s], want * sizeof(float)));
    HIP_TRY(hipMalloc(&q.d_out[s], want * sizeof(float)));
  }
  q.depth = d, q.slot_floats = want;
  return RPSF_OK;
}

// Parts of one pool job over `bytes` of staging: a few per thread, so that a thread on a slow core (or the caller's, which may
// sit on the other socket) simply takes fewer of them; at least 128 KiB each
static int host_parts_for(size_t bytes) {
  return (int)std::min<size_t>((size_t)HostPool::get().width() * 4, std::max<size_t>(1, bytes >> 17));
}

// Frames the caller keeps in page-locked memory (rpsf_host_alloc, or memory registered with hipHostRegister) need no staging when no dtype
// conversion is due: the copy engines read / write them directly and the host touches no pixel.
static bool is_pinned_host(const void* ptr) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
    (void)hipGetLastError();  // (an ordinary pageable pointer is "invalid value" to the runtime: not an error of ours)
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}
static bool all_pinned(const void* const* ptrs, int n) {
  for (int i = 0; i < n; ++i)
    if (!is_pinned_host(ptrs[i])) return false;
  return true;
}

// The sweep kernel bounds every wait of a job for its predecessors (a protocol error must not hang the GPU) and reports a wait that ran out through a
// page-locked word; every entry point that has just waited for the plan's work looks at it, so such an apply fails instead of returning a wrong image.
static int sweep_check(rpsf_plan* p) {
  if (!p->h_err3 || *reinterpret_cast<volatile uint32_t*>(p->h_err3) == 0) return RPSF_OK;
  *reinterpret_cast<volatile uint32_t*>(p->h_err3) = 0;
  return fail(RPSF_E_HIP, "sweep kernel: a job waited for the jobs it follows beyond the bound (internal protocol error; the output of this apply is not valid)");
}

static int drain_after_error(rpsf_plan* p, hipError_t err, const char* where) {
  (void)hipStreamSynchronize(p->pipe->st_in);
  (void)hipStreamSynchronize(p->stream);
  (void)hipStreamSynchronize(p->pipe->st_out);
  if (err == hipErrorUnknown && !g_err.empty()) return RPSF_E_HIP;  // launch_apply already set the message
  return fail(RPSF_E_HIP, std::string(where) + ": " + hipGetErrorString(err));
}

// Row bands of one large host frame: the lattice rows are cut into `want` groups; band b owns the output rows from its first lattice
// row to the next band's and runs every patch that reaches into them - its own lattice rows and the one above, which both neighbours
// compute (the collective-free seam of regularizepsf_amd/sharding.py, on one GPU) - as a view of the plan.  A band can run as soon as the
// image rows its patches read are on the device, and its rows can leave while the next band is still arriving: upload, patches and download
// of one frame overlap.  Returns the number of bands (0: this plan / frame is not cut - the caller takes the whole-frame path).
static int ensure_bands(rpsf_plan* p, const rpsf_geometry& g, int want) {
  if (p->bands_h == g.height && p->bands_w == g.width && p->bands_mode == g.pad_mode && p->bands_want == want) return (int)p->bands.size();
  drop_bands(p);
  p->bands_h = g.height, p->bands_w = g.width, p->bands_mode = g.pad_mode, p->bands_want = want;  // (remembered also when the answer is "no bands")
  if (p->generic || p->parent || !p->lattice || (overlap_kind(p) != OV_PLANES && overlap_kind(p) != OV_SWEEP) || g.pad_mode == RPSF_PAD_WRAP || want < 2) return 0;
  const int N = p->N, H = g.height;
  std::vector<int> rows;
  for (int i = 0; i < p->n_patches; ++i) rows.push_back(p->h_coords[2 * i]);
  std::sort(rows.begin(), rows.end());
  rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
  const int L = (int)rows.size();
  const int B = std::min({want, (int)HostPipe::MAX_BANDS, L / 2});  // at least two lattice rows per band (it runs a third: the one above)
  if (B < 2) return 0;
  std::vector<int> cut(B + 1);
  // (equal bands: a first band of two lattice rows - an earlier first download - bought nothing, profiles/r06y_host_frame_knobs.log)
  for (int b = 0; b < B; ++b) cut[b] = b == 0 ? 0 : std::min(H, std::max(0, rows[(size_t)L * b / B]));
  cut[B] = H;
  for (int b = 0; b < B; ++b)
    if (cut[b + 1] <= cut[b]) return 0;
  std::vector<rpsf_plan*> bands;
  std::vector<int> in_rows;
  for (int b = 0; b < B; ++b) {
    std::vector<int32_t> idx, coords;
    int in_hi = 0;
    for (int i = 0; i < p->n_patches; ++i) {
      const int r = p->h_coords[2 * i];
      if (r < cut[b + 1] && r + N > cut[b]) {
        idx.push_back(i), coords.push_back(r), coords.push_back(p->h_coords[2 * i + 1]);
        in_hi = std::max(in_hi, std::min(H, r + N));
      }
    }
    rpsf_plan* view = nullptr;
    // (every np.pad mode but 'wrap' maps a row beyond the image edge to a row within the patch's own reach, so rows [0, in_hi) suffice)
    const int rc = idx.empty() ? fail(RPSF_E_STATE, "empty row band") :
                                 plan_create_impl(&view, p->device, N, (int)idx.size(), coords.data(), p, idx.data());
    if (rc != RPSF_OK) {
      for (rpsf_plan* v : bands) rpsf_plan_destroy(v);
      return 0;
    }
    bands.push_back(view), in_rows.push_back(std::max(in_hi, cut[b + 1]));
  }
  p->bands = bands, p->band_rows = cut, p->band_in_rows = in_rows;
  return B;
}

// One large frame in row bands, ONE pool job for both directions.  The workers stage the input piece by piece in row order (256 KiB
// pieces, claimed with fetch_add: no barrier between chunks, so the sixteen of them stream at the rate scripts/micro/host_copy.hip measures
// instead of the 78 GB/s of a job per 4 MiB chunk) and count finished pieces per chunk; the calling thread watches the counters, enqueues
// a chunk's H2D the moment it is complete, launches a band behind the chunk that completes the rows it reads, enqueues the band's D2H
// in pieces, polls the pieces' events and publishes each landed piece to the same workers, which widen it into the caller's array once
// no staging piece is left.  Widening of band 0 therefore runs while the last chunks are still going in (round 5 staged everything, then
// widened: "staged 1.0" + "out: waits 0.5 + conversions 0.9" one after the other, profiles/r05l_host_frame_row_bands.log).
static int host_one_frame_banded(rpsf_plan* p, const void* image, int in_f64, void* out, int out_f64, const rpsf_geometry& g, int B,
                                 bool direct_in, bool direct_out) {
  HostPipe& q = *p->pipe;
  HostPool& pool = HostPool::get(p->device);
  const int W = g.width, H = g.height;
  const size_t count = (size_t)H * W, bytes = count * sizeof(float);
  constexpr size_t PIECE = (size_t)1 << 16;                      // floats per unit of work of a worker
  static const bool trace = dev_env("RPSF_HOST_TRACE") != nullptr;
  static const int frame_every = dev_env("RPSF_FRAME_EVERY") ? std::atoi(dev_env("RPSF_FRAME_EVERY")) : 0;
  static const int frame_workers = dev_env("RPSF_FRAME_WORKERS") ? std::atoi(dev_env("RPSF_FRAME_WORKERS")) : 0;
  const auto t_start = std::chrono::steady_clock::now();
  auto ms_since = [&](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  struct Range {
    size_t lo, hi;
  };
  // input pieces: chunk by chunk, in row order
  std::vector<Range> in_pieces;
  std::vector<int> piece_chunk, chunk_pieces;
  std::vector<Range> chunks;
  // One upload per band - the rows it reads beyond the band before it - because every copy on a stream costs about 10 us of dead time
  // before the next one starts (sixteen 4 MiB uploads: 45 GB/s against the 57 GB/s of one copy, profiles/r06v_host_frame_timeline.txt); a
  // staged frame sends the first band's rows in four copies, so that the first one leaves when a sixteenth of the band is staged.
  std::vector<int> cuts{0};
  for (int b = 0; b < B; ++b) {
    const int hi = b + 1 == B ? H : p->band_in_rows[b], lo = cuts.back();
    if (hi <= lo) continue;
    if (b == 0 && !direct_in)
      for (int part = 1; part < 4; ++part) cuts.push_back(lo + (int)((long)(hi - lo) * part / 4));
    cuts.push_back(hi);
  }
  for (size_t i = 0; i + 1 < cuts.size(); ++i) {
    const size_t lo = (size_t)cuts[i] * W, hi = (size_t)cuts[i + 1] * W;
    if (hi <= lo) continue;
    int n = 0;
    for (size_t a = lo; a < hi && !direct_in; a += PIECE, ++n) in_pieces.push_back({a, std::min(hi, a + PIECE)}), piece_chunk.push_back((int)chunks.size());
    chunks.push_back({lo, hi}), chunk_pieces.push_back(n);
  }
  std::vector<std::atomic<int>> chunk_done(chunks.size());
  for (auto& c : chunk_done) c.store(0, std::memory_order_relaxed);
  // output pieces: appended by the calling thread as D2H pieces land (storage reserved up front: the workers index it while it grows)
  std::vector<Range> out_pieces(direct_out ? 0 : count / PIECE + HostPipe::MAX_PIECES + 2);
  std::atomic<size_t> next_in{0}, next_out{0}, out_avail{0};
  std::atomic<int> closed{0};
  const size_t n_in = in_pieces.size();
  auto worker = [&](int) {
    for (;;) {
      if (next_in.load(std::memory_order_relaxed) < n_in) {
        const size_t i = next_in.fetch_add(1, std::memory_order_relaxed);
        if (i < n_in) {
          rpsf_host::narrow_or_copy(q.h_in[0], image, in_f64 != 0, in_pieces[i].lo, in_pieces[i].hi);
          chunk_done[piece_chunk[i]].fetch_add(1, std::memory_order_release);
          continue;
        }
      }
      size_t o = next_out.load(std::memory_order_relaxed);
      if (o < out_avail.load(std::memory_order_acquire)) {
        if (next_out.compare_exchange_weak(o, o + 1, std::memory_order_relaxed))
          rpsf_host::widen_or_copy(out, out_f64 != 0, q.h_out[0], out_pieces[o].lo, out_pieces[o].hi);
        continue;
      }
      if (closed.load(std::memory_order_acquire) && next_out.load(std::memory_order_relaxed) >= out_avail.load(std::memory_order_acquire)) return;
      __builtin_ia32_pause();
    }
  };
  hipError_t err = hipSuccess;
  const bool inline_mode = pool.width() < 2;  // (HostPool::run with one part runs `meanwhile` BEFORE the part: the conductor would wait for staging that has not begun)
  std::vector<Range> landed;  // D2H pieces in the order they were enqueued; event c is q.ev_chunk[c]
  double t_enqueued = 0.0, t_first_out = 0.0, t_last_out = 0.0;
  auto conductor = [&] {
    size_t c = 0, published = 0, n_avail = 0;
    int next_band = 0;
    while (err == hipSuccess && (c < chunks.size() || published < landed.size())) {
      bool progress = false;
      if (c < chunks.size() && (direct_in || chunk_done[c].load(std::memory_order_acquire) == chunk_pieces[c])) {
        const size_t lo = chunks[c].lo, hi = chunks[c].hi;
        const int r1 = (int)(hi / W);
        err = hipMemcpyAsync(q.d_in[0] + lo, (direct_in ? static_cast<const float*>(image) : q.h_in[0]) + lo, (hi - lo) * sizeof(float),
                             hipMemcpyHostToDevice, q.st_in);
        while (err == hipSuccess && next_band < B && p->band_in_rows[next_band] <= r1) {
          const int b = next_band++;
          const int R0 = p->band_rows[b], R1 = p->band_rows[b + 1];
          rpsf_geometry gb = g;
          gb.out_row0 = R0, gb.out_rows = R1 - R0;
          err = hipEventRecord(q.ev_band_in[b], q.st_in);
          if (err == hipSuccess) err = hipStreamWaitEvent(p->stream, q.ev_band_in[b], 0);
          float* const host_out = direct_out ? static_cast<float*>(out) : q.h_out[0];
          if (err == hipSuccess && launch_apply(p->bands[b], q.d_in[0], q.d_out[0] + (size_t)R0 * W, gb, p->stream, nullptr) != RPSF_OK)
            err = hipErrorUnknown;
          if (err == hipSuccess) err = hipEventRecord(q.ev_band_k[b], p->stream);
          // (one download stream: a second one for every other band - to hide the 20 us between two copies of a stream - made the frame 8 % slower,
          // and letting the band's kernel write the page-locked rows itself instead of a download 5 %: profiles/r06y, r06z_host_frame_knobs.log)
          const hipStream_t so = q.st_out;
          if (err == hipSuccess) err = hipStreamWaitEvent(so, q.ev_band_k[b], 0);
          // one download per band; the last band of a frame that is widened afterwards in four, so that only a quarter of it is left
          // to widen when the last byte has landed
          const int sub = (b + 1 == B && !direct_out) ? 4 : 1;
          const int rows_per_piece = (R1 - R0 + sub - 1) / sub;
          for (int s0 = R0; s0 < R1 && err == hipSuccess; s0 += rows_per_piece) {
            const size_t plo = (size_t)s0 * W, phi = (size_t)std::min(R1, s0 + rows_per_piece) * W;
            err = hipMemcpyAsync(host_out + plo, q.d_out[0] + plo, (phi - plo) * sizeof(float), hipMemcpyDeviceToHost, so);
            if (err == hipSuccess) err = hipEventRecord(q.ev_chunk[landed.size()], so);
            landed.push_back({plo, phi});
          }
        }
        if (++c == chunks.size()) t_enqueued = ms_since(t_start);
        progress = true;
      }
      if (err == hipSuccess && published < landed.size()) {
        // (while chunks are still to go in: a look; afterwards: a wait - nothing else is left for this thread to do)
        const hipError_t qe = c < chunks.size() ? hipEventQuery(q.ev_chunk[published]) : hipEventSynchronize(q.ev_chunk[published]);
        if (qe == hipSuccess) {
          if (published == 0) t_first_out = ms_since(t_start);
          if (!direct_out && inline_mode) {  // (no workers: this thread widens the piece itself)
            rpsf_host::widen_or_copy(out, out_f64 != 0, q.h_out[0], landed[published].lo, landed[published].hi);
          } else if (!direct_out) {
            for (size_t a = landed[published].lo; a < landed[published].hi; a += PIECE) out_pieces[n_avail++] = {a, std::min(landed[published].hi, a + PIECE)};
            out_avail.store(n_avail, std::memory_order_release);
          }
          ++published, progress = true;
        } else if (qe != hipErrorNotReady) {
          err = qe;
        }
      }
      if (!progress) __builtin_ia32_pause();
    }
    t_last_out = ms_since(t_start);
    closed.store(1, std::memory_order_release);
  };
  if (direct_in && direct_out) conductor();  // nothing for the pool: the copy engines read and write the caller's pages
  else if (inline_mode) {  // RPSF_HOST_THREADS=1: no workers - stage everything, then conduct and widen on this thread (no overlap of the host's own steps)
    for (size_t i = 0; i < n_in; ++i) {
      rpsf_host::narrow_or_copy(q.h_in[0], image, in_f64 != 0, in_pieces[i].lo, in_pieces[i].hi);
      chunk_done[piece_chunk[i]].fetch_add(1, std::memory_order_release);
    }
    conductor();
  } else {
    // every second worker (one per CCD at the default width): sixteen threads streaming beside the copy engines slow the copies down more than they
    // gain (H2D busy 1.78 ms of a 67 MB frame with 16 workers, 1.48 with 8, 1.28 with 4 - which then cannot keep up: profiles/r06x_host_frame_matrix.log)
    const int every = frame_every > 0 ? frame_every : (pool.width() >= 16 ? 2 : 1);
    pool.run(std::max(1, (frame_workers > 0 ? std::min(frame_workers, pool.width()) : pool.width()) / every), worker, conductor, every);
  }
  if (trace)
    std::fprintf(stderr, "[rpsf host frame] %zu chunks, %d bands, %zu + %zu pieces: last chunk enqueued %.3f ms, first piece back %.3f, last %.3f, total %.3f ms\\n",
                 chunks.size(), B, n_in, (size_t)out_avail.load(), t_enqueued, t_first_out, t_last_out, ms_since(t_start));
  if (err != hipSuccess) return drain_after_error(p, err, "host frame");
  return sweep_check(p);
}

// One frame.  The conversions run chunk by chunk (>= 4 MiB) on the pool, each chunk's H2D copy starts as soon as it is staged,
// and on the way back each chunk is widened as soon as it has landed: conversion and PCIe overlap inside the frame.
static int host_one_frame(rpsf_plan* p, const void* image, int in_f64, void* out, int out_f64, const rpsf_geometry& g) {
  const size_t count = (size_t)g.height * g.width, bytes = count * sizeof(float);
  int rc = pipe_ensure(p, count, 1);
  if (rc != RPSF_OK) return rc;
  HostPipe& q = *p->pipe;
  HostPool& pool = HostPool::get(p->device);
  const int T = host_parts_for(bytes / std::max<size_t>(1, std::min<size_t>(HostPipe::MAX_CHUNKS, std::max<size_t>(1, bytes >> 22))));
  const int n_chunks = (int)std::min<size_t>(HostPipe::MAX_CHUNKS, std::max<size_t>(1, bytes >> 22));
  const size_t per_chunk = ((count + n_chunks - 1) / n_chunks + 1023) & ~(size_t)1023;
  auto chunk_range = [&](int c, size_t& lo, size_t& hi) { lo = std::min(count, c * per_chunk), hi = std::min(count, lo + per_chunk); };
  hipError_t err = hipSuccess;
  static const bool trace = dev_env("RPSF_HOST_TRACE") != nullptr;  // development: where a host frame's milliseconds go (stderr)
  const auto t_start = std::chrono::steady_clock::now();
  auto ms_since = [&](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  double t_conv_in = 0.0, t_conv_out = 0.0, t_wait_out = 0.0;
  const bool direct_in = !in_f64 && is_pinned_host(image), direct_out = !out_f64 && is_pinned_host(out);
  // Large frames are cut into row bands (views of this plan) so that the upload of band b + 1, the patches of band b and the download of
  // band b - 1 overlap (RPSF_HOST_BANDS=0: off, =n: that many).
  // (a band per 8 MiB, at most eight: 4096^2 end to end 2.30 / 2.21 / 2.20 ms at 4 / 6 / 8 bands, page-locked 1.83 / 1.80 / 1.79, profiles/r06w_host_bands.log)
  int want_bands = bytes >= ((size_t)24 << 20) ? (int)std::min<size_t>(8, bytes >> 23) : 0;
  if (p->host_bands_opt >= 0) want_bands = p->host_bands_opt;  // (RPSF_OPT_HOST_BANDS)
  const int B = ensure_bands(p, g, want_bands);
  p->last_host_bands = B;
  struct OutChunk {
    size_t lo, hi;
  };
  std::vector<OutChunk> out_chunks;  // in the order their download was enqueued; event c is q.ev_chunk[c]
  if (B >= 2) return host_one_frame_banded(p, image, in_f64, out, out_f64, g, B, direct_in, direct_out);
  {
  // (a frame of one chunk - under 8 MiB - has nothing to overlap with itself: everything on the plan's stream, no events to tie three together;
  // 512^2: 0.179 -> see profiles/r05v)
  const bool one_stream = n_chunks == 1;
  const hipStream_t s_in = one_stream ? p->stream : q.st_in, s_out = one_stream ? p->stream : q.st_out;
  if (direct_in) err = hipMemcpyAsync(q.d_in[0], image, bytes, hipMemcpyHostToDevice, s_in);
  for (int c = 0; c < n_chunks && err == hipSuccess && !direct_in; ++c) {
    size_t lo, hi;
    chunk_range(c, lo, hi);
    const auto t0 = std::chrono::steady_clock::now();
    pool.run(T, [&](int t) {
      size_t a, b;
      rpsf_host::split_range(lo, hi, t, T, a, b);
      rpsf_host::narrow_or_copy(q.h_in[0], image, in_f64 != 0, a, b);
    });
    t_conv_in += ms_since(t0);
    if (hi > lo) err = hipMemcpyAsync(q.d_in[0] + lo, q.h_in[0] + lo, (hi - lo) * sizeof(float), hipMemcpyHostToDevice, s_in);
  }
  if (!one_stream) {
    if (err == hipSuccess) err = hipEventRecord(q.ev_in[0], s_in);
    if (err == hipSuccess) err = hipStreamWaitEvent(p->stream, q.ev_in[0], 0);
  }
  // A small frame through the sweep kernel: the kernel's one store per output pixel goes straight to the page-locked result rows (the staging buffer, or
  // the caller's own page-locked array) - no download behind the launch, whose start-up and 1 MiB are a fifth of such a frame's time (512^2 / 64:
  // 0.140 -> see profiles/r06zu_small_frame_zero_copy_out.log).  Only where every pixel of the window is written by the launch (the lattice covers it).
  float* zc_out = nullptr;
  if (one_stream && err == hipSuccess && overlap_kind(p) == OV_SWEEP && p->sweep_ok && !dev_env("RPSF_NO_ZC_OUT")) {
    const int half = p->N / 2;
    const long r0 = (long)p->lat_r0 + g.origin_row, c0 = (long)p->lat_c0 + g.origin_col;
    const bool covered = r0 <= g.out_row0 && r0 + (long)p->nti * half >= (long)g.out_row0 + g.out_rows && c0 <= 0 && c0 + (long)p->ntj * half >= g.width;
    void* dev = nullptr;
    if (covered && hipHostGetDevicePointer(&dev, direct_out ? out : static_cast<void*>(q.h_out[0]), 0) == hipSuccess) zc_out = static_cast<float*>(dev);
    else (void)hipGetLastError();
  }
  if (err == hipSuccess && launch_apply(p, q.d_in[0], zc_out ? zc_out : q.d_out[0], g, p->stream, nullptr) != RPSF_OK) err = hipErrorUnknown;
  if (zc_out && err == hipSuccess) {
    err = hipEventRecord(q.ev_chunk[0], p->stream);
    out_chunks.push_back({0, count});
  }
  if (!one_stream || trace) {
    if (err == hipSuccess) err = hipEventRecord(q.ev_k[0], p->stream);
    if (err == hipSuccess && !one_stream) err = hipStreamWaitEvent(s_out, q.ev_k[0], 0);
    if (err == hipSuccess && one_stream) err = hipEventRecord(q.ev_in[0], p->stream);  // (trace only)
  }
  if (direct_out && !zc_out && err == hipSuccess) {
    err = hipMemcpyAsync(out, q.d_out[0], bytes, hipMemcpyDeviceToHost, s_out);
    if (err == hipSuccess) err = hipEventRecord(q.ev_chunk[0], s_out);
    out_chunks.push_back({0, count});
  }
  for (int c = 0; c < n_chunks && err == hipSuccess && !direct_out && !zc_out; ++c) {
    size_t lo, hi;
    chunk_range(c, lo, hi);
    if (hi > lo) err = hipMemcpyAsync(q.h_out[0] + lo, q.d_out[0] + lo, (hi - lo) * sizeof(float), hipMemcpyDeviceToHost, s_out);
    if (err == hipSuccess) err = hipEventRecord(q.ev_chunk[out_chunks.size()], s_out);
    out_chunks.push_back({lo, hi});
  }
  }
  const double t_enqueued = ms_since(t_start);
  double t_in_done = 0.0, t_kernel_done = 0.0;
  if (trace && err == hipSuccess && B < 2) {
    (void)hipEventSynchronize(q.ev_in[0]);
    t_in_done = ms_since(t_start);
    (void)hipEventSynchronize(q.ev_k[0]);
    t_kernel_done = ms_since(t_start);
  }
  for (size_t c = 0; c < out_chunks.size() && err == hipSuccess; ++c) {
    auto t0 = std::chrono::steady_clock::now();
    err = hipEventSynchronize(q.ev_chunk[c]);
    t_wait_out += ms_since(t0);
    if (err != hipSuccess || direct_out) continue;
    const size_t lo = out_chunks[c].lo, hi = out_chunks[c].hi;
    t0 = std::chrono::steady_clock::now();
    pool.run(T, [&](int t) {
      size_t a, b;
      rpsf_host::split_range(lo, hi, t, T, a, b);
      rpsf_host::widen_or_copy(out, out_f64 != 0, q.h_out[0], a, b);
    });
    t_conv_out += ms_since(t0);
  }
  if (trace)
    std::fprintf(stderr, "[rpsf host frame] %d chunks x %d parts, %d bands: staged+enqueued %.3f ms (conversions %.3f), H2D done %.3f, kernel done %.3f, "
                 "out: waits %.3f + conversions %.3f, total %.3f ms\n", n_chunks, T, B, t_enqueued, t_conv_in, t_in_done, t_kernel_done, t_wait_out,
                 t_conv_out, ms_since(t_start));
  if (err != hipSuccess) return drain_after_error(p, err, "host frame");
  return sweep_check(p);
}

// Frames per group of the streamed pipeline: small frames go through the shared-K batch launch a few at a time (the packed K is
// read once per group, and a launch / a copy of a few hundred KiB is all overhead: up to 32 frames or 16 MiB per group); from 16 MiB per
// frame on (2048^2), where a frame's copies take several times its kernel, one by one - a short batch then pays the shortest ramp.
static int stream_group_frames(size_t frame_bytes, int n_frames, int stream_group_opt) {
  int g = (int)std::max<size_t>(1, std::min<size_t>(32, ((size_t)16 << 20) / std::max<size_t>(1, frame_bytes)));
  if (stream_group_opt > 0) g = std::max(1, std::min(64, stream_group_opt));  // (RPSF_OPT_STREAM_GROUP)
  return std::min(g, n_frames);
}

// A sequence of frames of one geometry: `depth` groups in flight.  The calling thread stages group i (conversion on the pool),
// enqueues H2D(i) on the copy-in stream, the shared-K launch of the group on the plan's stream behind it, D2H(i) on the copy-out
// stream behind that, and - in the same pool job as the staging of the next group - widens whichever earlier group has landed.
// H2D of group i + 1, the kernel of group i and D2H of group i - 1 therefore run at the same time (PCIe is full duplex), and the
// host conversions of both directions share the pool.
static int host_frames(rpsf_plan* p, const void* const* images, int in_f64, void* const* outs, int out_f64, int n_frames,
                       const rpsf_geometry& g) {
  if (n_frames == 1) return host_one_frame(p, images[0], in_f64, outs[0], out_f64, g);
  const size_t count = (size_t)g.height * g.width, bytes = count * sizeof(float);
  // Groups: `first[i]` ... `first[i + 1]` are the frames of group i.  Small frames: equal groups.  Frames that go one by one (16 MiB and more) go two
  // by two in the MIDDLE of a long sequence - a staging job and an enqueue per two frames instead of per frame: 32 x 2048^2 float32 0.473 -> 0.415 ms
  // per frame - while the first and the last groups stay single frames, so that the ramps (the first upload, the last download) stay short
  // (groups of two throughout cost an 8-frame batch 0.48 -> 0.57 ms per frame; profiles/r05y).  RPSF_OPT_STREAM_GROUP fixes one size for all.
  const int G0 = stream_group_frames(bytes, n_frames, p->stream_group_opt);
  std::vector<int> first{0};
  {
    const bool pinned_size = p->stream_group_opt > 0;
    int pairs_from = 12;
    if (const char* e = dev_env("RPSF_STREAM_PAIRS_FROM")) pairs_from = std::max(4, std::atoi(e));  // development sweeps
    const bool pairs = !pinned_size && G0 == 1 && n_frames >= pairs_from && bytes * 2 <= ((size_t)128 << 20);
    while (first.back() < n_frames) {
      const int at = first.back(), left = n_frames - at;
      const int size = pairs ? ((at >= 2 && left >= 4) ? 2 : 1) : G0;
      first.push_back(at + std::min(size, left));
    }
  }
  const int n_groups = (int)first.size() - 1;
  int G = 1;
  for (int i = 0; i < n_groups; ++i) G = std::max(G, first[i + 1] - first[i]);
  int depth = bytes * G >= ((size_t)128 << 20) ? 3 : HostPipe::MAX_DEPTH;
  if (p->stream_depth_opt > 0) depth = std::max(1, std::min((int)HostPipe::MAX_DEPTH, p->stream_depth_opt));  // (RPSF_OPT_STREAM_DEPTH)
  depth = std::min(depth, n_groups);
  int rc = pipe_ensure(p, count * G, depth);
  if (rc != RPSF_OK) return rc;
  HostPipe& q = *p->pipe;
  HostPool& pool = HostPool::get(p->device);
  const int T = host_parts_for(bytes * G);  // (a job stages / unstages a whole group: small frames still give every worker a part)
  auto frames_of = [&](int grp) { return first[grp + 1] - first[grp]; };
  const bool direct_in = !in_f64 && all_pinned(images, n_frames);
  const bool direct_out = !out_f64 && all_pinned(const_cast<const void* const*>(outs), n_frames);
  hipError_t err = hipSuccess;
  int next_in = 0, next_out = 0;
  static const bool trace = dev_env("RPSF_HOST_TRACE") != nullptr;  // development: where the host's time goes (stderr)
  double t_jobs = 0.0, t_enqueue = 0.0, t_wait = 0.0;
  const auto t_start = std::chrono::steady_clock::now();
  auto ms_since = [&](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  // enqueue group `grp` (staged in slot grp % depth): H2D -> shared-K launch -> D2H on the three streams
  auto enqueue = [&](int grp) {
    const int sg = grp % depth, fg = frames_of(grp);
    const size_t gb = (size_t)fg * bytes;
    if (direct_in) {
      for (int f = 0; f < fg && err == hipSuccess; ++f)
        err = hipMemcpyAsync(q.d_in[sg] + (size_t)f * count, images[first[grp] + f], bytes, hipMemcpyHostToDevice, q.st_in);
    } else {
      err = hipMemcpyAsync(q.d_in[sg], q.h_in[sg], gb, hipMemcpyHostToDevice, q.st_in);
    }
    if (err == hipSuccess) err = hipEventRecord(q.ev_in[sg], q.st_in);
    if (err == hipSuccess) err = hipStreamWaitEvent(p->stream, q.ev_in[sg], 0);
    if (err == hipSuccess && launch_batch(p, q.d_in[sg], q.d_out[sg], fg, count, count, g, p->stream) != RPSF_OK) err = hipErrorUnknown;
    if (err == hipSuccess) err = hipEventRecord(q.ev_k[sg], p->stream);
    if (err == hipSuccess) err = hipStreamWaitEvent(q.st_out, q.ev_k[sg], 0);
    if (direct_out) {
      for (int f = 0; f < fg && err == hipSuccess; ++f)
        err = hipMemcpyAsync(outs[first[grp] + f], q.d_out[sg] + (size_t)f * count, bytes, hipMemcpyDeviceToHost, q.st_out);
    } else if (err == hipSuccess) {
      err = hipMemcpyAsync(q.h_out[sg], q.d_out[sg], gb, hipMemcpyDeviceToHost, q.st_out);
    }
    if (err == hipSuccess) err = hipEventRecord(q.ev_out[sg], q.st_out);
  };
  int staged = -1, next_enq = 0;  // a group that is staged but not yet enqueued: its enqueue runs on this thread WHILE the workers do the next job
  while (next_out < n_groups && err == hipSuccess) {
    auto t_phase = std::chrono::steady_clock::now();
    const bool can_in = next_in < n_groups && next_in - next_out < depth;
    bool out_ready = false;
    if (next_out < next_enq) {
      const hipError_t qe = hipEventQuery(q.ev_out[next_out % depth]);
      if (qe == hipSuccess) out_ready = true;
      else if (qe != hipErrorNotReady) err = qe;
      if (err == hipSuccess && !out_ready && !can_in && staged < 0) {  // nothing to stage or enqueue: wait for the oldest group in flight
        err = hipEventSynchronize(q.ev_out[next_out % depth]);
        out_ready = err == hipSuccess;
      }
    }
    if (err != hipSuccess) break;
    t_wait += ms_since(t_phase), t_phase = std::chrono::steady_clock::now();
    const int si = next_in % depth, so = next_out % depth;
    const int fi = can_in ? frames_of(next_in) : 0, fo = out_ready ? frames_of(next_out) : 0;
    const int ci = direct_in ? 0 : fi, co = direct_out ? 0 : fo;  // frames this job stages / unstages
    const int to_enqueue = staged;
    double t_meanwhile = 0.0;
    pool.run(ci + co > 0 ? T : 0,
             [&](int t) {
               // the group's frames laid end to end, cut into T parts: a part covers a run of pixels that may span frames
               const size_t total_in = (size_t)ci * count, total_out = (size_t)co * count;
               size_t a, b;
               rpsf_host::split_range(0, total_in, t, T, a, b);
               while (a < b) {
                 const size_t f = a / count, lo = a % count, hi = std::min(count, lo + (b - a));
                 rpsf_host::narrow_or_copy(q.h_in[si] + f * count, images[first[next_in] + f], in_f64 != 0, lo, hi);
                 a += hi - lo;
               }
               rpsf_host::split_range(0, total_out, t, T, a, b);
               while (a < b) {
                 const size_t f = a / count, lo = a % count, hi = std::min(count, lo + (b - a));
                 rpsf_host::widen_or_copy(outs[first[next_out] + f], out_f64 != 0, q.h_out[so] + f * count, lo, hi);
                 a += hi - lo;
               }
             },
             [&] {
               if (to_enqueue >= 0) {
                 const auto t0 = std::chrono::steady_clock::now();
                 enqueue(to_enqueue);
                 t_meanwhile = ms_since(t0);
               }
             });
    if (to_enqueue >= 0) staged = -1, next_enq = to_enqueue + 1;
    t_enqueue += t_meanwhile;
    t_jobs += ms_since(t_phase) - t_meanwhile;
    if (can_in) staged = next_in++;
    if (out_ready) ++next_out;
  }
  if (trace)
    std::fprintf(stderr, "[rpsf streamed] %d frames in %d groups of up to %d, depth %d, %d parts per job: staging jobs %.3f ms, enqueues %.3f ms, waits %.3f ms, total %.3f ms\n",
                 n_frames, n_groups, G, depth, T, t_jobs, t_enqueue, t_wait, ms_since(t_start));
  if (err != hipSuccess) return drain_after_error(p, err, "streamed frames");
  return sweep_check(p);
}

static int check_host_call(rpsf_plan* p, const void* a, const void* b, int n_frames, int height, int width, int pad_mode, float pad_value,
                           rpsf_geometry* g) {
  if (!p || !a || !b) return fail(RPSF_E_BADARG, "null argument");
  if (n_frames <= 0) return fail(RPSF_E_BADARG, "n_frames must be positive");
  if (!p->have_k) return fail(RPSF_E_STATE, "no transfer kernel installed (call rpsf_plan_set_transfer first)");
  *g = rpsf_geometry{height, width, pad_mode, pad_value, 0, 0, 0, height, width, 0, height, width};
  int rc = check_geometry(p, g);
  if (rc != RPSF_OK) return rc;
  HIP_TRY(hipSetDevice(p->device));
  return RPSF_OK;
}

extern "C" int rpsf_apply_frames_host(rpsf_plan* p, const void* const* images_host, int image_is_f64, int n_frames, int height,
                                      int width, int pad_mode, float pad_value, void* const* outs_host, int out_is_f64) {
  rpsf_geometry g;
  int rc = check_host_call(p, images_host, outs_host, n_frames, height, width, pad_mode, pad_value, &g);
  if (rc != RPSF_OK) return rc;
  for (int f = 0; f < n_frames; ++f)
    if (!images_host[f] || !outs_host[f]) return fail(RPSF_E_BADARG, "null frame pointer");
  return host_frames(p, images_host, image_is_f64, outs_host, out_is_f64, n_frames, g);
}

extern "C" int rpsf_apply_batch_host(rpsf_plan* p, const void* images_host, int image_is_f64, int n_frames, int height, int width,
                                     int pad_mode, float pad_value, void* outs_host, int out_is_f64) {
  rpsf_geometry g;
  int rc = check_host_call(p, images_host, outs_host, n_frames, height, width, pad_mode, pad_value, &g);
  if (rc != RPSF_OK) return rc;
  const size_t count = (size_t)height * width;
  std::vector<const void*> in(n_frames);
  std::vector<void*> out(n_frames);
  for (int f = 0; f < n_frames; ++f) {
    in[f] = static_cast<const char*>(images_host) + (size_t)f * count * (image_is_f64 ? 8 : 4);
    out[f] = static_cast<char*>(outs_host) + (size_t)f * count * (out_is_f64 ? 8 : 4);
  }
  return host_frames(p, in.data(), image_is_f64, out.data(), out_is_f64, n_frames, g);
}

extern "C" int rpsf_apply_batch(rpsf_plan* p, const float* images_host, int n_frames, int height, int width, int pad_mode,
                                float pad_value, float* outs_host) {
  return rpsf_apply_batch_host(p, images_host, 0, n_frames, height, width, pad_mode, pad_value, outs_host, 0);
}

extern "C" int rpsf_apply_host(rpsf_plan* p, const void* image_host, int image_is_f64, int height, int width, int pad_mode,
                               float pad_value, void* out_host, int out_is_f64) {
  rpsf_geometry g;
  int rc = check_host_call(p, image_host, out_host, 1, height, width, pad_mode, pad_value, &g);
  if (rc != RPSF_OK) return rc;
  return host_one_frame(p, image_host, image_is_f64, out_host, out_is_f64, g);
}

extern "C" int rpsf_apply(rpsf_plan* p, const float* image_host, int height, int width, int pad_mode, float pad_value,
                          float* out_host) {
  return rpsf_apply_host(p, image_host, 0, height, width, pad_mode, pad_value, out_host, 0);
}

// Page-locked host memory for callers that keep their frames in it (acquisition buffers, result rings): float32 frames in such memory
// cross PCIe without the staging copy of the pageable path.
extern "C" int rpsf_host_alloc(int device, size_t bytes, void** out) {
  if (!out) return fail(RPSF_E_BADARG, "null argument");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return RPSF_OK;
}
extern "C" int rpsf_host_free(void* ptr) {
  if (ptr) HIP_TRY(hipHostFree(ptr));
  return RPSF_OK;
}

// The NUMA node the device hangs off (-1: unknown / a single-node host): where a process that feeds this GPU from host arrays should
// run and allocate (the pool's workers are placed there by themselves).
extern "C" int rpsf_device_numa_node(int device, int* node) {
  if (!node) return fail(RPSF_E_BADARG, "null argument");
  *node = -1;
  char bdf[64] = {};
  HIP_TRY(hipDeviceGetPCIBusId(bdf, sizeof(bdf), device));
  for (char* c = bdf; *c; ++c) *c = (char)std::tolower(*c);
  const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
  if (FILE* f = std::fopen(path.c_str(), "r")) {
    if (std::fscanf(f, "%d", node) != 1) *node = -1;
    std::fclose(f);
  }
  return RPSF_OK;
}

// ------------------------------------------------------------------------------------------------
// ArrayPSFTransform.apply with a finite saturation threshold, whole (regularizepsf/transform.py:117-138,171-177): float64 copy, np.pad by 2N
// (the mode's index map, evaluated here on the host), mask = padded > threshold, binary dilation with the cross element (`dilation` >= 1
// iterations = a diamond of that radius; scipy treats the outside as background), NaN, the sequential row-major nanmean fill, the correction of
// the PADDED frame on the GPU (corners shifted by 2N, constant padding: exactly what the reference's slices see), raw values restored on the mask,
// crop.  Everything the reference does on the host stays on the host and in its order; what changed against the NumPy / SciPy route of rounds 1-4
// is the cost: no np.pad / astype / copy temporaries (three passes over a (H + 4N)^2 float64 frame), no scipy.ndimage pass over the whole mask for
// a handful of pixels, no device allocation per call, only the rows the patches read and the rows the caller gets cross PCIe.
// A sequence of frames (rpsf_apply_frames_host_saturated - the reference's example corrects a list of frames this way, docs/source/example.ipynb
// cell 25) alternates between two staging slots: the host steps of frame i + 1 run while the GPU corrects frame i.
// ------------------------------------------------------------------------------------------------
struct SatRun {
  rpsf_plan* p;
  int H, W, N, pad_mode, dilation, width, in_f64, out_f64;
  double threshold;
  long PH, PW;
  int r_lo, r_hi, o_lo, o_rows;
  rpsf_geometry g;
  std::vector<int> colmap;

  int init(rpsf_plan* plan, int height, int w, int mode, double thr, int dil, int nbw, int image_is_f64, int out_is_f64, int slots) {
    p = plan, H = height, W = w, N = plan->N, pad_mode = mode, dilation = dil, width = nbw, in_f64 = image_is_f64, out_f64 = out_is_f64, threshold = thr;
    PH = (long)H + 4L * N, PW = (long)W + 4L * N;
    if (PH * PW >= ((long)1 << 31)) return fail(RPSF_E_UNSUPPORTED, "padded frame too large for this entry point");
    // rows of the padded frame the patches read, and the geometry of the correction on it (the hipFFT fallback takes whole frames only)
    r_lo = p->generic ? 0 : (int)std::max<long>(0, 2L * N + std::min(0, p->corner_min[0]));
    r_hi = p->generic ? (int)PH : (int)std::min<long>(PH, 2L * N + p->corner_max[0] + N);
    o_lo = p->generic ? 0 : 2 * N, o_rows = p->generic ? (int)PH : H;  // output rows the device hands back
    g = rpsf_geometry{(int)PH, (int)PW, RPSF_PAD_CONSTANT, 0.f, 2 * N, 2 * N, r_lo, r_hi - r_lo, (int)PW, o_lo, o_rows, (int)PW};
    int rc = check_geometry(p, &g);
    if (rc != RPSF_OK) return rc;
    HIP_TRY(hipSetDevice(p->device));
    rc = pipe_ensure(p, (size_t)PH * PW, slots);
    if (rc != RPSF_OK) return rc;
    if ((int)p->sat.size() < slots) p->sat.resize(slots);
    colmap.resize(PW);
    for (long c = 0; c < PW; ++c) colmap[c] = pad_index((int)(c - 2L * N), W, pad_mode);
    return RPSF_OK;
  }

  // host steps of one frame into slot s (pad, threshold, dilation, raw values, NaN, sequential fill, narrowing into the pinned staging)
  void prepare(int s, const void* image_host) {
    HostPipe& q = *p->pipe;
    HostPool& pool = HostPool::get(p->device);
    SatScratch& sc = p->sat[s];
    sc.padded.resize((size_t)PH * PW);
    sc.mask.assign((size_t)PH * PW, 0);
    sc.raw.clear();
    double* const padded = sc.padded.data();
    uint8_t* const mask = sc.mask.data();
    const int parts = (int)std::min<long>(PH, (long)pool.width() * 4);
    std::vector<std::vector<int64_t>> hot(parts);  // per part: flat indices of the pixels above the threshold
    // ---- pad (:119-123) + threshold (:129), row blocks in parallel
    pool.run(parts, [&](int part) {
      const long ra = PH * part / parts, rb = PH * (part + 1) / parts;
      for (long r = ra; r < rb; ++r) {
        const int sr = pad_index((int)(r - 2L * N), H, pad_mode);
        double* dst = padded + r * PW;
        if (sr < 0) {
          for (long c = 0; c < PW; ++c) dst[c] = 0.0;  // np.pad(mode="constant") pads with 0
        } else if (in_f64) {
          const double* src = static_cast<const double*>(image_host) + (size_t)sr * W;
          for (long c = 0; c < PW; ++c) dst[c] = colmap[c] < 0 ? 0.0 : src[colmap[c]];
        } else {
          const float* src = static_cast<const float*>(image_host) + (size_t)sr * W;
          for (long c = 0; c < PW; ++c) dst[c] = colmap[c] < 0 ? 0.0 : (double)src[colmap[c]];
        }
        for (long c = 0; c < PW; ++c)
          if (dst[c] > threshold) hot[part].push_back(r * PW + c);
      }
    });
    // ---- dilation (:133): every pixel within `dilation` city-block steps of a saturated one
    std::vector<uint8_t> row_has(PH, 0);
    size_t n_hot = 0;
    for (const auto& list : hot) {
      n_hot += list.size();
      for (const int64_t idx : list) {
        const long r = idx / PW, c = idx % PW;
        for (long dr = -dilation; dr <= dilation; ++dr) {
          const long rr = r + dr;
          if (rr < 0 || rr >= PH) continue;
          const long span = dilation - std::labs(dr);
          const long c0 = std::max<long>(0, c - span), c1 = std::min<long>(PW - 1, c + span);
          std::memset(mask + rr * PW + c0, 1, (size_t)(c1 - c0 + 1));
          row_has[rr] = 1;
        }
      }
    }
    // ---- raw values of the masked pixels the caller will see (:126, :172), NaN (:134), sequential fill (:135-138)
    if (n_hot) {
      for (long r = 0; r < PH; ++r) {
        if (!row_has[r]) continue;
        for (long c = 0; c < PW; ++c)
          if (mask[r * PW + c]) {
            if (r >= 2L * N && r < 2L * N + H && c >= 2L * N && c < 2L * N + W) sc.raw.push_back({r * PW + c, padded[r * PW + c]});
            padded[r * PW + c] = std::nan("");
          }
      }
      const long hw = width / 2;
      for (long i = 0; i < PH; ++i) {
        if (!row_has[i]) continue;
        for (long j = 0; j < PW; ++j) {
          if (!mask[i * PW + j]) continue;
          long r0, r1, c0, c1;
          py_slice(i - hw, i + hw, PH, &r0, &r1);
          py_slice(j - hw, j + hw, PW, &c0, &c1);
          double sum = 0.0;
          long cnt = 0;
          for (long r = r0; r < r1; ++r)
            for (long c = c0; c < c1; ++c) {
              const double v = padded[r * PW + c];
              if (v == v) sum += v, ++cnt;
            }
          padded[i * PW + j] = cnt ? sum / (double)cnt : std::nan("");
        }
      }
    }
    const size_t in_lo = (size_t)r_lo * PW, in_hi = (size_t)r_hi * PW;
    const int T = host_parts_for((in_hi - in_lo) * sizeof(float));
    pool.run(T, [&](int t) {
      size_t a, b;
      rpsf_host::split_range(in_lo, in_hi, t, T, a, b);
      rpsf_host::narrow_or_copy(q.h_in[s], padded, true, a, b);
    });
  }

  // the correction of the padded frame of slot s: rows the patches read in, the caller's rows out; asynchronous on the plan's stream
  hipError_t enqueue(int s) {
    HostPipe& q = *p->pipe;
    const size_t in_lo = (size_t)r_lo * PW, in_hi = (size_t)r_hi * PW, out_lo = (size_t)o_lo * PW, out_hi = out_lo + (size_t)o_rows * PW;
    hipError_t err = hipMemcpyAsync(q.d_in[s], q.h_in[s] + in_lo, (in_hi - in_lo) * sizeof(float), hipMemcpyHostToDevice, p->stream);
    if (err == hipSuccess && launch_apply(p, q.d_in[s], q.d_out[s], g, p->stream, nullptr) != RPSF_OK) err = hipErrorUnknown;
    if (err == hipSuccess) err = hipMemcpyAsync(q.h_out[s] + out_lo, q.d_out[s], (out_hi - out_lo) * sizeof(float), hipMemcpyDeviceToHost, p->stream);
    if (err == hipSuccess) err = hipEventRecord(q.ev_out[s], p->stream);
    return err;
  }

  // wait for slot s, then the crop (:174-177) with the raw values back on the mask (:172)
  hipError_t finish(int s, void* out_host) {
    HostPipe& q = *p->pipe;
    HostPool& pool = HostPool::get(p->device);
    hipError_t err = hipEventSynchronize(q.ev_out[s]);
    if (err != hipSuccess) return err;
    const int rows_parts = std::min(H, pool.width() * 4);
    pool.run(rows_parts, [&](int part) {
      const long ra = (long)H * part / rows_parts, rb = (long)H * (part + 1) / rows_parts;
      for (long r = ra; r < rb; ++r) {
        const float* src = q.h_out[s] + (size_t)(2 * N + r) * PW + 2 * N;  // (h_out is indexed by rows of the padded frame)
        if (out_f64) {
          double* dst = static_cast<double*>(out_host) + (size_t)r * W;
          for (long c = 0; c < W; ++c) dst[c] = (double)src[c];
        } else {
          std::memcpy(static_cast<float*>(out_host) + (size_t)r * W, src, (size_t)W * sizeof(float));
        }
      }
    });
    for (const SatScratch::Raw& x : p->sat[s].raw) {
      const long r = x.idx / PW - 2L * N, c = x.idx % PW - 2L * N;
      if (out_f64) static_cast<double*>(out_host)[(size_t)r * W + c] = x.value;
      else static_cast<float*>(out_host)[(size_t)r * W + c] = (float)x.value;
    }
    return hipSuccess;
  }
};

static int check_saturated_call(rpsf_plan* p, const void* a, const void* b, int height, int width, int pad_mode, int dilation, int neighborhood_width) {
  if (!p || !a || !b) return fail(RPSF_E_BADARG, "null argument");
  if (height <= 0 || width <= 0) return fail(RPSF_E_BADARG, "image shape must be positive");
  if (pad_mode < 0 || pad_mode > RPSF_PAD_WRAP) return fail(RPSF_E_BADARG, "unknown pad mode");
  if (dilation < 1 || neighborhood_width < 0) return fail(RPSF_E_BADARG, "dilation must be >= 1 and the neighbourhood width >= 0 (other values: the NumPy route)");
  if (!p->have_k) return fail(RPSF_E_STATE, "no transfer kernel installed (call rpsf_plan_set_transfer first)");
  return RPSF_OK;
}

extern "C" int rpsf_apply_host_saturated(rpsf_plan* p, const void* image_host, int image_is_f64, int height, int width, int pad_mode,
                                         double threshold, int dilation, int neighborhood_width, void* out_host, int out_is_f64) {
  int rc = check_saturated_call(p, image_host, out_host, height, width, pad_mode, dilation, neighborhood_width);
  if (rc != RPSF_OK) return rc;
  SatRun run;
  rc = run.init(p, height, width, pad_mode, threshold, dilation, neighborhood_width, image_is_f64, out_is_f64, 1);
  if (rc != RPSF_OK) return rc;
  run.prepare(0, image_host);
  hipError_t err = run.enqueue(0);
  if (err == hipSuccess) err = run.finish(0, out_host);
  if (err != hipSuccess) return drain_after_error(p, err, "saturated host frame");
  return sweep_check(p);
}

extern "C" int rpsf_apply_frames_host_saturated(rpsf_plan* p, const void* const* images_host, int image_is_f64, int n_frames, int height, int width,
                                                int pad_mode, double threshold, int dilation, int neighborhood_width, void* const* outs_host,
                                                int out_is_f64) {
  int rc = check_saturated_call(p, images_host, outs_host, height, width, pad_mode, dilation, neighborhood_width);
  if (rc != RPSF_OK) return rc;
  if (n_frames <= 0) return fail(RPSF_E_BADARG, "n_frames must be positive");
  for (int f = 0; f < n_frames; ++f)
    if (!images_host[f] || !outs_host[f]) return fail(RPSF_E_BADARG, "null frame pointer");
  SatRun run;
  rc = run.init(p, height, width, pad_mode, threshold, dilation, neighborhood_width, image_is_f64, out_is_f64, std::min(2, n_frames));
  if (rc != RPSF_OK) return rc;
  hipError_t err = hipSuccess;
  for (int f = 0; f < n_frames && err == hipSuccess; ++f) {
    run.prepare(f & 1, images_host[f]);  // (the slot's previous frame, f - 2, was finished in the iteration before)
    err = run.enqueue(f & 1);
    if (err == hipSuccess && f > 0) err = run.finish((f - 1) & 1, outs_host[f - 1]);  // the GPU has had the host steps of frame f to correct frame f - 1
  }
  if (err == hipSuccess) err = run.finish((n_frames - 1) & 1, outs_host[n_frames - 1]);
  if (err != hipSuccess) return drain_after_error(p, err, "saturated host frames");
  return sweep_check(p);
}

// Self-test of the host worker pool (no GPU involved: the CPU test suite calls it): `jobs` jobs of `parts` parts from each of `callers` threads at
// once; every part adds its index into a per-job cell and the job must see the exact total when run() returns.  Returns the number of jobs
// whose total was wrong (0 = pass) through *failures.
extern "C" int rpsf_host_pool_selftest(int callers, int jobs, int parts, int* failures) {
  if (!failures || callers <= 0 || jobs <= 0 || parts <= 0) return fail(RPSF_E_BADARG, "bad argument");
  HostPool& pool = HostPool::get();
  std::atomic<int> bad{0};
  auto caller = [&](int id) {
    for (int j = 0; j < jobs; ++j) {
      const int n = 1 + (parts + id + j) % parts;  // varying job sizes, single-part jobs included
      std::atomic<long> sum{0};
      std::vector<int> hits(n, 0);
      pool.run(n, [&](int i) {
        sum.fetch_add(i + 1, std::memory_order_relaxed);
        ++hits[i];  // every part exactly once, by exactly one thread
      });
      bool ok = sum.load() == (long)n * (n + 1) / 2;
      for (int i = 0; i < n && ok; ++i) ok = hits[i] == 1;
      if (!ok) bad.fetch_add(1);
    }
  };
  std::vector<std::thread> threads;
  for (int c = 1; c < callers; ++c) threads.emplace_back(caller, c);
  caller(0);
  for (auto& t : threads) t.join();
  *failures = bad.load();
  return RPSF_OK;
}

extern "C" int rpsf_host_threads(int* threads) {
  if (!threads) return fail(RPSF_E_BADARG, "null argument");
  *threads = HostPool::get().width();
  return RPSF_OK;
}

// What PCIe gives a frame on this box: `bytes` from pinned host memory to the device, back, and both at once on two streams,
// `iters` times each (best).  The floor that the streamed entry points are measured against (end-to-end is reported
// separately from the device-resident figure).
extern "C" int rpsf_pcie_probe(int device, size_t bytes, int iters, double* h2d_ms, double* d2h_ms, double* duplex_ms) {
  if (bytes == 0 || iters <= 0) return fail(RPSF_E_BADARG, "bad argument");
  HIP_TRY(hipSetDevice(device));
  struct Res {
    void *h0 = nullptr, *h1 = nullptr;
    DevBuf d0, d1;
    hipStream_t s0 = nullptr, s1 = nullptr;
    hipEvent_t e[4] = {};
    ~Res() {
      (void)hipHostFree(h0);
      (void)hipHostFree(h1);
      for (auto& x : e)
        if (x) (void)hipEventDestroy(x);
      if (s0) (void)hipStreamDestroy(s0);
      if (s1) (void)hipStreamDestroy(s1);
    }
  } r;
  HIP_TRY(hipHostMalloc(&r.h0, bytes, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc(&r.h1, bytes, hipHostMallocDefault));
  std::memset(r.h0, 1, bytes);
  std::memset(r.h1, 2, bytes);
  HIP_TRY(r.d0.alloc(bytes));
  HIP_TRY(r.d1.alloc(bytes));
  HIP_TRY(hipStreamCreateWithFlags(&r.s0, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&r.s1, hipStreamNonBlocking));
  for (auto& x : r.e) HIP_TRY(hipEventCreate(&x));
  double best[3] = {1e30, 1e30, 1e30};
  for (int mode = 0; mode < 3; ++mode)
    for (int i = 0; i < iters + 1; ++i) {  // (the first pass of a mode is a warm-up)
      HIP_TRY(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      if (mode != 1) HIP_TRY(hipMemcpyAsync(r.d0.p, r.h0, bytes, hipMemcpyHostToDevice, r.s0));
      if (mode != 0) HIP_TRY(hipMemcpyAsync(r.h1, r.d1.p, bytes, hipMemcpyDeviceToHost, r.s1));
      HIP_TRY(hipStreamSynchronize(r.s0));
      HIP_TRY(hipStreamSynchronize(r.s1));
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (i > 0) best[mode] = std::min(best[mode], ms);
    }
  if (h2d_ms) *h2d_ms = best[0];
  if (d2h_ms) *d2h_ms = best[1];
  if (duplex_ms) *duplex_ms = best[2];
  return RPSF_OK;
}

extern "C" int rpsf_apply_device_timed(rpsf_plan* p, const void* image_dev, void* out_dev, const rpsf_geometry* geom,
                                       int iters, float* total_ms, float* kernel_ms) {
  if (!p || !image_dev || !out_dev || iters <= 0) return fail(RPSF_E_BADARG, "bad argument");
  if (!p->have_k) return fail(RPSF_E_STATE, "no transfer kernel installed (call rpsf_plan_set_transfer first)");
  int rc = check_geometry(p, geom);
  if (rc != RPSF_OK) return rc;
  HIP_TRY(hipSetDevice(p->device));
  return timed_loop(p, iters, total_ms, kernel_ms, [&](hipEvent_t k0, hipEvent_t k1) {
    return launch_apply(p, reinterpret_cast<const float*>(image_dev), reinterpret_cast<float*>(out_dev), *geom, p->stream, k0, k1);
  });
}

// `iters` applies back to back between ONE pair of events on the plan's stream: the average apply as the device sees it, with nothing
// between two launches that a caller's loop would not put there (per-apply event pairs cost a marker packet each: ~8 us of the 184 here).
extern "C" int rpsf_apply_device_loop_ms(rpsf_plan* p, const void* image_dev, void* out_dev, const rpsf_geometry* geom, int iters,
                                         double* ms_per_apply) {
  if (!p || !image_dev || !out_dev || iters <= 0 || !ms_per_apply) return fail(RPSF_E_BADARG, "bad argument");
  if (!p->have_k) return fail(RPSF_E_STATE, "no transfer kernel installed (call rpsf_plan_set_transfer first)");
  int rc = check_geometry(p, geom);
  if (rc != RPSF_OK) return rc;
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipEventRecord(p->ev[0], p->stream));
  for (int i = 0; i < iters; ++i) {
    rc = launch_apply(p, reinterpret_cast<const float*>(image_dev), reinterpret_cast<float*>(out_dev), *geom, p->stream, nullptr);
    if (rc != RPSF_OK) return rc;
  }
  HIP_TRY(hipEventRecord(p->ev[3], p->stream));
  HIP_TRY(hipEventSynchronize(p->ev[3]));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, p->ev[0], p->ev[3]));
  *ms_per_apply = (double)ms / iters;
  return sweep_check(p);
}

// ------------------------------------------------------------------------------------------------
// K2 entry points
// ------------------------------------------------------------------------------------------------
extern "C" int rpsf_build_transfer_device(int device, size_t count, const void* s_dev, const void* t_dev, int is_f64,
                                          double alpha, double epsilon, void* k_dev, void* stream) {
  if (!s_dev || !t_dev || !k_dev) return fail(RPSF_E_BADARG, "null argument");
  if (count == 0) return RPSF_OK;
  HIP_TRY(hipSetDevice(device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int block = 256;
  size_t grid = (count + block - 1) / block;
  if (is_f64)
    build_transfer_kernel<double><<<dim3((unsigned)grid), dim3(block), 0, st>>>(
        (const double*)s_dev, (const double*)t_dev, (double*)k_dev, count, alpha, epsilon);
  else
    build_transfer_kernel<float><<<dim3((unsigned)grid), dim3(block), 0, st>>>(
        (const float*)s_dev, (const float*)t_dev, (float*)k_dev, count, (float)alpha, (float)epsilon);
  HIP_TRY(hipGetLastError());
  return RPSF_OK;
}

extern "C" int rpsf_build_transfer(int device, size_t count, const void* s_host, const void* t_host, int is_f64,
                                   double alpha, double epsilon, void* k_host) {
  if (!s_host || !t_host || !k_host) return fail(RPSF_E_BADARG, "null argument");
  if (count == 0) return RPSF_OK;
  HIP_TRY(hipSetDevice(device));
  const size_t esz = is_f64 ? 16 : 8;
  const size_t chunk = std::min<size_t>(count, (size_t)8 << 20);  // 8 Mi elements per round
  DevBuf bs, bt, bk;
  HIP_TRY(bs.alloc(chunk * esz));
  HIP_TRY(bt.alloc(chunk * esz));
  HIP_TRY(bk.alloc(chunk * esz));
  char *ds = bs.as<char>(), *dt = bt.as<char>(), *dk = bk.as<char>();
  int rc = RPSF_OK;
  for (size_t first = 0; first < count && rc == RPSF_OK; first += chunk) {
    size_t cnt = std::min(chunk, count - first);
    hipError_t e = hipMemcpy(ds, (const char*)s_host + first * esz, cnt * esz, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dt, (const char*)t_host + first * esz, cnt * esz, hipMemcpyHostToDevice);
    if (e != hipSuccess) { rc = fail(RPSF_E_HIP, hipGetErrorString(e)); break; }
    rc = rpsf_build_transfer_device(device, cnt, ds, dt, is_f64, alpha, epsilon, dk, nullptr);
    if (rc != RPSF_OK) break;
    e = hipMemcpy((char*)k_host + first * esz, dk, cnt * esz, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(RPSF_E_HIP, hipGetErrorString(e));
  }
  return rc;
}

// ------------------------------------------------------------------------------------------------
// K3 entry point
// ------------------------------------------------------------------------------------------------
// fft_host / fft_dev: exactly one is non-null - where the spectra go
// model < 0: the samples come from values_host; otherwise they are rasterised on the device from params_host
// (count x RPSF_MODEL_PARAMS doubles) and, if values_dev is given, kept there as float32 as well
static int psf_fft_impl(int device, int patch_size, int count, const float* values_host, float* fft_host, void* fft_dev,
                        int model = -1, const double* params_host = nullptr, int normalize = 0, void* values_dev = nullptr) {
  if ((model < 0 ? !values_host : !params_host) || (!fft_host && !fft_dev)) return fail(RPSF_E_BADARG, "null argument");
  if (model > MODEL_MOFFAT) return fail(RPSF_E_BADARG, "unknown PSF model");
  if (count <= 0) return count == 0 ? RPSF_OK : fail(RPSF_E_BADARG, "negative count");
  return dispatch_n(patch_size, [&]<class C>() -> int {
    HIP_TRY(hipSetDevice(device));
    uint16_t* d_tab = nullptr;
    cf* d_tw = nullptr;
    int rc = upload_tables<C>(device, &d_tab, &d_tw, nullptr);
    if (rc != RPSF_OK) return rc;
    DevBuf keep_tab, keep_tw, b_in, b_out, b_par;
    keep_tab.p = d_tab, keep_tw.p = d_tw;
    const size_t per = (size_t)C::N * C::N;
    int chunk = (int)std::max<size_t>(1, (size_t)(64u << 20) / (per * sizeof(cf)));
    if (chunk > count) chunk = count;
    if (!(model >= 0 && values_dev)) HIP_TRY(b_in.alloc(per * sizeof(float) * chunk));  // (rasterised straight into values_dev otherwise)
    if (!fft_dev) HIP_TRY(b_out.alloc(per * sizeof(cf) * chunk));
    if (model >= 0) {
      HIP_TRY(b_par.alloc(sizeof(double) * RPSF_MODEL_PARAMS_DEV * (size_t)count));
      HIP_TRY(hipMemcpy(b_par.p, params_host, sizeof(double) * RPSF_MODEL_PARAMS_DEV * (size_t)count, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&psf_fft_kernel<C>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)Launch<C>::LDS_BYTES));
    for (int first = 0; first < count && rc == RPSF_OK; first += chunk) {
      int cnt = std::min(chunk, count - first);
      cf* d_out = fft_dev ? static_cast<cf*>(fft_dev) + (size_t)first * per : b_out.as<cf>();
      float* d_in = model >= 0 && values_dev ? static_cast<float*>(values_dev) + (size_t)first * per : b_in.as<float>();
      hipError_t e = hipSuccess;
      if (model < 0) {
        e = hipMemcpy(d_in, values_host + (size_t)first * per, per * sizeof(float) * cnt, hipMemcpyHostToDevice);
      } else {
        rasterize_kernel<<<dim3((unsigned)cnt), dim3(256), 0, nullptr>>>(model, C::N, b_par.as<double>() + (size_t)first * RPSF_MODEL_PARAMS_DEV,
                                                                      normalize, d_in);
        e = hipGetLastError();
      }
      if (e != hipSuccess) { rc = fail(RPSF_E_HIP, hipGetErrorString(e)); break; }
      constexpr int TEAMS = Launch<C>::TEAMS;
      unsigned grid = (unsigned)((cnt + TEAMS - 1) / TEAMS);
      psf_fft_kernel<C><<<dim3(grid), dim3(Launch<C>::WG), Launch<C>::LDS_BYTES, nullptr>>>(d_in, cnt, d_tab, d_tw, d_out);
      e = hipGetLastError();
      if (e == hipSuccess && fft_host)
        e = hipMemcpy(reinterpret_cast<cf*>(fft_host) + (size_t)first * per, d_out, per * sizeof(cf) * cnt, hipMemcpyDeviceToHost);
      if (e == hipSuccess && !fft_host) e = hipDeviceSynchronize();  // d_in is reused by the next chunk
      if (e != hipSuccess) rc = fail(RPSF_E_HIP, hipGetErrorString(e));
    }
    return rc;
  });
}

extern "C" int rpsf_psf_fft(int device, int patch_size, int count, const float* values_host, float* fft_host) {
  return psf_fft_impl(device, patch_size, count, values_host, fft_host, nullptr);
}
extern "C" int rpsf_psf_fft_device(int device, int patch_size, int count, const float* values_host, void* fft_c64_dev) {
  return psf_fft_impl(device, patch_size, count, values_host, nullptr, fft_c64_dev);
}
static_assert(RPSF_MODEL_PARAMS == RPSF_MODEL_PARAMS_DEV && RPSF_MODEL_ELLIPTICAL_GAUSSIAN == MODEL_ELLIPTICAL_GAUSSIAN &&
              RPSF_MODEL_MOFFAT == MODEL_MOFFAT, "rpsf.h and the kernel agree on the model table");
extern "C" int rpsf_psf_model_fft_device(int device, int model, int patch_size, int count, const double* params_host,
                                         int normalize, void* values_f32_dev, void* fft_c64_dev) {
  if (model < 0) return fail(RPSF_E_BADARG, "unknown PSF model");
  return psf_fft_impl(device, patch_size, count, nullptr, nullptr, fft_c64_dev, model, params_host, normalize, values_f32_dev);
}

// ------------------------------------------------------------------------------------------------
// Host-side saturation fill (regularizepsf/transform.py:135-138).  Sequential by definition: masked pixels
// are visited in row-major order and each is replaced by the NaN-ignoring mean of the window
// [i - w/2, i + w/2) x [j - w/2, j + w/2), so later pixels see earlier fills.  Window bounds follow Python
// slice semantics (a negative start wraps once; an empty window gives NaN), as in the reference.
// ------------------------------------------------------------------------------------------------

extern "C" int rpsf_saturation_fill(double* padded, int rows, int cols, const uint8_t* mask, int neighborhood_width) {
  if (!padded || !mask || rows <= 0 || cols <= 0) return fail(RPSF_E_BADARG, "bad argument");
  const long h = neighborhood_width / 2;  // floor division, like Python's // for the non-negative widths in use
  const double nan = std::nan("");
  for (long i = 0; i < rows; ++i)
    for (long j = 0; j < cols; ++j) {
      if (!mask[i * cols + j]) continue;
      long r0, r1, c0, c1;
      py_slice(i - h, i + h, rows, &r0, &r1);
      py_slice(j - h, j + h, cols, &c0, &c1);
      double sum = 0.0;
      long cnt = 0;
      for (long r = r0; r < r1; ++r)
        for (long c = c0; c < c1; ++c) {
          double v = padded[r * cols + c];
          if (v == v) sum += v, ++cnt;
        }
      padded[i * cols + j] = cnt ? sum / (double)cnt : nan;
    }
  return RPSF_OK;
}

// ------------------------------------------------------------------------------------------------
// device memory helpers
// ------------------------------------------------------------------------------------------------
extern "C" int rpsf_dev_alloc(int device, size_t bytes, void** out) {
  if (!out) return fail(RPSF_E_BADARG, "null argument");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
  return RPSF_OK;
}
extern "C" int rpsf_dev_free(int device, void* ptr) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipFree(ptr));
  return RPSF_OK;
}
extern "C" int rpsf_memcpy_h2d(int device, void* dst, const void* src, size_t bytes) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return RPSF_OK;
}
extern "C" int rpsf_memcpy_d2h(int device, void* dst, const void* src, size_t bytes) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return RPSF_OK;
}
extern "C" int rpsf_device_synchronize(int device) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipDeviceSynchronize());
  return RPSF_OK;
}

// ------------------------------------------------------------------------------------------------
// RCCL neighbour exchange (loaded lazily so that single-GPU use has no RCCL dependency)
// ------------------------------------------------------------------------------------------------
struct NcclId { char b[128]; };
typedef int (*fn_get_uid)(NcclId*);
typedef int (*fn_comm_init)(void**, int, NcclId, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_sendrecv)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_recv)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_group)(void);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_errstr)(int);
typedef int (*fn_comm_count)(void*, int*);

static struct {
  void* h;
  fn_get_uid get_uid;
  fn_comm_init comm_init;
  fn_comm_destroy comm_destroy;
  fn_sendrecv send;
  fn_recv recv;
  fn_group group_start, group_end;
  fn_allreduce allreduce;
  fn_errstr errstr;
  fn_comm_count comm_count;
} g_rccl;

static int load_rccl() {
  static std::mutex guard;  // communicators may be created from several threads
  std::lock_guard<std::mutex> lock(guard);
  if (g_rccl.h) return RPSF_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return fail(RPSF_E_RCCL, std::string("cannot load librccl: ") + dlerror());
  g_rccl.get_uid = (fn_get_uid)dlsym(h, "ncclGetUniqueId");
  g_rccl.comm_init = (fn_comm_init)dlsym(h, "ncclCommInitRank");
  g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
  g_rccl.send = (fn_sendrecv)dlsym(h, "ncclSend");
  g_rccl.recv = (fn_recv)dlsym(h, "ncclRecv");
  g_rccl.group_start = (fn_group)dlsym(h, "ncclGroupStart");
  g_rccl.group_end = (fn_group)dlsym(h, "ncclGroupEnd");
  g_rccl.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
  g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
  g_rccl.comm_count = (fn_comm_count)dlsym(h, "ncclCommCount");
  if (!g_rccl.get_uid || !g_rccl.comm_init || !g_rccl.comm_destroy || !g_rccl.send || !g_rccl.recv ||
      !g_rccl.group_start || !g_rccl.group_end || !g_rccl.allreduce)
    return fail(RPSF_E_RCCL, "librccl is missing expected symbols");
  g_rccl.h = h;
  return RPSF_OK;
}
#define NCCL_TRY(expr)                                                                               \
  do {                                                                                               \
    int r_ = (expr);                                                                                 \
    if (r_ != 0)                                                                                     \
      return fail(RPSF_E_RCCL, std::string(#expr) + ": " + (g_rccl.errstr ? g_rccl.errstr(r_) : "rccl error")); \
  } while (0)

struct rpsf_comm {
  void* comm = nullptr;
  int device = 0, rank = 0, world = 1;
  hipStream_t stream = nullptr;
  double* d_scalar = nullptr;
};

extern "C" int rpsf_comm_unique_id(void* id128) {
  if (!id128) return fail(RPSF_E_BADARG, "null argument");
  int rc = load_rccl();
  if (rc != RPSF_OK) return rc;
  NCCL_TRY(g_rccl.get_uid(reinterpret_cast<NcclId*>(id128)));
  return RPSF_OK;
}

extern "C" int rpsf_comm_create(rpsf_comm** out, int device, int rank, int world, const void* id128) {
  if (!out || !id128 || rank < 0 || rank >= world) return fail(RPSF_E_BADARG, "bad argument");
  int rc = load_rccl();
  if (rc != RPSF_OK) return rc;
  HIP_TRY(hipSetDevice(device));
  auto* c = new rpsf_comm;
  c->device = device, c->rank = rank, c->world = world;
  NcclId id;
  std::memcpy(&id, id128, sizeof(id));
  int r = g_rccl.comm_init(&c->comm, world, id, rank);
  if (r != 0) {
    delete c;
    return fail(RPSF_E_RCCL, std::string("ncclCommInitRank: ") + (g_rccl.errstr ? g_rccl.errstr(r) : "error"));
  }
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&c->d_scalar, 2 * sizeof(double));
  if (e != hipSuccess) {
    rpsf_comm_destroy(c);
    return fail(RPSF_E_HIP, hipGetErrorString(e));
  }
  *out = c;
  return RPSF_OK;
}

extern "C" void rpsf_comm_destroy(rpsf_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.comm_destroy) g_rccl.comm_destroy(c->comm);
  (void)hipFree(c->d_scalar);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int rpsf_comm_seam_exchange_add(rpsf_comm* c, const void* send_dev, size_t send_count, void* recv_dev,
                                           size_t recv_count, void* accum_dev, void* stream) {
  if (!c) return fail(RPSF_E_BADARG, "null comm");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
  const bool do_send = c->rank + 1 < c->world && send_count > 0;
  const bool do_recv = c->rank > 0 && recv_count > 0;
  if (do_send || do_recv) {
    NCCL_TRY(g_rccl.group_start());
    if (do_send) NCCL_TRY(g_rccl.send(send_dev, send_count, /*ncclFloat32*/ 7, c->rank + 1, c->comm, st));
    if (do_recv) NCCL_TRY(g_rccl.recv(recv_dev, recv_count, 7, c->rank - 1, c->comm, st));
    NCCL_TRY(g_rccl.group_end());
  }
  if (do_recv) return rpsf_add_rows(c->device, accum_dev, recv_dev, recv_count, st);
  return RPSF_OK;
}

// The exchange alone (no add): for callers that overlap it with the rest of their band's work on another stream
extern "C" int rpsf_comm_seam_exchange(rpsf_comm* c, const void* send_dev, size_t send_count, void* recv_dev, size_t recv_count,
                                       void* stream) {
  if (!c) return fail(RPSF_E_BADARG, "null comm");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
  const bool do_send = c->rank + 1 < c->world && send_count > 0;
  const bool do_recv = c->rank > 0 && recv_count > 0;
  if ((do_send && !send_dev) || (do_recv && !recv_dev)) return fail(RPSF_E_BADARG, "null buffer");
  if (do_send || do_recv) {
    NCCL_TRY(g_rccl.group_start());
    if (do_send) NCCL_TRY(g_rccl.send(send_dev, send_count, /*ncclFloat32*/ 7, c->rank + 1, c->comm, st));
    if (do_recv) NCCL_TRY(g_rccl.recv(recv_dev, recv_count, 7, c->rank - 1, c->comm, st));
    NCCL_TRY(g_rccl.group_end());
  }
  return RPSF_OK;
}
extern "C" void* rpsf_comm_stream(rpsf_comm* c) { return c ? (void*)c->stream : nullptr; }

// How many ranks RCCL itself counts in the communicator (ncclCommCount): what a harness prints next to its timings to show that the
// collective path really spans the GPUs it was launched on
extern "C" int rpsf_comm_ranks(rpsf_comm* c, int* ranks) {
  if (!c || !ranks) return fail(RPSF_E_BADARG, "null argument");
  if (!g_rccl.comm_count) return fail(RPSF_E_RCCL, "librccl has no ncclCommCount");
  NCCL_TRY(g_rccl.comm_count(c->comm, ranks));
  return RPSF_OK;
}

// Work enqueued on `waiter` after this call starts only when everything enqueued on `signaller` so far has finished
extern "C" int rpsf_stream_wait(int device, void* waiter, void* signaller) {
  HIP_TRY(hipSetDevice(device));
  hipEvent_t ev = nullptr;
  HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, reinterpret_cast<hipStream_t>(signaller));
  if (e == hipSuccess) e = hipStreamWaitEvent(reinterpret_cast<hipStream_t>(waiter), ev, 0);
  (void)hipEventDestroy(ev);  // released once the recorded work has completed
  if (e != hipSuccess) return fail(RPSF_E_HIP, hipGetErrorString(e));
  return RPSF_OK;
}

// Events for callers that order work across their streams at a distance (the sharded step: "this launch may start once the add of two
// steps ago has read the buffer it writes" - a dependency that rpsf_stream_wait, which records at the time of the call, cannot express)
extern "C" int rpsf_event_create(int device, void** event) {
  if (!event) return fail(RPSF_E_BADARG, "null argument");
  HIP_TRY(hipSetDevice(device));
  hipEvent_t ev = nullptr;
  HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  *event = ev;
  return RPSF_OK;
}
extern "C" int rpsf_event_record(void* event, void* stream) {
  if (!event) return fail(RPSF_E_BADARG, "null event");
  HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(event), reinterpret_cast<hipStream_t>(stream)));
  return RPSF_OK;
}
extern "C" int rpsf_stream_wait_event(void* stream, void* event) {
  if (!event) return fail(RPSF_E_BADARG, "null event");
  HIP_TRY(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), reinterpret_cast<hipEvent_t>(event), 0));
  return RPSF_OK;
}
extern "C" int rpsf_event_destroy(void* event) {
  if (event) HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
  return RPSF_OK;
}

// accum[0:count] += src[0:count] on `device` (kernel K4; the add of the seam exchange, exported for callers that move
// the seam rows themselves)
static int add_rows_impl(int device, void* accum_dev, const void* src_dev, size_t count, int max_workgroups, void* stream) {
  if (!accum_dev || !src_dev) return fail(RPSF_E_BADARG, "null argument");
  if (count == 0) return RPSF_OK;
  HIP_TRY(hipSetDevice(device));
  const int block = 256;
  size_t grid = ((count + 3) / 4 + block - 1) / block;
  if (max_workgroups > 0) grid = std::min<size_t>(grid, (size_t)max_workgroups);
  add_rows_kernel<<<dim3((unsigned)grid), dim3(block), 0, reinterpret_cast<hipStream_t>(stream)>>>(
      reinterpret_cast<float*>(accum_dev), reinterpret_cast<const float*>(src_dev), count);
  HIP_TRY(hipGetLastError());
  return RPSF_OK;
}
extern "C" int rpsf_add_rows(int device, void* accum_dev, const void* src_dev, size_t count, void* stream) {
  return add_rows_impl(device, accum_dev, src_dev, count, 0, stream);
}
// The same add on at most `max_workgroups` workgroups (grid-stride): for callers that run it BESIDE a persistent patch launch (the pipelined
// seam exchange) - a patch workgroup needs a whole CU, and a thousand small workgroups dispatched a moment before it would each hold one
extern "C" int rpsf_add_rows_narrow(int device, void* accum_dev, const void* src_dev, size_t count, int max_workgroups, void* stream) {
  if (max_workgroups <= 0) return fail(RPSF_E_BADARG, "max_workgroups must be positive");
  return add_rows_impl(device, accum_dev, src_dev, count, max_workgroups, stream);
}

extern "C" int rpsf_comm_allreduce_max(rpsf_comm* c, double* value) {
  if (!c || !value) return fail(RPSF_E_BADARG, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(c->d_scalar, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
  NCCL_TRY(g_rccl.allreduce(c->d_scalar, c->d_scalar + 1, 1, /*ncclFloat64*/ 8, /*ncclMax*/ 2, c->comm, c->stream));
  HIP_TRY(hipMemcpyAsync(value, c->d_scalar + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RPSF_OK;
}

extern "C" int rpsf_comm_barrier(rpsf_comm* c, void* stream) {
  if (!c) return fail(RPSF_E_BADARG, "null comm");
  HIP_TRY(hipSetDevice(c->device));
  if (stream) HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
  double v = 0.0;
  return rpsf_comm_allreduce_max(c, &v);
}
