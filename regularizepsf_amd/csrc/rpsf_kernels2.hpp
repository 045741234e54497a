// rpsf_kernels2.hpp - K1 of the three-stage plans (N = 128, 256), second generation: 16-byte global units, 8-byte
// LDS units, the two column-parity halves of the patch pipelined through the stages (rpsf_core2.hpp), and its K pack
// kernel.  Included by rpsf.hip after rpsf_kernels.hpp (PatchParams, barriers, STAMP, the direct-mode protocol).
#pragma once

template <class C>
struct Launch2 {
  static constexpr int WG = C::T;
  static constexpr int OT_WORDS = C::ORBIT_ROUNDS * 64;
  static constexpr int TABLE_FLOATS = 3 * C::N + OT_WORDS;  // twiddles (N complex) + window (N) + bin pairs of the self-paired groups
  static constexpr size_t LDS_BYTES = (size_t)TABLE_FLOATS * sizeof(float) + (size_t)C::LDS_UNITS * sizeof(cf);
  static_assert(TABLE_FLOATS % 4 == 0, "the exchange buffer must stay 16-byte aligned");
};

// 16-byte load that bypasses this CU's L1 (agent scope), through a raw buffer descriptor so that the compiler tracks it
__device__ __forceinline__ f32x4 load16_sc1(const float* base, size_t span_bytes, const float* p) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, span_bytes < 0x7fffffffu ? (int)span_bytes : -1, 0x00020000);
  const i32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)((p - base) * sizeof(float)), 0, /*sc1*/ 16);
  return f32x4{__int_as_float(q.x), __int_as_float(q.y), __int_as_float(q.z), __int_as_float(q.w)};
}

// Development-only timing ablations (results are wrong; never set by regularizepsf_amd/build.py):
//   RPSF2_ABL_NOGATHER / _NOK / _NOSTORE: no pixel loads / no K loads / no output stores;
//   RPSF2_ABL_NOLDS: no LDS exchanges (barriers stay); RPSF2_ABL_NOBAR: no workgroup barriers either;
//   RPSF2_ABL_NOVALU: no butterflies and no pair operations;
//   RPSF_STAMPS (+ RPSF_WAVE_STAMPS): per-phase timestamps (rpsf_kernels.hpp, STAMP).
// (In a fused / persistent launch RPSF2_ABL_NOSTORE turns the plane stores into no-ops that keep the values live, so that the protocol runs on.)
// Inside the kernels these are `if constexpr (dev::...)` branches - every build parses them; tests/test_cabi.py compiles the ablation
// set so that they cannot rot.  The switches of the closed round-2 ... round-4 experiments (split / wide skeletons, carry buffers, K
// depth, naps, packed arithmetic ...) are gone from the sources; what they measured is in DESIGN.md 5.1-5.5 and profiles/r02* ... r04*.
namespace dev {
#if defined(RPSF2_ABL_NOGATHER)
constexpr bool NOGATHER = true;
#else
constexpr bool NOGATHER = false;
#endif
#if defined(RPSF2_ABL_NOK)
constexpr bool NOK = true;
#else
constexpr bool NOK = false;
#endif
#if defined(RPSF2_ABL_NOSTORE)
constexpr bool NOSTORE = true;
#else
constexpr bool NOSTORE = false;
#endif
#if defined(RPSF2_ABL_NOVALU)
constexpr bool NOVALU = true;
#else
constexpr bool NOVALU = false;
#endif
// timing skeleton of VERDICT round 5 item 2 (wrong results): the right half of every patch is not stored and the tile sums read the two planes that a
// lattice-row pre-sum would leave - the plane traffic of "right half of patch j added into the left half of patch j + 1 on chip", at no cost
#if defined(RPSF2_SKEL_PRESUM)
constexpr bool SKEL_PRESUM = true;
#else
constexpr bool SKEL_PRESUM = false;
#endif
#if defined(RPSF_STAMPS)
constexpr bool STAMPS = true;
#else
constexpr bool STAMPS = false;
#endif
}  // namespace dev
#if defined(RPSF2_ABL_NOLDS)
#define ABL_LDS(...) ((void)0)
#else
#define ABL_LDS(...) __VA_ARGS__
#endif
#if defined(RPSF2_ABL_NOVALU)
#define ABL_VALU(...) ((void)0)
#else
#define ABL_VALU(...) __VA_ARGS__
#endif
// (RPSF2_ABL_NOBAR_MASK: timing builds without some of the six exchange barriers of a pass - bit i = barrier i of ABL_BARI(i), 2 ... 7; results wrong)
#if defined(RPSF2_ABL_NOBAR_MASK)
#define ABL_BARI(i) do { if constexpr (!((RPSF2_ABL_NOBAR_MASK >> (i)) & 1)) lds_barrier(); } while (0)
#else
#define ABL_BARI(i) ABL_BAR()
#endif
#if defined(RPSF2_ABL_NOBAR)
#define ABL_BAR() ((void)0)
#else
#define ABL_BAR() lds_barrier()
#endif

// Persistent form (fused plane sum only): a workgroup that has finished its patch takes the next slot of its XCD's chunk from
// a queue and JUMPS BACK TO THE FIRST INSTRUCTION OF THE KERNEL with the state the hardware hands a fresh workgroup
// (s[0:1] = kernarg segment, s2 = workgroup id x - the virtual block of the new slot -, v0 = workitem id x, exec = all lanes:
// the kernel descriptor asks for nothing else; regularizepsf_amd/build.py checks that against the built code object).  To the
// compiler the kernel stays straight-line code - a source-level loop costs it 60-650 B of scratch per lane and 15 % of the
// kernel - and the 5 us between the last store of one workgroup and the first load of its successor (s_endpgm,
// dispatch, kernarg and descriptor loads) shrink to the queue draw, which is in flight under the stores.
// SYM must be the kernel's own (extern "C") symbol.
#if defined(__HIP_DEVICE_COMPILE__)
#define RPSF_REENTER(SYM, BLK_, TID_)                                                                      \
  do {                                                                                                     \
    const void* ka_ = __builtin_amdgcn_kernarg_segment_ptr();                                              \
    asm volatile(                                                                                          \
        "s_mov_b64 s[92:93], %[ka]\n\t"                                                                    \
        "s_mov_b32 s94, %[blk]\n\t"                                                                        \
        "v_mov_b32 v0, %[tid]\n\t"                                                                         \
        "s_getpc_b64 s[90:91]\n"                                                                           \
        "1:\n\t"                                                                                           \
        "s_add_u32 s90, s90, " #SYM "-1b\n\t"                                                              \
        "s_addc_u32 s91, s91, -1\n\t"                                                                      \
        "s_mov_b64 s[0:1], s[92:93]\n\t"                                                                   \
        "s_mov_b32 s2, s94\n\t"                                                                            \
        "s_mov_b64 exec, -1\n\t"                                                                           \
        "s_setpc_b64 s[90:91]\n\t" ::[ka] "s"(ka_),                                                        \
        [blk] "s"(BLK_), [tid] "v"(TID_)                                                                   \
        : "s90", "s91", "s92", "s93", "s94", "s0", "s1", "s2", "v0", "memory", "scc");                     \
    __builtin_unreachable();                                                                               \
  } while (0)
#else
#define RPSF_REENTER(SYM, BLK_, TID_) ((void)0)
#endif
struct NoReenter {
  static constexpr bool enabled = false;
  __device__ __forceinline__ void operator()(unsigned, unsigned) const {}
};

// Image prefetch by the head summing workgroups of a persistent launch (256-pixel plan).  A frame that was not corrected a moment ago is
// not in the Infinity Cache: every first touch of a pixel by a gather then pays the HBM latency on the patch's critical path (8 different
// frames in rotation: +10 % per apply, all of it the image - scripts/distinct_frames.py).  Head workgroup h walks the entries h / 8,
// h / 8 + sum_first / 8, ... of chunk h % 8's tile list - the lattice tiles (N/2 x N/2 pixels) in the order in which the patch workgroups
// of that XCD first gather them, every tile once - and touches one dword per 128-byte line, one line per thread and tile, PF_TILES tiles
// per step and a step every PF_GAP_TICKS (10 ns ticks), as the side job of its summing loop (sum_tiles_worker): paced like this the lines
// arrive in the memory-side cache (and the XCD's L2) ahead of the gathers of every round but the first without a burst at the start of the
// launch (all tiles at once: profiles/r03v, +3 % on a frame that is cached anyway).  OPT-IN (rpsf_plan_set_image_prefetch): worth -4 % per
// apply on a stream of new 4096^2 frames and nothing on a repeated one, but +8 % at 8192^2 - that frame is larger than the cache, with or
// without a limit on how far the prefetch may run ahead of the chunk's queue - and +3 % on a band of 585 patches (profiles/r03w).  Nobody
// needs the values: a step's loads are consumed
// (added to a sink) at the beginning of the next step, when they have long arrived, so no wave stalls on them.
template <class C>
struct ImagePrefetch {
  static constexpr int HALF = C::N / 2, LPR = HALF / 32;  // 128-byte lines per tile row
  static constexpr int PF_TILES = 4;
  static constexpr unsigned long long PF_GAP_TICKS = 200;
  const PatchParams& p;
  uint32_t first, count, e;
  int x, slots, step, dr, dc;
  unsigned long long last = 0;
  float d[PF_TILES] = {};
  float sink = 0.f;
  __device__ __forceinline__ ImagePrefetch(const PatchParams& pp, int blk, bool on) : p(pp) {
    static_assert(HALF * LPR == C::T, "one line per thread and tile");
    x = blk & 7;
    const int left = p.n_patches - x * p.chunk;
    slots = left < p.chunk ? (left > 0 ? left : 0) : p.chunk;
    first = on ? p.prefetch_first[x] : 0, count = on && slots > 0 ? p.prefetch_first[x + 1] - first : 0;
    e = (uint32_t)(blk >> 3), step = p.sum_first >> 3;
    dr = (int)threadIdx.x / LPR, dc = ((int)threadIdx.x % LPR) * 32;
  }
  __device__ __forceinline__ void land() {  // the previous step's loads have arrived: their registers may go
    StaticFor<0, PF_TILES>::run([&]<int U>() RPSF_AI { sink += d[U], d[U] = 0.f; });
  }
  __device__ __forceinline__ void finish() {
    land();
    asm volatile("" ::"v"(sink));  // (the loads must happen; their values do not matter)
  }
  __device__ __forceinline__ bool operator()() {
    if (e >= count) {
      land();
      return false;
    }
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    if (now - last < PF_GAP_TICKS) return true;
    last = now;
    land();
    const ImageView& im = p.im;
    StaticFor<0, PF_TILES>::run([&]<int U>() RPSF_AI {
      if (e < count) {
        const uint32_t tile = p.prefetch_tiles[first + e];
        const int r = p.ts.lat_r0 + (int)(tile / (uint32_t)p.ts.ntj) * HALF + dr, c = p.ts.lat_c0 + (int)(tile % (uint32_t)p.ts.ntj) * HALF + dc;
        if (r >= 0 && r < im.H && r >= im.row0 && r < im.row0 + im.rows && c >= 0 && c < im.W)
          d[U] = *reinterpret_cast<const volatile float*>(im.img + (size_t)(r - im.row0) * im.ld + c);  // (plain policy: the line is to stay)
      }
      e += (uint32_t)step;
    });
    return true;
  }
};

// HOT: the instantiation the persistent kernels are compiled from - a fused launch on colour planes whose geometry the launcher has
// checked (hot_geometry, rpsf.hip): no float atomics, no direct mode, no pixel-by-pixel rim paths in the code (they cost the 256-pixel
// kernel 29 spilled SGPRs and half of its 200 KB).  Everything else runs the one-patch-per-workgroup kernel patch_kernel2.


// The second-generation packer fed from the two PSF spectra (KFromSpectra, rpsf_kernels.hpp)
template <class C>
__global__ void pack_spectra_kernel2(const cf* __restrict__ s_fft, const cf* __restrict__ t_fft, float alpha, float eps, int n_patches,
                                     const uint16_t* __restrict__ tab, const uint32_t* __restrict__ ot, cf* __restrict__ g, cf* __restrict__ gs);

template <class C, class REENTER, bool HOT = false, bool KNT = true, bool PLANE_NT = false>
__device__ __forceinline__ void patch_body2(const PatchParams& p, REENTER&& reenter) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int T = C::T, N = C::N;
  constexpr bool PERSIST = std::remove_reference_t<REENTER>::enabled;
  const int tu = threadIdx.x, t = tu;
  // Fused plane sum: a few workgroups at the head of the grid sum finished tiles beside the patches for the whole
  // launch (the patches leave half of the HBM bandwidth unused), the ones at its tail take the CUs the partial last
  // round of patches leaves idle.  All of them draw tiles from one queue.
  // (persistent form: bit 30 of the block index says that this is a re-entry - the tables are in LDS already and the park
  // words hold the tiles of the previous patch, still to be counted)
  if constexpr (dev::STAMPS)  // diagnostic builds: when did this pass reach the kernel's first instructions (stamp 14, through an LDS word)
    if (threadIdx.x == 0) *reinterpret_cast<unsigned long long*>(reinterpret_cast<cf*>(smem + Launch2<C>::TABLE_FLOATS) + C::BUF_UNITS + 4) = __builtin_amdgcn_s_memrealtime();
  // (the tile sums of the persistent kernels are compiled for "fused" - no run-time choice between two kinds of load in front of each of the 32 of a pass:
  // config 2 -2 %; in the 256-pixel kernel -1.2 % at 4096^2 and nothing at 8192^2 once the sums index by shift and mask, profiles/r04t, r04aa)
  constexpr bool SUM_KNOWN_FUSED = HOT;
  const bool again = PERSIST && ((blockIdx.x >> 30) & 1u);
  if constexpr (dev::STAMPS)  // ... and which workgroup this is (its block index at dispatch, kept in LDS across re-entries: stamp 15)
    // (an unused word of the bin-pair table: bit 31 clear, so the walk of the self-paired bins ignores it; written again behind the table staging below)
    if (threadIdx.x == 0 && !again) reinterpret_cast<uint32_t*>(smem + 3 * C::N)[Launch2<C>::OT_WORDS - 1] = blockIdx.x;
  // (persistent form: a summing workgroup at the head of the grid has nothing to sum while the first patches are still being
  // computed - no tile is complete before a full patch period - so it computes ONE patch of its XCD's chunk first)
  const int blk = (int)(blockIdx.x & 0x3fffffffu);
  const bool head_patch = PERSIST && !again && blk < p.sum_first && p.head_patches > 0;  // workgroup-uniform
  const int pb = head_patch ? (blk & 7) : blk - p.sum_first;                             // workgroup-uniform
  bool patchy = pb >= 0 && pb < p.patch_blocks;
  int frame = 0, xrow = 0, seq = 0;
  if (patchy) {
    xrow = p.slot0 + (pb >> 3);
    if constexpr (PERSIST) {
      // Every slot comes from the XCD chunk's queue, the first one included (position 0 = slot 0): no slot is tied to a workgroup
      // that may not be resident yet, so whichever workgroups of a chunk ARE resident drain it (forward progress: rpsf.hip, launch_patches).
      // (the head summing workgroups - the first of the grid to be dispatched - take the first positions of their chunks without a draw,
      // so that the number of draws of a launch, which the never-reset queue counters are accounted by, does not depend on a race)
      if (head_patch) {
        xrow = blk >> 3;
      } else if (!again) {
        unsigned* const word = reinterpret_cast<unsigned*>(reinterpret_cast<cf*>(smem + Launch2<C>::TABLE_FLOATS) + C::BUF_UNITS);
        if (tu == 0) *word = __hip_atomic_fetch_add(p.xq + (pb & 7) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - p.xq_base[pb & 7];
        lds_barrier();
        xrow = (int)__builtin_amdgcn_readfirstlane(*word);
        if (xrow < 0 || xrow > 0x0fffffff) xrow = 0x0fffffff;  // (queue positions stay far below; keeps the frame arithmetic in range)
        xrow += p.head_patches ? p.sum_first >> 3 : 0;
      }
    }
    if (p.n_frames > 1) {
      if (p.frame_major) {  // persistent batches of large frames: one frame after the other (its planes stay in the Infinity Cache)
        const int left = p.n_patches - (pb & 7) * p.chunk, mine = left < p.chunk ? (left > 0 ? left : 1) : p.chunk;  // slots of this XCD
        frame = xrow / mine;
        xrow = frame < p.n_frames ? xrow % mine : p.chunk;  // (past the last frame: no slot)
      } else {  // the frames of one patch slot side by side: its K is fetched once and served from L2 to the others
        frame = xrow % p.n_frames;
        xrow /= p.n_frames;
      }
    }
    seq = (pb & 7) * p.chunk + xrow;
    if (xrow >= p.chunk || seq >= p.n_patches) {  // workgroup-uniform
      if constexpr (!PERSIST) return;
      patchy = false;  // every workgroup of a persistent launch ends as a summing one
    }
  }
  cf* const lds = reinterpret_cast<cf*>(smem + Launch2<C>::TABLE_FLOATS);
  cf* const park = lds + C::BUF_UNITS;
  // persistent form: the previous patch of this workgroup is counted on its tiles once its plane stores have drained
  [[maybe_unused]] auto count_previous = [&]() RPSF_AI {
    if (tu < 4) __hip_atomic_fetch_add(p.tile_done + reinterpret_cast<const unsigned*>(park)[1 + t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  if (!patchy) {
    if constexpr (PERSIST) {
      if (again) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        count_previous();
      }
    }
    if constexpr (PERSIST && C::T == 512) {
      if (!again && blk < p.sum_first) {  // a head summing workgroup: the image prefetch is its side job
        ImagePrefetch<C> prefetch(p, blk, p.prefetch && p.n_frames <= 1);
        sum_tiles_worker<ImagePrefetch<C>&, 8, SUM_KNOWN_FUSED>(p.ts, 0, 1, prefetch);
        prefetch.finish();
        return;
      }
    }
    sum_tiles_worker<NoSideJob, 8, SUM_KNOWN_FUSED>(p.ts, 0, 1);
    return;
  }
  ImageView im = p.im;
  OutView ov = p.ov;
  im.img += (size_t)frame * p.im_frame_floats;
  ov.out += (size_t)frame * p.ov_frame_floats;
  // The slot descriptor through the scalar cache (a constant-address-space load of a uniform address): as a vector load it
  // would queue behind the plane stores of the workgroup's previous patch - VMEM returns in order - and the gather, which
  // needs the corner, would not even be issued before those stores are acknowledged.
  // (fetching it a pass ahead, into LDS, measured nothing: profiles/r04p)
  typedef const int __attribute__((address_space(4))) cint_as4;
  const cint_as4* dptr = (const cint_as4*)(const void*)(p.desc + (p.seq_base + seq));
  const int4 dsc = make_int4(dptr[0], dptr[1], dptr[2], dptr[3]);
  const int patch = dsc.z;
  // Start-up stagger.  The workgroups of one round move in lock step otherwise - all CUs stream K at one moment, store at
  // another, and the memory system alternates between idle and saturated.  The first resident workgroup of each CU is
  // held back by a different fraction of stagger_ticks (later workgroups inherit the offset of the one they replace).
  if (p.stagger_ticks > 0 && pb < p.stagger_blocks && !again && !head_patch) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // evenly spaced delays in bit-reversed order of the chunk row (profiles/r02av: -1 % against hashed delays at 12 us)
    // (a second, longer range of delays for the workgroups that will run one patch fewer - they have a period of slack - measured worse:
    // profiles/r04h, 0.195 ... 0.203 against 0.190 ... 0.194 ms)
    const unsigned long long wait = (unsigned long long)(__brev((unsigned)pb >> 3) >> 22) * p.stagger_ticks >> 10;
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
  STAMP(0);
  if constexpr (dev::STAMPS) {
    if (threadIdx.x == 0) p.stamps[(size_t)patch * 16 + 14] = *reinterpret_cast<const unsigned long long*>(reinterpret_cast<cf*>(smem + Launch2<C>::TABLE_FLOATS) + C::BUF_UNITS + 4);
    if (threadIdx.x == 0) p.stamps[(size_t)patch * 16 + 15] = 1 + (again ? reinterpret_cast<const uint32_t*>(smem + 3 * C::N)[Launch2<C>::OT_WORDS - 1] : blockIdx.x);
  }
  const int pr = dsc.x + p.origin_row, pc = dsc.y + p.origin_col;
  const int kpatch = patch;
  const cf* g = p.g + (size_t)kpatch * C::G_PER_PATCH;
  cf* tw = reinterpret_cast<cf*>(smem);
  float* win = smem + 2 * N;
  uint32_t* ot = reinterpret_cast<uint32_t*>(smem + 3 * N);
  // Table values first (VMEM returns in order: they arrive ahead of the pixels requested right behind them and are
  // put into LDS while the gather is in flight).
  static_assert(N <= 2 * T && Launch2<C>::OT_WORDS <= T, "one or two table entries per thread");
  cf tw0 = {0.f, 0.f}, tw1 = {0.f, 0.f};
  float wn0 = 0.f, wn1 = 0.f;
  uint32_t ot0 = 0;
  if (!again) {
    tw0 = p.tw[t < N ? t : 0], tw1 = p.tw[t + T < N ? t + T : 0];
    wn0 = p.win[t < N ? t : 0], wn1 = p.win[t + T < N ? t + T : 0];
    ot0 = p.pairtab[t < Launch2<C>::OT_WORDS ? t : 0];
  }
  GroupIds<C> gids;
  gids.load(p.tab, t);
  cf v[64];
  const bool fast = patch_inside2<C>(pr, pc, im.H, im.W, im.row0, im.rows) && quads_aligned(im.img, im.ld, pc);
  int* maps = reinterpret_cast<int*>(lds);
  if (!fast) {
    build_pad_maps<C>(t, maps, im, pr, pc);
    lds_barrier();
  }
  if constexpr (dev::NOGATHER) {
#pragma unroll
    for (int j = 0; j < 64; ++j) v[j] = cf{(float)(t + j), (float)(t - j)};
  } else {
    load_raw2<C, HOT>(t, v, im, pr, pc, fast, maps);
  }
  if (!again) {
    if (t < N) tw[t] = tw0, win[t] = wn0;
    if (t + T < N) tw[t + T] = tw1, win[t + T] = wn1;
    if (t < Launch2<C>::OT_WORDS) ot[t] = ot0;
  }
  // tables staged; the maps (which share LDS with the exchange buffer) are no longer needed.  (A re-entered pass over an interior patch has
  // neither to wait for: its first LDS operations are wave-local, and the previous pass ended with a barrier.)
  if (!(PERSIST && again && fast)) lds_barrier();  // (workgroup-uniform)
  if constexpr (dev::STAMPS)  // the workgroup's block index at dispatch, kept in an unused word of the bin-pair table
    if (PERSIST && !again && t == 0) ot[Launch2<C>::OT_WORDS - 1] = blockIdx.x & 0x3fffffffu;
  // Persistent launches: the next slot of this XCD's chunk (xq counters are never reset: this launch owns the positions from xq_base[xcd]
  // on, position 0 = slot 0) and the tile this lane will count the patch on.
  // 256-pixel plan (EARLY_DRAW): REQUESTED right after the frequency step and PUT INTO LDS IN FRONT OF THE PLANE STORES, a whole inverse
  // transform later, when the two round trips - an atomic with return and a load, both in wave 0 - are long over.  Used behind the stores,
  // as until round 4, the values cost a drain: the stores sit in divergent branches (interior / rim paths), the compiler cannot count them,
  // so the wait in front of the use was `s_waitcnt vmcnt(0)` - wave 0 waited for the acknowledgement of its 32 write-through stores, and
  // the seven other waves for wave 0 at the barrier behind it (the "No drain here" below was not true of the code object).
  constexpr bool EARLY_DRAW = PERSIST && C::SPLIT_ROWS;
  // (the 128-pixel plan, MID_DRAW: requested behind the last barrier of the inverse exchange - across its slot-by-slot frequency step the two values cost 44
  // spilled registers - and put into LDS in front of the stores all the same: configs 2 / 5 -0.7 % / -0.5 %, profiles/r04r)
  constexpr bool MID_DRAW = PERSIST && !EARLY_DRAW;
  unsigned drawn = 0, qword = 0;  // (one register each: lane tu < 4 holds the word of its own tile)
  // (Tried on top of it, round 4, profiles/r04o: every wave reads the drawn position behind the last barrier of the inverse exchange, fetches the next
  // slot's descriptor and touches one dword of each of the next patch's 2048 lines, a transform stage ahead of its gather - new frames 0.2095 -> 0.204 ms,
  // but the repeated frame of the headline loop 0.1916 -> 0.200: the descriptor's scalar load sits on the chain right behind a barrier.  The opt-in
  // prefetch by the head summing workgroups does better on new frames, 0.201, at no cost to the repeated one.  Not kept.)
  [[maybe_unused]] auto draw_next = [&]() RPSF_AI {
    if constexpr (PERSIST) {
      if (HOT || p.tile_done) {
        if (tu < 4) qword = reinterpret_cast<const uint32_t*>(p.quads + (p.seq_base + seq))[tu];
        if (tu == 0 && !head_patch)
          drawn = __hip_atomic_fetch_add(p.xq + (pb & 7) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - p.xq_base[pb & 7] +
                  (p.head_patches ? (unsigned)(p.sum_first >> 3) : 0u);
      }
    }
  };
  window_patch2<C>(t, v, win);
  STAMP(1);
  // ---- forward: the halves leapfrog through stage 1, X1 (wave-local) and stage 2 ----
  ABL_VALU(stage1h<C, 0, false>(t, v, tw));
  ABL_LDS(x1_write2<C, 0>(t, v, lds));
  ABL_VALU(stage1h<C, 1, false>(t, v, tw));
  wave_lds_sync();
  ABL_LDS(x1_read2<C, 0>(t, v, lds));
  asm volatile("" ::: "memory");  // (compiler-only: the hardware keeps a wave's DS operations in order, the optimiser must too - other lanes of the wave read what this lane writes)
  ABL_LDS(x1_write2<C, 1>(t, v, lds));  // (a wave's DS operations complete in order: these writes cannot overtake the reads)
  STAMP(2);
  ABL_VALU(stage2h<C, 0, false>(t, v, tw));
  wave_lds_sync();
  ABL_LDS(x1_read2<C, 1>(t, v, lds));
  STAMP(3);
  cf k[2 * C::KCH];
  if constexpr (dev::NOK) {
#pragma unroll
    for (int j = 0; j < 2 * C::KCH; ++j) k[j] = cf{1.0f + j, 0.5f * t};
  } else {
    load_k_chunk2<C, 0, KNT>(t, k, g);  // in flight across the exchange below (raw barriers do not drain VMEM)
  }
  cf ko[2 * C::ORBIT_ROUNDS];
  if (t < 64) {
    const cf* gs = p.gs + (size_t)kpatch * C::GS_PER_PATCH;
    StaticFor<0, C::ORBIT_ROUNDS>::run([&]<int R>() RPSF_AI { load_stream16(gs + (size_t)(R * 64 + t) * 2, ko[2 * R], ko[2 * R + 1]); });
  }
  // No barrier between X1 and X2: a wave's X1 region IS its own two planes of the X2 image (Cfg2::X1_OWN_ROWS), which only its own
  // x2_mid_write2 touches - its DS operations complete in order.
  asm volatile("" ::: "memory");  // (compiler-only: the hardware keeps a wave's DS operations in order, the optimiser must too - other lanes of the wave read what this lane writes)
  ABL_LDS(x2_mid_write2<C, 0>(t, v, lds));
  ABL_VALU(stage2h<C, 1, false>(t, v, tw));
  ABL_BARI(2);
  if constexpr (PERSIST) {
    // every wave has its pixels, so - VMEM returns in order - the plane stores of the workgroup's previous patch, issued
    // ahead of them, are acknowledged: that patch can be counted on its tiles
    if (again) count_previous();
  }
  ABL_LDS(x2_last_read2<C, 0>(gids, v, lds));
  ABL_BARI(3);
  ABL_LDS(x2_mid_write2<C, 1>(t, v, lds));
  if constexpr (C::SPLIT_ROWS) ABL_VALU(stage3_rows<C, false, 0, 0>(t, gids, v));  // the row DFTs of the half that has arrived, under the exchange of the other
  ABL_BARI(4);
  ABL_LDS(x2_last_read2<C, 1>(gids, v, lds));
  // no barrier: every X2 unit is read by exactly one thread, the same one that rewrites it below
  STAMP(4);
  // ---- frequency step ----
  if constexpr (C::SPLIT_ROWS) ABL_VALU(stage3_rows<C, false, 1, 0>(t, gids, v));
  ABL_VALU(freq_a<C>(t, gids, v, park));
  if (t < 64) {  // wave 0: the bin pairs of the four self-paired groups, one pair per lane
    wave_lds_sync();
    StaticFor<0, C::ORBIT_ROUNDS>::run([&]<int R>() RPSF_AI { self_orbit<C>(t, R, ot, ko[2 * R], ko[2 * R + 1], tw, park); });
    wave_lds_sync();
  }
  STAMP(5);
  if constexpr (dev::NOVALU) {
    StaticFor<1, C::NCHUNK>::run([&]<int CI>() RPSF_AI {  // keep the K stream: every chunk is requested and consumed
      cf acc = k[0];
      StaticFor<1, 2 * C::KCH>::run([&]<int I>() RPSF_AI { acc = acc + k[I]; });
      v[CI] = v[CI] + acc;
      load_k_chunk2<C, CI>(t, k, g);
    });
    v[0] = v[0] + k[0] + k[15];
  } else {
    freq_b<C, KNT>(t, gids, v, k, g, tw, park);
  }
  if constexpr (C::SPLIT_ROWS) ABL_VALU(stage3_rows<C, true, 0, 0>(t, gids, v));
  STAMP(6);
  if constexpr (EARLY_DRAW) draw_next();
  // ---- inverse ----
  ABL_LDS(x2_last_write2<C, 0>(gids, v, lds));
  if constexpr (C::SPLIT_ROWS) ABL_VALU(stage3_rows<C, true, 1, 0>(t, gids, v));
  ABL_BARI(5);
  ABL_LDS(x2_mid_read2<C, 0>(t, v, lds));
  ABL_BARI(6);
  ABL_LDS(x2_last_write2<C, 1>(gids, v, lds));
  ABL_VALU(stage2h<C, 0, true>(t, v, tw));
  // (park: idle since the frequency step; the previous pass's words were read by count_previous long ago)
  auto park_draw = [&]() RPSF_AI {
    if (HOT || p.tile_done) {
      if (tu == 0) *reinterpret_cast<unsigned*>(park) = drawn;
      if (tu < 4) reinterpret_cast<unsigned*>(park)[1 + tu] = (unsigned)frame * p.n_tiles + quad_tile(qword);  // (this frame's counters)
    }
  };
  ABL_BARI(7);
  ABL_LDS(x2_mid_read2<C, 1>(t, v, lds));
  if constexpr (EARLY_DRAW) park_draw();
  // (no barrier: the wave's X1 region is the planes it has just read, see above)
  if constexpr (MID_DRAW) draw_next();
  STAMP(7);
  asm volatile("" ::: "memory");  // (compiler-only: the hardware keeps a wave's DS operations in order, the optimiser must too - other lanes of the wave read what this lane writes)
  ABL_LDS(x1_write2<C, 0>(t, v, lds));
  ABL_VALU(stage2h<C, 1, true>(t, v, tw));
  wave_lds_sync();
  ABL_LDS(x1_read2<C, 0>(t, v, lds));
  asm volatile("" ::: "memory");  // (compiler-only: the hardware keeps a wave's DS operations in order, the optimiser must too - other lanes of the wave read what this lane writes)
  ABL_LDS(x1_write2<C, 1>(t, v, lds));
  STAMP(8);
  ABL_VALU(stage1h<C, 0, true>(t, v, tw));
  wave_lds_sync();
  ABL_LDS(x1_read2<C, 1>(t, v, lds));
  ABL_VALU(stage1h<C, 1, true>(t, v, tw));
  STAMP(9);
  // ---- overlap-add ----
  if constexpr (dev::NOSTORE) {
    if (!p.tile_done) {  // keep every value live but store (almost) nothing
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 64; ++j) acc += v[j].x * v[j].y;
      if (acc == 123456.789f) ov.out[threadIdx.x] = acc;
      return;
    }  // (fused / persistent launches keep their protocol: the plane stores below become no-ops that keep the values live)
  }
  const int plane = HOT || ov.plane_stride ? dsc.w : 0;
  auto add = [](float* a, float val) { unsafeAtomicAdd(a, val); };
  auto pstore4 = [](float* a, f32x4 val) RPSF_AI { __builtin_nontemporal_store(val, reinterpret_cast<f32x4*>(a)); };
  auto pstore1 = [](float* a, float val) RPSF_AI { *a = val; };
  if (!HOT && p.dv.out) {  // direct overlap-add (opt-in; moves as many bytes as the planes do and waits on top - DESIGN.md)
    OutView dv = p.dv;
    dv.out += (size_t)frame * p.dv_frame_floats;
    uint32_t qw[4];
    direct_begin(p, frame, seq, qw, reinterpret_cast<uint32_t*>(park));
    STAMP(10);
    const float* dbase = dv.out;
    const size_t dspan = ((size_t)(dv.rows - 1) * dv.ld + dv.W) * sizeof(float);
    store_patch2<C>(
        t, v, ov, dv, plane, pr, pc, win, qw, add, [=]<int R1, int C1>(const float* a) RPSF_AI { return load16_sc1(dbase, dspan, a); },
        [](const float* a) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); },
        pstore4, pstore1);
    STAMP(11);
    direct_end(p, frame, plane, qw);
    STAMP(12);
  } else if (HOT || p.tile_done) {
    // Plane sum fused into this launch: the plane stores are write-through, and once they have drained the patch is
    // counted on its four tiles; the workgroups behind the patches in the grid sum a tile as soon as its count is complete.
    const float* pbase = ov.out;
    const __amdgpu_buffer_rsrc_t rsrc = plane_rsrc(pbase);
    if constexpr (!EARLY_DRAW && !MID_DRAW) draw_next();
    if constexpr (MID_DRAW) {
      if (tu == 0) *reinterpret_cast<unsigned*>(park) = drawn;
      if (tu < 4) reinterpret_cast<unsigned*>(park)[1 + tu] = (unsigned)frame * p.n_tiles + quad_tile(qword);
    }
    store_patch2<C, HOT>(
        t, v, ov, ov, plane, pr, pc, win, nullptr, add, []<int R1, int C1>(const float* a) RPSF_AI { return *reinterpret_cast<const f32x4*>(a); },
        [](const float* a) { return *a; },
        [=](float* a, f32x4 val) RPSF_AI {
          // (PLANE_NT: the instantiation for batches of frames side by side, whose planes are live all at once - see rpsf.hip, plane_nt)
          if constexpr (dev::NOSTORE) asm volatile("" ::"v"(val.x), "v"(val.y), "v"(val.z), "v"(val.w), "v"(a));
          else if constexpr (PLANE_NT) plane_store16_aux<16 | 2>(rsrc, (size_t)(a - pbase), val);
          else plane_store16_wt(rsrc, (size_t)(a - pbase), val);
        },
        [](float* a, float val) RPSF_AI { __hip_atomic_store(reinterpret_cast<unsigned*>(a), __float_as_uint(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
    STAMP(10);
    if constexpr (PERSIST) {
      // No drain here: the next patch's loads queue behind these stores anyway, and the patch is counted on its tiles from
      // inside the next pass (count_previous), when the stores are known to have been acknowledged.
      if constexpr (!EARLY_DRAW && !MID_DRAW) {
        if (tu == 0) *reinterpret_cast<unsigned*>(park) = drawn;  // (park: idle since the frequency step)
        if (tu < 4) reinterpret_cast<unsigned*>(park)[1 + tu] = (unsigned)frame * p.n_tiles + quad_tile(qword);  // (this frame's counters)
      }
      lds_barrier();
      STAMP(12);
      const unsigned nx = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const unsigned*>(park));
      const int left = p.n_patches - (pb & 7) * p.chunk;  // slots of this XCD's chunk that hold a patch (x frames: queue positions)
      // ... or, once the chunk is exhausted, a block index behind the patches: the workgroup sums tiles with the others
      const bool more = (int)nx < (left < p.chunk ? left : p.chunk) * (p.n_frames > 1 ? p.n_frames : 1);
      // (a head summing workgroup has had its patch: it re-enters under its own block index and sums from now on)
      reenter(0x40000000u | (head_patch ? (unsigned)blk : (unsigned)p.sum_first + (more ? ((nx << 3) | (unsigned)(pb & 7)) : (unsigned)p.patch_blocks)),
              (unsigned)tu);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (t < 4) {
      const uint4 q4 = p.quads[p.seq_base + seq];
      const uint32_t w = t == 0 ? q4.x : t == 1 ? q4.y : t == 2 ? q4.z : q4.w;
      __hip_atomic_fetch_add(p.tile_done + (size_t)frame * p.n_tiles + quad_tile(w), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    STAMP(12);
  } else {
    store_patch2<C>(t, v, ov, ov, plane, pr, pc, win, nullptr, add, []<int R1, int C1>(const float* a) RPSF_AI { return *reinterpret_cast<const f32x4*>(a); },
                    [](const float* a) { return *a; }, pstore4, pstore1);
  }
  STAMP(13);
}

template <class C>
__global__ __launch_bounds__(Launch2<C>::WG, 2) void patch_kernel2(PatchParams p) {  // 2 waves per SIMD: 256 registers
  patch_body2<C>(p, NoReenter());
}
// the persistent form of the 256-pixel plan (instantiated in k2_256p.hip)
// (RPSF_VGPR_CAP: development builds that leave registers for co-resident waves of another kernel; the attribute counts the
// unified register file in halves, so 124 caps the kernel at 248)
#define RPSF_VGPR_ATTR
extern "C" __global__ __launch_bounds__(512, 2) RPSF_VGPR_ATTR void patch_kernel2_256p(PatchParams p);
// ... and of the 128-pixel plan (k2_128p.hip): four 128-thread workgroups per CU hide the dispatch of one another, but only
// persistent ones keep the phase offsets of the start-up stagger
extern "C" __global__ __launch_bounds__(128, 2) void patch_kernel2_128p(PatchParams p);
// ... the same kernel with plain instead of streaming loads of the pair words (k2_128pc.hip), for launches whose K fits the Infinity Cache beside
// everything else (rpsf.hip, k_cached): the next apply, or the next frame of a batch, finds K there.  2048^2: -6 % per kernel, 8 x 2048^2: -2.3 %;
// from 3072^2 (160 MB of K) on the streaming form wins (+6 %), and the 256-pixel plan at 4096^2 loses 16 % with plain loads (profiles/r04av).
extern "C" __global__ __launch_bounds__(128, 2) void patch_kernel2_128pc(PatchParams p);
// ... and with streaming plane stores on top (k2_128pcs.hip), for batches of frames side by side whose planes - live all at once - exceed the Infinity Cache:
// 8 x 2048^2 0.3244 -> 0.3096 ms; 2 ... 4 frames, single frames of any size and the 256-pixel plan lose 2 ... 8 % with them (profiles/r04bd)
extern "C" __global__ __launch_bounds__(128, 2) void patch_kernel2_128pcs(PatchParams p);
template <class C>
struct PersistentKernel2;
template <>
struct PersistentKernel2<Cfg256v2> {
  static constexpr auto fn = &patch_kernel2_256p;
  static constexpr auto fn_k_cached = &patch_kernel2_256p;  // (no such forms)
  static constexpr auto fn_k_cached_planes_nt = &patch_kernel2_256p;
};
template <>
struct PersistentKernel2<Cfg128v2> {
  static constexpr auto fn = &patch_kernel2_128p;
  static constexpr auto fn_k_cached = &patch_kernel2_128pc;
  static constexpr auto fn_k_cached_planes_nt = &patch_kernel2_128pcs;
};

// K pack: the caller's full complex64 K (n, N, N) -> folded pair words in the stream layout [word][thread], plus the
// side array of the self-paired bin pairs
template <class C>
__global__ void pack_kernel2(const cf* __restrict__ kfull, int n_patches, const uint16_t* __restrict__ tab,
                             const uint32_t* __restrict__ ot, cf* __restrict__ g, cf* __restrict__ gs) {
  const size_t per = (size_t)C::G_PER_PATCH + C::GS_PER_PATCH;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per * n_patches) return;
  const int patch = (int)(idx / per);
  const int rem = (int)(idx % per);
  const cf* kf = kfull + (size_t)patch * C::N * C::N;
  if (rem < C::G_PER_PATCH) {
    const int b = rem & 1, t = (rem >> 1) % C::T, w = (rem >> 1) / C::T;
    g[(size_t)patch * C::G_PER_PATCH + rem] = pack_value2<C>(kf, tab, t, w, b);
  } else {
    const int r2 = rem - C::G_PER_PATCH;
    gs[(size_t)patch * C::GS_PER_PATCH + r2] = pack_orbit2<C>(kf, tab, ot, r2 >> 1, r2 & 1);
  }
}

template <class C>
__global__ void pack_spectra_kernel2(const cf* __restrict__ s_fft, const cf* __restrict__ t_fft, float alpha, float eps, int n_patches,
                                     const uint16_t* __restrict__ tab, const uint32_t* __restrict__ ot, cf* __restrict__ g, cf* __restrict__ gs) {
  const size_t per = (size_t)C::G_PER_PATCH + C::GS_PER_PATCH;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per * n_patches) return;
  const int patch = (int)(idx / per);
  const int rem = (int)(idx % per);
  const KFromSpectra kf{s_fft + (size_t)patch * C::N * C::N, t_fft + (size_t)patch * C::N * C::N, alpha, eps};
  if (rem < C::G_PER_PATCH) {
    const int b = rem & 1, t = (rem >> 1) % C::T, w = (rem >> 1) / C::T;
    g[(size_t)patch * C::G_PER_PATCH + rem] = pack_value2<C>(kf, tab, t, w, b);
  } else {
    const int r2 = rem - C::G_PER_PATCH;
    gs[(size_t)patch * C::GS_PER_PATCH + r2] = pack_orbit2<C>(kf, tab, ot, r2 >> 1, r2 & 1);
  }
}
