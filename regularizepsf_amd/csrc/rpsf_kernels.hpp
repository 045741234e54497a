// rpsf_kernels.hpp - the device code of librpsf_hip.so: K1 (patch kernel), its K pack kernel and K3 (PSF spectra) as
// templates over the plan geometry - instantiated in the k1_*.hip translation units, one group of plans each, so
// that the library builds in parallel - and, for the host translation unit only (RPSF_HOST_TU), the plan-independent
// kernels: K5 (colour-plane sum), K5' (fix-up of the direct overlap-add), K2 (transfer-kernel build), K4 (seam add)
// and the three small kernels of the hipFFT fallback.  The per-thread phase functions the kernels are made of live
// in rpsf_core.hpp / rpsf_core2.hpp, which the CPU emulators (tests/emu) compile as well.
#pragma once

// ------------------------------------------------------------------------------------------------
// Colour-plane sum by lattice tile.  out_tile = sum of the planes that have a patch over the tile (4-bit cover mask,
// fixed colour order: deterministic); a tile nobody covers is zeroed.  Used tile-list by tile-list: the tiles whose
// contributors have all finished are summed by the CUs the partial last round of patches leaves idle (same launch as
// those patches), the remaining ones by a small kernel afterwards.
// ------------------------------------------------------------------------------------------------
struct TileSum {
  const float* planes;
  size_t plane_stride;
  int ld_planes;
  float* out;
  int ld_out;
  int rows, W, row0;         // resident output window
  int lat_r0, lat_c0, half;  // full-image coordinates of lattice tile (0, 0); tile edge
  int ntj;
  const uint8_t* cover;      // per tile: 4-bit colour mask
  const uint32_t* tiles;     // the list
  int count;
  // fused with the patch launch: a tile is summed once its done counter has reached epoch * (number of contributors);
  // done == nullptr: the planes are complete (separate launch)
  const uint32_t* done;
  uint32_t epoch;
  uint32_t* queue;           // fused: next position in the list (never reset: this launch owns [queue_base, queue_base + count))
  uint32_t queue_base;
  // batches (frames that share the transfer kernel): the list has count = tiles x frames positions, position i = tile list entry
  // i / n_frames of frame i % n_frames (the frames of one patch slot run side by side, so their tiles complete together);
  // a list entry in flight is (frame << 24) | tile
  int n_frames;              // 0 or 1: one frame
  uint32_t n_tiles;          // done counters of frame f start at done + f * n_tiles
  size_t planes_frame_floats, out_frame_floats;
  int frame_major;           // the list runs frame after frame (persistent batches of large frames) instead of tile after tile
  uint32_t n_tiles_listed;   // entries of `tiles` (= count / frames)
};
__device__ __forceinline__ uint32_t sum_entry(const TileSum& p, uint32_t i) {
  if (p.n_frames <= 1) return p.tiles[i];
  if (p.frame_major) return ((i / p.n_tiles_listed) << 24) | p.tiles[i % p.n_tiles_listed];  // frame after frame
  return ((i % (uint32_t)p.n_frames) << 24) | p.tiles[i / (uint32_t)p.n_frames];
}

// 16-byte plane accesses of the fused mode: write-through stores and L1-bypassing loads (agent scope), so that a
// workgroup on another XCD reads what the patch workgroups wrote (MI355X_MICROARCH.md, inter-workgroup visibility).
// Raw buffer accesses, so that the compiler tracks them; a descriptor spans up to 4 GiB from `base`.
typedef float rpsf_f4 __attribute__((ext_vector_type(4)));
typedef int rpsf_i4 __attribute__((ext_vector_type(4)));
// (the base is the same for every lane; saying so - readfirstlane - keeps the descriptor in scalar registers, otherwise the
// compiler wraps every buffer access in a waterfall loop over the descriptor values it believes may differ between lanes)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const float* base) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)hi << 32) | lo), 0, -1, 0x00020000);
}
// Cache policy of the fused mode's plane accesses.  Both are sc1 (write-through stores, L1/L2-bypassing loads: what makes them
// visible across XCDs).  The stores are NOT marked streaming: a plane line is read by the sum a few tens of microseconds after it was
// written, and without `nt` it is still in the Infinity Cache then; the loads ARE (each line is read once).  Measured
// (profiles/r02av): 4096^2 0.2066 -> 0.192 ms; with nt on both, or on the stores only, or on neither: 0.207 / 0.215 / 0.212 ms;
// 8192^2, whose planes are four times the cache, unchanged.
constexpr int PLANE_AUX_STORE = 16;     /* sc1 */
constexpr int PLANE_AUX_LOAD = 16 | 2;  /* sc1 | nt */
__device__ __forceinline__ rpsf_f4 plane_load16_wt(__amdgpu_buffer_rsrc_t r, size_t float_offset) {
  const rpsf_i4 q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(float_offset * sizeof(float)), 0, PLANE_AUX_LOAD);
  return rpsf_f4{__int_as_float(q.x), __int_as_float(q.y), __int_as_float(q.z), __int_as_float(q.w)};
}
template <int AUX>
__device__ __forceinline__ void plane_store16_aux(__amdgpu_buffer_rsrc_t r, size_t float_offset, rpsf_f4 v) {
  const rpsf_i4 q = {__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(q, r, (int)(float_offset * sizeof(float)), 0, AUX);
}
__device__ __forceinline__ void plane_store16_wt(__amdgpu_buffer_rsrc_t r, size_t float_offset, rpsf_f4 v) {
  const rpsf_i4 q = {__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(q, r, (int)(float_offset * sizeof(float)), 0, PLANE_AUX_STORE);
}

// UN: groups in flight per thread (4 x UN sixteen-byte loads);  KNOWN_FUSED: the caller is a fused launch (the persistent kernels), so the
// plane loads are the write-through-aware ones without a run-time choice in front of each
template <int UN = 8, bool KNOWN_FUSED = false>
__device__ __forceinline__ void sum_tile(const TileSum& p0, uint32_t entry_any_lane, int tid, int nthreads, bool known_complete = false) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  // The entry is the same in every lane (the whole workgroup sums one tile), but it comes out of LDS or a strided loop, so the compiler has to assume
  // otherwise - and then the tile's coverage mask is a vector value and every "is this plane present" choice of the 32 loads a v_cndmask on VCC, which
  // gfx950 issues at 23 cycles apiece (scripts/micro/valu_mix.hip, profiles/r04y).  Saying that it is uniform makes them scalar selects.
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t entry = __builtin_amdgcn_readfirstlane(entry_any_lane);
#else
  const uint32_t entry = entry_any_lane;
#endif
  const uint32_t tile = entry & 0xffffffu, frame = entry >> 24;
  TileSum p = p0;  // this frame's planes, output and counters
  p.planes += (size_t)frame * p0.planes_frame_floats;
  p.out += (size_t)frame * p0.out_frame_floats;
  if (p.done) p.done += (size_t)frame * p0.n_tiles;
  const int ti = tile / p.ntj, tj = tile % p.ntj;
  const int cov_all = p.cover[tile];
#if defined(RPSF2_SKEL_PRESUM)  // (timing skeleton, rpsf_kernels2.hpp: only the planes of the patches whose LEFT half lies over this tile are read)
  const int cov = cov_all & ((tj & 1) ? 10 : 5);
#else
  const int cov = cov_all;
#endif
  const bool fused = KNOWN_FUSED || p.done != nullptr;
  if (fused && !known_complete) {  // wait until every contributor of the tile has published its stores
    if (tid == 0) {
      const uint32_t want = p.epoch * (uint32_t)__builtin_popcount(cov_all & 15);
      while (__hip_atomic_load(p.done + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
  }
  const int y0 = max(p.lat_r0 + ti * p.half, p.row0), y1 = min(p.lat_r0 + (ti + 1) * p.half, p.row0 + p.rows);
  const int x0 = max(p.lat_c0 + tj * p.half, 0), x1 = min(p.lat_c0 + (tj + 1) * p.half, p.W);
  if (y0 >= y1 || x0 >= x1) return;
  const bool vec = ((x0 | x1 | p.ld_planes | p.ld_out) & 3) == 0 && (p.plane_stride & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(p.planes) | reinterpret_cast<uintptr_t>(p.out)) & 15) == 0;
  const int gw = (x1 - x0) >> 2;
  // Row and column group of element i: a tile is 2^k groups of four pixels wide except where the image clips its last column, and an integer
  // division by a run-time value is a dozen instructions, three of them v_cndmask on VCC (23 cycles apiece on gfx950): the power-of-two case
  // shifts and masks; the clipped tiles (e.g. 96 of 128 pixels at the right edge of a 4064-wide image) keep the 16-byte accesses in a compact loop with
  // the division (until round 5 they fell to the pixel-by-pixel path: four agent-scope loads per pixel).
  if (vec) {
    const __amdgpu_buffer_rsrc_t rsrc = plane_rsrc(p.planes);
    const int total = gw * (y1 - y0);
    // (UN = 8: 32 sixteen-byte loads per thread - a whole tile per pass for the patch kernels' workgroup sizes)
    if ((gw & (gw - 1)) == 0) {
      const int sh = 31 - __builtin_clz((unsigned)gw);
      auto row_of = [&](int i) RPSF_AI { return i >> sh; };
      auto col_of = [&](int i) RPSF_AI { return i & (gw - 1); };
      for (int i0 = tid; i0 < total; i0 += UN * nthreads) {
        f4 v[UN][4];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int i = min(i0 + u * nthreads, total - 1);
          const size_t off = (size_t)(y0 + row_of(i) - p.row0) * p.ld_planes + x0 + (col_of(i) << 2);
#pragma unroll
          for (int k = 0; k < 4; ++k) {  // unconditional loads (redirected to plane 0's line when unused) so that they overlap
            const size_t o = ((cov >> k) & 1) ? off + k * p.plane_stride : off;
            v[u][k] = fused ? plane_load16_wt(rsrc, o) : __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.planes + o));
          }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int i = i0 + u * nthreads;
          if (i >= total) break;
          f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if ((cov >> k) & 1) acc += v[u][k];
          __builtin_nontemporal_store(acc, reinterpret_cast<f4*>(p.out + (size_t)(y0 + row_of(i) - p.row0) * p.ld_out + x0 + (col_of(i) << 2)));
        }
      }
    } else {  // a clipped tile: the same 16-byte accesses, one group per thread and step (a compact loop: these are a few tiles per frame, and a second
              // unrolled instantiation would cost the persistent kernels 19 KB of code and 13 more spilled SGPRs)
      for (int i = tid; i < total; i += nthreads) {
        const int row = i / gw, col = i - row * gw;
        const size_t off = (size_t)(y0 + row - p.row0) * p.ld_planes + x0 + (col << 2);
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if ((cov >> k) & 1) {  // (workgroup-uniform)
            const size_t o = off + k * p.plane_stride;
            acc += fused ? plane_load16_wt(rsrc, o) : __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.planes + o));
          }
        __builtin_nontemporal_store(acc, reinterpret_cast<f4*>(p.out + (size_t)(y0 + row - p.row0) * p.ld_out + x0 + (col << 2)));
      }
    }
  } else {
    const int w = x1 - x0, total = w * (y1 - y0);
    for (int i = tid; i < total; i += nthreads) {
      const size_t yl = (size_t)(y0 + i / w - p.row0);
      const int x = x0 + i % w;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((cov >> k) & 1) {
          const float* src = p.planes + k * p.plane_stride + yl * p.ld_planes + x;
          acc += fused ? __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : *src;
        }
      p.out[yl * p.ld_out + x] = acc;
    }
  }
}
// workgroup `block` of `nblocks` co-operating ones works through the list: in fixed strides, or - fused mode, where the
// workgroups start at different times (most while the last patches still run, the rest after them) - from a queue
// BETWEEN: a side job of the calling workgroup (the image prefetch of the head summing workgroups, rpsf_kernels2.hpp), called by every thread
// at the top of each round of the queue loop and whenever no drawn tile is complete yet; returns whether it has steps left.  It must not
// contain a workgroup barrier (waves may disagree by a round about when a step is due).
struct NoSideJob {
  __device__ __forceinline__ bool operator()() const { return false; }
};
template <class BETWEEN = NoSideJob, int UN = 8, bool KNOWN_FUSED = false>
__device__ __forceinline__ void sum_tiles_worker(const TileSum& p, int block, int nblocks, BETWEEN&& between = BETWEEN()) {
  if (!p.queue) {
    for (int i = block; i < p.count; i += nblocks) sum_tile<UN, KNOWN_FUSED>(p, p.tiles[i], threadIdx.x, blockDim.x);
    return;
  }
  // A workgroup holds up to SUM_LOOKAHEAD drawn tiles and sums whichever of them is complete first, instead of waiting for the
  // head of the (predicted) order while later tiles are complete already - with the plane stores kept in the Infinity Cache,
  // the sooner a complete tile is summed the likelier its planes are still there.  Thread 0 keeps the list.
#define RPSF_SUM_LOOKAHEAD 4
  constexpr int SUM_LOOKAHEAD = RPSF_SUM_LOOKAHEAD;
  __shared__ uint32_t pend[SUM_LOOKAHEAD];
  __shared__ uint32_t pick;
  int npend = 0;
  bool exhausted = false;
  bool side_job = true;
  for (;;) {
    if (side_job) side_job = between();
    if (threadIdx.x == 0) {
      uint32_t chosen = 0xffffffffu;
      for (;;) {
        for (int j = 0; j < npend; ++j) {
          const uint32_t tile = pend[j];  // (frame << 24) | tile
          const uint32_t want = p.epoch * (uint32_t)__builtin_popcount(p.cover[tile & 0xffffffu] & 15);
          if (__hip_atomic_load(p.done + (size_t)(tile >> 24) * p.n_tiles + (tile & 0xffffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) {
            chosen = tile;
            for (int m = j + 1; m < npend; ++m) pend[m - 1] = pend[m];
            --npend;
            break;
          }
        }
        if (chosen != 0xffffffffu) break;
        if (!exhausted && npend < SUM_LOOKAHEAD) {
          const uint32_t i = __hip_atomic_fetch_add(p.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - p.queue_base;
          if (i >= (uint32_t)p.count) exhausted = true; else pend[npend++] = sum_entry(p, i);
          continue;
        }
        if (npend == 0) break;  // every position drawn, nothing left to wait for
        if (side_job) {         // nothing complete yet: a step of the side job instead of a nap
          chosen = 0xfffffffeu;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
      pick = chosen;
    }
    __syncthreads();
    const uint32_t tile = pick;
    __syncthreads();
    if (tile == 0xffffffffu) return;
    if (tile == 0xfffffffeu) continue;
    sum_tile<UN, KNOWN_FUSED>(p, tile, threadIdx.x, blockDim.x, true);
  }
}

// ------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------
struct PatchParams {
  ImageView im;
  OutView ov;
  int origin_row, origin_col;
  const int4* desc;         // per processing-order slot: {corner row, corner col, patch index, colour plane}
                            // (one 16-byte load instead of the order -> coords -> plane pointer chase)
  int chunk;                // patches per XCD chunk
  int seq_base;             // first processing-order slot of this launch
  int slot0;                // first slot of every chunk this launch covers (the apply may be cut into a main and a tail launch)
  int sum_first;            // fused plane sum: workgroups [0, sum_first) sum tiles beside the patches for the whole launch,
  int patch_blocks;         // [sum_first, sum_first + patch_blocks) process patches, the ones beyond sum tiles again
  TileSum ts;               // ... of this list
  uint32_t* tile_done;      // fused plane sum: per (frame, tile) count of contributors whose plane stores are complete (nullptr: off)
  unsigned long long* stamps;  // diagnostic builds (RPSF_STAMPS): 16 phase timestamps per patch
  int stagger_ticks;        // start-up stagger of the first resident workgroups, in 10 ns ticks (0 = off)
  int stagger_blocks;       // how many leading blocks are staggered (= resident workgroup capacity)
  int n_patches;
  int n_frames;             // batch: every patch slot is processed for n_frames frames that share the transfer kernel
  size_t im_frame_floats;   // frame f reads im.img + f * im_frame_floats ...
  size_t ov_frame_floats;   // ... and writes ov.out + f * ov_frame_floats
  const uint16_t* tab;
  const uint32_t* pairtab;  // two-stage plans: bin pairs of the special slots (build_pair_table)
  const cf* tw;
  const float* win;
  const cf* g;
  const cf* gs;
  // direct overlap-add (rpsf_core.hpp, store_patch_direct); dv.out == nullptr: off
  OutView dv;                 // the output image itself
  size_t dv_frame_floats;     // batch: frame f accumulates into dv.out + f * dv_frame_floats
  const uint4* quads;         // per processing-order slot: the four quadrant words
  uint32_t* flags;            // per (frame, tile): (epoch << 8) | (tile initialised << 3) | direct contributors done
  uint32_t* dyn_side;         // per (frame, tile): (epoch << 8) | colour bits of contributors demoted to their plane at run time
  uint32_t* chunk_xcc;        // per chunk: (epoch << 8) | (XCC id + 1) of the first workgroup that registered
  uint32_t flag_epoch;        // 1 .. 2^24-1, new for every apply
  uint32_t n_tiles;
  int orphan_mod;             // testing aid: workgroups with seq % orphan_mod == 1 behave as if they ran on a foreign XCD
  // persistent launch (patch_kernel2_256p): the grid is sum_first + 8 * persist workgroups; the resident ones start on
  // slots [0, persist) of their XCD's chunk and draw the later ones from xq[xcd * 32] (one counter per XCD on a line of its own)
  int frame_major;  // batches: queue position = frame * (slots of the XCD) + slot instead of slot * frames + frame
  int plane_nt;     // (host side only: the launch takes patch_kernel2_128pcs - streaming plane stores - see rpsf.hip)
  int persist;
  uint32_t* xq;
  uint32_t xq_base[8];
  int prefetch;      // persistent launches: the head summing workgroups first touch the image in processing order (rpsf_kernels2.hpp, prefetch_chunk)
  const uint32_t* prefetch_tiles;  // per chunk: the lattice tiles of the image in the order the chunk's patches first need them ...
  uint32_t prefetch_first[9];      // ... chunk x owns entries [prefetch_first[x], prefetch_first[x + 1])
  int head_patches;  // persistent launches: the summing workgroups at the head of the grid compute this many (0 or 1) patches before they turn to summing
};

template <class C>
struct Launch {
  static constexpr int WG = C::T < 64 ? 64 : C::T;
  static constexpr int TEAMS = WG / C::T;
  static constexpr int PT_FLOATS = (C::PT_WORDS + 3) / 4 * 4;  // pair table of the special slots (two-stage plans), one copy per workgroup
  static constexpr int TABLE_FLOATS = 3 * C::N + PT_FLOATS;    // twiddles (N complex) + window (N) + pair table, shared by the workgroup
  static constexpr size_t LDS_BYTES = (size_t)(TABLE_FLOATS + TEAMS * C::LDS_FLOATS) * sizeof(float);
  // Waves per SIMD the kernel is compiled for.  Where LDS already limits a CU to four single-wave workgroups
  // (N = 64: each team parks its whole patch), one wave per SIMD may as well use the other half of the
  // register file: spills then go to AGPRs instead of scratch memory.
  static constexpr int WAVES_PER_SIMD = (WG == 64 && 4 * LDS_BYTES <= 160 * 1024 && 5 * LDS_BYTES > 160 * 1024) ? 1 : 2;
  static_assert(C::KDEPTH == 1 || WAVES_PER_SIMD == 1, "two K chunks in flight need the 512-register budget");
};


// Workgroup barrier that orders LDS traffic only.  __syncthreads() also makes hipcc drain every
// outstanding global load/store (s_waitcnt vmcnt(0)), which would expose the HBM latency of the K
// prefetch and of the plane stores at each of the ~20 exchange barriers of a patch.
// Orders one wave's own LDS traffic (a wave's DS operations complete in order; no other wave is involved).
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#if defined(RPSF_WAVE_STAMPS)  // diagnostic: lane 0 of EVERY wave (up to 8 per patch) stamps - scripts/dev_wave_stamps.py
#define STAMP(i)                                                                                                                              \
  do {                                                                                                                                        \
    if ((threadIdx.x & 63) == 0) p.stamps[((size_t)patch * 8 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();             \
  } while (0)
#elif defined(RPSF_STAMPS)
#define STAMP(i)                                                                                         \
  do {                                                                                                   \
    if (threadIdx.x == 0) p.stamps[(size_t)patch * 16 + (i)] = __builtin_amdgcn_s_memrealtime();          \
  } while (0)
#else
#define STAMP(i) ((void)0)
#endif

// Direct overlap-add protocol of one workgroup (= one patch; three-stage plans only).  See rpsf_core.hpp,
// store_patch_direct, for the scheme.  Same-XCD visibility needs no cache maintenance: stores go through to the
// XCD's L2 (complete once vmcnt has drained) and the successor reads them with L1-bypassing loads.  That the
// workgroups of one chunk really share an XCD is an observation about the dispatcher, not a guarantee, so it is
// checked: the first workgroup of a chunk registers its XCC id, and one that finds itself elsewhere ("orphan")
// sends all four quadrants to its colour plane, marks them in dyn_side for the fix-up kernel and only keeps
// the flags moving.
// direct_begin: registers / checks the XCD, waits for the predecessors of the quadrants that accumulate into the
// output and returns the run-time quadrant words.  scratch: 8 words of LDS no other wave is using.
__device__ __forceinline__ void direct_begin(const PatchParams& p, int frame, int seq, uint32_t (&qw)[4], uint32_t* scratch) {
  const uint4 q4 = p.quads[p.seq_base + seq];
  qw[0] = q4.x, qw[1] = q4.y, qw[2] = q4.z, qw[3] = q4.w;
  uint32_t* flags = p.flags + (size_t)frame * p.n_tiles;
  const uint32_t epoch = p.flag_epoch;
  if (threadIdx.x == 0) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    const uint32_t mine = (epoch << 8) | (xcc + 1);
    uint32_t* reg = p.chunk_xcc + (blockIdx.x & 7);
    uint32_t seen = __hip_atomic_load(reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while ((seen >> 8) != epoch) {
      if (__hip_atomic_compare_exchange_strong(reg, &seen, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        seen = mine;
    }
    bool orphan = seen != mine;
    if (p.orphan_mod > 0 && seq % p.orphan_mod == 1) orphan = true;
    scratch[0] = orphan ? 1u : 0u;
  }
  if (threadIdx.x < 4) {
    const uint32_t w = qw[threadIdx.x];
    uint32_t init = 0;
    if (quad_mode(w) == QUAD_DIRECT && quad_rank(w) > 0) {
      const uint32_t want = (epoch << 8) | quad_rank(w);
      uint32_t f;
      while ((((f = __hip_atomic_load(flags + quad_tile(w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & ~8u)) != want)
        __builtin_amdgcn_s_sleep(2);
      init = (f >> 3) & 1u;
    }
    scratch[1 + threadIdx.x] = init;
  }
  lds_barrier();
  const bool orphan = scratch[0] != 0;  // workgroup-uniform
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (quad_mode(qw[q]) != QUAD_DIRECT) continue;
    if (orphan) qw[q] = quad_word(QUAD_SIDE, quad_rank(qw[q]), quad_tile(qw[q])) | QUAD_DEMOTED | (scratch[1 + q] ? QUAD_ACC : 0u);
    else if (scratch[1 + q]) qw[q] |= QUAD_ACC;
  }
}
// direct_end: every wave drains its stores, then one lane per quadrant moves the tile's flag on
__device__ __forceinline__ void direct_end(const PatchParams& p, int frame, int plane, const uint32_t (&qw)[4]) {
  const uint32_t epoch = p.flag_epoch;
  uint32_t* flags = p.flags + (size_t)frame * p.n_tiles;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  if (threadIdx.x < 4) {
    const uint32_t w = qw[threadIdx.x];
    const bool demoted = (w & QUAD_DEMOTED) != 0;
    if (demoted) {  // tell the fix-up kernel that this colour's plane holds a contribution to the tile
      uint32_t* dyn = p.dyn_side + (size_t)frame * p.n_tiles + quad_tile(w);
      uint32_t seen = __hip_atomic_load(dyn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), next;
      do next = ((seen >> 8) == epoch ? seen : (epoch << 8)) | (1u << plane);
      while (!__hip_atomic_compare_exchange_strong(dyn, &seen, next, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    if (demoted || quad_mode(w) == QUAD_DIRECT) {
      const uint32_t init = demoted ? ((w & QUAD_ACC) ? 1u : 0u) : 1u;
      __hip_atomic_store(flags + quad_tile(w), (epoch << 8) | (init << 3) | (quad_rank(w) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <class C>
__device__ __forceinline__ void direct_store(const PatchParams& p, int t, const cf* v, const OutView& ov, int frame, int seq,
                                             int plane, int pr, int pc, const float* win, uint32_t* scratch) {
  OutView dv = p.dv;
  dv.out += (size_t)frame * p.dv_frame_floats;
  uint32_t qw[4];
  direct_begin(p, frame, seq, qw, scratch);
  store_patch_direct<C>(
      t, v, ov, dv, plane, pr, pc, win, qw,
      [](const float* a) {
        const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return cf{__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32))};
      },
      [](const float* a) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); });
  direct_end(p, frame, plane, qw);
}

template <class C>
__global__ __launch_bounds__(Launch<C>::WG, Launch<C>::WAVES_PER_SIMD) void patch_kernel(PatchParams p) {  // 2 waves per SIMD: 256 registers
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int T = C::T;
  const int team = threadIdx.x / T, t = threadIdx.x % T;
  // Blocks are dealt round-robin over the 8 XCDs (b and b+8 share one).  The plan's processing order is cut
  // into 8 contiguous chunks of compact lattice regions: XCD x works through chunk x, so the four patches that
  // overlap a pixel usually read it through the same L2 (and, with the direct overlap-add, accumulate it there;
  // that placement is checked at run time, see direct_store).
  // Batch: the n_frames workgroups of one patch slot are consecutive on their XCD, so the slot's packed
  // K comes from HBM once and from that XCD's L2 for the other frames.
  int frame = 0, xrow = blockIdx.x >> 3;
  if (p.n_frames > 1) {
    frame = xrow % p.n_frames;
    xrow /= p.n_frames;
  }
  ImageView im = p.im;
  OutView ov = p.ov;
  im.img += (size_t)frame * p.im_frame_floats;
  ov.out += (size_t)frame * p.ov_frame_floats;
  const int slot = xrow * Launch<C>::TEAMS + team;
  const int seq = (blockIdx.x & 7) * p.chunk + slot;
  const bool active = slot < p.chunk && seq < p.n_patches;
  const int4 dsc = p.desc[p.seq_base + (active ? seq : p.n_patches - 1)];  // inactive teams stay in step with the barriers
  const int patch = dsc.z;
  // Optional start-up stagger (rpsf_plan_set_stagger, off by default): every CU gathers, streams K and stores
  // at the same instants; delaying the first resident workgroup of each CU by a different amount spreads the
  // phases for the whole launch (later workgroups inherit the offset of the one they replace).  Measured: a patch
  // gets 9 % shorter with a 50 us spread, and the delay itself costs as much - kept as a diagnostic knob.
  if (p.stagger_ticks > 0 && (int)blockIdx.x < p.stagger_blocks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long wait = (unsigned long long)(((blockIdx.x >> 3) * 0x9E3779B1u >> 22) & 1023) * p.stagger_ticks >> 10;
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
  float* lds = smem + Launch<C>::TABLE_FLOATS + team * C::LDS_FLOATS;
  STAMP(0);
  const int pr = dsc.x + p.origin_row, pc = dsc.y + p.origin_col;
  const cf* g = p.g + (size_t)patch * C::G_PER_PATCH;
  cf v[64];
  const bool fast = patch_inside<C>(pr, pc, im.H, im.W, im.row0, im.rows) && pairs_aligned(im.img, im.ld, pc);
  // twiddle and window tables live in LDS: their reads must not queue behind the patch's global loads
  cf* tw = reinterpret_cast<cf*>(smem);
  float* win = smem + 2 * C::N;
  for (int i = threadIdx.x; i < C::N; i += Launch<C>::WG) {
    tw[i] = p.tw[i];
    win[i] = p.win[i];
  }
  uint32_t* pt = reinterpret_cast<uint32_t*>(smem + 3 * C::N);
  for (int i = threadIdx.x; i < C::PT_WORDS; i += Launch<C>::WG) pt[i] = p.pairtab[i];
  GroupIds<C> gids;
  gids.load(p.tab, t);
  {
    int* maps = reinterpret_cast<int*>(lds);
    if (!fast) build_pad_maps<C>(t, maps, im, pr, pc);
    lds_barrier();
    // (VMEM returns in order: issuing these loads before the table staging above would make the tables
    //  wait for all 64 of them and lose the load -> stage-1 overlap; measured 9.0 -> 12.8 us)
    load_patch<C>(t, v, im, pr, pc, win, fast, maps);
    lds_barrier();  // the maps share LDS with the exchange buffer
  }
  STAMP(1);
  stage1<C, false>(t, v, tw);
  STAMP(2);
  if constexpr (C::S3) {
    // X1 is a register<->lane transpose inside each wave (private LDS region): wave-level ordering is enough
    x1_write<C, 0>(t, v, lds);
    wave_lds_sync();
    x1_read<C, 0>(t, v, lds);
    wave_lds_sync();
    x1_write<C, 1>(t, v, lds);
    wave_lds_sync();
    x1_read<C, 1>(t, v, lds);
    STAMP(3);
    stage2<C, false>(t, v, tw);
    STAMP(4);
  }
  KRing<C> kring;
  kring_fill<C>(t, kring, g);  // first K slots: in flight across the exchange below (raw barriers do not drain VMEM)
  lds_barrier();  // every wave has left its X1 region (X2 uses the whole buffer)
  x2_mid_write<C, 0>(t, v, lds);
  lds_barrier();
  x2_last_read<C, 0>(gids, v, lds);
  lds_barrier();
  x2_mid_write<C, 1>(t, v, lds);
  lds_barrier();
  x2_last_read<C, 1>(gids, v, lds);
  // no barrier: every X2 word is read by exactly one thread, the same one that rewrites it below

  STAMP(5);
  {
    STAMP(6);
    STAMP(7);
    freq_step<C>(t, gids, v, kring, g, p.gs + (size_t)patch * C::GS_PER_PATCH, tw, reinterpret_cast<cf*>(lds + C::PARK_OFFSET), pt);
    STAMP(8);
  }

  x2_last_write<C, 0>(gids, v, lds);
  lds_barrier();
  x2_mid_read<C, 0>(t, v, lds);
  lds_barrier();
  x2_last_write<C, 1>(gids, v, lds);
  lds_barrier();
  x2_mid_read<C, 1>(t, v, lds);
  lds_barrier();
  STAMP(9);
  if constexpr (C::S3) {
    stage2<C, true>(t, v, tw);
    STAMP(10);
    x1_write<C, 0>(t, v, lds);
    wave_lds_sync();
    x1_read<C, 0>(t, v, lds);
    wave_lds_sync();
    x1_write<C, 1>(t, v, lds);
    wave_lds_sync();
    x1_read<C, 1>(t, v, lds);
  }
  STAMP(11);
  stage1<C, true>(t, v, tw);
  STAMP(12);
  if (active) {
    const int plane = ov.plane_stride ? dsc.w : 0;
    if constexpr (C::S3) {
      if (p.dv.out) {
        direct_store<C>(p, t, v, ov, frame, seq, plane, pr, pc, win, reinterpret_cast<uint32_t*>(lds));
        STAMP(13);
        return;
      }
    }
    store_patch<C>(t, v, ov, plane, pr, pc, win, [](float* a, float val) { unsafeAtomicAdd(a, val); });
  }
  STAMP(13);
}

// ------------------------------------------------------------------------------------------------
// K2: transform.py:78-82.  Mirrors NumPy's evaluation order: |.| by hypot, scalar powers with
// NumPy's fast paths (0, 1, 2, 0.5, -1), complex * real as a full complex product with (r, 0),
// complex / real as NumPy's scaled division, then a plain complex product with T.
// ------------------------------------------------------------------------------------------------
template <class R>
__device__ __forceinline__ R np_pow(R x, R e) {
  if (e == R(0)) return R(1);
  if (e == R(1)) return x;
  if (e == R(2)) return x * x;
  if (e == R(0.5)) return sqrt(x);
  if (e == R(-1)) return R(1) / x;
  // Small integer exponents (the reference's alpha = 3 asks for x^4: transform.py:78-82 with NumPy's float32 power loop, i.e. glibc's powf, which is
  // correctly rounded in nearly all cases).  In float32 the product is formed in float64 - x^2 is exact there, every further factor rounds at 53 bits - and
  // rounded once to float32: the correctly rounded power up to double rounding, for two or three multiplications instead of a pow() of 100+ instructions.
  // Measured against the reference's own K (tests/golden/construct.npz, scripts/construct_bits.py): see profiles/r06zr_construct_pow.log.
  if constexpr (sizeof(R) == 4) {
    if (e == R(3) || e == R(4) || e == R(5) || e == R(6) || e == R(8)) {
      const double d = (double)x, d2 = d * d;
      const double r = e == R(3) ? d2 * d : e == R(4) ? d2 * d2 : e == R(5) ? d2 * d2 * d : e == R(6) ? d2 * d2 * d2 : (d2 * d2) * (d2 * d2);
      return (R)r;
    }
  }
  return pow(x, e);
}

template <class R>
__device__ __forceinline__ void transfer_value(R sr, R si, R tr, R ti, R alpha, R eps, R& kr, R& ki) {
#pragma clang fp contract(off)  // (NumPy rounds every product; and the value must not depend on which kernel this is inlined into: K2 and the one-pass packers)
  R sabs = hypot(sr, si), tabs = hypot(tr, ti);
  R pw = np_pow(sabs, alpha - R(1));
  // conj(S) * (pw + 0i)
  R cr = sr, ci = -si;
  R nr = cr * pw - ci * R(0), ni = cr * R(0) + ci * pw;
  R den = np_pow(sabs, alpha + R(1)) + np_pow(eps * tabs, alpha + R(1));
  // (nr + i ni) / (den + 0i), NumPy's algorithm for |re| >= |im|
  R qr, qi;
  if (fabs(den) == R(0)) {
    qr = nr / fabs(den);
    qi = ni / fabs(den);
  } else {
    R rat = R(0) / den;
    R scl = R(1) / (den + R(0) * rat);
    qr = (nr + ni * rat) * scl;
    qi = (ni - nr * rat) * scl;
  }
  kr = qr * tr - qi * ti;
  ki = qr * ti + qi * tr;
}

template <class R>
__global__ void build_transfer_kernel(const R* __restrict__ s, const R* __restrict__ t, R* __restrict__ k,
                                      size_t count, R alpha, R eps) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  transfer_value<R>(s[2 * i], s[2 * i + 1], t[2 * i], t[2 * i + 1], alpha, eps, k[2 * i], k[2 * i + 1]);
}

// One patch of K evaluated from the two PSF spectra where the packers ask for it (complex64 spectra, float32 arithmetic: the very operations of
// build_transfer_kernel<float>, so the packed K is bit-identical to K2 followed by the pack kernel): `construct` on device-resident spectra goes
// spectra -> packed K in ONE pass - 2 x n N^2 x 8 B read, n N (N/2 + 1) x 8 B written - instead of writing the full K (n N^2 x 8 B) and reading it back.
struct KFromSpectra {
  const cf* s;
  const cf* t;
  float alpha, eps;
  __device__ __forceinline__ cf operator()(int i) const {
    const cf a = s[i], b = t[i];
    cf k;
    transfer_value<float>(a.x, a.y, b.x, b.y, alpha, eps, k.x, k.y);
    return k;
  }
};

// ------------------------------------------------------------------------------------------------
// K-pack
// ------------------------------------------------------------------------------------------------
template <class C>
__global__ void pack_kernel(const cf* __restrict__ kfull, int n_patches, const uint16_t* __restrict__ tab,
                            const uint32_t* __restrict__ pt, cf* __restrict__ g, cf* __restrict__ gs) {
  const size_t per = (size_t)C::G_PER_PATCH;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per * n_patches) return;
  int patch = (int)(idx / per);
  int rem = (int)(idx % per);
  int b = rem & 1, t = (rem >> 1) % C::T, i = (rem >> 1) / C::T;
  int rho = 2 * i + b;
  const cf* kf = kfull + (size_t)patch * C::N * C::N;
  g[idx] = pack_value<C>(kf, tab, pt, t, rho, 0);
  if constexpr (!C::INLINE_GS) {
    const int w = rho >> 1, s = w / C::E, e = w % C::E;
    if (slot_is_special<C>(s, t))
      gs[(size_t)patch * C::GS_PER_PATCH + (size_t)C::spec_prefix(s) * 2 * C::E + ((size_t)e * C::spec_t(s) + t) * 2 + b] =
          pack_value<C>(kf, tab, pt, t, rho, 1);
  }
}

// The same packer fed from the two PSF spectra (see KFromSpectra)
template <class C>
__global__ void pack_spectra_kernel(const cf* __restrict__ s_fft, const cf* __restrict__ t_fft, float alpha, float eps, int n_patches,
                                    const uint16_t* __restrict__ tab, const uint32_t* __restrict__ pt, cf* __restrict__ g, cf* __restrict__ gs) {
  const size_t per = (size_t)C::G_PER_PATCH;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per * n_patches) return;
  int patch = (int)(idx / per);
  int rem = (int)(idx % per);
  int b = rem & 1, t = (rem >> 1) % C::T, i = (rem >> 1) / C::T;
  int rho = 2 * i + b;
  const KFromSpectra kf{s_fft + (size_t)patch * C::N * C::N, t_fft + (size_t)patch * C::N * C::N, alpha, eps};
  g[idx] = pack_value<C>(kf, tab, pt, t, rho, 0);
  if constexpr (!C::INLINE_GS) {
    const int w = rho >> 1, s = w / C::E, e = w % C::E;
    if (slot_is_special<C>(s, t))
      gs[(size_t)patch * C::GS_PER_PATCH + (size_t)C::spec_prefix(s) * 2 * C::E + ((size_t)e * C::spec_t(s) + t) * 2 + b] =
          pack_value<C>(kf, tab, pt, t, rho, 1);
  }
}

// ------------------------------------------------------------------------------------------------
// K3: batched 2-D FFT of real N x N arrays -> full N x N complex spectrum (psf.py:216-219)
// ------------------------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(Launch<C>::WG) void psf_fft_kernel(const float* __restrict__ values, int count,
                                                                 const uint16_t* __restrict__ tab,
                                                                 const cf* __restrict__ tw, cf* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int T = C::T, N = C::N, NC = C::NC, E = C::E;
  const int team = threadIdx.x / T, t = threadIdx.x % T;
  int item = blockIdx.x * Launch<C>::TEAMS + team;
  const bool active = item < count;
  if (!active) item = count - 1;
  float* lds = smem + team * C::LDS_FLOATS;
  GroupIds<C> gids;
  gids.load(tab, t);
  const float* src = values + (size_t)item * N * N;
  cf v[64];
  {
    ThreadPos<C> tp(t);
    constexpr int NR = 1 << C::A1, NCOL = 1 << C::B1;
    StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
      int r = (R1 << (C::A2 + C::AL)) + tp.r_rest;
      StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
        int c = (C1 << (C::B2 + C::BL)) + tp.c_rest;
        v[R1 * NCOL + C1] = cf{src[(size_t)r * N + 2 * c], src[(size_t)r * N + 2 * c + 1]};
      });
    });
  }
  stage1<C, false>(t, v, tw);
  if constexpr (C::S3) {
    x1_write<C, 0>(t, v, lds);
    __syncthreads();
    x1_read<C, 0>(t, v, lds);
    __syncthreads();
    x1_write<C, 1>(t, v, lds);
    __syncthreads();
    x1_read<C, 1>(t, v, lds);
    __syncthreads();
    stage2<C, false>(t, v, tw);
  }
  x2_mid_write<C, 0>(t, v, lds);
  __syncthreads();
  x2_last_read<C, 0>(gids, v, lds);
  __syncthreads();
  x2_mid_write<C, 1>(t, v, lds);
  __syncthreads();
  x2_last_read<C, 1>(gids, v, lds);
  stage_last<C, false>(v);
  if (!active) return;
  cf* dst = out + (size_t)item * N * N;
  // unpack X[kr][kc] = E + W^kc O and X[kr][kc + N/2] = E - W^kc O for every bin this thread holds
  StaticFor<0, C::NSLOT>::run([&]<int S>() RPSF_AI {
    cf* za = v + (2 * S) * E;
    cf* zb = za + E;
    int ga = gids[2 * S], gb = gids[2 * S + 1];
    bool self = partner_gid<C>(ga) == ga;
    int qa, ma, qb, mb;
    gid_to_qm<C>(ga, qa, ma);
    gid_to_qm<C>(gb, qb, mb);
    const bool qza = qa == 0, qzb = qb == 0, mza = ma == 0, mzb = mb == 0;
    StaticFor<0, E>::run([&]<int EE>() RPSF_AI {
      constexpr int EA = C::EA, EB = C::EB, K3 = EE / EB, L3 = EE % EB;
      constexpr int RR = (EA - 1 - K3) * EB + (EB - 1 - L3), ZR = ((EA - K3) % EA) * EB + (EB - 1 - L3);
      constexpr int RZ = (EA - 1 - K3) * EB + (EB - L3) % EB, ZZ = ((EA - K3) % EA) * EB + (EB - L3) % EB;
      auto pick = [&](const cf* src, bool qz, bool mz) RPSF_AI {
        cf rr = src[RR], zr = src[ZR], rz = src[RZ], zz = src[ZZ];
        return sel(mz, sel(qz, zz, rz), sel(qz, zr, rr));
      };
      cf pa = sel(self, pick(za, qza, mza), pick(zb, qza, mza));
      cf pb = sel(self, pick(zb, qzb, mzb), pick(za, qzb, mzb));
      {
        const int kr = qa + C::Q * K3, kc = ma + C::M * L3;
        cf zc = cconj(pa);
        cf e2 = (za[EE] + zc) * 0.5f, wo = cmul(tw[kc], mul_mi(za[EE] - zc)) * 0.5f;
        dst[(size_t)kr * N + kc] = e2 + wo;
        dst[(size_t)kr * N + kc + NC] = e2 - wo;
      }
      {
        const int kr = qb + C::Q * K3, kc = mb + C::M * L3;
        cf zc = cconj(pb);
        cf e2 = (zb[EE] + zc) * 0.5f, wo = cmul(tw[kc], mul_mi(zb[EE] - zc)) * 0.5f;
        dst[(size_t)kr * N + kc] = e2 + wo;
        dst[(size_t)kr * N + kc + NC] = e2 - wo;
      }
    });
  });
}


#if defined(RPSF_HOST_TU)
__global__ __launch_bounds__(256) void sum_tiles_kernel(TileSum p) { sum_tiles_worker(p, blockIdx.x, gridDim.x); }

// ------------------------------------------------------------------------------------------------
// K5: out = sum of the colour planes that have a patch over the pixel (fixed order: deterministic)
// ------------------------------------------------------------------------------------------------
struct SumParams {
  const float* planes;
  size_t plane_stride;
  float* out;
  int rows, W, ld_planes, ld_out;
  int row_begin;       // first window row this launch sums
  int row0;            // full-image row of window row 0
  int lat_r0, lat_c0;  // full-image coordinates of lattice tile (0, 0)
  int half_shift;      // log2(N/2)
  int nti, ntj;
  const uint8_t* cover;  // nti x ntj, 4-bit class masks
  size_t planes_frame_floats, out_frame_floats;  // batch: frame f (= blockIdx.y) at planes + f*..., out + f*...
};

__device__ __forceinline__ int cover_at(const SumParams& p, int y, int x) {
  int ty = (y - p.lat_r0) >> p.half_shift, tx = (x - p.lat_c0) >> p.half_shift;
  if (y < p.lat_r0 || x < p.lat_c0 || ty >= p.nti || tx >= p.ntj) return 0;
  return p.cover[ty * p.ntj + tx];
}

__device__ __forceinline__ bool sum_vector_ok(const SumParams& p) {
  return ((p.ld_planes | p.ld_out) & 3) == 0 &&
         ((reinterpret_cast<uintptr_t>(p.planes) | reinterpret_cast<uintptr_t>(p.out) | (p.plane_stride * 4)) & 15) == 0;
}
// Branch-free 16-byte read of plane k: a plane without a patch over the tile is never read (its content is
// stale) - the load is redirected to one always-valid line and its result discarded.  Keeping the loads
// unconditional matters: a load under a branch is waited for at the join, which serialises the four planes.
__device__ __forceinline__ float4 load_plane4(const SumParams& p, int k, size_t off, int cov) {
  const bool on = (cov >> k) & 1;
  const float* src = on ? p.planes + k * p.plane_stride + off : p.planes;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 a = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src));
  return make_float4(on ? a.x : 0.f, on ? a.y : 0.f, on ? a.z : 0.f, on ? a.w : 0.f);
}
__device__ __forceinline__ void store_out4(float* o, float4 v) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<f4*>(o));
}

// four consecutive pixels (x .. x+3) of window row yl
__device__ __forceinline__ void sum_planes_group(const SumParams& p, int yl, int x, bool vector_ok) {
  int y = yl + p.row0;
  size_t off = (size_t)yl * p.ld_planes + x;
  float* o = p.out + (size_t)yl * p.ld_out + x;
  int c0 = cover_at(p, y, x), c3 = cover_at(p, y, x + 3);
  if (vector_ok && x + 3 < p.W && c0 == c3) {  // one tile, aligned: four 16-byte loads, one 16-byte store
    float4 a0 = load_plane4(p, 0, off, c0), a1 = load_plane4(p, 1, off, c0), a2 = load_plane4(p, 2, off, c0),
           a3 = load_plane4(p, 3, off, c0);
    store_out4(o, make_float4(((a0.x + a1.x) + a2.x) + a3.x, ((a0.y + a1.y) + a2.y) + a3.y,
                              ((a0.z + a1.z) + a2.z) + a3.z, ((a0.w + a1.w) + a2.w) + a3.w));
  } else {
    for (int i = 0; i < 4 && x + i < p.W; ++i) o[i] = sum_planes_at(p.planes, p.plane_stride, off + i, cover_at(p, y, x + i));
  }
}

__global__ void sum_planes_kernel(SumParams p) {
  p.planes += (size_t)blockIdx.y * p.planes_frame_floats;
  p.out += (size_t)blockIdx.y * p.out_frame_floats;
  const unsigned groups = (unsigned)(p.W + 3) >> 2;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)groups * p.rows) return;
  const bool small = (size_t)groups * p.rows < ((size_t)1 << 32);
  const int yl = small ? (int)((unsigned)idx / groups) : (int)(idx / groups);
  const int xg = small ? (int)((unsigned)idx % groups) : (int)(idx % groups);
  sum_planes_group(p, yl + p.row_begin, xg * 4, sum_vector_ok(p));
}

// ------------------------------------------------------------------------------------------------
// K5': fix-up after a direct overlap-add launch.  FIX_SUB workgroups per lattice tile; most exit at once.
//   out_tile = (tile initialised by its direct contributors ? out_tile : 0) + sum of the colour planes that hold a
//   side contribution (static: contributors of another chunk; dynamic: demoted at run time), in colour order.
// Tiles without any contributor are zeroed (the reference leaves uncovered output at 0, transform.py:167-169).
// ------------------------------------------------------------------------------------------------
struct FixParams {
  const float* planes;
  size_t plane_stride, planes_frame_floats;
  int ld_planes;
  float* out;
  int ld_out;
  size_t out_frame_floats;
  int rows, W, row0;           // resident output window: rows [row0, row0 + rows) of the image, W columns
  int lat_r0, lat_c0, half;    // full-image coordinates of lattice tile (0, 0); tile edge
  int ntj;
  const uint8_t* tile_info;    // bits 0-3 static side mask, bit 4 tile has contributors
  const uint32_t* flags;
  const uint32_t* dyn_side;
  uint32_t epoch, n_tiles;
};
constexpr int FIX_SUB = 8;

__global__ __launch_bounds__(256) void fixup_kernel(FixParams p) {
  const uint32_t tile = blockIdx.x / FIX_SUB, sub = blockIdx.x % FIX_SUB, frame = blockIdx.y;
  const uint32_t d = p.dyn_side[(size_t)frame * p.n_tiles + tile], f = p.flags[(size_t)frame * p.n_tiles + tile];
  const uint32_t side = (p.tile_info[tile] & 15u) | ((d >> 8) == p.epoch ? (d & 15u) : 0u);
  const bool init = (f >> 8) == p.epoch && (f & 8u);
  if (side == 0 && init) return;
  const int ti = tile / p.ntj, tj = tile % p.ntj;
  const int span = (p.half + FIX_SUB - 1) / FIX_SUB;
  const int ty0 = p.lat_r0 + ti * p.half + (int)sub * span;
  const int y0 = max(ty0, p.row0), y1 = min(min(ty0 + span, p.lat_r0 + (ti + 1) * p.half), p.row0 + p.rows);
  const int x0 = max(p.lat_c0 + tj * p.half, 0), x1 = min(p.lat_c0 + (tj + 1) * p.half, p.W);
  if (y0 >= y1 || x0 >= x1) return;
  const float* planes = p.planes + (size_t)frame * p.planes_frame_floats;
  float* out = p.out + (size_t)frame * p.out_frame_floats;
  const bool vec = ((x0 | x1 | p.ld_planes | p.ld_out) & 3) == 0 && (p.plane_stride & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  typedef float f4 __attribute__((ext_vector_type(4)));
  if (vec) {
    const int gw = (x1 - x0) >> 2, total = gw * (y1 - y0);
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      const int yl = y0 + i / gw - p.row0, x = x0 + ((i % gw) << 2);
      f4* o = reinterpret_cast<f4*>(out + (size_t)yl * p.ld_out + x);
      const f4* pl = reinterpret_cast<const f4*>(planes + (size_t)yl * p.ld_planes + x);
      f4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)  // unconditional loads (redirected to plane 0's line when unused) so they overlap
        v[k] = __builtin_nontemporal_load(((side >> k) & 1) ? pl + k * (p.plane_stride >> 2) : pl);
      f4 acc = init ? *o : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((side >> k) & 1) acc += v[k];
      *o = acc;
    }
  } else {
    const int w = x1 - x0, total = w * (y1 - y0);
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      const int yl = y0 + i / w - p.row0, x = x0 + i % w;
      float* o = out + (size_t)yl * p.ld_out + x;
      const float* pl = planes + (size_t)yl * p.ld_planes + x;
      float acc = init ? *o : 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((side >> k) & 1) acc += pl[k * p.plane_stride];
      *o = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K6: parametric PSF models rasterised on the sample grid of every patch (the built-in device models of
// regularizepsf_amd/functional.py; the reference evaluates a Python callable per patch on the host,
// regularizepsf/psf.py:65-70,159-165).  One workgroup per patch; element [i][j] of a patch is the model at row = j, col = i:
// the reference hands np.meshgrid(arange, arange) - 'xy' indexing - to the model as (row, col).  Evaluated in float64 from
// float64 parameters, stored as float32 (what the spectrum kernel K3 takes).  normalize: every patch is scaled to unit sum.
enum PsfModel : int { MODEL_ELLIPTICAL_GAUSSIAN = 0, MODEL_MOFFAT = 1 };
constexpr int RPSF_MODEL_PARAMS_DEV = 8;
__device__ __forceinline__ double psf_model_value(int model, const double* __restrict__ q, double row, double col) {
  const double dr = row - q[1], dc = col - q[2];
  if (model == MODEL_ELLIPTICAL_GAUSSIAN) {  // q: amplitude, row0, col0, sigma_row, sigma_col, theta, background, -
    const double ct = cos(q[5]), st = sin(q[5]);
    const double u = dr * ct + dc * st, v = dc * ct - dr * st;
    return q[6] + q[0] * exp(-0.5 * ((u * u) / (q[3] * q[3]) + (v * v) / (q[4] * q[4])));
  }
  // Moffat.  q: amplitude, row0, col0, alpha, beta, -, background, -
  return q[6] + q[0] * pow(1.0 + (dr * dr + dc * dc) / (q[3] * q[3]), -q[4]);
}
__global__ __launch_bounds__(256) void rasterize_kernel(int model, int n, const double* __restrict__ params, int normalize,
                                                        float* __restrict__ out) {
  __shared__ double part[256];
  const double* q = params + (size_t)blockIdx.x * RPSF_MODEL_PARAMS_DEV;
  float* dst = out + (size_t)blockIdx.x * n * n;
  double scale = 1.0;
  if (normalize) {
    double acc = 0.0;
    for (int e = threadIdx.x; e < n * n; e += 256) acc += psf_model_value(model, q, (double)(e % n), (double)(e / n));
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {  // fixed tree: the sum does not depend on the launch
      if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
      __syncthreads();
    }
    scale = 1.0 / part[0];
  }
  for (int e = threadIdx.x; e < n * n; e += 256) dst[e] = (float)(psf_model_value(model, q, (double)(e % n), (double)(e / n)) * scale);
}


// K4: accum[i] += src[i]
// ------------------------------------------------------------------------------------------------
// (grid-stride: a caller that runs it beside a persistent patch launch gives it a handful of workgroups, so that it lives on the CUs that
// launch leaves free instead of spreading a thousand small workgroups over CUs the patch workgroups are waiting for)
__global__ void add_rows_kernel(float* __restrict__ accum, const float* __restrict__ src, size_t count) {
  const size_t step = (size_t)gridDim.x * blockDim.x * 4;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < count; i += step) {
    if (i + 3 < count) {
      float4 a = *reinterpret_cast<float4*>(accum + i);
      float4 b = *reinterpret_cast<const float4*>(src + i);
      a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
      *reinterpret_cast<float4*>(accum + i) = a;
    } else {
      for (size_t j = i; j < count; ++j) accum[j] += src[j];
    }
  }
}


// ------------------------------------------------------------------------------------------------
// hipFFT fallback (any patch size without a plan): gather, multiply, scatter
// ------------------------------------------------------------------------------------------------
struct GenericGeom {
  int N, first, count;         // patches [first, first + count) of the plan
  int origin_row, origin_col;
  ImageView im;
  OutView ov;
};
// one thread per (patch, r, c) of the chunk
__global__ void generic_gather_kernel(GenericGeom gg, const int32_t* __restrict__ coords, const float* __restrict__ win,
                                      cf* __restrict__ buf) {
  const size_t per = (size_t)gg.N * gg.N;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per * gg.count) return;
  const int k = (int)(idx / per), rem = (int)(idx % per), r = rem / gg.N, c = rem % gg.N;
  const int y = pad_index(coords[2 * (gg.first + k)] + gg.origin_row + r, gg.im.H, gg.im.pad_mode);
  const int x = pad_index(coords[2 * (gg.first + k) + 1] + gg.origin_col + c, gg.im.W, gg.im.pad_mode);
  const float px = (y < 0 || x < 0) ? gg.im.pad_value : gg.im.img[(size_t)(y - gg.im.row0) * gg.im.ld + x];
  buf[idx] = cf{px * (win[r] * win[c]), 0.0f};
}
__global__ void generic_multiply_kernel(cf* __restrict__ buf, const cf* __restrict__ k, size_t count, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) buf[i] = cmul(buf[i], k[i]) * scale;
}
// `colour` < 0: every patch of the chunk, atomic adds (any corner list).  `colour` 0 .. 3: only the patches of that colour class, which do not
// overlap one another (rpsf.hip generic_colours), with plain read-add-write: the four passes of a chunk run one after the other on the
// stream, so every pixel receives its contributions in a fixed order and the fallback is bit-reproducible like the compiled sizes.
__global__ void generic_scatter_kernel(GenericGeom gg, const int32_t* __restrict__ coords, const float* __restrict__ win,
                                       const cf* __restrict__ buf, const uint8_t* __restrict__ colours, int colour) {
  const size_t per = (size_t)gg.N * gg.N;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per * gg.count) return;
  const int k = (int)(idx / per), rem = (int)(idx % per), r = rem / gg.N, c = rem % gg.N;
  if (colour >= 0 && colours[gg.first + k] != colour) return;
  const int y = coords[2 * (gg.first + k)] + gg.origin_row + r, x = coords[2 * (gg.first + k) + 1] + gg.origin_col + c;
  if (y < 0 || y >= gg.ov.H || x < 0 || x >= gg.ov.W) return;  // the crop of transform.py:174-177
  float* dst = gg.ov.out + (size_t)(y - gg.ov.row0) * gg.ov.ld + x;
  const float v = buf[idx].x * (win[r] * win[c]);
  if (colour >= 0) *dst += v;
  else unsafeAtomicAdd(dst, v);
}


#endif  // RPSF_HOST_TU
