// rpsf_kernels3.hpp - third kernel generation, N = 16, 32, 64 (the patch sizes of the reference's own example,
// docs/source/example.ipynb): sweep_kernel<C>.  One launch is the whole apply (regularizepsf/transform.py:117-177 without the saturation
// branch): no colour planes, no plane-sum kernel, every output pixel written exactly once.
//
// A workgroup owns a REGION of output pixels (rpsf_plan3.hpp) and keeps N rows of it in LDS (the ring).  Its waves draw JOBS from the
// region's list - one slab of 128 / N patches each - and run a job from the gather to the inverse transform without meeting the
// other waves (rpsf_core3.hpp).  Only the last step is ordered: a job adds its slab into the ring after the jobs it overlaps have
// added theirs (two flags per job, LDS words, polled by the wave), which fixes the order of the four additions of every pixel.
// Phase-B jobs then write the band of H rows that their slab has just completed to the output image.  Patches on the border between two
// regions are computed by both (nothing is handed from one workgroup to another).
#pragma once

struct SweepParams {
  ImageView im;
  Flush3 fl;
  int lat_r0, lat_c0;  // image coordinates of the lattice origin
  const Job3* jobs;
  const Region3* regions;
  int n_regions, group;  // group = regions per XCD (blocks b and b + 8 share an XCD)
  const float* k3;
  const float* win;    // N window weights
  const float* zeros;  // 16 bytes of zeros
  size_t im_frame_floats, out_frame_floats;
  int aligned_in;  // 16-byte gathers allowed (image pointer, row stride and column origin multiples of four floats)
  unsigned long long* stamps;  // development builds: phase timestamps / dumps, else null
  uint32_t* err;   // set to non-zero if a dependency wait ran into its bound (never, unless the job lists are wrong)
};

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ void lds_fence_wave() {
  // the lanes of a wave exchange data through LDS without a barrier: DS instructions of one wave execute in order, the compiler only has to keep them so
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t lds_u32_offset(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

// Development builds (-DRPSF3_STAMPS, scripts/stamps_sweep.py): 15 timestamps (10 ns ticks) of each of the first eight jobs of the first eight waves of
// every region: [region][wave][job slot][16]
#if defined(RPSF3_STAMPS)
#define STAMP3(i)                                                                                                                         \
  do { /* (scheduling barriers: the compiler must not move a phase's arithmetic across its stamp) */                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                                    \
    if (lane0 == 0 && wave < 8 && jslot < 8) P.stamps[(((size_t)region * 8 + wave) * 8 + jslot) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                                                                    \
  } while (0)
// ... behind arithmetic on the lane's values: two of them are tied to the stamp, or the optimiser sinks a phase's transforms past it (they feed nothing before
// the next transpose) and the phase reads 0.05 us
#define STAMP3T(i)                                                      \
  do {                                                                  \
    asm volatile("" : "+v"(v[0].x), "+v"(v[N - 1].y) : : "memory");     \
    STAMP3(i);                                                          \
  } while (0)
#else
#define STAMP3(i) ((void)0)
#define STAMP3T(i) ((void)0)
#endif

template <class C, bool NT>
__device__ __forceinline__ void sweep_body(const SweepParams& P) {
  constexpr int N = C::N, H = C::H;
  extern __shared__ __attribute__((aligned(16))) float lds3[];
  float* ring = lds3;
  const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* xb = ring + C::RINGF + wave * C::XF;
  float* side = ring + C::RINGF + C::WAVES * C::XF + wave * C::SIDEF;
  uint32_t* flags = reinterpret_cast<uint32_t*>(ring + C::RINGF + C::WAVES * (C::XF + C::SIDEF));
  uint32_t* next = flags + C::NFLAGS;
  const int b = blockIdx.x;
  const int region = (b & 7) * P.group + (b >> 3);
  if (region >= P.n_regions) return;
  const Region3 reg = P.regions[region];
  for (int i = tid; i < C::NFLAGS + 4; i += C::WG) flags[i] = 0;
  if (lane0 < 4) side[C::SIDE_ZERO + lane0] = 0.0f;
  __syncthreads();
  ImageView im = P.im;
  Flush3 fl = P.fl;
  im.img += (size_t)blockIdx.y * P.im_frame_floats;
  fl.out += (size_t)blockIdx.y * P.out_frame_floats;
  const float w_re = P.win[lane0 % H], w_im = P.win[lane0 % H + H];
  const uint32_t flags_off = lds_u32_offset(flags), next_off = lds_u32_offset(next);
#if defined(RPSF3_STAGGER_TICKS)
  {  // development: the waves of a workgroup start RPSF3_STAGGER_TICKS x 10 ns apart, RPSF3_STAGGER_GROUP of them together (the jobs of one phase)
#if !defined(RPSF3_STAGGER_GROUP)
#define RPSF3_STAGGER_GROUP 1
#endif
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)(wave / RPSF3_STAGGER_GROUP) * RPSF3_STAGGER_TICKS) __builtin_amdgcn_s_sleep(8);
  }
#endif
  [[maybe_unused]] int jslot = -1;
  for (;;) {
    ++jslot;
    STAMP3(0);
    // (per-lane addresses are recomputed every pass: hoisted out of the loop - the compiler's choice otherwise - they stay live across the
    // whole body and cost registers)
    int ln = lane0;
    asm volatile("" : "+v"(ln));
    const int lane = ln, q = lane / H, p = lane % H;
    // ---- draw the next job of the region ----
    // (one lane draws; the exec mask is narrowed inside the asm statement so that the compiler sees uniform control flow around it)
    uint32_t drawn;
    {
      const uint32_t one = 1;
      uint64_t saved;
      asm volatile(
          "s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %2, %3\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, %1"
          : "=&v"(drawn), "=&s"(saved)
          : "v"(next_off), "v"(one)
          : "memory");
    }
    const int j = (int)__builtin_amdgcn_readfirstlane(drawn);
    if (j >= reg.njobs) break;
    STAMP3(1);
    const Job3* jd = P.jobs + reg.job0 + j;
    // The descriptor this wave will most likely draw next (WAVES jobs on) is requested now and thrown away: a job's 64 bytes are read once per launch, so
    // they come from HBM (0.8 us of a 14 us job, profiles/r06k) unless somebody has asked for the line before.  The eight scalar registers stay reserved
    // until the wait for this job's own descriptor - which is a wait for every scalar load in flight - has passed (the statement behind it).
    // 4096^2: N = 64 -1.5 %, N = 32 -0.5 %, N = 16 nothing (profiles/r06zo_sweep_descriptor_prefetch.log).
    typedef int int8v __attribute__((ext_vector_type(8)));
    int8v ahead = {};
    if constexpr (C::DESC_PREFETCH) {
      const int jn = j + C::WAVES < reg.njobs ? j + C::WAVES : reg.njobs - 1;
      const Job3* jp = P.jobs + reg.job0 + jn;
      asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(ahead) : "s"(jp) : "memory");
    }
    // the descriptor through the scalar cache (a uniform address in the constant address space)
    typedef const int __attribute__((address_space(4))) cint_as4;
    const cint_as4* jds = (const cint_as4*)(const void*)jd;
    const int jrow = jds[0], jcol = jds[1], jring_col = jds[2];
    const uint32_t jflags = (uint32_t)jds[3];
    const int jdep0 = jds[4], jdep1 = jds[5], jown0 = jds[6], jown1 = jds[7];
    const int row0 = P.lat_r0 + jrow, col0 = P.lat_c0 + jcol;
    if constexpr (C::DESC_PREFETCH) asm volatile("s_waitcnt lgkmcnt(0)" : : "s"(ahead), "s"(row0), "s"(col0) : "memory");
    STAMP3(2);
#if defined(RPSF3_ABL_ONE_K)  // ablation (wrong results): every patch multiplies by the transfer kernel of slot q - K comes from the caches
    const int kslot = q;
#else
    const int kslot = jd->kslot[q];
#endif
    // ---- gather ----
    f32x4 g[C::DIRECT_GATHER ? 1 : H];
    cf v[N];
    const bool fast = P.aligned_in && row0 >= im.row0 && row0 + N <= im.row0 + im.rows && row0 >= 0 && row0 + N <= im.H && col0 >= 0 &&
                      col0 + C::SLABW <= im.W;
#if defined(RPSF3_ABL_ONE_SLAB)  // ablation (wrong results): every job reads the image's first slab - pixels come from the caches
    const float* slab = im.img + (size_t)(wave * 2) * im.ld;
#else
    const float* slab = im.img + (size_t)(row0 - im.row0) * im.ld + col0;
#endif
    if constexpr (C::DIRECT_GATHER) {
      if (fast) r3_load_fast<C, 0>(lane, v, slab, im.ld);
      else r3_load_generic<C, 0>(lane, v, im, row0, col0);
      if constexpr (!C::SPLIT_GATHER) {
        if (fast) r3_load_fast<C, 1>(lane, v, slab, im.ld);
        else r3_load_generic<C, 1>(lane, v, im, row0, col0);
      }
    } else if constexpr (C::SPLIT_GATHER) {
      if (fast) g3_load_fast<C, 0, 1>(lane, g, slab, im.ld);
      else g3_load_generic<C, 0, 1>(lane, g, im, row0, col0);
    } else {
      if (fast) g3_load_fast<C>(lane, g, slab, im.ld);
      else g3_load_generic<C>(lane, g, im, row0, col0);
    }
    // this job's transfer kernel: the first a words of the lane's column on their way while the rows are transformed; the b words of the slab's
    // column-0 lanes - one word per lane, coalesced - go through LDS (every other lane multiplies by zeros from there)
    const float* kp = P.k3 + (size_t)kslot * C::K_FLOATS;
    f32x4 ka[C::KPRE > 0 ? C::KPRE : 1];
    auto request_k = [&]() RPSF_AI {
      StaticFor<0, C::KPRE>::run([&]<int J>() RPSF_AI {
        if constexpr (NT) ka[J] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kp + (size_t)(J * H + p) * 4));
        else ka[J] = *reinterpret_cast<const f32x4*>(kp + (size_t)(J * H + p) * 4);
      });
    };
    if constexpr (!C::KPRE_LATE) request_k();
    const f32x4 kbw = *reinterpret_cast<const f32x4*>(kp + C::KA_FLOATS + p * 4);
    STAMP3(3);
    // ---- rows ----
    if constexpr (C::DIRECT_GATHER) {
      if constexpr (C::SPLIT_GATHER) {  // (N = 64: the second row of the lane only now - 64 registers of pixels in flight at a time)
        if (fast) r3_load_fast<C, 1>(lane, v, slab, im.ld);
        else r3_load_generic<C, 1>(lane, v, im, row0, col0);
      }
    } else if constexpr (C::SPLIT_GATHER) {
      StaticFor<0, C::NSUB>::run([&]<int S>() RPSF_AI {
        t0_write<C, 0, S>(lane, g, xb);
        lds_fence_wave();
        t0_read<C, 0, S>(lane, v, xb);
        lds_fence_wave();
      });
      if (fast) g3_load_fast<C, 1, 1>(lane, g, slab, im.ld);
      else g3_load_generic<C, 1, 1>(lane, g, im, row0, col0);
      StaticFor<0, C::NSUB>::run([&]<int S>() RPSF_AI {
        t0_write<C, 1, S>(lane, g, xb);
        lds_fence_wave();
        t0_read<C, 1, S>(lane, v, xb);
        lds_fence_wave();
      });
    } else {
      StaticFor<0, C::NSUB>::run([&]<int S>() RPSF_AI {
        t0_write<C, 0, S>(lane, g, xb);
        lds_fence_wave();
        t0_read<C, 0, S>(lane, v, xb);
        lds_fence_wave();
        t0_write<C, 1, S>(lane, g, xb);
        lds_fence_wave();
        t0_read<C, 1, S>(lane, v, xb);
        lds_fence_wave();
      });
    }
    STAMP3T(4);
    window_in_fft_rows<C>(v, w_re, w_im);
    unpack_rows<C>(v);
    STAMP3T(5);
    // ---- columns ----
    StaticFor<0, C::NSUB>::run([&]<int S>() RPSF_AI {
      t1_write<C, 0, S>(lane, v, xb);
      lds_fence_wave();
      t1_read<C, 0, S>(lane, v, xb);
      lds_fence_wave();
      t1_write<C, 1, S>(lane, v, xb);
      lds_fence_wave();
      t1_read<C, 1, S>(lane, v, xb);
      lds_fence_wave();
    });
    STAMP3T(6);
    if constexpr (C::KPRE_LATE) request_k();
    lds_st4(side + 4 * lane, kbw);
    lds_fence_wave();
#if !defined(RPSF3_ABL_NO_COLUMN_FFTS)  // ablation (wrong results): the two column transforms left out - how much of a job is arithmetic?
    FftSmall<C::LOGN, false>::run(v);
#endif
    STAMP3T(7);
    {
      const bool col0lane = p == 0;
      kmul3<C, NT>(v, ka, kp + p * 4, col0lane ? side + 4 * (q * H) : side + C::SIDE_ZERO, col0lane ? 4 : 0);
    }
    STAMP3T(8);
#if !defined(RPSF3_ABL_NO_COLUMN_FFTS)
    FftSmall<C::LOGN, true>::run(v);
#endif
    STAMP3T(9);
    // ---- back to rows ----
    StaticFor<0, C::NSUB>::run([&]<int S>() RPSF_AI {
      t2_write<C, 0, S>(lane, v, xb);
      lds_fence_wave();
      t2_read<C, 0, S>(lane, v, xb);
      lds_fence_wave();
      t2_write<C, 1, S>(lane, v, xb);
      lds_fence_wave();
      t2_read<C, 1, S>(lane, v, xb);
      lds_fence_wave();
    });
    STAMP3T(10);
    repack_rows<C>(v);
    FftSmall<C::LOGN, true>::run(v);
    // second window before the wait: what follows the wait is what the jobs behind this one wait for
    window_out<C>(v, w_re, w_im, ((jflags >> (J3_VALID_SHIFT + q)) & 1u) != 0);
    STAMP3T(11);
    // ---- wait for the jobs this one overlaps, then add ----
    {
      const int d0 = jdep0, d1 = jdep1;
      // (bounded: a protocol error must not hang the GPU - it is reported through P.err instead; 2^22 polls of ~0.1 us are far beyond any real wait)
      // Both flags are read by one pair of LDS instructions and one wait (a job without a second - or any - predecessor asks slot 0 for "at least 0").
      auto wait_for_both = [&](int da, int db) RPSF_AI {
        if (da < 0 && db < 0) return;
        const uint32_t a0 = flags_off + 4u * ((uint32_t)(da < 0 ? 0 : da) & (C::NFLAGS - 1)), want0 = da < 0 ? 0u : (uint32_t)da + 1;
        const uint32_t a1 = flags_off + 4u * ((uint32_t)(db < 0 ? 0 : db) & (C::NFLAGS - 1)), want1 = db < 0 ? 0u : (uint32_t)db + 1;
        for (int spin = 0; spin < (1 << 22); ++spin) {
          uint32_t s0, s1;
          asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(s0), "=&v"(s1) : "v"(a0), "v"(a1) : "memory");
          if (__builtin_amdgcn_readfirstlane(s0) >= want0 && __builtin_amdgcn_readfirstlane(s1) >= want1) return;
          __builtin_amdgcn_s_sleep(1);
        }
        if (lane0 == 0) __hip_atomic_store(P.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (a word in page-locked host memory: rpsf.hip sweep_check)
      };
#if !defined(RPSF3_ABL_NO_WAIT)  // ablation (races: wrong results): the adds are not ordered - what does the order cost?
      wait_for_both(d0, d1);
#endif
    }
    STAMP3T(12);
#if defined(RPSF3_ABL_FORCE_ERR)  // development: exercise the report path of a wait that ran out (scripts/sweep_force_error.py)
    if (lane0 == 0 && j == 0 && region == 0) __hip_atomic_store(P.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
    __builtin_amdgcn_s_setprio(3);  // (the adds of a job are what the jobs after it wait for: they go first in the SIMD's arbitration)
    const int hs = (jflags & J3_RING_HALF) ? 1 : 0;
    {
      float* ru = ring + (hs * H + p) * C::RP + jring_col + q * N;
      float* rl = ring + ((hs ^ 1) * H + p) * C::RP + jring_col + q * N;
      accumulate3<C>(v, ru, rl, (int)((jflags >> J3_UPPER_SHIFT) & 3u), (int)((jflags >> J3_LOWER_SHIFT) & 3u));
    }
    STAMP3T(13);
    // ---- phase B: the band(s) this slab has completed go to the output image ----
    // (read and stored unit by unit: a version that read the band into a register array first, released the flag and stored afterwards
    // gave wrong images on the GPU - and right ones in the emulator - in every form tried, profiles/r06i)
#if defined(RPSF3_ABL_NO_FLUSH)  // ablation (wrong results): nothing is written to the output image
    if (false) {
#else
    if (jflags & (J3_FLUSH_UPPER | J3_FLUSH_LOWER)) {
#endif
      lds_fence_wave();
      auto st4 = [](float* dst, f32x4 x) RPSF_AI { __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(dst)); };
      auto st1 = [](float* dst, float x) RPSF_AI { __builtin_nontemporal_store(x, dst); };
      const int oc0 = P.lat_c0 + jown0, oc1 = P.lat_c0 + jown1;
      if (jflags & J3_FLUSH_UPPER) flush3<C>(lane, ring + (hs * H) * C::RP + jring_col, fl, row0, col0, oc0, oc1, st4, st1);
      if (jflags & J3_FLUSH_LOWER) flush3<C>(lane, ring + ((hs ^ 1) * H) * C::RP + jring_col, fl, row0 + H, col0, oc0, oc1, st4, st1);
    }
    STAMP3T(14);
    // ---- done: LDS executes a wave's instructions in order, so whoever sees the flag sees the adds (and the flush has read its rows) ----
    {  // (every lane stores the same word: no branch)
      const uint32_t a = flags_off + 4u * ((uint32_t)j & (C::NFLAGS - 1)), val = (uint32_t)j + 1;
      asm volatile("ds_write_b32 %0, %1" : : "v"(a), "v"(val) : "memory");
      __builtin_amdgcn_s_setprio(0);
    }
  }
}

template <class C>
__global__ __launch_bounds__(C::WG) void sweep_kernel(SweepParams P) {
  sweep_body<C, true>(P);
}
// (K by plain loads: batches of frames share it, and small transfer kernels stay in the Infinity Cache from one apply to the next)
template <class C>
__global__ __launch_bounds__(C::WG) void sweep_kernel_kc(SweepParams P) {
  sweep_body<C, false>(P);
}

// K pack for the sweep kernel (one-time): fold + reorder, see Cfg3 / pack_value3
template <class C>
__global__ void pack_kernel3(const cf* __restrict__ kfull, int n_patches, cf* __restrict__ k3) {
  constexpr int per = C::K_FLOATS / 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)per * n_patches) return;
  const int patch = (int)(idx / per), rem = (int)(idx % per);
  const cf* kf = kfull + (size_t)patch * C::N * C::N;
  k3[idx] = pack_value3<C>(kf, rem);
}
template <class C>
__global__ void pack_spectra_kernel3(const cf* __restrict__ s_fft, const cf* __restrict__ t_fft, float alpha, float eps, int n_patches,
                                     cf* __restrict__ k3) {
  constexpr int per = C::K_FLOATS / 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)per * n_patches) return;
  const int patch = (int)(idx / per), rem = (int)(idx % per);
  const size_t off = (size_t)patch * C::N * C::N;
  k3[idx] = pack_value3<C>(KFromSpectra{s_fft + off, t_fft + off, alpha, eps}, rem);
}
#endif
