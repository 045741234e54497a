// k2_256s.hip - development only (RPSF_DEV_SPLIT): the split-patch timing skeleton of VERDICT round 3, item 1 (A) - two independent
// 256-thread workgroups per CU, each on half a 256-pixel patch (see Cfg2's LOGR_ and patch_body2's C::HALF).  Empty in the product build.
#include "rpsf_device.hpp"

#if defined(RPSF_DEV_SPLIT)
struct Reenter256s {
  static constexpr bool enabled = true;
  __device__ __forceinline__ void operator()(unsigned block, unsigned tid) const { RPSF_REENTER(patch_kernel2_256s, block, tid); }
};

extern "C" __global__ __launch_bounds__(256, 2) void patch_kernel2_256s(PatchParams p) { patch_body2<Cfg256half, Reenter256s, /*HOT*/ true>(p, Reenter256s()); }
#endif

#if defined(RPSF_DEV_WIDE)
// ... and of VERDICT round 2's 1024-thread layout: one workgroup of 16 waves per CU, 32 values per thread, 128 registers (Cfg256wide)
struct Reenter256w {
  static constexpr bool enabled = true;
  __device__ __forceinline__ void operator()(unsigned block, unsigned tid) const { RPSF_REENTER(patch_kernel2_256w, block, tid); }
};

extern "C" __global__ __launch_bounds__(1024, 4) void patch_kernel2_256w(PatchParams p) { patch_body2<Cfg256wide, Reenter256w, /*HOT*/ true>(p, Reenter256w()); }
#endif
