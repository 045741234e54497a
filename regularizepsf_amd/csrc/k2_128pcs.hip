// k2_128pcs.hip - patch_kernel2_128pc with streaming plane stores, see rpsf_kernels2.hpp
#include "rpsf_device.hpp"

struct Reenter128pcs {
  static constexpr bool enabled = true;
  __device__ __forceinline__ void operator()(unsigned block, unsigned tid) const { RPSF_REENTER(patch_kernel2_128pcs, block, tid); }
};

extern "C" __global__ __launch_bounds__(128, 2) void patch_kernel2_128pcs(PatchParams p) {
  patch_body2<Cfg128v2, Reenter128pcs, /*HOT*/ true, /*KNT*/ false, /*PLANE_NT*/ true>(p, Reenter128pcs());
}
