// k2_256p.hip - the persistent form of the 256-pixel second-generation patch kernel (see RPSF_REENTER in rpsf_kernels2.hpp)
#include "rpsf_device.hpp"

struct Reenter256p {
  static constexpr bool enabled = true;
  __device__ __forceinline__ void operator()(unsigned block, unsigned tid) const { RPSF_REENTER(patch_kernel2_256p, block, tid); }
};

extern "C" __global__ __launch_bounds__(512, 2) RPSF_VGPR_ATTR void patch_kernel2_256p(PatchParams p) { patch_body2<Cfg256v2, Reenter256p, /*HOT*/ true>(p, Reenter256p()); }
