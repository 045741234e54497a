// k2_128p.hip - the persistent form of the 128-pixel second-generation patch kernel (see RPSF_REENTER in rpsf_kernels2.hpp)
#include "rpsf_device.hpp"

struct Reenter128p {
  static constexpr bool enabled = true;
  __device__ __forceinline__ void operator()(unsigned block, unsigned tid) const { RPSF_REENTER(patch_kernel2_128p, block, tid); }
};

extern "C" __global__ __launch_bounds__(128, 2) void patch_kernel2_128p(PatchParams p) { patch_body2<Cfg128v2, Reenter128p, /*HOT*/ true>(p, Reenter128p()); }
