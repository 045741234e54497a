// k2_128pc.hip - patch_kernel2_128p with plain (cacheable) loads of the pair words, see rpsf_kernels2.hpp
#include "rpsf_device.hpp"

struct Reenter128pc {
  static constexpr bool enabled = true;
  __device__ __forceinline__ void operator()(unsigned block, unsigned tid) const { RPSF_REENTER(patch_kernel2_128pc, block, tid); }
};

extern "C" __global__ __launch_bounds__(128, 2) void patch_kernel2_128pc(PatchParams p) {
  patch_body2<Cfg128v2, Reenter128pc, /*HOT*/ true, /*KNT*/ false>(p, Reenter128pc());
}
