// rpsf_device.hpp - what every translation unit of librpsf_hip.so includes: the C ABI, the per-thread phases and the
// kernel templates, plus the list of plans compiled into the library (explicitly instantiated in the k1_*.hip units and
// declared extern everywhere else).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <type_traits>

#include "../../include/rpsf.h"
#include "rpsf_core.hpp"
#include "rpsf_core2.hpp"
#include "rpsf_core3.hpp"
#include "rpsf_plan3.hpp"

using namespace rpsf;

#include "rpsf_kernels.hpp"
#include "rpsf_kernels2.hpp"
#include "rpsf_kernels3.hpp"

#define RPSF_PLANS_V1(X) X(Cfg256) X(Cfg128) X(Cfg64) X(Cfg32) X(Cfg16)
#define RPSF_PLANS_V2(X) X(Cfg256v2) X(Cfg128v2)
#define RPSF_PLANS_V3(X) X(Cfg3_64) X(Cfg3_32) X(Cfg3_16)
#define RPSF_DECL_V1(C)                                                                                        \
  extern template __global__ void patch_kernel<C>(PatchParams);                                                \
  extern template __global__ void pack_kernel<C>(const cf*, int, const uint16_t*, const uint32_t*, cf*, cf*);  \
  extern template __global__ void pack_spectra_kernel<C>(const cf*, const cf*, float, float, int, const uint16_t*, const uint32_t*, cf*, cf*); \
  extern template __global__ void psf_fft_kernel<C>(const float*, int, const uint16_t*, const cf*, cf*);
#define RPSF_DECL_V2(C)                                           \
  extern template __global__ void patch_kernel2<C>(PatchParams); \
  extern template __global__ void pack_kernel2<C>(const cf*, int, const uint16_t*, const uint32_t*, cf*, cf*); \
  extern template __global__ void pack_spectra_kernel2<C>(const cf*, const cf*, float, float, int, const uint16_t*, const uint32_t*, cf*, cf*);
#define RPSF_INST_V1(C)                                                                                \
  template __global__ void patch_kernel<C>(PatchParams);                                               \
  template __global__ void pack_kernel<C>(const cf*, int, const uint16_t*, const uint32_t*, cf*, cf*); \
  template __global__ void pack_spectra_kernel<C>(const cf*, const cf*, float, float, int, const uint16_t*, const uint32_t*, cf*, cf*); \
  template __global__ void psf_fft_kernel<C>(const float*, int, const uint16_t*, const cf*, cf*);
#define RPSF_INST_V2(C)                                    \
  template __global__ void patch_kernel2<C>(PatchParams); \
  template __global__ void pack_kernel2<C>(const cf*, int, const uint16_t*, const uint32_t*, cf*, cf*); \
  template __global__ void pack_spectra_kernel2<C>(const cf*, const cf*, float, float, int, const uint16_t*, const uint32_t*, cf*, cf*);
#define RPSF_DECL_V3(C)                                                                  \
  extern template __global__ void sweep_kernel<C>(SweepParams);                          \
  extern template __global__ void sweep_kernel_kc<C>(SweepParams);                       \
  extern template __global__ void pack_kernel3<C>(const cf*, int, cf*);                  \
  extern template __global__ void pack_spectra_kernel3<C>(const cf*, const cf*, float, float, int, cf*);
#define RPSF_INST_V3(C)                                                           \
  template __global__ void sweep_kernel<C>(SweepParams);                          \
  template __global__ void sweep_kernel_kc<C>(SweepParams);                       \
  template __global__ void pack_kernel3<C>(const cf*, int, cf*);                  \
  template __global__ void pack_spectra_kernel3<C>(const cf*, const cf*, float, float, int, cf*);
