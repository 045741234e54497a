// k1_256.hip - explicit instantiation of the kernels of one group of plans (see rpsf_device.hpp)
#include "rpsf_device.hpp"

RPSF_INST_V1(Cfg256)
