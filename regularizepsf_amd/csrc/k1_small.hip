// k1_small.hip - explicit instantiation of the kernels of one group of plans (see rpsf_device.hpp)
#include "rpsf_device.hpp"

RPSF_INST_V1(Cfg64)
RPSF_INST_V1(Cfg32)
RPSF_INST_V1(Cfg16)
