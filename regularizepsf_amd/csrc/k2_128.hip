// k2_128.hip - explicit instantiation of the kernels of one group of plans (see rpsf_device.hpp)
#include "rpsf_device.hpp"

RPSF_INST_V2(Cfg128v2)
