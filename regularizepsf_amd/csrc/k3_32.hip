// k3_32.hip - explicit instantiation of the third-generation (sweep) kernels of the 32-pixel plan (see rpsf_device.hpp)
#include "rpsf_device.hpp"

RPSF_INST_V3(Cfg3_32)
