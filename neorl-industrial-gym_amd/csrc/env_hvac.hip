// kernels of HVACControl (nig_envs.hpp) -- one translation unit per environment
#include "nig_kernels.hpp"
NIG_DEFINE_ENV_LAUNCH(HVACControl, nig_launch_hvac)
