// kernels of AdvancedPowerGrid (nig_envs.hpp) -- one translation unit per environment
#include "nig_kernels.hpp"
NIG_DEFINE_ENV_LAUNCH(AdvancedPowerGrid, nig_launch_apg)
