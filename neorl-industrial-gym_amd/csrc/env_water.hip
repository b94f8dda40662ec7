// kernels of WaterTreatment (nig_envs.hpp) -- one translation unit per environment
#include "nig_kernels.hpp"
NIG_DEFINE_ENV_LAUNCH(WaterTreatment, nig_launch_water)
