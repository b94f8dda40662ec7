// kernels of SteelAnnealing (nig_envs.hpp) -- one translation unit per environment
#include "nig_kernels.hpp"
NIG_DEFINE_ENV_LAUNCH(SteelAnnealing, nig_launch_steel)
