// kernels of PowerGrid (nig_envs.hpp) -- one translation unit per environment
#include "nig_kernels.hpp"
NIG_DEFINE_ENV_LAUNCH(PowerGrid, nig_launch_pg)
