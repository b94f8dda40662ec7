// kernels of SupplyChain (nig_envs.hpp) -- one translation unit per environment
#include "nig_kernels.hpp"
NIG_DEFINE_ENV_LAUNCH(SupplyChain, nig_launch_supply)
