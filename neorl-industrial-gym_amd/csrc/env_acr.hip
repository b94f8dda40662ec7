// kernels of AdvancedChemicalReactor (nig_envs.hpp) -- one translation unit per environment
#include "nig_kernels.hpp"
NIG_DEFINE_ENV_LAUNCH(AdvancedChemicalReactor, nig_launch_acr)
