// kernels of ChemicalReactor (nig_envs.hpp) -- one translation unit per environment
#include "nig_kernels.hpp"
NIG_DEFINE_ENV_LAUNCH(ChemicalReactor, nig_launch_cr)
