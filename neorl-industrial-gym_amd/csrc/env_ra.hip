// kernels of RobotAssembly (nig_envs.hpp) -- one translation unit per environment
#include "nig_kernels.hpp"
NIG_DEFINE_ENV_LAUNCH(RobotAssembly, nig_launch_ra)
