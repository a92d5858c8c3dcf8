// demc_resobs.cpp -- the instances of k_res_obs (demc_resobs.hpp: the lean resident kernel of the default sampler on the per-observation
// families), in a translation unit of their own so that `make -j` compiles them beside the rest of the library.
#define DEMC_DEVICE_HELPERS_ONLY
#include "demc_resobs.hpp"

namespace demc {
#define DEMC_X_(...) template __global__ void k_res_obs<__VA_ARGS__>(KParams);
DEMC_RESOBS_INSTANCES(DEMC_X_)
#undef DEMC_X_
}  // namespace demc
