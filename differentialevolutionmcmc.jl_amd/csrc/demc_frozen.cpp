// demc_frozen.cpp -- the instances of k_frozen_sweep (demc_frozen.hpp: the row-streaming block sweeps of long hierarchical rows), in
// a translation unit of their own so that `make -j` compiles them beside the rest of the library.
#define DEMC_DEVICE_HELPERS_ONLY
#include "demc_frozen.hpp"

namespace demc {
#define DEMC_X_(...) template __global__ void k_frozen_sweep<__VA_ARGS__>(KParams);
DEMC_FROZEN_INSTANCES(DEMC_X_)
DEMC_FROZEN_INSTANCES_EXP(DEMC_X_)
#undef DEMC_X_
}  // namespace demc
