// demc_k1_res.cpp -- the instances of k_propose (demc_kernels.hpp) in its resident form (one workgroup per group, the iterations between two migrations in one launch),
// in a translation unit of their own so that `make -j` compiles them beside the rest of the library.
#define DEMC_DEVICE_HELPERS_ONLY
#include "demc_kernels.hpp"

namespace demc {
#define DEMC_X_(...) template __global__ void k_propose<__VA_ARGS__>(KParams);
DEMC_K1_RES_INSTANCES(DEMC_X_)
#undef DEMC_X_
}  // namespace demc
