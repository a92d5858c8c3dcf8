// demc_resmvn.cpp -- the instances of k_res_mvn (demc_resmvn.hpp: the lean resident / streaming-resident / DE-MC_Z bodies of the
// default sampler on MvNormal), in a translation unit of their own: 43 instances of a 800-line kernel are a third of the library's
// compile time, and `make -j` builds the units side by side.  No device code crosses the units (no relocatable device code).
#define DEMC_DEVICE_HELPERS_ONLY
#include "demc_resmvn.hpp"

namespace demc {
#define DEMC_X_(...) template __global__ void k_res_mvn<__VA_ARGS__>(KParams);
DEMC_RESMVN_INSTANCES(DEMC_X_)
DEMC_RESMVN_INSTANCES_DIR(DEMC_X_)
DEMC_RESMVN_INSTANCES_EXP(DEMC_X_)
#undef DEMC_X_
}  // namespace demc
