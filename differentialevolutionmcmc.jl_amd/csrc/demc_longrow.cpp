// demc_longrow.cpp -- the instances of k_longrow (demc_longrow.hpp), in a translation unit of their own.
//
// Why: the kernel is PERSISTENT -- a loop over the workgroup's particles around the whole body -- and with the compiler's
// machine-level loop-invariant code motion on, every FP64 literal of every span loop (the softplus polynomials alone hold
// ~30) and every invariant address is hoisted out of that outer loop and stays live across all of it: 256 VGPRs + scratch
// spills + `v_readlane` reloads inside the span loops, against 191 VGPRs with the pass off (the one-particle kernel of round
// 3: 220).  The flag is global to a compilation, so the kernel gets its own (see the Makefile); the other kernels keep it.
#define DEMC_DEVICE_HELPERS_ONLY
#include "demc_longrow.hpp"

namespace demc {
template __global__ void k_longrow<256>(KParams);
template __global__ void k_longrow<512>(KParams);
#ifdef DEMC_EXPERIMENTS  // A/B builds only (DEMC_LR_WG = 384 / 1024): three and four waves per SIMD
template __global__ void k_longrow<384, 2>(KParams);
template __global__ void k_longrow<512, 2>(KParams);
#endif
}  // namespace demc
