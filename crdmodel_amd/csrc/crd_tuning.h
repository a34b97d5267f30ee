// crd_tuning.h -- the experiment switches of the kernel launchers.  A production build has none: every function here is a
// constant, and libcrd.so reads no tuning variable from the environment.  A build with -DCRD_TUNING_BUILD (tools/build_variant.sh;
// bound with CRD_LIBRARY=... by tools/tune_fused.py, tools/ring_ab.py) takes them from tools/tuning/crd_tuning_knobs.h instead.
#pragma once

#ifdef CRD_TUNING_BUILD
#include "../../tools/tuning/crd_tuning_knobs.h"
#else
namespace crd {
namespace tuning {
constexpr bool enabled() { return false; }
inline const char *knob(const char *) { return nullptr; }
constexpr bool verbose() { return false; }
inline long plane_skew(long dflt) { return dflt; }
}  // namespace tuning
}  // namespace crd
#endif
