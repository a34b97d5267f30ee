// crd_tuning_knobs.h -- environment knobs of a TUNING build of libcrd (-DCRD_TUNING_BUILD, tools/build_variant.sh); never part of
// crdmodel_amd/libcrd.so.  The launch code asks for a knob on every launch, so tools/tune_fused.py / tools/ring_ab.py can flip
// them between launches of one process (interleaved A/B timing).  Honoured only when CRD_TUNING is set when the process starts;
// the launch-plan measurement then stays out.
//   CRD_FUSED_CHUNK     rows per work item            CRD_FUSED_STRIPS   wavefronts (adjacent strips) per block, 1..4
//   CRD_FUSED_ONEROUND  chunks stretched to one round CRD_FUSED_REMAP    block -> XCD mapping 0 / 1 / 2
//   CRD_FUSED_COLS      columns per lane 1 / 2        CRD_FUSED_NT       non-temporal stores of the new state 0 / 1
//   CRD_AUTOTUNE_VERBOSE=1  one line per timed candidate on stderr (production: crd_set_autotune(ctx, 2))
//   CRD_PLANE_SKEW=bytes    stagger of the field planes inside their allocations (production: 16640)
#pragma once

#include <cstdlib>

namespace crd {
namespace tuning {
inline bool enabled()
{
	static const bool on = std::getenv("CRD_TUNING") != nullptr;
	return on;
}
inline const char *knob(const char *name) { return enabled() ? std::getenv(name) : nullptr; }
inline bool verbose() { return std::getenv("CRD_AUTOTUNE_VERBOSE") != nullptr; }
inline long plane_skew(long dflt)
{
	const char *e = std::getenv("CRD_PLANE_SKEW");
	return e ? (std::atol(e) > 0 ? std::atol(e) : 0) : dflt;
}
}  // namespace tuning
}  // namespace crd
