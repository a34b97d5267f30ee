// crd_fused.hip -- the one-launch RK4 step kernels (crd_fused_impl.h): fp64 instantiations, the precision dispatch and the launch
// interface's small functions.  The fp32 instantiations live in crd_fused_f32.hip.
#include "crd_fused_impl.h"

namespace crd {

bool fused_step_supported(int, const SlabDesc &d) { return d.nyl >= 2 * kStepHalo; }

bool fused_two_steps_supported(const SlabDesc &d) { return d.nyl >= 4 * kStepHalo && kernel_model(d) != kModelDiffusionOnly; }

// Steps one launch of this slab takes when the plan asks for `want` (1 .. 3): three only where the three-step kernel exists (FHN:
// kCanThreeSteps) and on a single slab (rows wrap; the exchange cycles of multi-slab runs step pairs), else two where the two-step
// kernels do, else one.
int fused_steps_supported(int precision, const SlabDesc &d, int want)
{
	// (FHN: fp64 with one column per lane, fp32 with two -- an even nx; crd_fused_impl.h: kThreeStepCols)
	bool three = kernel_model(d) == CRD_MODEL_FHN && (precision == CRD_PRECISION_F64 || d.nx % 2 == 0);
#ifdef CRD_THREE_STEPS_GOLDBETER
	three = three || (precision == CRD_PRECISION_F64 && kernel_model(d) == CRD_MODEL_GOLDBETER);
#endif
	// (single slabs; in fp32 -- a strip per wavefront -- the slabs of a multi-slab run too, inside their exchange cycles: the fp64 block strip
	// measured nothing on a rank's share and keeps pairs there)
	if (want >= 3 && three && (d.wrap || precision != CRD_PRECISION_F64) && d.nyl >= 6 * kStepHalo) return 3;
	return (want >= 2 && fused_two_steps_supported(d)) ? 2 : 1;
}

int fused_max_items(const SlabDesc &d)
{
	// upper bound on the work items of any launch on this slab: the narrowest strips, the shortest chunks the heuristic uses
	// (4 rows: launches that leave most CUs idle, fused_chunk_rows)
	const int strips = (d.nx + (kValid - 2) - 1) / (kValid - 2);
	return strips * ((d.nyl + 2 * kGhost + 3) / 4 + 2);
}

const char *fused_kernel_name(int, int) { return "crd_rk4_fused_step_kernel"; }

int fused_default_columns(int precision, int nx) { return (precision == CRD_PRECISION_F32 && nx % 2 == 0) ? 2 : 1; }

int fused_plan_candidates() { return kNumPlanCandidates; }

bool fused_plan_candidate(int index, int *chunk_mode, int *mapping, int *cols, int *nt, int *steps)
{
	if (index < 0 || index >= kNumPlanCandidates) return false;
	*steps = kPlanCandidates[index].steps;
	*chunk_mode = kPlanCandidates[index].one_round;
	*mapping = kPlanCandidates[index].remap;
	*cols = kPlanCandidates[index].cols;
	*nt = kPlanCandidates[index].nt;
	return true;
}

hipError_t launch_sum_partials(const double *partials, int n, double *out, hipEvent_t done, hipStream_t s)
{
	clear_launch_status();
	if (done) hipExtLaunchKernelGGL(crd_sum_partials_kernel, dim3(1), dim3(256), 0, s, nullptr, done, 0, partials, n, out);
	else crd_sum_partials_kernel<<<1, 256, 0, s>>>(partials, n, out);
	return launch_status();
}

hipError_t launch_fused_step(int precision, const SlabDesc &d, const FusedCall &c, int row_begin, int row_end, int row_begin2, int row_end2,
                             hipStream_t s)
{
	static_assert(kApron == kStepHalo && kGhost >= kStepHalo, "the planes carry at least the ghost rows one fused step consumes");
	const int model = kernel_model(d);
	if (precision != CRD_PRECISION_F64) return launch_fused_step_f32(d, c, row_begin, row_end, row_begin2, row_end2, s);
	switch (model) {
	case CRD_MODEL_FHN: return launch_fused_t<double, CRD_MODEL_FHN>(d, c, row_begin, row_end, row_begin2, row_end2, d.js, d.ny, s);
	case CRD_MODEL_GOLDBETER: return launch_fused_t<double, CRD_MODEL_GOLDBETER>(d, c, row_begin, row_end, row_begin2, row_end2, d.js, d.ny, s);
	default: return launch_fused_t<double, kModelDiffusionOnly>(d, c, row_begin, row_end, row_begin2, row_end2, d.js, d.ny, s);
	}
}

}  // namespace crd
