// crd_fused_f32.hip -- the fp32 instantiations of the one-launch RK4 step kernels (crd_fused_impl.h); crd_fused.hip has the fp64 ones
// and the precision dispatch.
#include "crd_fused_impl.h"

namespace crd {

hipError_t launch_fused_step_f32(const SlabDesc &d, const FusedCall &c, int row_begin, int row_end, int row_begin2, int row_end2, hipStream_t s)
{
	switch (kernel_model(d)) {
	case CRD_MODEL_FHN: return launch_fused_t<float, CRD_MODEL_FHN>(d, c, row_begin, row_end, row_begin2, row_end2, d.js, d.ny, s);
	case CRD_MODEL_GOLDBETER: return launch_fused_t<float, CRD_MODEL_GOLDBETER>(d, c, row_begin, row_end, row_begin2, row_end2, d.js, d.ny, s);
	default: return launch_fused_t<float, kModelDiffusionOnly>(d, c, row_begin, row_end, row_begin2, row_end2, d.js, d.ny, s);
	}
}

}  // namespace crd
