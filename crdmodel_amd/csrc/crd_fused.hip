// crd_fused.hip -- whole-RK4-step kernel (all four stages on chip).  Placeholder until the kernel lands: the
// context falls back to the staged stepper while fused_step_supported() is false.
#include <hip/hip_runtime.h>

#include "crd_internal.h"
#include "crd_kernels.h"

namespace crd {

bool fused_step_supported(int, const SlabDesc &) { return false; }
const char *fused_kernel_name(int, int) { return "crd_rk4_fused_step_kernel"; }
hipError_t launch_fused_step(int, const SlabDesc &, const FusedCall &, int, int, hipStream_t) { return hipErrorNotSupported; }

}  // namespace crd
