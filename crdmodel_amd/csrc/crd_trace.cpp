// crd_trace.cpp -- roctx ranges around step batches, halo exchanges, state transfers and output rows (the
// reference has a shell `time` and a one-second ETA, util/ShellScripts/runFHNmodelTorus.sh:6, src/FHNmodel_torus.cpp:457-477).
// The marker library is bound with dlopen at first use and only when somebody is listening -- rocprofv3 announces itself through
// ROCP_TOOL_LIBRARIES; CRD_ROCTX=1 / 0 forces the binding on / off -- so an ordinary run neither needs nor maps it.
#include <dlfcn.h>

#include <cstdlib>
#include <mutex>

#include "crd_internal.h"

namespace crd {

namespace {

// Bound once, by whichever thread traces first (crd_run's writer thread and its stepping thread may both be that thread, so may
// two issuing threads of a LOCAL group): std::call_once orders the binding before every reader of the two pointers.
struct Roctx {
	int (*push)(const char *) = nullptr;
	int (*pop)() = nullptr;
};
Roctx g_roctx;
std::once_flag g_roctx_once;

void bind_roctx()
{
	const char *force = std::getenv("CRD_ROCTX");
	const bool wanted = force ? std::atoi(force) != 0 : std::getenv("ROCP_TOOL_LIBRARIES") != nullptr;
	if (!wanted) return;
	// rocprofv3 --marker-trace listens to the rocprofiler-sdk build of roctx; the roctracer one is the fallback for older tools
	for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
		void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
		if (!h) continue;
		auto push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
		auto pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
		if (push && pop) {
			g_roctx.push = push;
			g_roctx.pop = pop;
			return;
		}
	}
}

}  // namespace

void trace_push(const char *name)
{
	std::call_once(g_roctx_once, bind_roctx);
	if (g_roctx.push && name) (void)g_roctx.push(name);
}

void trace_pop()
{
	std::call_once(g_roctx_once, bind_roctx);
	if (g_roctx.pop) (void)g_roctx.pop();
}

}  // namespace crd

extern "C" {

void crd_trace_range_push(const char *name) { crd::trace_push(name); }
void crd_trace_range_pop(void) { crd::trace_pop(); }

}  // extern "C"
