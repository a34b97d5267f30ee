// crd_kernels.h -- launch interface between the context code (crd_context.cpp, crd_steppers.cpp) and the HIP kernels
// (crd_kernels.hip).  Plain structs; every pointer is a device pointer on the context's device.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace crd {

// A field plane on the device: (nyl + 2*kGhost) rows of nx reals, theta fastest; local row j (j in
// [-kGhost, nyl+kGhost)) starts at base + (j + kGhost) * nx.  Two planes (var0 = diffusing, var1 = local) make a state.
struct Planes {
	void *u;
	void *v;
};

// Everything a kernel needs to know about the slab; passed by value as a kernel argument.
struct SlabDesc {
	const void *cE;    // nx: cX + cA[i], coefficient of uE - uC   (cA = D (-sin th / (r rho)) / (2 dx), 0 for flat;
	const void *cWn;   // nx: cA[i] - cX, coefficient of uC - uW    cX = D / r^2 / dx^2, D/dx^2 for flat)
	const void *cP;    // nx: phi-diffusion coefficient D / rho^2 / dy^2                 (D/dy^2 for flat)
	const void *brow;  // nyl + 2*kGhost: the kinetics' row parameter (FHN: EPSILON b(j); Goldbeter: v0 + v1 b(j)), index j + kGhost
	double ka4;        // Goldbeter pow(KA, p)
	int nx;
	int nyl;
	int wrap;          // 1: single slab, phi neighbours wrap inside the slab; 0: read ghost rows
	int wrap_x = 1;    // 1: the context spans all of theta, column -1 is column nx-1; 0: a theta-block of a 2-D decomposition --
	                   // the columns beside the block come from ghost-column strips (staged kernels / f() only)
	int has_row0;      // slab owns global row 0      (js == 0)
	int has_rowN;      // slab owns global row ny-1   (je == ny-1)
	int js;            // global index of local row 0
	int ny;            // global row count
	int model;
	int just_diffusion;
};

struct StageCall {
	int stage;       // 0: RHS only (out = f(yin)); 1..4: fused RK4 stage
	int absorb;      // t_stage < tBoundary
	double dt;
	Planes yin;      // stencil input of this stage
	Planes y0;       // state at the start of the step (stages 1-3)
	Planes acc;      // running combination (stages 1-4)
	Planes yout;     // stage output (next stage's input; stage 4: the new state; stage 0: ydot)
	// theta-blocks (SlabDesc::wrap_x = 0): var0 of the columns west / east of the block for the rows of yin, nyl reals each
	const void *gcol_w = nullptr, *gcol_e = nullptr;
};

// One evaluation of f on AoS device vectors (the ARKRhsFn boundary): y, ydot are [nyl][nx][2];
// ghost_lo / ghost_hi hold var0 of rows -1 and nyl (ignored when d.wrap).  Rows [row_begin, row_end) are produced; rows
// row_begin-1 and row_end of y must be resident as well (the pipelined host path computes band by band).
hipError_t launch_rhs_aos(int precision, const SlabDesc &d, int absorb, const void *y, void *ydot, const void *ghost_lo,
                          const void *ghost_hi, int row_begin, int row_end, hipStream_t s, const void *gcol_w = nullptr, const void *gcol_e = nullptr);

// One RK4 stage (or a bare RHS) on SoA planes, rows [row_begin, row_end) of the slab.
hipError_t launch_stage(int precision, const SlabDesc &d, const StageCall &c, int row_begin, int row_end, hipStream_t s);
const char *stage_kernel_name(int precision, int model);

// Whole RK4 step in one launch (all four stages on chip); reads y0 with kGhost ghost rows, writes yout rows
// [row_begin, row_end) and, if non-empty, [row_begin2, row_end2).  absorb[k] = t_stage_k < tBoundary for the four stages.
// How the one-launch step cuts a slab into work items and deals them to the XCDs (crd_fused.hip: kPlanCandidates), measured
// on the first full-size launch of a context when autotune is set.  Every plan computes bit-identical results.
struct FusedPlan {
	int autotune = 1;                      // 0: never measure; 1: measure on the first full-size launch; 2: ... and print every timing to stderr
	int tuned = 0;
	int pinned = 0;                        // set by crd_set_launch_plan / CRD_LAUNCH_PLAN: applies to launches of every size
	int one_round = 0, remap = 0;
	int cols = 1;                          // grid columns per lane (1 or 2)
	int nt = 0;                            // the new state stored with the non-temporal hint
	int steps = 1;                         // RK4 steps per launch (2: single slabs only; the stepping loop then issues pairs)
	int rows = 0;                          // height of the launch the plan was measured on
	float ms_default = 0.f, ms_best = 0.f;  // measured launch times of the plain plan and of the chosen one
};

// What a launch_fused_step call WOULD launch (filled instead of launching when FusedCall::geometry is set): the instantiation and how
// the rows are cut.  With the kernel table of the build (step_kernel_stats) it prices a launch's vector issue without a profiler.
struct FusedGeometry {
	int real_bytes, model, absorb, embed, cols, nt, steps;  // template arguments of the kernel
	int strips, chunk_rows, chunks, blocks, waves_per_block, fill_iterations, iterations_per_trip, lanes, lanes_valid, mapping;
	long wave_iterations;  // pipeline iterations of all wavefronts of the launch: strips x (rows + chunks x fill)
};
// One step kernel of this build as the assembler printed it (crd_kernel_table.cpp, generated at build time by tools/kernel_regs.py):
// registers, occupancy, and the static instruction mix of one trip of the steady-state loop.
struct KernelStats {
	int real_bytes, model, absorb, embed, cols, nt, steps, vgprs, sgprs, lds_bytes, scratch_bytes, wavefronts_per_simd, loop_valu, loop_salu, loop_vmem, loop_lds, loop_total;
	int exec_skipped_vmem;  // vector-memory regions a wavefront can skip on its execution mask (multi-step kernels: 0, or the build stops)
};
const KernelStats *step_kernel_stats(int real_bytes, int model, int absorb, int embed, int cols, int nt, int steps);  // nullptr: the build has no table
const char *step_kernel_table_digest();  // 16 hex digits over the table's rows ("" without a table): the build the profile tables are stamped with

struct FusedCall {
	double dt;
	int absorb[5];  // four stages + the embedded pair's fifth (embed = 1: t + dt; embed = 2: t + 3/4 dt)
	int absorb2[4] = {0, 0, 0, 0};  // the stages of the step after this one (t + dt + c_k dt): read by a two-step launch
	int absorb3[4] = {0, 0, 0, 0};  // ... and of the one after that ((t + dt) + dt + c_k dt): a three-step launch
	int steps = 1;  // 2 / 3: this launch advances the rows by TWO / THREE steps (no error estimate; three: single slabs, FHN fp64)
	Planes y0;
	Planes yout;
	// embedded error estimate (adaptive stepping): weighted square sum of the local error over the launch's rows
	int embed = 0;  // 0: none; 1: RK4(3), k5 = f(t + dt, y_new); 2: Zonneveld 5(3)4 (ARKode's default fourth-order table)
	double rtol = 0.0, atol = 0.0;
	double *err_partials = nullptr;  // device scratch, err_capacity doubles (>= fused_max_items)
	int err_capacity = 0;
	double *err_sum = nullptr;       // device: the sum, written by a follow-up reduction on the same stream
	// An attempt cut into two launches (the rows that read owned rows only go under a halo exchange, the rest behind it): the first launch
	// leaves its partials at the front and sums nothing (err_defer_sum; it says how many it wrote in *err_items_out), the second appends
	// its own behind them (err_offset) and the one sum kernel behind it adds all of them, in that order.
	int err_offset = 0;
	bool err_defer_sum = false;
	int *err_items_out = nullptr;
	FusedPlan *plan = nullptr;       // launch plan of the context (nullptr: plain plan)
	// A third plane set the plan measurement may overwrite: candidates are then timed stepping yout -> scratch -> yout ..., every
	// launch reading what the previous one wrote as real stepping does, instead of repeating y0 -> yout (whose input, never
	// overwritten, stays in the 256 MB memory-side cache on slabs that fit: a one-GPU share of an 8-GPU run does).
	Planes tune_scratch{nullptr, nullptr};
	FusedGeometry *geometry = nullptr;  // set: nothing is launched or measured, *geometry says what the call would launch
	// An event that is to be set when the call's LAST kernel has run (the sum kernel of an embedded pair, else the step kernel): bound to
	// that kernel's own completion signal (hipExtLaunchKernel), not recorded behind it -- a record is a barrier packet of its own, and
	// the next kernel of the stream starts ~7 us later for it (kernel-trace timelines, profiles/r05/ring_cycle_timeline.txt).
	hipEvent_t done_event = nullptr;
	// ... and one set when the call's FIRST kernel starts (hipExtLaunchKernel's start event): the pair times a launch with no record on
	// either side of it (crd_step_rk4_timed's sampled launches).
	hipEvent_t start_event = nullptr;
};
hipError_t launch_fused_step(int precision, const SlabDesc &d, const FusedCall &c, int row_begin, int row_end, int row_begin2, int row_end2,
                             hipStream_t s);
// The fixed-order sum of an attempt's error partials (what launch_fused_step launches behind an embedded-pair kernel unless
// FusedCall::err_defer_sum), for callers that run it on another stream; `done`: an event set by the kernel's own completion, or null.
hipError_t launch_sum_partials(const double *partials, int n, double *out, hipEvent_t done, hipStream_t s);
const char *fused_kernel_name(int precision, int model);
bool fused_step_supported(int precision, const SlabDesc &d);
int fused_default_columns(int precision, int nx);  // columns per lane of launches without a measured plan
int fused_plan_candidates();                       // the plans the tuner times (crd_launch_plan_candidate)
bool fused_plan_candidate(int index, int *chunk_mode, int *mapping, int *cols, int *nt, int *steps);
bool fused_two_steps_supported(const SlabDesc &d);
int fused_steps_supported(int precision, const SlabDesc &d, int want);  // steps per launch a plan that asks for `want` gets on this slab
int fused_max_items(const SlabDesc &d);

// Layout adaptors between the AoS boundary layout (host precision: f64 or device precision) and SoA planes.
hipError_t launch_aos_to_planes(int precision, int src_is_f64, const void *aos, Planes dst, int nx, int nyl, hipStream_t s);
hipError_t launch_planes_to_aos(int precision, int dst_is_f64, Planes src, void *aos, int nx, int nyl, hipStream_t s);
// Extract var0 of one AoS row into a contiguous row (halo packing for crd_rhs_*).
hipError_t launch_aos_row_extract(int precision, const void *aos, void *row, int nx, int j, hipStream_t s);
// var0 of the first and the last column (rows 0 .. nyl-1) into two contiguous strips: from an AoS vector / from a field plane
// (the E / W strips of Exchange(), src/FHNmodel_torus.cpp:854-876, for theta-blocks).
hipError_t launch_aos_cols_extract(int precision, const void *aos, void *col_w, void *col_e, int nx, int nyl, hipStream_t s);
hipError_t launch_plane_cols_extract(int precision, const void *u_plane, void *col_w, void *col_e, int nx, int nyl, hipStream_t s);

// out = cubic Hermite interpolant at t_n + theta h of the step (yn, fn) -> (yp, fp); owned rows of both fields.
hipError_t launch_hermite(int precision, Planes yn, Planes yp, Planes fn, Planes fp, Planes out, int nx, int nyl, double theta, double h, hipStream_t s);

// The vector operations of ARKode's initial-step estimate on the owned rows of both fields (arkHin; oracle/arkode_erk.py):
//   launch_hin_bound   *out = max_i |f_i| / (0.1 |y_i| + rtol |y_i| + atol)
//   launch_axpy_planes out = y + h f
//   launch_ydd_sumsq   *out = sum_i (((f2_i - f0_i) / h) / (rtol |y_i| + atol))^2, added in a fixed order (partials_dev: >= 256 doubles)
hipError_t launch_hin_bound(int precision, Planes y, Planes f, int nx, int nyl, double rtol, double atol, double *out_dev, hipStream_t s);
hipError_t launch_axpy_planes(int precision, Planes y, Planes f, double h, Planes out, int nx, int nyl, hipStream_t s);
hipError_t launch_ydd_sumsq(int precision, Planes y, Planes f0, Planes f2, double h, double rtol, double atol, int nx, int nyl, double *partials_dev, double *out_dev,
                            hipStream_t s);

// One double from device memory to page-locked, device-visible host memory by a one-thread kernel, `done` set by that kernel's own
// completion (no record): what the host waits for after a reduction over the ranks.  (hipMemcpyAsync of 8 bytes is a blit kernel of
// 7 - 20 us on this stack, and an event record behind it a barrier packet of its own.)
hipError_t launch_scalar_to_host(const double *src_dev, double *dst_host_mapped, hipEvent_t done, hipStream_t s);

// max |u| over the owned rows, written to *out (device double).
hipError_t launch_max_abs(int precision, const void *u_plane, int nx, int nyl, double *out_dev, hipStream_t s);

}  // namespace crd
