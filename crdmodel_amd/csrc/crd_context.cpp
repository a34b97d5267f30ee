// crd_context.cpp -- the device context behind the C ABI: memory, streams, halo transports, the RK4 drivers.
// Host code only (compiled by hipcc for the HIP runtime API); kernels live in crd_kernels.hip / crd_fused.hip.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only; the library is bound with dlopen at first use

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>

#include "crd_internal.h"
#include "crd_kernels.h"

using namespace crd;

namespace {
thread_local std::string g_create_error;  // crd_last_error(NULL)

// RCCL is bound at first use, not at load time: single-GPU runs never map the library, and a process that already
// carries an RCCL (PyTorch bundles one under the same SONAME) shares that copy instead of loading a second one.
struct RcclApi {
	void *handle = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	std::string error;

	bool load()
	{
		if (handle) return true;
		if (!error.empty()) return false;
		for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (handle) break;
		}
		if (!handle) {
			error = std::string("cannot load librccl: ") + dlerror();
			return false;
		}
		auto sym = [&](const char *n) {
			void *p = dlsym(handle, n);
			if (!p && error.empty()) error = std::string("librccl lacks ") + n;
			return p;
		};
		GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
		CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
		CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
		GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
		GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
		Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
		Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
		AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
		GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
		if (!error.empty()) {
			handle = nullptr;
			return false;
		}
		return true;
	}
};
RcclApi g_rccl;
}

// The three streams of a context: interior sweeps, edge bands (high priority), halo exchange.  Contexts of a LOCAL group
// that share a device share one set (crd_comm_attach_local), so a device never carries more than three of this library's
// streams however many slabs it hosts.
struct StreamSet {
	int device = 0;
	hipStream_t compute = nullptr, comm = nullptr, band = nullptr;
	~StreamSet()
	{
		(void)hipSetDevice(device);
		for (hipStream_t s : {compute, comm, band})
			if (s) {
				(void)hipStreamSynchronize(s);
				(void)hipStreamDestroy(s);
			}
	}
};

struct crd_ctx {
	crd_params p{};
	crd_grid g{};
	int slab = 0, n_slabs = 1, device = 0;
	int64_t js = 0, je = 0;
	int nx = 0, nyl = 0;
	size_t real_size = 8;
	size_t plane_bytes = 0;

	// State planes: Y (current), SA / SB (stage ping-pong), ACC; [k][0] = var0, [k][1] = var1.
	enum { Y = 0, SA = 1, SB = 2, ACC = 3, NPLANES = 4 };
	void *plane[NPLANES][2] = {};
	void *cA = nullptr, *cP = nullptr, *brow = nullptr;
	void *stage_in = nullptr, *stage_out = nullptr;  // AoS staging for the *_host entry points (lazy)
	size_t stage_bytes = 0;
	void *ghost_lo = nullptr, *ghost_hi = nullptr;   // var0 of rows -1 / nyl for the AoS RHS (multi-slab)
	void *edge_lo = nullptr, *edge_hi = nullptr;     // var0 of rows 0 / nyl-1 packed from an AoS vector
	double *scalar_dev = nullptr;
	double *err_partials = nullptr;  // adaptive stepping: per-item error sums (lazy)
	int err_capacity = 0;

	std::shared_ptr<StreamSet> streams;                               // owner of the handles below
	hipStream_t compute = nullptr, comm = nullptr, band = nullptr;
	// Edge bands are launched on the compute stream ahead of the interior sweep.  CRD_BAND_STREAM=1 moves them to a third,
	// high-priority stream (measured on one GPU with an RCCL self-ring: no gain at 8192 x 1024..4096 slabs, and the time then
	// depends on when the runtime maps that stream to a hardware queue), kept as a knob for multi-GPU experiments.
	bool bands_on_own_stream = false;
	hipEvent_t ev_edges = nullptr, ev_halo = nullptr, ev_interior = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
	std::vector<hipEvent_t> ev_k;  // per-launch timing events
	hipStream_t down = nullptr;    // device-to-host stream of the pipelined crd_rhs_host (lazy)
	std::vector<hipEvent_t> ev_band;  // its per-band events (lazy)

	SlabDesc desc{};
	int stepper = CRD_STEPPER_AUTO;

	int halo = CRD_HALO_SELF;
	std::vector<crd_ctx *> group;  // LOCAL: all contexts of the run, by slab index
	ncclComm_t nccl = nullptr;

	std::string err;

	Planes planes(int k) const { return Planes{plane[k][0], plane[k][1]}; }
	void *row_ptr(void *base, int64_t j) const { return static_cast<char *>(base) + (size_t)(j + kGhost) * (size_t)nx * real_size; }
};

namespace {

int fail(crd_ctx *c, int code, const std::string &msg)
{
	if (c) c->err = msg; else g_create_error = msg;
	return code;
}

#define HIP_TRY(ctx, expr)                                                                                      \
	do {                                                                                                        \
		hipError_t e_ = (expr);                                                                                 \
		if (e_ != hipSuccess) return fail((ctx), CRD_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
	} while (0)

#define NCCL_TRY(ctx, expr)                                                                                        \
	do {                                                                                                           \
		ncclResult_t r_ = (expr);                                                                                  \
		if (r_ != ncclSuccess) return fail((ctx), CRD_ERCCL, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
	} while (0)

int set_device(crd_ctx *c)
{
	HIP_TRY(c, hipSetDevice(c->device));
	return CRD_OK;
}

// Upload a host double table converted to the device precision.
int upload_table(crd_ctx *c, const std::vector<double> &src, void **dst)
{
	const size_t n = src.size();
	HIP_TRY(c, hipMalloc(dst, std::max<size_t>(n, 1) * c->real_size));
	if (c->p.precision == CRD_PRECISION_F64) {
		HIP_TRY(c, hipMemcpy(*dst, src.data(), n * sizeof(double), hipMemcpyHostToDevice));
	} else {
		std::vector<float> f(n);
		for (size_t i = 0; i < n; i++) f[i] = (float)src[i];
		HIP_TRY(c, hipMemcpy(*dst, f.data(), n * sizeof(float), hipMemcpyHostToDevice));
	}
	return CRD_OK;
}

int ensure_staging(crd_ctx *c, size_t bytes)
{
	if (c->stage_bytes >= bytes) return CRD_OK;
	if (c->stage_in) (void)hipFree(c->stage_in);
	if (c->stage_out) (void)hipFree(c->stage_out);
	c->stage_in = c->stage_out = nullptr;
	c->stage_bytes = 0;
	HIP_TRY(c, hipMalloc(&c->stage_in, bytes));
	HIP_TRY(c, hipMalloc(&c->stage_out, bytes));
	c->stage_bytes = bytes;
	return CRD_OK;
}

bool absorbing(const crd_ctx *c, double t_stage) { return t_stage < c->p.t_boundary; }  // strict <, src/FHNmodel_torus.cpp:643

int resolve_stepper(const crd_ctx *c)
{
	if (c->stepper == CRD_STEPPER_STAGED) return CRD_STEPPER_STAGED;
	const bool can_fuse = fused_step_supported(c->p.precision, c->desc);
	if (c->stepper == CRD_STEPPER_FUSED) return can_fuse ? CRD_STEPPER_FUSED : -1;
	return can_fuse ? CRD_STEPPER_FUSED : CRD_STEPPER_STAGED;
}

// ---- halo transports ------------------------------------------------------------------------------------------
// Fill the ghost rows [-depth, 0) and [nyl, nyl+depth) of `which` planes of every context from its ring neighbours.
// Enqueued on each context's comm stream; the caller orders it against the compute streams with events.

int exchange_rccl(crd_ctx *c, Planes pl, int depth, bool with_v)
{
	const size_t count = (size_t)depth * (size_t)c->nx;
	const ncclDataType_t dt = c->p.precision == CRD_PRECISION_F64 ? ncclDouble : ncclFloat;
	void *fields[2] = {pl.u, pl.v};
	crd_halo_op ops[4];
	if (crd_halo_plan(c->slab, c->n_slabs, c->nyl, depth, ops) != CRD_OK) return fail(c, CRD_EINVAL, "bad halo plan");
	NCCL_TRY(c, g_rccl.GroupStart());
	for (int f = 0; f < (with_v ? 2 : 1); f++)
		for (const crd_halo_op &op : ops) {
			void *rows = c->row_ptr(fields[f], op.row_begin);
			if (op.is_send) NCCL_TRY(c, g_rccl.Send(rows, count, dt, op.peer, c->nccl, c->comm));
			else NCCL_TRY(c, g_rccl.Recv(rows, count, dt, op.peer, c->nccl, c->comm));
		}
	NCCL_TRY(c, g_rccl.GroupEnd());
	return CRD_OK;
}

// LOCAL: pull the rows from the neighbours' planes with device-to-device copies (peer copies across devices).
int exchange_local_pull(crd_ctx *c, int plane_index, int depth, bool with_v)
{
	crd_ctx *prev = c->group[(size_t)((c->slab + c->n_slabs - 1) % c->n_slabs)];
	crd_ctx *next = c->group[(size_t)((c->slab + 1) % c->n_slabs)];
	const size_t bytes = (size_t)depth * (size_t)c->nx * c->real_size;
	for (int f = 0; f < (with_v ? 2 : 1); f++) {
		void *mine = c->plane[plane_index][f];
		HIP_TRY(c, hipMemcpyPeerAsync(c->row_ptr(mine, -depth), c->device, prev->row_ptr(prev->plane[plane_index][f], prev->nyl - depth), prev->device,
		                              bytes, c->comm));
		HIP_TRY(c, hipMemcpyPeerAsync(c->row_ptr(mine, c->nyl), c->device, next->row_ptr(next->plane[plane_index][f], 0), next->device, bytes, c->comm));
	}
	return CRD_OK;
}

// ---- steppers -------------------------------------------------------------------------------------------------

struct StagePlan {
	int in, out;
	double c;  // stage time = t + c dt
};
const StagePlan kStages[4] = {{crd_ctx::Y, crd_ctx::SA, 0.0}, {crd_ctx::SA, crd_ctx::SB, 0.5}, {crd_ctx::SB, crd_ctx::SA, 0.5}, {crd_ctx::SA, crd_ctx::Y, 1.0}};

StageCall make_stage_call(const crd_ctx *c, int stage, double t, double dt)
{
	const StagePlan &sp = kStages[stage - 1];
	StageCall call{};
	call.stage = stage;
	call.dt = dt;
	call.absorb = absorbing(c, t + sp.c * dt) ? 1 : 0;
	call.yin = c->planes(sp.in);
	call.y0 = c->planes(crd_ctx::Y);
	call.acc = c->planes(crd_ctx::ACC);
	call.yout = c->planes(sp.out);
	return call;
}

// Single slab: four launches per step, phi wrap inside the kernel.
int staged_step_self(crd_ctx *c, double t, double dt, hipEvent_t *k_begin, hipEvent_t *k_end)
{
	for (int stage = 1; stage <= 4; stage++) {
		const StageCall call = make_stage_call(c, stage, t, dt);
		if (stage == 2 && k_begin) HIP_TRY(c, hipEventRecord(*k_begin, c->compute));
		HIP_TRY(c, launch_stage(c->p.precision, c->desc, call, 0, c->nyl, c->compute));
		if (stage == 2 && k_end) HIP_TRY(c, hipEventRecord(*k_end, c->compute));
	}
	return CRD_OK;
}

FusedCall make_fused_call(const crd_ctx *c, double t, double dt, int src, int dst)
{
	FusedCall call{};
	call.dt = dt;
	const double cs[4] = {0.0, 0.5, 0.5, 1.0};
	for (int k = 0; k < 4; k++) call.absorb[k] = absorbing(c, t + cs[k] * dt) ? 1 : 0;
	call.y0 = c->planes(src);
	call.yout = c->planes(dst);
	return call;
}

int fused_step_self(crd_ctx *c, double t, double dt, int src, int dst, hipEvent_t *k_begin, hipEvent_t *k_end)
{
	const FusedCall call = make_fused_call(c, t, dt, src, dst);
	if (k_begin) HIP_TRY(c, hipEventRecord(*k_begin, c->compute));
	HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, 0, c->nyl, 0, 0, c->compute));
	if (k_end) HIP_TRY(c, hipEventRecord(*k_end, c->compute));
	return CRD_OK;
}

// Several slabs (LOCAL group driven by one thread, or this rank's slab under RCCL).  Per stage:
//   compute: [wait halo(in)] boundary rows 0 and nyl-1 -> record edges(out) -> interior rows
//   comm:    wait edges(out) -> exchange one ghost row of out.u -> record halo(out)
// so the exchange of stage s+1's input overlaps stage s's interior sweep.  The step's first input (Y) has its
// halo exchanged at the end of the previous step's stage 4 (or by prime_halo before the first step).
int exchange_stage_input(crd_ctx *const *cs, int n, int plane_index, int depth, bool with_v)
{
	// comm streams wait for the producers of the edge rows
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (int rc = set_device(c)) return rc;
		if (c->halo == CRD_HALO_LOCAL) {
			// pulling from neighbours: their edge rows must be complete too
			crd_ctx *prev = c->group[(size_t)((c->slab + c->n_slabs - 1) % c->n_slabs)];
			crd_ctx *next = c->group[(size_t)((c->slab + 1) % c->n_slabs)];
			HIP_TRY(c, hipStreamWaitEvent(c->comm, prev->ev_edges, 0));
			HIP_TRY(c, hipStreamWaitEvent(c->comm, next->ev_edges, 0));
		}
		HIP_TRY(c, hipStreamWaitEvent(c->comm, c->ev_edges, 0));
	}
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (int rc = set_device(c)) return rc;
		int rc = (c->halo == CRD_HALO_RCCL) ? exchange_rccl(c, c->planes(plane_index), depth, with_v) : exchange_local_pull(c, plane_index, depth, with_v);
		if (rc) return rc;
		HIP_TRY(c, hipEventRecord(c->ev_halo, c->comm));
	}
	return CRD_OK;
}

int prime_halo(crd_ctx *const *cs, int n, int plane_index, int depth, bool with_v)
{
	for (int k = 0; k < n; k++) {
		if (int rc = set_device(cs[k])) return rc;
		HIP_TRY(cs[k], hipEventRecord(cs[k]->ev_edges, cs[k]->compute));
		HIP_TRY(cs[k], hipEventRecord(cs[k]->ev_interior, cs[k]->compute));
	}
	return exchange_stage_input(cs, n, plane_index, depth, with_v);
}

int staged_step_multi(crd_ctx *const *cs, int n, double t, double dt, bool timed_step)
{
	for (int stage = 1; stage <= 4; stage++) {
		const StagePlan &sp = kStages[stage - 1];
		for (int k = 0; k < n; k++) {
			crd_ctx *c = cs[k];
			if (int rc = set_device(c)) return rc;
			const StageCall call = make_stage_call(c, stage, t, dt);
			HIP_TRY(c, hipStreamWaitEvent(c->compute, c->ev_halo, 0));
			HIP_TRY(c, launch_stage(c->p.precision, c->desc, call, 0, 1, c->compute));
			HIP_TRY(c, launch_stage(c->p.precision, c->desc, call, c->nyl - 1, c->nyl, c->compute));
			HIP_TRY(c, hipEventRecord(c->ev_edges, c->compute));
		}
		// start moving the edge rows of `out` while the interiors run
		if (int rc = exchange_stage_input(cs, n, sp.out, 1, false)) return rc;
		for (int k = 0; k < n; k++) {
			crd_ctx *c = cs[k];
			if (int rc = set_device(c)) return rc;
			const StageCall call = make_stage_call(c, stage, t, dt);
			const bool timed = timed_step && stage == 2 && !c->ev_k.empty();
			if (timed) HIP_TRY(c, hipEventRecord(c->ev_k[0], c->compute));
			HIP_TRY(c, launch_stage(c->p.precision, c->desc, call, 1, c->nyl - 1, c->compute));
			if (timed) HIP_TRY(c, hipEventRecord(c->ev_k[1], c->compute));
		}
	}
	return CRD_OK;
}

// Fused stepper on several slabs: ONE exchange every kExchangeEvery steps, kGhost = 4 * kExchangeEvery ghost rows of both
// fields.  Step q of a cycle (q = 0 right after an exchange) produces rows [-e, nyl + e) with e = 4 (kExchangeEvery-1-q):
// the still-valid part of the ghost region is recomputed redundantly (same kernel, same inputs, so bit-identical to what
// the owning slab computes) instead of being communicated.  Only the last step of a cycle (e = 0) is split:
//   band:    [wait previous step] edge bands [0, B) and [nyl-B, nyl) in one launch -> record edges
//   comm:    wait edges -> exchange kGhost rows of u and v with the ring neighbours -> record halo
//   compute: interior [B, nyl-B) (needs neither ghost rows nor the bands)         -> record interior
// so the exchange overlaps an interior sweep, and the first step of the next cycle waits for edges + halo.  Per step that
// is 1.25 launches and a quarter of an RCCL group on the host, against 3 launches + 1 group for a per-step exchange.
constexpr int kFusedBand = 32;
static_assert(kFusedBand >= kGhost, "the edge bands must contain every row the exchange sends");

// The edge-band stream exists only in contexts that step a multi-slab run with the fused stepper; it is created on first use.
int ensure_band_stream(crd_ctx *c)
{
	if (c->band || !c->bands_on_own_stream) return CRD_OK;
	if (!c->streams->band) {
		int lo = 0, hi = 0;  // the band launch is tiny and on the critical path of the exchange: give it priority over the interior sweep
		HIP_TRY(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
		HIP_TRY(c, hipStreamCreateWithPriority(&c->streams->band, hipStreamNonBlocking, hi));
	}
	c->band = c->streams->band;
	return CRD_OK;
}

int fused_step_multi(crd_ctx *const *cs, int n, double t, double dt, int src, int dst, int q, bool timed_step)
{
	const int ext = kStepHalo * (kExchangeEvery - 1 - q);
	if (q < kExchangeEvery - 1) {
		for (int k = 0; k < n; k++) {
			crd_ctx *c = cs[k];
			if (int rc = set_device(c)) return rc;
			const FusedCall call = make_fused_call(c, t, dt, src, dst);
			const bool timed = timed_step && !c->ev_k.empty();
			if (q > 0) {
				if (timed) HIP_TRY(c, hipEventRecord(c->ev_k[0], c->compute));
				HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, -ext, c->nyl + ext, 0, 0, c->compute));
				if (timed) HIP_TRY(c, hipEventRecord(c->ev_k[1], c->compute));
				HIP_TRY(c, hipEventRecord(c->ev_interior, c->compute));
				continue;
			}
			// First step after an exchange.  Output rows [kStepHalo, nyl - kStepHalo) read owned rows only, so they are launched
			// as soon as the bands of the previous step are in: the exchange gets this sweep as extra time to land.
			const bool split = c->nyl >= 4 * kFusedBand;
			HIP_TRY(c, hipStreamWaitEvent(c->compute, c->ev_edges, 0));
			if (split) HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, kStepHalo, c->nyl - kStepHalo, 0, 0, c->compute));
			HIP_TRY(c, hipStreamWaitEvent(c->compute, c->ev_halo, 0));
			if (c->halo == CRD_HALO_LOCAL) {
				// LOCAL halos are PULLED by the neighbours from this context's planes: the next step that overwrites those
				// rows (q = 1) must not start before both neighbours have finished copying them
				HIP_TRY(c, hipStreamWaitEvent(c->compute, c->group[(size_t)((c->slab + c->n_slabs - 1) % c->n_slabs)]->ev_halo, 0));
				HIP_TRY(c, hipStreamWaitEvent(c->compute, c->group[(size_t)((c->slab + 1) % c->n_slabs)]->ev_halo, 0));
			}
			// the rows that read ghost rows: [-ext, kStepHalo) and [nyl - kStepHalo, nyl + ext) in one launch
			if (split) HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, -ext, kStepHalo, c->nyl - kStepHalo, c->nyl + ext, c->compute));
			else HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, -ext, c->nyl + ext, 0, 0, c->compute));
			HIP_TRY(c, hipEventRecord(c->ev_interior, c->compute));
		}
		return CRD_OK;
	}
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (int rc = set_device(c)) return rc;
		const FusedCall call = make_fused_call(c, t, dt, src, dst);
		const bool split = c->nyl >= 4 * kFusedBand;
		if (int rc = ensure_band_stream(c)) return rc;
		hipStream_t bs = c->bands_on_own_stream ? c->band : c->compute;
		if (kExchangeEvery == 1) {  // per-step exchange: this step's inputs were produced by the previous split step
			HIP_TRY(c, hipStreamWaitEvent(c->compute, c->ev_edges, 0));
			HIP_TRY(c, hipStreamWaitEvent(bs, c->ev_halo, 0));
		}
		HIP_TRY(c, hipStreamWaitEvent(bs, c->ev_interior, 0));  // previous step done: its output is read, its input plane is overwritten
		if (split) HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, 0, kFusedBand, c->nyl - kFusedBand, c->nyl, bs));
		else HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, 0, c->nyl, 0, 0, bs));
		HIP_TRY(c, hipEventRecord(c->ev_edges, bs));
	}
	if (int rc = exchange_stage_input(cs, n, dst, kGhost, true)) return rc;
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (int rc = set_device(c)) return rc;
		if (c->nyl >= 4 * kFusedBand) {
			const FusedCall call = make_fused_call(c, t, dt, src, dst);
			HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, kFusedBand, c->nyl - kFusedBand, 0, 0, c->compute));
		}
		HIP_TRY(c, hipEventRecord(c->ev_interior, c->compute));
	}
	return CRD_OK;
}

constexpr int kMaxTimedLaunches = 64;
// Step of an exchange cycle whose single full-slab launch is the one timed in a multi-slab fused run (step 0 is split in two).
constexpr int kTimedCycleStep = (kExchangeEvery > 2) ? 1 : 0;

int ensure_timing_events(crd_ctx *c)
{
	while ((int)c->ev_k.size() < 2 * kMaxTimedLaunches) {
		hipEvent_t e;
		HIP_TRY(c, hipEventCreate(&e));
		c->ev_k.push_back(e);
	}
	return CRD_OK;
}

// The stepping loop shared by crd_step_rk4 / crd_step_rk4_timed / the group call.
int run_steps(crd_ctx *const *cs, int n, double t0, double dt, int64_t nsteps, int *timed_launches)
{
	crd_ctx *lead = cs[0];
	if (nsteps < 0 || !(dt > 0.0) || !std::isfinite(t0)) return fail(lead, CRD_EINVAL, "bad t0 / dt / nsteps");
	const int stepper = resolve_stepper(lead);
	if (stepper < 0) return fail(lead, CRD_EINVAL, "fused stepper not available for this configuration");
	for (int k = 0; k < n; k++)
		if (resolve_stepper(cs[k]) != stepper) return fail(lead, CRD_EINVAL, "contexts of one run disagree on the stepper");
	int timed = 0;
	const bool single = (lead->halo == CRD_HALO_SELF);
	if (single) {
		crd_ctx *c = lead;
		if (int rc = set_device(c)) return rc;
		int cur = crd_ctx::Y;
		for (int64_t s = 0; s < nsteps; s++) {
			const double t = t0 + (double)s * dt;
			hipEvent_t *kb = nullptr, *ke = nullptr;
			if (timed_launches && timed < kMaxTimedLaunches && (s * kMaxTimedLaunches / std::max<int64_t>(nsteps, 1)) >= timed) {
				kb = &c->ev_k[(size_t)(2 * timed)];
				ke = &c->ev_k[(size_t)(2 * timed + 1)];
				timed++;
			}
			int rc;
			if (stepper == CRD_STEPPER_STAGED) {
				rc = staged_step_self(c, t, dt, kb, ke);
			} else {
				const int dst = (cur == crd_ctx::Y) ? crd_ctx::SA : crd_ctx::Y;
				rc = fused_step_self(c, t, dt, cur, dst, kb, ke);
				cur = dst;
			}
			if (rc) return rc;
		}
		if (cur != crd_ctx::Y) {  // odd number of fused steps: the result sits in SA; swap the plane pointers
			std::swap(c->plane[crd_ctx::Y][0], c->plane[crd_ctx::SA][0]);
			std::swap(c->plane[crd_ctx::Y][1], c->plane[crd_ctx::SA][1]);
		}
	} else {
		const bool fused = (stepper == CRD_STEPPER_FUSED);
		if (nsteps > 0)
			if (int rc = prime_halo(cs, n, crd_ctx::Y, fused ? kGhost : 1, fused)) return rc;
		int cur = crd_ctx::Y;
		for (int64_t s = 0; s < nsteps; s++) {
			// time one launch of the dominant kernel mid-run (fused: the first one-launch step of a cycle)
			const bool timed_step = timed_launches && !timed &&
			                        (fused ? (s % kExchangeEvery == kTimedCycleStep && (s >= nsteps / 2 || s + kExchangeEvery >= nsteps)) : s >= nsteps / 2);
			const double t = t0 + (double)s * dt;
			if (fused) {
				const int dst = (cur == crd_ctx::Y) ? crd_ctx::SA : crd_ctx::Y;
				if (int rc = fused_step_multi(cs, n, t, dt, cur, dst, (int)(s % kExchangeEvery), timed_step)) return rc;
				cur = dst;
			} else if (int rc = staged_step_multi(cs, n, t, dt, timed_step)) {
				return rc;
			}
			if (timed_step) timed = 1;
		}
		if (cur != crd_ctx::Y)
			for (int k = 0; k < n; k++) {
				std::swap(cs[k]->plane[crd_ctx::Y][0], cs[k]->plane[crd_ctx::SA][0]);
				std::swap(cs[k]->plane[crd_ctx::Y][1], cs[k]->plane[crd_ctx::SA][1]);
			}
		// leave the compute stream of every context ordered behind its last band launch and exchange
		for (int k = 0; k < n; k++) {
			if (int rc = set_device(cs[k])) return rc;
			HIP_TRY(cs[k], hipStreamWaitEvent(cs[k]->compute, cs[k]->ev_edges, 0));
			HIP_TRY(cs[k], hipStreamWaitEvent(cs[k]->compute, cs[k]->ev_halo, 0));
		}
	}
	if (timed_launches) *timed_launches = timed;
	return CRD_OK;
}

int check_group(crd_ctx *const *ctxs, int n)
{
	if (!ctxs || n < 1 || !ctxs[0]) return CRD_EINVAL;
	for (int k = 0; k < n; k++) {
		if (!ctxs[k]) return CRD_EINVAL;
		if (ctxs[k]->n_slabs != n || ctxs[k]->slab != k) return fail(ctxs[0], CRD_EINVAL, "group must hold every slab of the run in slab order");
	}
	return CRD_OK;
}

}  // namespace

extern "C" {

const char *crd_last_error(const crd_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int crd_create(const crd_params *p, int slab, int n_slabs, int device, crd_ctx **out)
{
	if (!out) return CRD_EINVAL;
	*out = nullptr;
	std::string why;
	if (!p || !validate_params(*p, &why)) return fail(nullptr, CRD_EINVAL, p ? why : "null params");
	crd_ctx *c = new (std::nothrow) crd_ctx;
	if (!c) return fail(nullptr, CRD_ENOMEM, "host allocation failed");
	auto bail = [&](int rc) {
		g_create_error = c->err;
		crd_destroy(c);
		return rc;
	};
	c->p = *p;
	int rc = crd_grid_from_params(p, &c->g);
	if (rc) return bail(fail(c, rc, "bad geometry"));
	rc = crd_slab_extents(c->g.ny, slab, n_slabs, &c->js, &c->je);
	if (rc) return bail(fail(c, rc, "bad slab index / count for this ny"));
	c->slab = slab;
	c->n_slabs = n_slabs;
	c->device = device;
	c->nx = (int)c->g.nx;
	if (c->je - c->js + 1 > INT32_MAX / 2) return bail(fail(c, CRD_EINVAL, "slab too tall"));
	c->nyl = (int)(c->je - c->js + 1);
	if (c->nyl < 2 * kStepHalo) return bail(fail(c, CRD_EINVAL, "every slab needs at least 8 rows"));
	if (n_slabs > 1 && c->nyl < kGhost) return bail(fail(c, CRD_EINVAL, "every slab of a multi-slab run needs at least " + std::to_string(kGhost) + " rows (one exchange moves that many ghost rows)"));
	c->real_size = p->precision == CRD_PRECISION_F64 ? 8 : 4;
	c->plane_bytes = (size_t)(c->nyl + 2 * kGhost) * (size_t)c->nx * c->real_size;
	c->halo = n_slabs == 1 ? CRD_HALO_SELF : -1;  // multi-slab contexts must be wired before use
	if (const char *e = std::getenv("CRD_BAND_STREAM")) c->bands_on_own_stream = std::atoi(e) != 0;

	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return bail(fail(c, CRD_EHIP, "no HIP device available (libcrd has no CPU fallback)"));
	if (device < 0 || device >= ndev) return bail(fail(c, CRD_EINVAL, "device ordinal out of range"));
	if ((rc = set_device(c))) return bail(rc);

#define CREATE_TRY(expr)                                                                                \
	do {                                                                                                \
		hipError_t e_ = (expr);                                                                         \
		if (e_ != hipSuccess) return bail(fail(c, e_ == hipErrorOutOfMemory ? CRD_ENOMEM : CRD_EHIP,    \
		                                       std::string(#expr) + ": " + hipGetErrorString(e_)));     \
	} while (0)

	c->streams = std::make_shared<StreamSet>();
	c->streams->device = device;
	// Measured on a world-size-1 RCCL ring (tools/ring_overhead.py): leaving CUs out of the compute stream's CU mask to keep
	// room for the exchange kernel costs 7-20% of the sweep, and a high-priority exchange stream changes nothing, so both
	// streams are plain: the exchange kernel gets its CUs as the interior sweep drains and finishes under the next sweep.
	CREATE_TRY(hipStreamCreateWithFlags(&c->streams->compute, hipStreamNonBlocking));
	CREATE_TRY(hipStreamCreateWithFlags(&c->streams->comm, hipStreamNonBlocking));
	c->compute = c->streams->compute;
	c->comm = c->streams->comm;
	CREATE_TRY(hipEventCreateWithFlags(&c->ev_edges, hipEventDisableTiming));
	CREATE_TRY(hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
	CREATE_TRY(hipEventCreateWithFlags(&c->ev_interior, hipEventDisableTiming));
	CREATE_TRY(hipEventCreate(&c->ev_t0));
	CREATE_TRY(hipEventCreate(&c->ev_t1));
	for (int k = 0; k < crd_ctx::NPLANES; k++)
		for (int f = 0; f < 2; f++) {
			CREATE_TRY(hipMalloc(&c->plane[k][f], c->plane_bytes));
			CREATE_TRY(hipMemsetAsync(c->plane[k][f], 0, c->plane_bytes, c->compute));
		}
	CREATE_TRY(hipMalloc(&c->ghost_lo, (size_t)c->nx * c->real_size));
	CREATE_TRY(hipMalloc(&c->ghost_hi, (size_t)c->nx * c->real_size));
	CREATE_TRY(hipMalloc(&c->edge_lo, (size_t)c->nx * c->real_size));
	CREATE_TRY(hipMalloc(&c->edge_hi, (size_t)c->nx * c->real_size));
	CREATE_TRY(hipMalloc((void **)&c->scalar_dev, sizeof(double)));
#undef CREATE_TRY

	Coefficients co;
	build_coefficients(c->p, c->g, &co);
	std::vector<double> brow;
	build_beta_rows(c->p, c->g, c->js - kGhost, c->je + 1 + kGhost, &brow);
	if ((rc = upload_table(c, co.cA, &c->cA)) || (rc = upload_table(c, co.cP, &c->cP)) || (rc = upload_table(c, brow, &c->brow))) return bail(rc);

	SlabDesc &d = c->desc;
	d.cA = c->cA;
	d.cP = c->cP;
	d.brow = c->brow;
	d.cX = co.cX;
	d.ka4 = std::pow(kGbKa, 4.0);  // pow(KA, p), src/GoldbeterModel_torus.cpp:695
	d.nx = c->nx;
	d.nyl = c->nyl;
	d.wrap = (n_slabs == 1);
	d.has_row0 = (c->js == 0);
	d.has_rowN = (c->je == c->g.ny - 1);
	d.js = (int)c->js;
	d.ny = (int)c->g.ny;
	d.model = p->model;
	d.just_diffusion = (p->model == CRD_MODEL_GOLDBETER && p->just_diffusion != 0);
	if (hipStreamSynchronize(c->compute) != hipSuccess) return bail(fail(c, CRD_EHIP, "device initialisation failed"));
	*out = c;
	return CRD_OK;
}

void crd_destroy(crd_ctx *c)
{
	if (!c) return;
	if (c->streams) (void)hipSetDevice(c->device);  // a context refused at creation (e.g. bad device ordinal) owns nothing on any device
	(void)hipGetLastError();
	if (c->compute) (void)hipStreamSynchronize(c->compute);
	if (c->comm) (void)hipStreamSynchronize(c->comm);
	if (c->band) (void)hipStreamSynchronize(c->band);
	if (c->nccl && g_rccl.handle) (void)g_rccl.CommDestroy(c->nccl);
	for (auto &pl : c->plane)
		for (void *q : pl)
			if (q) (void)hipFree(q);
	for (void *q : {c->cA, c->cP, c->brow, c->stage_in, c->stage_out, c->ghost_lo, c->ghost_hi, c->edge_lo, c->edge_hi, (void *)c->scalar_dev,
	                (void *)c->err_partials})
		if (q) (void)hipFree(q);
	for (hipEvent_t e : c->ev_k) (void)hipEventDestroy(e);
	for (hipEvent_t e : c->ev_band) (void)hipEventDestroy(e);
	if (c->down) {
		(void)hipStreamSynchronize(c->down);
		(void)hipStreamDestroy(c->down);
	}
	for (hipEvent_t e : {c->ev_edges, c->ev_halo, c->ev_interior, c->ev_t0, c->ev_t1})
		if (e) (void)hipEventDestroy(e);
	c->streams.reset();  // destroys the streams unless another context of a LOCAL group on this device still uses them
	// detach from a LOCAL group so the survivors do not dereference this context
	for (crd_ctx *o : c->group)
		if (o && o != c) {
			o->group.clear();
			o->halo = -1;
		}
	delete c;
}

int crd_get_grid(const crd_ctx *c, crd_grid *g)
{
	if (!c || !g) return CRD_EINVAL;
	*g = c->g;
	return CRD_OK;
}

int crd_get_slab(const crd_ctx *c, int64_t *js, int64_t *je)
{
	if (!c || !js || !je) return CRD_EINVAL;
	*js = c->js;
	*je = c->je;
	return CRD_OK;
}

int crd_comm_attach_local(crd_ctx *const *ctxs, int n)
{
	if (int rc = check_group(ctxs, n)) return rc;
	if (n == 1) return CRD_OK;
	for (int k = 0; k < n; k++) {
		crd_ctx *c = ctxs[k];
		if (c->p.precision != ctxs[0]->p.precision || c->nx != ctxs[0]->nx) return fail(ctxs[0], CRD_EINVAL, "contexts of one run must share nx and precision");
		c->group.assign(ctxs, ctxs + n);
		c->halo = CRD_HALO_LOCAL;
		for (int j = 0; j < k; j++)
			if (ctxs[j]->device == c->device) {  // slabs on one device run on one set of streams
				if (hipSetDevice(c->device) != hipSuccess) return fail(ctxs[0], CRD_EHIP, "hipSetDevice failed");
				for (hipStream_t s : {c->compute, c->comm, c->band})
					if (s) (void)hipStreamSynchronize(s);
				c->streams = ctxs[j]->streams;
				c->compute = c->streams->compute;
				c->comm = c->streams->comm;
				c->band = nullptr;  // picked up from the shared set on first use
				break;
			}
	}
	// enable peer access between distinct devices (ignore "already enabled")
	for (int a = 0; a < n; a++)
		for (int b = 0; b < n; b++)
			if (ctxs[a]->device != ctxs[b]->device) {
				if (hipSetDevice(ctxs[a]->device) != hipSuccess) return fail(ctxs[0], CRD_EHIP, "hipSetDevice failed");
				hipError_t e = hipDeviceEnablePeerAccess(ctxs[b]->device, 0);
				if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(ctxs[0], CRD_EHIP, "hipDeviceEnablePeerAccess failed");
				(void)hipGetLastError();
			}
	return CRD_OK;
}

int crd_comm_unique_id(void *id128)
{
	if (!id128) return CRD_EINVAL;
	static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
	ncclUniqueId id;
	if (!g_rccl.load()) return fail(nullptr, CRD_ERCCL, g_rccl.error);
	if (g_rccl.GetUniqueId(&id) != ncclSuccess) return fail(nullptr, CRD_ERCCL, "ncclGetUniqueId failed");
	std::memcpy(id128, &id, sizeof id);
	return CRD_OK;
}

int crd_comm_init_rccl(crd_ctx *c, const void *id128)
{
	if (!c || !id128) return CRD_EINVAL;
	if (c->nccl) return fail(c, CRD_ESTATE, "RCCL communicator already initialised");
	if (!g_rccl.load()) return fail(c, CRD_ERCCL, g_rccl.error);
	if (int rc = set_device(c)) return rc;
	ncclUniqueId id;
	std::memcpy(&id, id128, sizeof id);
	NCCL_TRY(c, g_rccl.CommInitRank(&c->nccl, c->n_slabs, id, c->slab));
	c->halo = CRD_HALO_RCCL;
	c->desc.wrap = 0;  // ghosts come from the ring, also when the ring is this rank alone
	return CRD_OK;
}

int crd_state_upload(crd_ctx *c, const void *y, int host_is_f64)
{
	if (!c || !y) return CRD_EINVAL;
	if (c->p.precision == CRD_PRECISION_F64 && !host_is_f64) return fail(c, CRD_EINVAL, "an fp64 context takes double host buffers");
	if (int rc = set_device(c)) return rc;
	const size_t bytes = 2 * (size_t)c->nx * (size_t)c->nyl * (host_is_f64 ? 8 : 4);
	if (int rc = ensure_staging(c, 2 * (size_t)c->nx * (size_t)c->nyl * 8)) return rc;
	HIP_TRY(c, hipMemcpyAsync(c->stage_in, y, bytes, hipMemcpyHostToDevice, c->compute));
	HIP_TRY(c, launch_aos_to_planes(c->p.precision, host_is_f64, c->stage_in, c->planes(crd_ctx::Y), c->nx, c->nyl, c->compute));
	HIP_TRY(c, hipStreamSynchronize(c->compute));
	return CRD_OK;
}

int crd_state_download(crd_ctx *c, void *y, int host_is_f64)
{
	if (!c || !y) return CRD_EINVAL;
	if (c->p.precision == CRD_PRECISION_F64 && !host_is_f64) return fail(c, CRD_EINVAL, "an fp64 context fills double host buffers");
	if (int rc = set_device(c)) return rc;
	const size_t bytes = 2 * (size_t)c->nx * (size_t)c->nyl * (host_is_f64 ? 8 : 4);
	if (int rc = ensure_staging(c, 2 * (size_t)c->nx * (size_t)c->nyl * 8)) return rc;
	HIP_TRY(c, launch_planes_to_aos(c->p.precision, host_is_f64, c->planes(crd_ctx::Y), c->stage_out, c->nx, c->nyl, c->compute));
	HIP_TRY(c, hipMemcpyAsync(y, c->stage_out, bytes, hipMemcpyDeviceToHost, c->compute));
	HIP_TRY(c, hipStreamSynchronize(c->compute));
	return CRD_OK;
}

int crd_rhs_device(crd_ctx *c, double t, const void *y, void *ydot)
{
	if (!c || !y || !ydot) return CRD_EINVAL;
	if (c->halo < 0) return fail(c, CRD_ESTATE, "multi-slab context is not wired (crd_comm_attach_local / crd_comm_init_rccl)");
	if (c->halo == CRD_HALO_LOCAL) return fail(c, CRD_ESTATE, "LOCAL groups evaluate f through crd_group_rhs_device");
	if (int rc = set_device(c)) return rc;
	if (c->halo == CRD_HALO_RCCL) {
		// Exchange(): pack var0 of the first / last row, swap with the ring neighbours (src/FHNmodel_torus.cpp:775-950;
		// only var0 is ever read from the strips, :548,:570).
		const ncclDataType_t dt = c->p.precision == CRD_PRECISION_F64 ? ncclDouble : ncclFloat;
		crd_halo_op ops[4];
		if (crd_halo_plan(c->slab, c->n_slabs, c->nyl, 1, ops) != CRD_OK) return fail(c, CRD_EINVAL, "bad halo plan");
		HIP_TRY(c, launch_aos_row_extract(c->p.precision, y, c->edge_lo, c->nx, 0, c->compute));
		HIP_TRY(c, launch_aos_row_extract(c->p.precision, y, c->edge_hi, c->nx, c->nyl - 1, c->compute));
		NCCL_TRY(c, g_rccl.GroupStart());
		for (const crd_halo_op &op : ops) {
			// the plan's rows map onto the packed strips: row 0 / nyl-1 are edge_lo / edge_hi, row -1 / nyl the ghost strips
			if (op.is_send) NCCL_TRY(c, g_rccl.Send(op.row_begin == 0 ? c->edge_lo : c->edge_hi, (size_t)c->nx, dt, op.peer, c->nccl, c->compute));
			else NCCL_TRY(c, g_rccl.Recv(op.row_begin < 0 ? c->ghost_lo : c->ghost_hi, (size_t)c->nx, dt, op.peer, c->nccl, c->compute));
		}
		NCCL_TRY(c, g_rccl.GroupEnd());
	}
	HIP_TRY(c, launch_rhs_aos(c->p.precision, c->desc, absorbing(c, t) ? 1 : 0, y, ydot, c->ghost_lo, c->ghost_hi, 0, c->nyl, c->compute));
	return CRD_OK;
}

namespace {

bool is_pinned_host(const void *p)
{
	hipPointerAttribute_t attr;
	const hipError_t e = hipPointerGetAttributes(&attr, p);
	(void)hipGetLastError();  // ordinary (pageable) host memory is reported as an error
	return e == hipSuccess && attr.type == hipMemoryTypeHost;
}

// crd_rhs_host on pinned vectors, single slab: bands of rows flow host -> device -> kernel -> host with the three legs
// of neighbouring bands overlapping (upload on the comm stream, kernels on the compute stream, download on its own stream).
int rhs_host_pipelined(crd_ctx *c, double t, const void *y, void *ydot)
{
	const size_t row_bytes = 2 * (size_t)c->nx * c->real_size;
	int band = (int)std::max<size_t>(64, ((size_t)32 << 20) / row_bytes);  // ~32 MiB per band
	if (band > c->nyl) band = c->nyl;
	const int nb = (c->nyl + band - 1) / band;
	if (!c->down) HIP_TRY(c, hipStreamCreateWithFlags(&c->down, hipStreamNonBlocking));
	while ((int)c->ev_band.size() < 2 * nb + 2) {
		hipEvent_t e;
		HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
		c->ev_band.push_back(e);
	}
	const char *src = static_cast<const char *>(y);
	char *dst = static_cast<char *>(ydot);
	char *din = static_cast<char *>(c->stage_in), *dout = static_cast<char *>(c->stage_out);
	const int absorb = absorbing(c, t) ? 1 : 0;
	// every stream starts behind whatever the context did before
	HIP_TRY(c, hipEventRecord(c->ev_band[(size_t)(2 * nb)], c->compute));
	HIP_TRY(c, hipStreamWaitEvent(c->comm, c->ev_band[(size_t)(2 * nb)], 0));
	HIP_TRY(c, hipStreamWaitEvent(c->down, c->ev_band[(size_t)(2 * nb)], 0));
	// the periodic neighbour of row 0 is the last row: it goes first
	HIP_TRY(c, hipMemcpyAsync(din + (size_t)(c->nyl - 1) * row_bytes, src + (size_t)(c->nyl - 1) * row_bytes, row_bytes, hipMemcpyHostToDevice, c->comm));
	for (int k = 0; k < nb; k++) {
		const int r0 = k * band, r1 = std::min(c->nyl, r0 + band);
		HIP_TRY(c, hipMemcpyAsync(din + (size_t)r0 * row_bytes, src + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes, hipMemcpyHostToDevice, c->comm));
		HIP_TRY(c, hipEventRecord(c->ev_band[(size_t)k], c->comm));
		if (k >= 1) {  // band k-1 has its upper neighbour row now
			const int q0 = (k - 1) * band, q1 = q0 + band;
			HIP_TRY(c, hipStreamWaitEvent(c->compute, c->ev_band[(size_t)k], 0));
			HIP_TRY(c, launch_rhs_aos(c->p.precision, c->desc, absorb, din, dout, c->ghost_lo, c->ghost_hi, q0, q1, c->compute));
			HIP_TRY(c, hipEventRecord(c->ev_band[(size_t)(nb + k - 1)], c->compute));
			HIP_TRY(c, hipStreamWaitEvent(c->down, c->ev_band[(size_t)(nb + k - 1)], 0));
			HIP_TRY(c, hipMemcpyAsync(dst + (size_t)q0 * row_bytes, dout + (size_t)q0 * row_bytes, (size_t)(q1 - q0) * row_bytes, hipMemcpyDeviceToHost, c->down));
		}
	}
	{  // last band: its upper neighbour is row 0, uploaded with band 0
		const int q0 = (nb - 1) * band, q1 = c->nyl;
		HIP_TRY(c, hipStreamWaitEvent(c->compute, c->ev_band[(size_t)(nb - 1)], 0));
		HIP_TRY(c, launch_rhs_aos(c->p.precision, c->desc, absorb, din, dout, c->ghost_lo, c->ghost_hi, q0, q1, c->compute));
		HIP_TRY(c, hipEventRecord(c->ev_band[(size_t)(2 * nb - 1)], c->compute));
		HIP_TRY(c, hipStreamWaitEvent(c->down, c->ev_band[(size_t)(2 * nb - 1)], 0));
		HIP_TRY(c, hipMemcpyAsync(dst + (size_t)q0 * row_bytes, dout + (size_t)q0 * row_bytes, (size_t)(q1 - q0) * row_bytes, hipMemcpyDeviceToHost, c->down));
	}
	HIP_TRY(c, hipStreamSynchronize(c->down));
	HIP_TRY(c, hipStreamSynchronize(c->compute));
	return CRD_OK;
}

}  // namespace

void *crd_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
		(void)hipGetLastError();
		return nullptr;
	}
	return p;
}

void crd_host_free(void *p)
{
	if (p) (void)hipHostFree(p);
}

int crd_rhs_host(crd_ctx *c, double t, const void *y, void *ydot)
{
	if (!c || !y || !ydot) return CRD_EINVAL;
	if (int rc = set_device(c)) return rc;
	const size_t bytes = 2 * (size_t)c->nx * (size_t)c->nyl * c->real_size;
	if (int rc = ensure_staging(c, 2 * (size_t)c->nx * (size_t)c->nyl * 8)) return rc;
	if (c->halo == CRD_HALO_SELF && c->nyl >= 128 && is_pinned_host(y) && is_pinned_host(ydot)) return rhs_host_pipelined(c, t, y, ydot);
	HIP_TRY(c, hipMemcpyAsync(c->stage_in, y, bytes, hipMemcpyHostToDevice, c->compute));
	if (int rc = crd_rhs_device(c, t, c->stage_in, c->stage_out)) return rc;
	HIP_TRY(c, hipMemcpyAsync(ydot, c->stage_out, bytes, hipMemcpyDeviceToHost, c->compute));
	HIP_TRY(c, hipStreamSynchronize(c->compute));
	return CRD_OK;
}

// LOCAL groups: f on every slab of the run, halos pulled from the neighbours' vectors.
int crd_group_rhs_device(crd_ctx *const *ctxs, int n, double t, const void *const *y, void *const *ydot)
{
	if (int rc = check_group(ctxs, n)) return rc;
	if (!y || !ydot) return CRD_EINVAL;
	if (n == 1) return crd_rhs_device(ctxs[0], t, y[0], ydot[0]);
	for (int k = 0; k < n; k++) {
		crd_ctx *c = ctxs[k];
		if (c->halo != CRD_HALO_LOCAL) return fail(ctxs[0], CRD_ESTATE, "group is not attached");
		if (int rc = set_device(c)) return rc;
		HIP_TRY(c, launch_aos_row_extract(c->p.precision, y[k], c->edge_lo, c->nx, 0, c->compute));
		HIP_TRY(c, launch_aos_row_extract(c->p.precision, y[k], c->edge_hi, c->nx, c->nyl - 1, c->compute));
		HIP_TRY(c, hipEventRecord(c->ev_edges, c->compute));
	}
	for (int k = 0; k < n; k++) {
		crd_ctx *c = ctxs[k];
		crd_ctx *prev = ctxs[(k + n - 1) % n], *next = ctxs[(k + 1) % n];
		if (int rc = set_device(c)) return rc;
		const size_t bytes = (size_t)c->nx * c->real_size;
		HIP_TRY(c, hipStreamWaitEvent(c->compute, prev->ev_edges, 0));
		HIP_TRY(c, hipStreamWaitEvent(c->compute, next->ev_edges, 0));
		HIP_TRY(c, hipMemcpyPeerAsync(c->ghost_lo, c->device, prev->edge_hi, prev->device, bytes, c->compute));
		HIP_TRY(c, hipMemcpyPeerAsync(c->ghost_hi, c->device, next->edge_lo, next->device, bytes, c->compute));
		HIP_TRY(c, launch_rhs_aos(c->p.precision, c->desc, absorbing(c, t) ? 1 : 0, y[k], ydot[k], c->ghost_lo, c->ghost_hi, 0, c->nyl, c->compute));
	}
	// the edge buffers may be repacked by the next call only after every neighbour has copied them
	for (int k = 0; k < n; k++) {
		if (int rc = set_device(ctxs[k])) return rc;
		HIP_TRY(ctxs[k], hipStreamSynchronize(ctxs[k]->compute));
	}
	return CRD_OK;
}

int crd_group_rhs_host(crd_ctx *const *ctxs, int n, double t, const void *const *y, void *const *ydot)
{
	if (int rc = check_group(ctxs, n)) return rc;
	if (!y || !ydot) return CRD_EINVAL;
	std::vector<const void *> din((size_t)n);
	std::vector<void *> dout((size_t)n);
	for (int k = 0; k < n; k++) {
		crd_ctx *c = ctxs[k];
		if (!y[k] || !ydot[k]) return CRD_EINVAL;
		if (int rc = set_device(c)) return rc;
		const size_t bytes = 2 * (size_t)c->nx * (size_t)c->nyl * c->real_size;
		if (int rc = ensure_staging(c, 2 * (size_t)c->nx * (size_t)c->nyl * 8)) return rc;
		HIP_TRY(c, hipMemcpyAsync(c->stage_in, y[k], bytes, hipMemcpyHostToDevice, c->compute));
		din[(size_t)k] = c->stage_in;
		dout[(size_t)k] = c->stage_out;
	}
	if (int rc = crd_group_rhs_device(ctxs, n, t, din.data(), dout.data())) return rc;
	for (int k = 0; k < n; k++) {
		crd_ctx *c = ctxs[k];
		if (int rc = set_device(c)) return rc;
		const size_t bytes = 2 * (size_t)c->nx * (size_t)c->nyl * c->real_size;
		HIP_TRY(c, hipMemcpyAsync(ydot[k], c->stage_out, bytes, hipMemcpyDeviceToHost, c->compute));
		HIP_TRY(c, hipStreamSynchronize(c->compute));
	}
	return CRD_OK;
}

int crd_set_stepper(crd_ctx *c, int stepper)
{
	if (!c) return CRD_EINVAL;
	if (stepper != CRD_STEPPER_AUTO && stepper != CRD_STEPPER_STAGED && stepper != CRD_STEPPER_FUSED) return fail(c, CRD_EINVAL, "unknown stepper");
	if (stepper == CRD_STEPPER_FUSED && !fused_step_supported(c->p.precision, c->desc)) return fail(c, CRD_EINVAL, "fused stepper not available for this configuration");
	c->stepper = stepper;
	return CRD_OK;
}

int crd_step_rk4(crd_ctx *c, double t0, double dt, int64_t nsteps)
{
	if (!c) return CRD_EINVAL;
	if (c->halo < 0) return fail(c, CRD_ESTATE, "multi-slab context is not wired (crd_comm_attach_local / crd_comm_init_rccl)");
	if (c->halo == CRD_HALO_LOCAL) return fail(c, CRD_ESTATE, "LOCAL groups step through crd_group_step_rk4");
	crd_ctx *one[1] = {c};
	return run_steps(one, 1, t0, dt, nsteps, nullptr);
}

int crd_group_step_rk4(crd_ctx *const *ctxs, int n, double t0, double dt, int64_t nsteps)
{
	if (int rc = check_group(ctxs, n)) return rc;
	if (n > 1)
		for (int k = 0; k < n; k++)
			if (ctxs[k]->halo != CRD_HALO_LOCAL) return fail(ctxs[0], CRD_ESTATE, "group is not attached");
	return run_steps(ctxs, n, t0, dt, nsteps, nullptr);
}

int crd_adaptive_defaults(crd_adaptive_options *o)
{
	if (!o) return CRD_EINVAL;
	o->rtol = 1.e-5;   // src/FHNmodel_torus.cpp:197
	o->atol = 1.e-10;  // :198
	o->h0 = 0.0;
	o->safety = 0.96;
	o->bias = 1.5;
	o->growth = 20.0;
	o->shrink = 0.1;
	o->max_steps = 200000;  // :372
	return CRD_OK;
}

// Error-controlled integration of all slabs of a run (n = 1: a single-slab or RCCL context; n > 1: a LOCAL group).
static int integrate_adaptive_impl(crd_ctx *const *cs, int n, double t0, double tout, const crd_adaptive_options *opt_in, crd_adaptive_stats *stats)
{
	crd_ctx *lead = cs[0];
	crd_adaptive_options o;
	crd_adaptive_defaults(&o);
	if (opt_in) o = *opt_in;
	if (!(o.rtol >= 0.0) || !(o.atol >= 0.0) || !(o.rtol + o.atol > 0.0) || !(o.safety > 0.0) || !(o.bias > 0.0) || !(o.growth >= 1.0) ||
	    !(o.shrink > 0.0 && o.shrink < 1.0) || o.max_steps < 1 || !(o.h0 >= 0.0) || !std::isfinite(t0) || !std::isfinite(tout) || tout < t0)
		return fail(lead, CRD_EINVAL, "bad adaptive options / time interval");
	const bool multi = lead->halo != CRD_HALO_SELF;
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (!fused_step_supported(c->p.precision, c->desc)) return fail(lead, CRD_EINVAL, "slab too small for the fused step kernel");
		if (int rc = set_device(c)) return rc;
		if (!c->err_partials) {
			c->err_capacity = fused_max_items(c->desc);
			HIP_TRY(c, hipMalloc((void **)&c->err_partials, sizeof(double) * (size_t)c->err_capacity));
		}
	}
	crd_adaptive_stats st{};
	double t = t0;
	double h = o.h0 > 0.0 ? o.h0 : 0.8 * crd_stable_dt(&lead->p);
	const double n_components = 2.0 * (double)lead->g.nx * (double)lead->g.ny;  // WRMS norm over the whole grid
	constexpr int kEmbedHalo = kStepHalo + 1;                                     // the fifth stage reads one more row
	int cur = crd_ctx::Y;
	bool after_reject = false;
	int rc = CRD_OK;
	while (t < tout) {
		if (st.accepted + st.rejected >= o.max_steps) {
			rc = fail(lead, CRD_ESTATE, "adaptive integration: max_steps attempts taken before reaching tout");
			break;
		}
		double hh = h;
		bool clipped = false;
		if (t + hh >= tout || tout - (t + hh) < 1e-12 * std::fabs(tout)) {  // land on tout exactly; absorb a sliver of a last step
			hh = tout - t;
			clipped = true;
		}
		if (!(hh > 1e-14 * std::fmax(std::fabs(t), 1e-300)) && !(t == 0.0 && hh > 0.0)) {
			rc = fail(lead, CRD_ESTATE, "adaptive integration: step size underflow");
			break;
		}
		const int dst = (cur == crd_ctx::Y) ? crd_ctx::SA : crd_ctx::Y;
		if (multi)  // every attempt starts from freshly exchanged ghost rows of the current state (no overlap: the host waits for the norm anyway)
			if ((rc = prime_halo(cs, n, cur, kEmbedHalo, true))) break;
		double sum = 0.0;
		for (int k = 0; k < n && rc == CRD_OK; k++) {
			crd_ctx *c = cs[k];
			if ((rc = set_device(c))) break;
			FusedCall call = make_fused_call(c, t, hh, cur, dst);
			call.embed = 1;
			call.rtol = o.rtol;
			call.atol = o.atol;
			call.err_partials = c->err_partials;
			call.err_capacity = c->err_capacity;
			call.err_sum = c->scalar_dev;
			if (multi) HIP_TRY(c, hipStreamWaitEvent(c->compute, c->ev_halo, 0));
			HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, 0, c->nyl, 0, 0, c->compute));
			if (c->halo == CRD_HALO_RCCL)  // every rank gets the same bits, hence takes the same decision
				NCCL_TRY(c, g_rccl.AllReduce(c->scalar_dev, c->scalar_dev, 1, ncclDouble, ncclSum, c->nccl, c->compute));
		}
		for (int k = 0; k < n && rc == CRD_OK; k++) {  // LOCAL groups: add the slabs' sums in slab order
			crd_ctx *c = cs[k];
			if ((rc = set_device(c))) break;
			double part = 0.0;
			HIP_TRY(c, hipMemcpyAsync(&part, c->scalar_dev, sizeof(double), hipMemcpyDeviceToHost, c->compute));
			HIP_TRY(c, hipStreamSynchronize(c->compute));
			sum += part;
		}
		if (rc != CRD_OK) break;
		const double err = o.bias * std::sqrt(sum / n_components);
		st.err_last = err;
		double eta;
		if (!(err == err) || std::isinf(err)) eta = o.shrink;  // NaN / inf: the step blew up
		else if (err <= 0.0) eta = o.growth;
		else eta = std::fmin(o.growth, std::fmax(o.shrink, o.safety * std::pow(err, -0.25)));
		if (err <= 1.0) {
			t = clipped ? tout : t + hh;
			cur = dst;
			st.accepted++;
			if (after_reject) eta = std::fmin(eta, 1.0);  // no growth right after a rejection
			after_reject = false;
			if (!clipped || st.accepted == 1) {
				st.h_last = hh;
				st.h_min = (st.h_min == 0.0) ? hh : std::fmin(st.h_min, hh);
				st.h_max = std::fmax(st.h_max, hh);
			}
			if (!clipped) h = hh * eta;
			else h = std::fmax(h, hh * eta);  // a step shortened to hit tout says nothing against the step it replaced
		} else {
			st.rejected++;
			after_reject = true;
			h = hh * std::fmin(eta, 0.9);
		}
	}
	if (cur != crd_ctx::Y)
		for (int k = 0; k < n; k++) {
			std::swap(cs[k]->plane[crd_ctx::Y][0], cs[k]->plane[crd_ctx::SA][0]);
			std::swap(cs[k]->plane[crd_ctx::Y][1], cs[k]->plane[crd_ctx::SA][1]);
		}
	st.t = t;
	st.h_next = h;
	if (stats) *stats = st;
	return rc;
}

int crd_integrate_adaptive(crd_ctx *c, double t0, double tout, const crd_adaptive_options *opt, crd_adaptive_stats *stats)
{
	if (!c) return CRD_EINVAL;
	if (c->halo < 0) return fail(c, CRD_ESTATE, "multi-slab context is not wired (crd_comm_attach_local / crd_comm_init_rccl)");
	if (c->halo == CRD_HALO_LOCAL) return fail(c, CRD_ESTATE, "LOCAL groups integrate through crd_group_integrate_adaptive");
	crd_ctx *one[1] = {c};
	return integrate_adaptive_impl(one, 1, t0, tout, opt, stats);
}

int crd_group_integrate_adaptive(crd_ctx *const *ctxs, int n, double t0, double tout, const crd_adaptive_options *opt, crd_adaptive_stats *stats)
{
	if (int rc = check_group(ctxs, n)) return rc;
	if (n > 1)
		for (int k = 0; k < n; k++)
			if (ctxs[k]->halo != CRD_HALO_LOCAL) return fail(ctxs[0], CRD_ESTATE, "group is not attached");
	return integrate_adaptive_impl(ctxs, n, t0, tout, opt, stats);
}

int crd_synchronize(crd_ctx *c)
{
	if (!c) return CRD_EINVAL;
	if (int rc = set_device(c)) return rc;
	HIP_TRY(c, hipStreamSynchronize(c->comm));
	if (c->band) HIP_TRY(c, hipStreamSynchronize(c->band));
	HIP_TRY(c, hipStreamSynchronize(c->compute));
	return CRD_OK;
}

int crd_step_rk4_timed(crd_ctx *c, double t0, double dt, int64_t nsteps, double *ms_total, double *kernel_ms, int *launches_per_step)
{
	if (!c) return CRD_EINVAL;
	if (c->halo < 0 || c->halo == CRD_HALO_LOCAL) return fail(c, CRD_ESTATE, "timed stepping needs a single-slab or RCCL context");
	if (int rc = set_device(c)) return rc;
	if (int rc = ensure_timing_events(c)) return rc;
	crd_ctx *one[1] = {c};
	int timed = 0;
	HIP_TRY(c, hipEventRecord(c->ev_t0, c->compute));
	if (int rc = run_steps(one, 1, t0, dt, nsteps, &timed)) return rc;
	HIP_TRY(c, hipEventRecord(c->ev_t1, c->compute));
	HIP_TRY(c, hipEventSynchronize(c->ev_t1));
	float ms = 0.f;
	HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
	if (ms_total) *ms_total = ms;
	double sum = 0.0;
	for (int k = 0; k < timed; k++) {
		float m = 0.f;
		HIP_TRY(c, hipEventElapsedTime(&m, c->ev_k[(size_t)(2 * k)], c->ev_k[(size_t)(2 * k + 1)]));
		sum += m;
	}
	if (kernel_ms) *kernel_ms = timed ? sum / timed : 0.0;
	if (launches_per_step) *launches_per_step = (resolve_stepper(c) == CRD_STEPPER_FUSED) ? 1 : 2;
	return CRD_OK;
}

int crd_dominant_kernel_rows(const crd_ctx *c, int64_t *rows)
{
	if (!c || !rows) return CRD_EINVAL;
	const int stepper = resolve_stepper(c);
	if (c->halo == CRD_HALO_SELF) *rows = c->nyl;
	else if (stepper == CRD_STEPPER_FUSED) *rows = c->nyl + 2 * kStepHalo * (kExchangeEvery - 1 - kTimedCycleStep);  // the timed step of an exchange cycle
	else *rows = c->nyl - 2;
	return CRD_OK;
}

const char *crd_dominant_kernel_name(const crd_ctx *c)
{
	if (!c) return "";
	return resolve_stepper(c) == CRD_STEPPER_FUSED ? fused_kernel_name(c->p.precision, c->p.model) : stage_kernel_name(c->p.precision, c->p.model);
}

int crd_state_max_abs(crd_ctx *c, double *out)
{
	if (!c || !out) return CRD_EINVAL;
	if (int rc = set_device(c)) return rc;
	HIP_TRY(c, launch_max_abs(c->p.precision, c->plane[crd_ctx::Y][0], c->nx, c->nyl, c->scalar_dev, c->compute));
	HIP_TRY(c, hipMemcpyAsync(out, c->scalar_dev, sizeof(double), hipMemcpyDeviceToHost, c->compute));
	HIP_TRY(c, hipStreamSynchronize(c->compute));
	return CRD_OK;
}

}  // extern "C"
