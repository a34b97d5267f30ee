// crd_ctx.h -- the device context behind the C ABI and the helpers its translation units share (not part of the ABI):
//   crd_context.cpp   lifecycle, communicator wiring, state transfer, the f() entry points
//   crd_halo.cpp      RCCL binding and the halo transports
//   crd_steppers.cpp  fixed-step RK4 drivers (staged / fused, single slab / deep-halo multi-slab) and the adaptive integrator
#pragma once

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only; the library is bound with dlopen at first use

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "crd_internal.h"
#include "crd_kernels.h"

namespace crd {

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
	decltype(&ncclCommCount) CommCount = nullptr;
	decltype(&ncclCommUserRank) CommUserRank = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	std::string error;
	std::string library;  // crd_comm_set_rccl_library: bind this file instead of librccl.so.1

	bool load();  // crd_halo.cpp
};
extern RcclApi g_rccl;

}  // namespace crd

// Rendezvous of the issuing threads of a LOCAL group (crd_group_step_rk4 with one host thread per device).  A stream may
// only be told to wait for an event AFTER the thread that records it has enqueued the record; the threads meet here between
// "everybody has recorded" and "everybody waits".  A thread that fails leaves the group and raises `abort`; the others see it
// at their next rendezvous and give up too, so nobody waits for a thread that is gone.
struct GroupBarrier {
	std::mutex m;
	std::condition_variable cv;
	int members = 0, arrived = 0;
	unsigned generation = 0;
	std::atomic<int> abort{0};

	bool wait()  // false: another thread has failed
	{
		std::unique_lock<std::mutex> lk(m);
		const unsigned g = generation;
		if (++arrived >= members) {
			arrived = 0;
			generation++;
			cv.notify_all();
		} else {
			cv.wait(lk, [&] { return generation != g; });
		}
		return abort.load() == 0;
	}
	void leave_failed()
	{
		std::unique_lock<std::mutex> lk(m);
		abort.store(1);
		members--;
		if (members > 0 && arrived >= members) {
			arrived = 0;
			generation++;
		}
		cv.notify_all();
	}
};

// The two streams of a context: sweeps, halo exchange.  Contexts of a LOCAL group that share a device share one set
// (crd_comm_attach_local), so a device never carries more than two of this library's streams however many slabs it hosts.
struct StreamSet {
	int device = 0;
	hipStream_t compute = nullptr, comm = nullptr;
	~StreamSet()
	{
		(void)hipSetDevice(device);
		for (hipStream_t s : {compute, comm})
			if (s) {
				(void)hipStreamSynchronize(s);
				(void)hipStreamDestroy(s);
			}
	}
};

struct crd_ctx {
	crd_params p{};
	crd_grid g{};
	int slab = 0, n_slabs = 1, device = 0;  // slab = the block's rank c0 d1 + c1, n_slabs = d0 d1
	int c0 = 0, d0 = 1, c1 = 0, d1 = 1;     // block (c0, c1) of a d0 x d1 decomposition; phi-slabs: d0 = 1, c1 = slab
	int64_t is = 0, ie = 0, js = 0, je = 0;
	int nx = 0, nyl = 0;                    // the block's own width and height (nx = nxl; g.nx is the grid's)
	size_t real_size = 8;
	size_t plane_bytes = 0;

	// State planes: Y (current), SA / SB (stage ping-pong), ACC; [k][0] = var0, [k][1] = var1.
	// OUT exists only in contexts that have produced dense output (lazy): the third state plane of crd_integrate_adaptive.
	enum { Y = 0, SA = 1, SB = 2, ACC = 3, OUT = 4, NPLANES = 5 };
	void *plane[NPLANES][2] = {};
	std::vector<void *> plane_allocs;  // what hipMalloc returned for them (a plane starts `plane_skew` x its index into its allocation)
	size_t plane_skew = 0;             // bytes; see alloc_plane (crd_context.cpp)
	void *cE = nullptr, *cWn = nullptr, *cP = nullptr, *brow = nullptr;
	void *stage_in = nullptr, *stage_out = nullptr;  // AoS staging for the *_host entry points (lazy)
	size_t stage_bytes = 0;
	void *ghost_lo = nullptr, *ghost_hi = nullptr;   // var0 of rows -1 / nyl for the AoS RHS (multi-slab)
	void *edge_lo = nullptr, *edge_hi = nullptr;     // var0 of rows 0 / nyl-1 packed from an AoS vector
	// theta-blocks (d0 > 1): var0 of the columns beside the block (ghost) and of its own first / last column (edge), nyl reals each,
	// [0] = west, [1] = east -- per state plane for the staged stepper, one pair for the AoS RHS
	void *gcol[NPLANES][2] = {}, *ecol[NPLANES][2] = {};
	void *ghost_col[2] = {}, *edge_col[2] = {};
	crd_ctx *neighbour(int dtheta, int dphi) const  // of a LOCAL group, periodic in both directions
	{
		return group[(size_t)(((c0 + dtheta + d0) % d0) * d1 + (c1 + dphi + d1) % d1)];
	}
	double *scalar_dev = nullptr;
	double *scalar_host = nullptr;  // page-locked, device-visible host memory
	// Where a reduction kernel leaves its scalar for the host: straight in host memory (no copy, one synchronisation) unless the
	// scalar still has to be reduced over the ranks of an RCCL run on the device.
	double *scalar_sink() const { return (scalar_host && halo != CRD_HALO_RCCL) ? scalar_host : scalar_dev; }
	double *err_partials = nullptr;  // adaptive stepping: per-item error sums, two slots of err_capacity (lazy)
	int err_capacity = 0;
	hipEvent_t ev_attempt[2] = {nullptr, nullptr}, ev_norm[2] = {nullptr, nullptr};  // ... per slot: kernel done (compute), scalar on the host

	std::shared_ptr<StreamSet> streams;                               // owner of the handles below
	hipStream_t compute = nullptr, comm = nullptr;
	hipEvent_t ev_edges = nullptr, ev_halo = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
	std::vector<hipEvent_t> ev_k;  // per-launch timing events
	hipStream_t down = nullptr;    // device-to-host stream of the pipelined crd_rhs_host (lazy)
	std::vector<hipEvent_t> ev_band;  // its per-band events (lazy)

	// Dense output of the adaptive integrator (ARK_NORMAL): after a call that overshot tout, plane Y holds the interpolant at
	// t_out, SA the integrator's own state y_{n+1} at t_np1, OUT y_n at t_n, SB / ACC f(t_n, y_n) / f(t_{n+1}, y_{n+1}).
	struct {
		bool pending = false;
		double t_out = 0.0, t_n = 0.0, t_np1 = 0.0;
	} dense;

	// Memory of the ARKode-style controller (CRD_ADAPT_ARKODE) between calls, as ARKodeMem keeps it between ARKode() calls: valid
	// while the resident state is the one the integrator left (`live`), dropped by whatever replaces that state.
	struct {
		bool live = false;
		int64_t nst = 0;         // steps taken since the state was new
		double tn = 0.0;         // time the integrator has reached
		double h = 0.0, hprime = 0.0, eta = 1.0, etamax = 0.0;
		double ehist[3] = {1.0, 1.0, 1.0};
	} ark;

	// Multi-slab fused stepping: position in the deep-halo exchange cycle the resident state is at (steps taken since the ghost
	// rows were last exchanged, 0 = just exchanged), or -1 when the ghost rows cannot be trusted (new state, another stepper,
	// an error): crd_step_rk4 then starts with an exchange, otherwise it carries on where the previous call stopped.
	// Halo slack: sweeps of owned-only rows the compute stream launches after an exchange before it waits for the halo -- 1: the
	// cycle's first step is split (rounds 1-2); 2: the first two are, for a third sweep of cover (crd_set_halo_slack).
	int halo_slack = 1;
	int exchange_every = crd::kDefaultExchangeEvery;  // E: fused steps per deep-halo exchange (crd_set_exchange_period), the same on every slab of a run
	int group_threads = 0;  // lead context of a LOCAL group: issuing threads of crd_group_step_rk4 (0 = one per device)
	bool ghost_deferred = false;  // step 0 of the cycle has launched its owned-only rows; its ghost readers wait for step 1
	double deferred_t = 0.0;      // that launch's time
	int deferred_nsub = 1;        // ... and how many steps it takes (1, or 2 under a two-steps-per-launch plan)
	int cycle_pos = -1;
	// time and step of the last fixed step a stepping call issued: crd_get_launch_geometry reports the kernel a step THERE runs (with the
	// absorbing rows on, t < tBoundary, a one-step launch runs the instantiation with the selects)
	double last_step_t = 0.0, last_step_dt = 0.0;
	int xchg_plane = 0;    // the state plane the exchange in progress sends this context's rows from (LOCAL neighbours pull from it)
	int timed_rows = 0;    // rows of the multi-slab fused launch crd_step_rk4_timed last put its events around
	int timed_steps = 0;   // steps per launch of the launches the last timed call put its events around (all alike: 1, or 2)
	int cycle_start = -1;  // the decision for the call in progress, taken for ALL slabs of the run before any thread issues (run_steps)
	// RCCL runs: the ranks AGREE on the cycle position at the start of every stepping call (one 2-value ncclAllReduce(min) of
	// (pos, -pos) on the comm stream, overlapped with the call's first step where that step involves no exchange): a rank whose
	// state is new -- an upload on that rank only, a failed call -- makes every rank start afresh with an exchange instead of
	// leaving the ring's send / receive sequences unpaired.
	double *agree_dev = nullptr;   // 2 doubles on the device
	double *agree_host = nullptr;  // 4 doubles, page-locked: [0..1] what this rank contributes, [2..3] the reduced values
	hipEvent_t ev_agree = nullptr;
	int64_t agreement_restarts = 0;  // calls that started afresh because the ranks disagreed (diagnostics)

	// Diagnostics of a timed multi-slab fused run (crd_set_diagnostics): per exchange of the timed call, event pairs around the
	// compute stream's wait for the halo (how long the sweep stood still for the exchange) and around the exchange itself on the
	// comm stream.  Off by default: the records sit between sweeps that otherwise run back to back.
	bool diagnostics = false, diag_active = false;  // requested / being collected by the crd_step_rk4_timed call in progress
	std::vector<hipEvent_t> ev_diag;  // 4 per exchange: wait begin / end (compute), exchange begin / end (comm)
	int diag_waits = 0, diag_exchanges = 0;
	crd_step_timing timing{};  // filled by crd_step_rk4_timed

	crd::SlabDesc desc{};
	int stepper = CRD_STEPPER_AUTO;
	crd::FusedPlan plan{}, plan_embed{}, plan_arkode{};  // launch plans of the one-launch step (plain / RK4(3) estimate / Zonneveld 5(3)4)

	int halo = CRD_HALO_SELF;
	std::vector<crd_ctx *> group;  // LOCAL: all contexts of the run, by slab index
	GroupBarrier *bar = nullptr;   // set while several host threads issue for the group (crd_group_step_rk4)
	ncclComm_t nccl = nullptr;

	std::string err;

	crd::Planes planes(int k) const { return crd::Planes{plane[k][0], plane[k][1]}; }
	void *row_ptr(void *base, int64_t j) const { return static_cast<char *>(base) + (size_t)(j + crd::kGhost) * (size_t)nx * real_size; }
};

namespace crd {

int fail(crd_ctx *c, int code, const std::string &msg);  // records msg (thread-local when c is null) and returns code

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
int set_device(crd_ctx *c);
int upload_table(crd_ctx *c, const std::vector<double> &src, void **dst);  // host doubles -> device precision
int ensure_staging(crd_ctx *c, size_t bytes);                              // AoS staging of the *_host entry points
int alloc_plane(crd_ctx *c, int k, int f);                                 // one field plane of state k, zeroed
inline bool absorbing(const crd_ctx *c, double t_stage) { return t_stage < c->p.t_boundary; }  // strict <, src/FHNmodel_torus.cpp:643
int resolve_stepper(const crd_ctx *c);  // CRD_STEPPER_STAGED / _FUSED, or -1 when the requested one is unavailable
int check_group(crd_ctx *const *ctxs, int n);

// crd_halo.cpp: fill ghost rows [-depth, 0) and [nyl, nyl + depth) of plane `plane_index` from the ring neighbours, on the comm streams
int exchange_stage_input(crd_ctx *const *cs, int n, int plane_index, int depth, bool with_v);
// ... 2-D blocks (LOCAL groups): one ghost row of var0 from the phi neighbours and one ghost column strip from the theta neighbours
int exchange_block_input(crd_ctx *const *cs, int n, int plane_index);
int prime_block_halo(crd_ctx *const *cs, int n, int plane_index);
int prime_halo(crd_ctx *const *cs, int n, int plane_index, int depth, bool with_v);

// crd_steppers.cpp
FusedCall make_fused_call(const crd_ctx *c, double t, double dt, int src, int dst);
int run_steps(crd_ctx *const *cs, int n, double t0, double dt, int64_t nsteps, int *timed_launches);
void decide_cycle_start(crd_ctx *const *all, int n_all);  // call once per stepping call, before run_steps (and before any issuing thread starts)

}  // namespace crd
