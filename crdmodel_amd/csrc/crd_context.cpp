// crd_context.cpp -- the device context behind the C ABI: lifecycle, communicator wiring, state transfer and the f() entry
// points.  Host code only (compiled by hipcc for the HIP runtime API); kernels live in crd_kernels.hip / crd_fused.hip, the
// halo transports in crd_halo.cpp, the steppers in crd_steppers.cpp.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "crd_ctx.h"
#include "crd_tuning.h"

namespace crd {

namespace {
thread_local std::string g_create_error;  // crd_last_error(NULL)
}

int fail(crd_ctx *c, int code, const std::string &msg)
{
	if (c) c->err = msg; else g_create_error = msg;
	return code;
}

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

// One field plane.  Planes are whole multiples of 64 KiB (a row of 8192 doubles) and of 512 KiB at the BASELINE sizes, so with
// back-to-back allocations element (j, i) of every plane a step touches -- two read, two written -- sits at the same offset
// modulo any power-of-two interleave of the memory channels.  plane_skew (bytes) staggers the planes inside
// their allocations, plane number x skew, to take that alignment away.
int alloc_plane(crd_ctx *c, int k, int f)
{
	const size_t index = (size_t)(2 * k + f), skew = c->plane_skew * index;
	void *base = nullptr;
	hipError_t e = hipMalloc(&base, c->plane_bytes + skew);
	if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? CRD_ENOMEM : CRD_EHIP, std::string("hipMalloc(plane): ") + hipGetErrorString(e));
	c->plane_allocs.push_back(base);
	c->plane[k][f] = static_cast<char *>(base) + skew;
	HIP_TRY(c, hipMemsetAsync(c->plane[k][f], 0, c->plane_bytes, c->compute));
	return CRD_OK;
}

int resolve_stepper(const crd_ctx *c)
{
	if (c->stepper == CRD_STEPPER_STAGED) return CRD_STEPPER_STAGED;
	if (c->d0 > 1) return c->stepper == CRD_STEPPER_FUSED ? -1 : CRD_STEPPER_STAGED;  // theta-blocks: the staged kernels only
	// The deep-halo exchange of a multi-slab run sends kGhost owned rows: shorter slabs step with the staged kernels -- and
	// since every slab of a run must take the same stepper, the SHORTEST slab of the run decides (each context, and each
	// rank of an RCCL run, works that out for itself from the slab formula).
	int64_t shortest = c->nyl;
	for (int k = 0; c->d1 > 1 && k < c->d1; k++) {
		int64_t js, je;
		if (crd_slab_extents(c->g.ny, k, c->d1, &js, &je) == CRD_OK) shortest = std::min<int64_t>(shortest, je - js + 1);
	}
	const bool can_fuse = fused_step_supported(c->p.precision, c->desc) && (c->halo == CRD_HALO_SELF || shortest >= kStepHalo * c->exchange_every);
	if (c->stepper == CRD_STEPPER_FUSED) return can_fuse ? CRD_STEPPER_FUSED : -1;
	return can_fuse ? CRD_STEPPER_FUSED : CRD_STEPPER_STAGED;
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

}  // namespace crd

using namespace crd;

extern "C" {

int crd_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) {
		(void)hipGetLastError();
		return 0;
	}
	return n;
}

const char *crd_last_error(const crd_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int crd_create(const crd_params *p, int slab, int n_slabs, int device, crd_ctx **out)
{
	return crd_create_block(p, 0, 1, slab, n_slabs, device, out);
}

int crd_get_block(const crd_ctx *c, int64_t *is, int64_t *ie, int64_t *js, int64_t *je)
{
	if (!c || !is || !ie || !js || !je) return CRD_EINVAL;
	*is = c->is;
	*ie = c->ie;
	*js = c->js;
	*je = c->je;
	return CRD_OK;
}

int crd_create_block(const crd_params *p, int c0, int d0, int c1, int d1, int device, crd_ctx **out)
{
	if (!out) return CRD_EINVAL;
	*out = nullptr;
	const int slab = c0 * d1 + c1, n_slabs = d0 * d1;
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
	rc = crd_block_extents(c->g.nx, c->g.ny, c0, d0, c1, d1, &c->is, &c->ie, &c->js, &c->je);
	if (rc) return bail(fail(c, rc, "bad block index / count for this grid"));
	c->slab = slab;
	c->n_slabs = n_slabs;
	c->c0 = c0;
	c->d0 = d0;
	c->c1 = c1;
	c->d1 = d1;
	c->device = device;
	c->nx = (int)(c->ie - c->is + 1);
	if (d0 > 1 && c->nx < 2) return bail(fail(c, CRD_EINVAL, "every theta-block needs at least 2 columns"));
	if (c->je - c->js + 1 > INT32_MAX / 2) return bail(fail(c, CRD_EINVAL, "slab too tall"));
	c->nyl = (int)(c->je - c->js + 1);
	if (c->nyl < 2 * kStepHalo) return bail(fail(c, CRD_EINVAL, "every slab needs at least 8 rows"));
	c->real_size = p->precision == CRD_PRECISION_F64 ? 8 : 4;
	c->plane_bytes = (size_t)(c->nyl + 2 * kGhost) * (size_t)c->nx * c->real_size;
	c->halo = n_slabs == 1 ? CRD_HALO_SELF : -1;  // multi-slab contexts must be wired before use
	// Exchange period of a multi-slab run (crd_set_exchange_period): 10 steps where every slab of the run is at least 256 rows tall, 8
	// otherwise -- a rule in the run's own numbers, the same on every slab.  (Through the self-ring on the round's final tree, E = 4 / 6 /
	// 8 / 10 / 12 / 16: the 8192 x 1024 share 39.4 / 40.0 / 37.6 / 37.5 / 38.1 / 38.3 us per step, the 8192 x 4096 one 131.1 / 122.4 /
	// 121.0 / 120.0 / 121.2 / 121.7; profiles/r05/ring_overhead_periods_final_tree.txt.  Mid-round, before the cycle's small launches
	// carried their own completion events, 8 cost 41.5 and 16 was the default.)  bench.py rehearses 8 / 10 / 16 on the run's own ring.
	if (n_slabs > 1 && d0 == 1 && c->g.ny / n_slabs >= 256) c->exchange_every = kTallSlabExchangeEvery;
	if (const char *e = std::getenv("CRD_AUTOTUNE"))  // 0 / 1 / 2, as crd_set_autotune
		c->plan.autotune = c->plan_embed.autotune = c->plan_arkode.autotune = std::atoi(e) <= 0 ? 0 : (std::atoi(e) >= 2 ? 2 : 1);
	// CRD_LAUNCH_PLAN=mode,mapping,columns,nt: what crd_set_launch_plan does, for a program one cannot change (crd_run under a profiler)
	if (const char *e = std::getenv("CRD_LAUNCH_PLAN")) {
		int m = -1, k = -1, cols = -1, nt = 0, st = 1;
		if (std::sscanf(e, "%d,%d,%d,%d,%d", &m, &k, &cols, &nt, &st) >= 3 && m >= 0 && m <= 2 && k >= 0 && k <= 2 && cols >= 1 && cols <= 2 && nt >= 0 && nt <= 1 && st >= 1 &&
		    st <= 3) {
			c->plan.tuned = c->plan.pinned = 1;
			c->plan.one_round = m;
			c->plan.remap = k;
			c->plan.cols = cols;
			c->plan.nt = nt;
			c->plan.steps = st;
			c->plan.rows = (int)(c->je - c->js + 1);
		}
	}

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
	CREATE_TRY(hipEventCreate(&c->ev_t0));
	CREATE_TRY(hipEventCreate(&c->ev_t1));
	// 16 KiB + 256 B (profiles/r03/plane_skew.txt: -3 % at 8192^2 fp64 on one box, -5 % Goldbeter 4096^2 on another, nowhere a loss beyond
	// run-to-run noise); rows stay 256-byte aligned
	c->plane_skew = (size_t)tuning::plane_skew(16640) & ~(size_t)255;
	for (int k = 0; k < crd_ctx::OUT; k++)  // (the OUT plane is allocated by the first dense-output call)
		for (int f = 0; f < 2; f++)
			if ((rc = alloc_plane(c, k, f))) return bail(rc);
	CREATE_TRY(hipMalloc(&c->ghost_lo, (size_t)c->nx * c->real_size));
	CREATE_TRY(hipMalloc(&c->ghost_hi, (size_t)c->nx * c->real_size));
	CREATE_TRY(hipMalloc(&c->edge_lo, (size_t)c->nx * c->real_size));
	CREATE_TRY(hipMalloc(&c->edge_hi, (size_t)c->nx * c->real_size));
	CREATE_TRY(hipMalloc((void **)&c->scalar_dev, 4 * sizeof(double)));  // [0] reductions, [2], [3] the adaptive integrator's attempt slots
	CREATE_TRY(hipHostMalloc((void **)&c->scalar_host, 8 * sizeof(double), hipHostMallocPortable | hipHostMallocMapped));
	if (d0 > 1) {  // theta-blocks: ghost / edge column strips of var0, per stage-input plane and one pair for the AoS RHS
		const size_t strip = (size_t)c->nyl * c->real_size;
		for (int side = 0; side < 2; side++) {
			for (int k = 0; k < crd_ctx::OUT; k++) {
				CREATE_TRY(hipMalloc(&c->gcol[k][side], strip));
				CREATE_TRY(hipMalloc(&c->ecol[k][side], strip));
				CREATE_TRY(hipMemsetAsync(c->gcol[k][side], 0, strip, c->compute));
			}
			CREATE_TRY(hipMalloc(&c->ghost_col[side], strip));
			CREATE_TRY(hipMalloc(&c->edge_col[side], strip));
		}
	}
#undef CREATE_TRY

	Coefficients co;
	build_coefficients(c->p, c->g, &co);
	std::vector<double> brow;
	build_beta_rows(c->p, c->g, c->js - kGhost, c->je + 1 + kGhost, &brow);
	if (p->model == CRD_MODEL_GOLDBETER)  // the kernels take the row-constant source term v0 + v1 b(j) of src/GoldbeterModel_torus.cpp:715 ready-made
		for (double &b : brow) b = std::fma(kGbV1, b, kGbV0);
	else  // FHN: EPSILON b(j), the addend of dv = fma(EPSILON, u, EPSILON b) (src/FHNmodel_torus.cpp:660)
		for (double &b : brow) b = kFhnEpsilon * b;
	if (d0 > 1) {  // the block's own columns of the per-column tables
		co.cE = std::vector<double>(co.cE.begin() + c->is, co.cE.begin() + c->ie + 1);
		co.cWn = std::vector<double>(co.cWn.begin() + c->is, co.cWn.begin() + c->ie + 1);
		co.cP = std::vector<double>(co.cP.begin() + c->is, co.cP.begin() + c->ie + 1);
	}
	if ((rc = upload_table(c, co.cE, &c->cE)) || (rc = upload_table(c, co.cWn, &c->cWn)) || (rc = upload_table(c, co.cP, &c->cP)) || (rc = upload_table(c, brow, &c->brow))) return bail(rc);

	SlabDesc &d = c->desc;
	d.cE = c->cE;
	d.cWn = c->cWn;
	d.cP = c->cP;
	d.brow = c->brow;
	d.ka4 = std::pow(kGbKa, 4.0);  // pow(KA, p), src/GoldbeterModel_torus.cpp:695
	d.nx = c->nx;
	d.nyl = c->nyl;
	d.wrap = (d1 == 1);    // phi wraps inside the block
	d.wrap_x = (d0 == 1);  // theta wraps inside the block
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
	if (c->nccl && g_rccl.handle) (void)g_rccl.CommDestroy(c->nccl);
	for (void *q : c->plane_allocs) (void)hipFree(q);
	if (c->scalar_host) (void)hipHostFree(c->scalar_host);
	for (void *q : {c->cE, c->cWn, c->cP, c->brow, c->stage_in, c->stage_out, c->ghost_lo, c->ghost_hi, c->edge_lo, c->edge_hi, (void *)c->scalar_dev,
	                (void *)c->err_partials, c->ghost_col[0], c->ghost_col[1], c->edge_col[0], c->edge_col[1]})
		if (q) (void)hipFree(q);
	for (auto &pl : c->gcol)
		for (void *q : pl)
			if (q) (void)hipFree(q);
	for (auto &pl : c->ecol)
		for (void *q : pl)
			if (q) (void)hipFree(q);
	for (hipEvent_t e : c->ev_k) (void)hipEventDestroy(e);
	for (hipEvent_t e : c->ev_diag) (void)hipEventDestroy(e);
	if (c->ev_agree) (void)hipEventDestroy(c->ev_agree);
	for (hipEvent_t e : {c->ev_attempt[0], c->ev_attempt[1], c->ev_norm[0], c->ev_norm[1]})
		if (e) (void)hipEventDestroy(e);
	if (c->agree_dev) (void)hipFree(c->agree_dev);
	if (c->agree_host) (void)hipHostFree(c->agree_host);
	for (hipEvent_t e : c->ev_band) (void)hipEventDestroy(e);
	if (c->down) {
		(void)hipStreamSynchronize(c->down);
		(void)hipStreamDestroy(c->down);
	}
	for (hipEvent_t e : {c->ev_edges, c->ev_halo, c->ev_t0, c->ev_t1})
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
		if (c->p.precision != ctxs[0]->p.precision || c->g.nx != ctxs[0]->g.nx || c->d0 != ctxs[0]->d0 || c->d1 != ctxs[0]->d1)
			return fail(ctxs[0], CRD_EINVAL, "contexts of one run must share the grid, the decomposition and the precision");
		c->group.assign(ctxs, ctxs + n);
		c->halo = CRD_HALO_LOCAL;
		for (int j = 0; j < k; j++)
			if (ctxs[j]->device == c->device) {  // slabs on one device run on one set of streams
				if (hipSetDevice(c->device) != hipSuccess) return fail(ctxs[0], CRD_EHIP, "hipSetDevice failed");
				for (hipStream_t s : {c->compute, c->comm})
					if (s) (void)hipStreamSynchronize(s);
				c->streams = ctxs[j]->streams;
				c->compute = c->streams->compute;
				c->comm = c->streams->comm;
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

int crd_comm_set_rccl_library(const char *path)
{
	if (g_rccl.handle) return fail(nullptr, CRD_ESTATE, "the RCCL entry points are already bound");
	g_rccl.library = path ? path : "";
	g_rccl.error.clear();
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
	if (c->d0 > 1) return fail(c, CRD_EINVAL, "the RCCL transport moves phi-slab halos: theta-blocks (d0 > 1) run as LOCAL groups");
	if (!g_rccl.load()) return fail(c, CRD_ERCCL, g_rccl.error);
	if (int rc = set_device(c)) return rc;
	ncclUniqueId id;
	std::memcpy(&id, id128, sizeof id);
	NCCL_TRY(c, g_rccl.CommInitRank(&c->nccl, c->n_slabs, id, c->slab));
	c->halo = CRD_HALO_RCCL;
	c->desc.wrap = 0;  // ghosts come from the ring, also when the ring is this rank alone
	return CRD_OK;
}

int crd_comm_info(const crd_ctx *c, int *halo, int *ranks, int *rank)
{
	if (!c) return CRD_EINVAL;
	if (halo) *halo = c->halo;
	int n = c->n_slabs, r = c->slab;
	if (c->halo == CRD_HALO_RCCL) {  // what the communicator itself says, not what this context was created with
		NCCL_TRY(const_cast<crd_ctx *>(c), g_rccl.CommCount(c->nccl, &n));
		NCCL_TRY(const_cast<crd_ctx *>(c), g_rccl.CommUserRank(c->nccl, &r));
	} else if (c->halo == CRD_HALO_SELF) {
		n = 1;
		r = 0;
	}
	if (ranks) *ranks = n;
	if (rank) *rank = r;
	return CRD_OK;
}

int crd_halo_exchange(crd_ctx *c, int depth)
{
	if (!c) return CRD_EINVAL;
	if (depth < 1 || depth > kGhost || depth > c->nyl) return fail(c, CRD_EINVAL, "halo depth out of range (1 .. 64, at most the slab's rows)");
	if (c->halo < 0) return fail(c, CRD_ESTATE, "multi-slab context is not wired (crd_comm_attach_local / crd_comm_init_rccl)");
	if (c->halo == CRD_HALO_LOCAL) return fail(c, CRD_ESTATE, "LOCAL groups exchange inside crd_group_step_rk4");
	if (int rc = set_device(c)) return rc;
	if (c->halo == CRD_HALO_SELF) {
		// a single slab wraps inside the kernels; fill the ghost rows with the periodic image anyway so that they can be read back
		const size_t bytes = (size_t)depth * (size_t)c->nx * c->real_size;
		for (int f = 0; f < 2; f++) {
			void *pl = c->plane[crd_ctx::Y][f];
			HIP_TRY(c, hipMemcpyAsync(c->row_ptr(pl, -depth), c->row_ptr(pl, c->nyl - depth), bytes, hipMemcpyDeviceToDevice, c->compute));
			HIP_TRY(c, hipMemcpyAsync(c->row_ptr(pl, c->nyl), c->row_ptr(pl, 0), bytes, hipMemcpyDeviceToDevice, c->compute));
		}
		HIP_TRY(c, hipStreamSynchronize(c->compute));
		return CRD_OK;
	}
	crd_ctx *one[1] = {c};
	c->cycle_pos = -1;
	if (int rc = prime_halo(one, 1, crd_ctx::Y, depth, true)) return rc;
	HIP_TRY(c, hipStreamWaitEvent(c->compute, c->ev_halo, 0));
	HIP_TRY(c, hipStreamSynchronize(c->comm));
	HIP_TRY(c, hipStreamSynchronize(c->compute));
	if (depth == kStepHalo * c->exchange_every) c->cycle_pos = 0;  // a full deep-halo exchange: the fused stepper may start its cycle from here
	return CRD_OK;
}

int crd_state_download_rows(crd_ctx *c, int var, int64_t row_begin, int64_t row_count, void *host)
{
	if (!c || !host) return CRD_EINVAL;
	if ((var != 0 && var != 1) || row_count < 0 || row_begin < -kGhost || row_begin + row_count > (int64_t)c->nyl + kGhost) return fail(c, CRD_EINVAL, "rows outside the plane");
	if (int rc = set_device(c)) return rc;
	if (row_count == 0) return CRD_OK;
	HIP_TRY(c, hipMemcpyAsync(host, c->row_ptr(c->plane[crd_ctx::Y][var], row_begin), (size_t)row_count * (size_t)c->nx * c->real_size, hipMemcpyDeviceToHost, c->compute));
	HIP_TRY(c, hipStreamSynchronize(c->compute));
	return CRD_OK;
}

int crd_state_upload(crd_ctx *c, const void *y, int host_is_f64)
{
	if (!c || !y) return CRD_EINVAL;
	if (c->p.precision == CRD_PRECISION_F64 && !host_is_f64) return fail(c, CRD_EINVAL, "an fp64 context takes double host buffers");
	TraceRange range("crd_state_upload");
	if (int rc = set_device(c)) return rc;
	const size_t bytes = 2 * (size_t)c->nx * (size_t)c->nyl * (host_is_f64 ? 8 : 4);
	if (int rc = ensure_staging(c, 2 * (size_t)c->nx * (size_t)c->nyl * 8)) return rc;
	c->dense.pending = false;  // a new state: nothing to resume
	c->ark.live = false;       // ... and a controller history that belongs to the old one
	c->cycle_pos = -1;         // ... and ghost rows that belong to the old one
	HIP_TRY(c, hipMemcpyAsync(c->stage_in, y, bytes, hipMemcpyHostToDevice, c->compute));
	HIP_TRY(c, launch_aos_to_planes(c->p.precision, host_is_f64, c->stage_in, c->planes(crd_ctx::Y), c->nx, c->nyl, c->compute));
	HIP_TRY(c, hipStreamSynchronize(c->compute));
	return CRD_OK;
}

int crd_state_download(crd_ctx *c, void *y, int host_is_f64)
{
	if (!c || !y) return CRD_EINVAL;
	if (c->p.precision == CRD_PRECISION_F64 && !host_is_f64) return fail(c, CRD_EINVAL, "an fp64 context fills double host buffers");
	TraceRange range("crd_state_download");
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
	if (c->d0 > 1) return fail(c, CRD_ESTATE, "theta-blocks evaluate f through crd_group_rhs_device");
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
	// Exchange() (src/FHNmodel_torus.cpp:775-950): pack var0 of the first / last row -- and, for theta-blocks, of the first / last
	// column -- of every block's vector, then every block pulls its four neighbours' strips (only var0 is ever read from them)
	for (int k = 0; k < n; k++) {
		crd_ctx *c = ctxs[k];
		if (c->halo != CRD_HALO_LOCAL) return fail(ctxs[0], CRD_ESTATE, "group is not attached");
		if (int rc = set_device(c)) return rc;
		HIP_TRY(c, launch_aos_row_extract(c->p.precision, y[k], c->edge_lo, c->nx, 0, c->compute));
		HIP_TRY(c, launch_aos_row_extract(c->p.precision, y[k], c->edge_hi, c->nx, c->nyl - 1, c->compute));
		if (c->d0 > 1) HIP_TRY(c, launch_aos_cols_extract(c->p.precision, y[k], c->edge_col[0], c->edge_col[1], c->nx, c->nyl, c->compute));
		HIP_TRY(c, hipEventRecord(c->ev_edges, c->compute));
	}
	for (int k = 0; k < n; k++) {
		crd_ctx *c = ctxs[k];
		if (int rc = set_device(c)) return rc;
		if (c->d1 > 1) {
			crd_ctx *prev = c->neighbour(0, -1), *next = c->neighbour(0, +1);
			const size_t bytes = (size_t)c->nx * c->real_size;
			HIP_TRY(c, hipStreamWaitEvent(c->compute, prev->ev_edges, 0));
			HIP_TRY(c, hipStreamWaitEvent(c->compute, next->ev_edges, 0));
			HIP_TRY(c, hipMemcpyPeerAsync(c->ghost_lo, c->device, prev->edge_hi, prev->device, bytes, c->compute));
			HIP_TRY(c, hipMemcpyPeerAsync(c->ghost_hi, c->device, next->edge_lo, next->device, bytes, c->compute));
		}
		if (c->d0 > 1) {
			crd_ctx *west = c->neighbour(-1, 0), *east = c->neighbour(+1, 0);
			const size_t bytes = (size_t)c->nyl * c->real_size;
			HIP_TRY(c, hipStreamWaitEvent(c->compute, west->ev_edges, 0));
			HIP_TRY(c, hipStreamWaitEvent(c->compute, east->ev_edges, 0));
			HIP_TRY(c, hipMemcpyPeerAsync(c->ghost_col[0], c->device, west->edge_col[1], west->device, bytes, c->compute));
			HIP_TRY(c, hipMemcpyPeerAsync(c->ghost_col[1], c->device, east->edge_col[0], east->device, bytes, c->compute));
		}
		HIP_TRY(c, launch_rhs_aos(c->p.precision, c->desc, absorbing(c, t) ? 1 : 0, y[k], ydot[k], c->ghost_lo, c->ghost_hi, 0, c->nyl, c->compute, c->ghost_col[0],
		                          c->ghost_col[1]));
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

int crd_synchronize(crd_ctx *c)
{
	if (!c) return CRD_EINVAL;
	if (int rc = set_device(c)) return rc;
	HIP_TRY(c, hipStreamSynchronize(c->comm));
	HIP_TRY(c, hipStreamSynchronize(c->compute));
	return CRD_OK;
}

int crd_set_autotune(crd_ctx *c, int on)
{
	if (!c) return CRD_EINVAL;
	c->plan.autotune = c->plan_embed.autotune = c->plan_arkode.autotune = on <= 0 ? 0 : (on >= 2 ? 2 : 1);  // 2: print the timings to stderr
	if (on <= 0) c->plan.tuned = c->plan_embed.tuned = c->plan_arkode.tuned = c->plan.pinned = 0;  // back to the plain plan
	return CRD_OK;
}

int crd_set_launch_plan(crd_ctx *c, int chunk_mode, int xcd_mapping, int columns_per_lane, int nontemporal_stores, int steps_per_launch)
{
	if (!c) return CRD_EINVAL;
	if (chunk_mode < 0 || chunk_mode > 2 || xcd_mapping < 0 || xcd_mapping > 2 || columns_per_lane < 1 || columns_per_lane > 2 || nontemporal_stores < 0 ||
	    nontemporal_stores > 1 || steps_per_launch < 1 || steps_per_launch > 3)
		return fail(c, CRD_EINVAL, "crd_set_launch_plan: chunk mode 0..2, mapping 0..2, columns per lane 1..2, non-temporal stores 0..1, steps per launch 1..3");
	// (the error-controlled integrators' instantiations take the plan too; they step one column per lane whatever it says)
	for (FusedPlan *pl : {&c->plan, &c->plan_embed, &c->plan_arkode}) {
		pl->tuned = pl->pinned = 1;
		pl->one_round = chunk_mode;
		pl->remap = xcd_mapping;
		pl->cols = columns_per_lane;
		pl->nt = nontemporal_stores;
		pl->steps = pl == &c->plan ? steps_per_launch : 1;
		pl->rows = c->nyl;  // (a pinned plan applies to launches of every height)
		pl->ms_default = pl->ms_best = 0.f;
	}
	return CRD_OK;
}

int crd_get_launch_plan(const crd_ctx *c, crd_launch_plan *out)
{
	if (!c || !out) return CRD_EINVAL;
	out->autotune = c->plan.autotune;
	out->tuned = c->plan.tuned;
	out->one_round = c->plan.one_round;
	out->xcd_mapping = c->plan.remap;
	out->rows = c->plan.rows;
	// (without a plan the launches take the default of their precision: two columns per lane in fp32 on an even nx)
	out->columns_per_lane = c->plan.tuned ? c->plan.cols : fused_default_columns(c->p.precision, c->nx);
	out->nontemporal_stores = c->plan.nt;
	out->steps_per_launch = c->plan.tuned ? fused_steps_supported(c->p.precision, c->desc, c->plan.steps) : 1;
	// (the three-step pipeline has ONE form per precision -- one column per lane in fp64, two in fp32 --, whatever the plan's columns say)
	if (out->steps_per_launch == 3) out->columns_per_lane = c->p.precision == CRD_PRECISION_F64 ? 1 : 2;
	out->ms_default = c->plan.ms_default;
	out->ms_chosen = c->plan.ms_best;
	return CRD_OK;
}

int crd_get_launch_geometry(crd_ctx *c, crd_launch_geometry *out)
{
	if (!c || !out) return CRD_EINVAL;
	*out = crd_launch_geometry{};
	if (resolve_stepper(c) != CRD_STEPPER_FUSED) return fail(c, CRD_ESTATE, "launch geometry: this context does not step with the one-launch kernel");
	if (int rc = set_device(c)) return rc;
	int64_t rows = c->nyl;
	if (int rc = crd_dominant_kernel_rows(c, &rows)) return rc;  // (a multi-slab context: the full-height launch of the exchange cycle that the timed calls time)
	const int ext = (int)(rows - c->nyl) / 2;
	FusedGeometry g{};
	FusedCall call{};
	call.dt = 1.0;
	// the stage flags of the last step the context took (none taken: of a step at t = 0): which instantiation a launch there runs
	{
		const double cs4[4] = {0.0, 0.5, 0.5, 1.0}, dt_last = c->last_step_dt > 0.0 ? c->last_step_dt : 0.0;
		for (int k = 0; k < 4; k++) call.absorb[k] = absorbing(c, c->last_step_t + cs4[k] * dt_last) ? 1 : 0;
		call.absorb[4] = call.absorb[3];
	}
	call.steps = c->plan.tuned ? fused_steps_supported(c->p.precision, c->desc, c->plan.steps) : 1;
	// (fp32 triples any stage of which has the absorbing rows on are stepped as a pair and a single step: run_steps, triple_absorbs)
	if (call.steps == 3 && c->p.precision != CRD_PRECISION_F64 && (call.absorb[0] || call.absorb[1] || call.absorb[2] || call.absorb[3])) call.steps = 2;
	call.plan = &c->plan;
	call.geometry = &g;
	HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, -ext, c->nyl + ext, 0, 0, c->compute));
	out->strips = g.strips;
	out->chunk_rows = g.chunk_rows;
	out->chunks = g.chunks;
	out->workgroups = g.blocks;
	out->wavefronts_per_workgroup = g.waves_per_block;
	out->fill_iterations = g.fill_iterations;
	out->iterations_per_trip = g.iterations_per_trip;
	out->lanes = g.lanes;
	out->lanes_valid = g.lanes_valid;
	out->rows = (int32_t)rows;
	out->wavefront_iterations = g.wave_iterations;
	out->wavefront_iterations_effective = g.wave_iterations - (int64_t)g.strips * g.chunks * (g.fill_iterations / 2 + 1);
	if (const KernelStats *k = step_kernel_stats(g.real_bytes, g.model, g.absorb, g.embed, g.cols, g.nt, g.steps)) {
		out->vgprs = k->vgprs;
		out->sgprs = k->sgprs;
		out->lds_bytes = k->lds_bytes;
		out->scratch_bytes = k->scratch_bytes;
		out->wavefronts_per_simd = k->wavefronts_per_simd;
		out->loop_valu = k->loop_valu;
		out->loop_salu = k->loop_salu;
		out->loop_vmem = k->loop_vmem;
		out->loop_lds = k->loop_lds;
		out->loop_instructions = k->loop_total;
		out->exec_skipped_vmem = k->exec_skipped_vmem;
	}
	hipDeviceProp_t prop;
	HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
	out->simds = 4 * prop.multiProcessorCount;
	out->clock_khz = prop.clockRate;
	return CRD_OK;
}

int crd_launch_plan_candidate(int index, crd_launch_plan *out)
{
	if (!out) return CRD_EINVAL;
	*out = crd_launch_plan{};
	if (!fused_plan_candidate(index, &out->one_round, &out->xcd_mapping, &out->columns_per_lane, &out->nontemporal_stores, &out->steps_per_launch)) return CRD_EINVAL;
	return CRD_OK;
}

int crd_state_max_abs(crd_ctx *c, double *out)
{
	if (!c || !out) return CRD_EINVAL;
	if (int rc = set_device(c)) return rc;
	double *sink = c->scalar_sink();
	HIP_TRY(c, launch_max_abs(c->p.precision, c->plane[crd_ctx::Y][0], c->nx, c->nyl, sink, c->compute));
	if (sink != c->scalar_host) HIP_TRY(c, hipMemcpyAsync(c->scalar_host, sink, sizeof(double), hipMemcpyDeviceToHost, c->compute));
	HIP_TRY(c, hipStreamSynchronize(c->compute));
	*out = c->scalar_host[0];
	return CRD_OK;
}

}  // extern "C"
