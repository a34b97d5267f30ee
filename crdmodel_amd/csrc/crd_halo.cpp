// crd_halo.cpp -- the RCCL binding and the halo transports of multi-slab runs.  Host code only.
#include <dlfcn.h>

#include <mutex>

#include "crd_ctx.h"

namespace crd {

RcclApi g_rccl;

bool RcclApi::load()
{
	static std::mutex once;  // contexts are per-thread objects, the binding is per-process
	std::lock_guard<std::mutex> lock(once);
	if (handle) return true;
	if (!error.empty()) return false;
	// crd_comm_set_rccl_library: bind that library instead (another RCCL build; the multi-process ring tests' stand-in, tests/native)
	if (!library.empty()) {
		handle = dlopen(library.c_str(), RTLD_NOW | RTLD_LOCAL);
		if (!handle) {
			error = std::string("cannot load ") + library + ": " + dlerror();
			return false;
		}
	}
	for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
		if (handle) break;
		handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
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
	CommCount = reinterpret_cast<decltype(CommCount)>(sym("ncclCommCount"));
	CommUserRank = reinterpret_cast<decltype(CommUserRank)>(sym("ncclCommUserRank"));
	GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
	if (!error.empty()) {
		handle = nullptr;
		return false;
	}
	return true;
}

namespace {

// The two transports: fill the ghost rows [-depth, 0) and [nyl, nyl + depth) of one context's planes from its ring
// neighbours, enqueued on the context's comm stream.

// RCCL: the four point-to-point operations of crd_halo_plan per field, in one group.
int exchange_rccl(crd_ctx *c, Planes pl, int depth, bool with_v)
{
	const size_t count = (size_t)depth * (size_t)c->nx;
	const ncclDataType_t dt = c->p.precision == CRD_PRECISION_F64 ? ncclDouble : ncclFloat;
	void *fields[2] = {pl.u, pl.v};
	crd_halo_op ops[4];
	if (crd_halo_plan(c->slab, c->n_slabs, c->nyl, depth, ops) != CRD_OK) return fail(c, CRD_EINVAL, "bad halo plan");
	// (diagnostics: the exchange's own duration on the comm stream, from the moment its inputs are ready to its last byte)
	const bool diag = c->diag_active && 4 * c->diag_exchanges + 3 < (int)c->ev_diag.size();
	if (diag) HIP_TRY(c, hipEventRecord(c->ev_diag[(size_t)(4 * c->diag_exchanges + 2)], c->comm));
	NCCL_TRY(c, g_rccl.GroupStart());
	for (int f = 0; f < (with_v ? 2 : 1); f++)
		for (const crd_halo_op &op : ops) {
			void *rows = c->row_ptr(fields[f], op.row_begin);
			if (op.is_send) NCCL_TRY(c, g_rccl.Send(rows, count, dt, op.peer, c->nccl, c->comm));
			else NCCL_TRY(c, g_rccl.Recv(rows, count, dt, op.peer, c->nccl, c->comm));
		}
	NCCL_TRY(c, g_rccl.GroupEnd());
	if (diag) HIP_TRY(c, hipEventRecord(c->ev_diag[(size_t)(4 * c->diag_exchanges++ + 3)], c->comm));
	return CRD_OK;
}

// LOCAL: pull the rows from the neighbours' planes with device-to-device copies (peer copies across devices).
int exchange_local_pull(crd_ctx *c, int plane_index, int depth, bool with_v)
{
	crd_ctx *prev = c->group[(size_t)((c->slab + c->n_slabs - 1) % c->n_slabs)];
	crd_ctx *next = c->group[(size_t)((c->slab + 1) % c->n_slabs)];
	const size_t bytes = (size_t)depth * (size_t)c->nx * c->real_size;
	// (a neighbour's state need not sit in the plane of the same index: slabs whose launch plans pair steps differently have made
	// different numbers of launches since the last exchange -- every context says where its rows are, xchg_plane)
	for (int f = 0; f < (with_v ? 2 : 1); f++) {
		void *mine = c->plane[plane_index][f];
		HIP_TRY(c, hipMemcpyPeerAsync(c->row_ptr(mine, -depth), c->device, prev->row_ptr(prev->plane[prev->xchg_plane][f], prev->nyl - depth), prev->device,
		                              bytes, c->comm));
		HIP_TRY(c, hipMemcpyPeerAsync(c->row_ptr(mine, c->nyl), c->device, next->row_ptr(next->plane[next->xchg_plane][f], 0), next->device, bytes, c->comm));
	}
	return CRD_OK;
}

}  // namespace

// Exchange for every context of the call: the comm streams first wait for the producers of the rows that travel, then carry
// the transfers and record each context's halo event; the caller orders the compute streams against those events.
int exchange_stage_input(crd_ctx *const *cs, int n, int plane_index, int depth, bool with_v)
{
	// Several issuing threads (one per device of a LOCAL group): every thread has enqueued the records of its contexts'
	// edge events before any thread makes a stream wait for a neighbour's.
	TraceRange range(depth > kStepHalo + 1 ? "crd_halo_exchange(deep)" : "crd_halo_exchange");
	for (int k = 0; k < n; k++) cs[k]->xchg_plane = plane_index;  // (read by the neighbours' pulls, behind the rendezvous below)
	GroupBarrier *bar = cs[0]->bar;
	if (bar && !bar->wait()) return fail(cs[0], CRD_ESTATE, "another slab's issuing thread failed");
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
	// ... and every halo event is on record before any thread makes a stream wait for a neighbour's
	if (bar && !bar->wait()) return fail(cs[0], CRD_ESTATE, "another slab's issuing thread failed");
	return CRD_OK;
}

// 2-D blocks of a LOCAL group (staged stepper): var0 of plane `plane_index` -- one ghost row from each phi neighbour (when phi is
// split) and one ghost column strip from each theta neighbour (when theta is), the four strips of the reference's Exchange()
// (src/FHNmodel_torus.cpp:775-950; the 5-point stencil reads no corner).  Every context has recorded ev_edges behind the launch
// that wrote the plane and behind the pack of its edge columns.
int exchange_block_input(crd_ctx *const *cs, int n, int plane_index)
{
	TraceRange range("crd_halo_exchange(blocks)");
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (int rc = set_device(c)) return rc;
		HIP_TRY(c, hipStreamWaitEvent(c->comm, c->ev_edges, 0));
		if (c->d1 > 1) {
			HIP_TRY(c, hipStreamWaitEvent(c->comm, c->neighbour(0, -1)->ev_edges, 0));
			HIP_TRY(c, hipStreamWaitEvent(c->comm, c->neighbour(0, +1)->ev_edges, 0));
		}
		if (c->d0 > 1) {
			HIP_TRY(c, hipStreamWaitEvent(c->comm, c->neighbour(-1, 0)->ev_edges, 0));
			HIP_TRY(c, hipStreamWaitEvent(c->comm, c->neighbour(+1, 0)->ev_edges, 0));
		}
	}
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (int rc = set_device(c)) return rc;
		if (c->d1 > 1) {
			crd_ctx *prev = c->neighbour(0, -1), *next = c->neighbour(0, +1);
			const size_t bytes = (size_t)c->nx * c->real_size;
			void *mine = c->plane[plane_index][0];
			HIP_TRY(c, hipMemcpyPeerAsync(c->row_ptr(mine, -1), c->device, prev->row_ptr(prev->plane[plane_index][0], prev->nyl - 1), prev->device, bytes, c->comm));
			HIP_TRY(c, hipMemcpyPeerAsync(c->row_ptr(mine, c->nyl), c->device, next->row_ptr(next->plane[plane_index][0], 0), next->device, bytes, c->comm));
		}
		if (c->d0 > 1) {
			crd_ctx *west = c->neighbour(-1, 0), *east = c->neighbour(+1, 0);
			const size_t bytes = (size_t)c->nyl * c->real_size;
			HIP_TRY(c, hipMemcpyPeerAsync(c->gcol[plane_index][0], c->device, west->ecol[plane_index][1], west->device, bytes, c->comm));
			HIP_TRY(c, hipMemcpyPeerAsync(c->gcol[plane_index][1], c->device, east->ecol[plane_index][0], east->device, bytes, c->comm));
		}
		HIP_TRY(c, hipEventRecord(c->ev_halo, c->comm));
	}
	return CRD_OK;
}

// ... in front of a run's first stage: the edge columns of the resident state, then the exchange
int prime_block_halo(crd_ctx *const *cs, int n, int plane_index)
{
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (int rc = set_device(c)) return rc;
		if (c->d0 > 1)
			HIP_TRY(c, launch_plane_cols_extract(c->p.precision, c->plane[plane_index][0], c->ecol[plane_index][0], c->ecol[plane_index][1], c->nx, c->nyl, c->compute));
		HIP_TRY(c, hipEventRecord(c->ev_edges, c->compute));
	}
	return exchange_block_input(cs, n, plane_index);
}

int prime_halo(crd_ctx *const *cs, int n, int plane_index, int depth, bool with_v)
{
	for (int k = 0; k < n; k++) {
		if (int rc = set_device(cs[k])) return rc;
		HIP_TRY(cs[k], hipEventRecord(cs[k]->ev_edges, cs[k]->compute));
	}
	return exchange_stage_input(cs, n, plane_index, depth, with_v);
}

}  // namespace crd
