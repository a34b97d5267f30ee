/* crd_arkode_shim.c -- see crd_arkode_shim.h.  Plain C; no HIP, no C++. */
#include "crd_arkode_shim.h"

#include <stddef.h>

int crd_arkode_f(realtype t, N_Vector y, N_Vector ydot, void *user_data)
{
	crd_ctx *ctx = (crd_ctx *)user_data;
	if (!ctx || !y || !ydot) return -1;
	/* realtype must be the context's precision: double for SUNDIALS_DOUBLE_PRECISION (the reference's build, :790-796) */
	return crd_rhs_host(ctx, (double)t, NV_DATA_P(y), NV_DATA_P(ydot)) == CRD_OK ? 0 : -1;
}

int crd_arkode_attach(const crd_run_config *cfg, int rank, int nprocs, int device, crd_bcast_fn bcast, void *comm, crd_ctx **out)
{
	if (!cfg || !out) return CRD_EINVAL;
	*out = NULL;
	crd_params p = cfg->params;
	p.precision = sizeof(realtype) == 8 ? CRD_PRECISION_F64 : CRD_PRECISION_F32;
	crd_ctx *ctx = NULL;
	int rc = crd_create(&p, rank, nprocs, device, &ctx);
	if (rc != CRD_OK) return rc;
	if (nprocs > 1) {
		unsigned char id[128];
		if (!bcast) {
			crd_destroy(ctx);
			return CRD_EINVAL;
		}
		if (rank == 0 && (rc = crd_comm_unique_id(id)) != CRD_OK) {
			crd_destroy(ctx);
			return rc;
		}
		if (bcast(id, (int)sizeof id, comm) != 0) {
			crd_destroy(ctx);
			return CRD_ERCCL;
		}
		if ((rc = crd_comm_init_rccl(ctx, id)) != CRD_OK) {
			crd_destroy(ctx);
			return rc;
		}
	}
	*out = ctx;
	return CRD_OK;
}

realtype *crd_arkode_alloc(long local_length) { return local_length > 0 ? (realtype *)crd_host_alloc((size_t)local_length * sizeof(realtype)) : NULL; }

void crd_arkode_free(realtype *data) { crd_host_free(data); }
