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
	if (nprocs > 1) {
		/* Every rank takes part in the broadcast whatever happened to it so far: a rank that returned early would leave the
		 * others blocked in bcast().  Byte 0 carries rank 0's status (its crd_create and crd_comm_unique_id), bytes 1..128 the id. */
		unsigned char msg[129] = {0};
		if (!bcast) {
			crd_destroy(ctx);
			return CRD_EINVAL;
		}
		if (rank == 0) {
			int rc0 = rc != CRD_OK ? rc : crd_comm_unique_id(msg + 1);
			msg[0] = (unsigned char)(-rc0); /* crd_status values are 0 .. -7 */
			if (rc == CRD_OK) rc = rc0;
		}
		if (bcast(msg, (int)sizeof msg, comm) != 0 && rc == CRD_OK) rc = CRD_ERCCL;
		if (rc == CRD_OK && msg[0] != 0) rc = rank == 0 ? -(int)msg[0] : CRD_ERCCL; /* rank 0 could not make an id: nobody joins */
		if (rc != CRD_OK) {
			crd_destroy(ctx);
			return rc;
		}
		/* (collective: if crd_create failed on a rank other than 0, that rank has returned its error above and the ranks that
		 * get here wait for it -- abort the job on any non-zero return, as the reference does after check_flag, :681-705) */
		if ((rc = crd_comm_init_rccl(ctx, msg + 1)) != CRD_OK) {
			crd_destroy(ctx);
			return rc;
		}
	} else if (rc != CRD_OK) {
		return rc;
	}
	*out = ctx;
	return CRD_OK;
}

realtype *crd_arkode_alloc(long local_length) { return local_length > 0 ? (realtype *)crd_host_alloc((size_t)local_length * sizeof(realtype)) : NULL; }

void crd_arkode_free(realtype *data) { crd_host_free(data); }
