/*
 * crd_arkode_shim.h -- CRDModel's SUNDIALS callback bound to libcrd.
 *
 * For a maintainer who keeps the reference's adaptive ARKode loop (src/FHNmodel_torus.cpp:356-372,423) and only replaces the
 * right-hand side: compile crd_arkode_shim.c into the reference program with the SUNDIALS headers that program already uses
 * (2.6 / 2.7: `ARKRhsFn` = int (*)(realtype, N_Vector, N_Vector, void *)), then
 *
 *     crd_ctx *gpu;  crd_arkode_attach(&cfg, rank, nprocs, device, bcast, comm, &gpu);
 *     ARKodeInit(arkode_mem, crd_arkode_f, NULL, T0, y);          // was f           (:362)
 *     ARKodeSetUserData(arkode_mem, (void *) gpu);                // was udata       (:369)
 *
 * INTEGRATION.md section 1 has the surrounding lines.
 */
#ifndef CRD_ARKODE_SHIM_H
#define CRD_ARKODE_SHIM_H

#ifdef CRD_SHIM_NVECTOR_HEADER /* the self-test supplies its own N_Vector (tests/native/mock_nvector.h) */
#include CRD_SHIM_NVECTOR_HEADER
#else
#include <nvector/nvector_parallel.h>
#include <sundials/sundials_types.h>
#endif

#include "crd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The ARKRhsFn: replaces `static int f(realtype t, N_Vector y, N_Vector ydot, void *user_data)` (src/FHNmodel_torus.cpp:126,
 * 504-667 and the three siblings).  user_data is the crd_ctx.  y / ydot are the rank's NVECTOR_PARALLEL vectors in the
 * reference's layout (IDX, :60); halo exchange, diffusion and kinetics happen on the GPU; every element of ydot is written.
 * Returns 0, or -1 as the reference does when its Exchange fails (:522) -- unrecoverable for ARKode. */
int crd_arkode_f(realtype t, N_Vector y, N_Vector ydot, void *user_data);

/* Creates the rank's context from the run configuration (crd_config_load_ini of the same ini file main() reads, :158-174) and,
 * for nprocs > 1, joins the RCCL ring: `bcast(buf, bytes, comm)` is any broadcast of `bytes` (129: a status byte + the 128-byte
 * id) from rank 0 -- with MPI, `MPI_Bcast(buf, bytes, MPI_BYTE, 0, comm)` wrapped in a two-line function.  Every rank always
 * takes part in that one broadcast, also a rank whose own set-up has failed, and a failure of rank 0 (context, RCCL library,
 * unique id) is returned by EVERY rank after it, so an error never turns into a hang in bcast.  A crd_create failure on another
 * rank is returned by that rank only: treat any non-zero return as fatal for the job (MPI_Abort), as the reference treats a
 * failing check_flag (src/FHNmodel_torus.cpp:681-705).  The reference's 2-D process grid becomes
 * phi-slabs: run it with dims = {1, nprocs} (:724-728).  Returns crd_status; *out is NULL on failure and crd_last_error(NULL)
 * has the text. */
typedef int (*crd_bcast_fn)(void *buf, int bytes, void *comm);
int crd_arkode_attach(const crd_run_config *cfg, int rank, int nprocs, int device, crd_bcast_fn bcast, void *comm, crd_ctx **out);

/* Page-locked storage for the N_Vector data (N_VMake_Parallel instead of N_VNew_Parallel, :281): with it a single-GPU
 * crd_arkode_f streams the slab band by band over the host link. */
realtype *crd_arkode_alloc(long local_length);
void crd_arkode_free(realtype *data);

#ifdef __cplusplus
}
#endif
#endif
