/* Test double for the two SUNDIALS names crd_arkode_shim.c uses (realtype, N_Vector + NV_DATA_P), so that the shim can be
 * compiled and its callback driven on a box without SUNDIALS.  Test infrastructure only: not a build of the reference, not
 * shipped; a real build includes <nvector/nvector_parallel.h> instead (integration/crd_arkode_shim.h). */
#ifndef MOCK_NVECTOR_H
#define MOCK_NVECTOR_H
typedef double realtype;
typedef struct mock_nvector {
	long local_length;
	realtype *data;
} *N_Vector;
#define NV_DATA_P(v) ((v)->data)
#endif
