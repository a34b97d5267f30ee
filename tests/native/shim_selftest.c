/* Drives integration/crd_arkode_shim.c the way the reference's main() + an explicit Runge-Kutta integrator would: the callback
 * is registered as an `ARKRhsFn`-typed function pointer, called once per stage on host vectors, and the trajectory is compared
 * with libcrd's own resident-state stepper (crd_step_rk4) on the same initial conditions.  argv[1] = ini file.
 * Built by tests/test_gpu_driver.py with gcc -DCRD_SHIM_NVECTOR_HEADER='"mock_nvector.h"'. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "crd_arkode_shim.h"

typedef int (*ARKRhsFn)(realtype t, N_Vector y, N_Vector ydot, void *user_data); /* SUNDIALS 2.x arkode.h */

static N_Vector make_vector(long n)
{
	N_Vector v = (N_Vector)malloc(sizeof *v);
	v->local_length = n;
	v->data = crd_arkode_alloc(n);
	return v;
}

int main(int argc, char **argv)
{
	if (argc != 2) return 2;
	crd_run_config cfg;
	char err[256];
	if (crd_config_load_ini(argv[1], CRD_MODEL_FHN, CRD_SURFACE_TORUS, &cfg, err, sizeof err) != CRD_OK) {
		fprintf(stderr, "%s\n", err);
		return 3;
	}
	crd_ctx *gpu = NULL;
	if (crd_arkode_attach(&cfg, 0, 1, 0, NULL, NULL, &gpu) != CRD_OK) {
		fprintf(stderr, "%s\n", crd_last_error(NULL));
		return 4;
	}
	crd_grid g;
	crd_get_grid(gpu, &g);
	const long n = 2 * g.nx * g.ny;
	N_Vector y = make_vector(n), k = make_vector(n), ys = make_vector(n), acc = make_vector(n);
	if (!y->data || !k->data || !ys->data || !acc->data) return 5;
	if (crd_initial_conditions(&cfg, 0, g.ny - 1, y->data) != CRD_OK) return 6;
	if (crd_state_upload(gpu, y->data, 1) != CRD_OK) return 7;

	ARKRhsFn f = crd_arkode_f; /* what ARKodeInit(arkode_mem, f, NULL, T0, y) stores (src/FHNmodel_torus.cpp:362) */
	const double dt = 0.5 * crd_stable_dt(&cfg.params);
	const int nsteps = 6; /* crosses nothing special; tBoundary of the ini is honoured through t */
	const double c[4] = {0.0, 0.5, 0.5, 1.0}, w[4] = {1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0};
	double t = 0.0;
	for (int s = 0; s < nsteps; s++) {
		for (long q = 0; q < n; q++) acc->data[q] = y->data[q], ys->data[q] = y->data[q];
		for (int st = 0; st < 4; st++) {
			if (f(t + c[st] * dt, ys, k, gpu) != 0) {
				fprintf(stderr, "f failed: %s\n", crd_last_error(gpu));
				return 8;
			}
			const double a = st < 3 ? c[st + 1] * dt : 0.0;
			for (long q = 0; q < n; q++) {
				acc->data[q] += w[st] * dt * k->data[q];
				if (st < 3) ys->data[q] = y->data[q] + a * k->data[q];
			}
		}
		for (long q = 0; q < n; q++) y->data[q] = acc->data[q];
		t += dt;
	}
	if (crd_step_rk4(gpu, 0.0, dt, nsteps) != CRD_OK || crd_state_download(gpu, ys->data, 1) != CRD_OK) return 9;
	double worst = 0.0, scale = 0.0, moved = 0.0;
	crd_initial_conditions(&cfg, 0, g.ny - 1, k->data);
	for (long q = 0; q < n; q++) {
		worst = fmax(worst, fabs(y->data[q] - ys->data[q]));
		scale = fmax(scale, fabs(ys->data[q]));
		moved = fmax(moved, fabs(ys->data[q] - k->data[q]));
	}
	printf("shim selftest: %ld unknowns, %d RK4 steps through the ARKRhsFn, max |diff| / max |y| = %.3e, state moved by %.3e\n", n, nsteps, worst / scale, moved);
	if (f(0.0, NULL, k, gpu) != -1) return 10; /* error convention of the callback */
	N_Vector all[4] = {y, k, ys, acc};
	for (int i = 0; i < 4; i++) {
		crd_arkode_free(all[i]->data);
		free(all[i]);
	}
	crd_destroy(gpu);
	return (worst / scale <= 1e-12 && moved > 1e-6) ? 0 : 1;
}
