/*
 * crd_oracle.c -- CPU restatement of CRDModel's RHS hot path.  TEST INFRASTRUCTURE ONLY (see crd_oracle.h).
 * PARITY UNPINNED against reference-run output: the reference cannot be built in this image (SUNDIALS, Boost
 * absent) and ships no tests or golden vectors; see crd_oracle.h for what pins this file instead.
 *
 * Plain C99 + libm (+ OpenMP for the timed multi-core baseline).  All citations are /root/reference paths.
 */
#include "crd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* src/FHNmodel_torus.cpp:63 -- PI as the reference spells it (same double as M_PI). */
#define ORACLE_PI 3.1415926535897932
/* src/FHNmodel_torus.cpp:68,71 */
#define ORACLE_EPSILON 0.36
#define NV 2
/* src/GoldbeterModel_torus.cpp:67-78 */
#define GB_V0 1.0
#define GB_K 10.0
#define GB_KF 1.0
#define GB_V1 7.3
#define GB_VM2 65.0
#define GB_VM3 500.0
#define GB_K2 1.0
#define GB_KR 2.0
#define GB_KA 0.9
#define GB_M 2.0
#define GB_N 2.0
#define GB_P 4.0

#define LIDX(i, j, nxl) (NV * (i) + NV * (j) * (nxl)) /* src/FHNmodel_torus.cpp:60 */

/* Geometry scalars: torus src/FHNmodel_torus.cpp:73-76,188-193,233-234; flat src/FHNmodel_flat.cpp:172-175,190-192,
 * 230-231.  ny is a double product truncated to long (torus) or an integer ratio times nx (flat). */
int crd_oracle_geometry(int surface, double surface_length, double surface_width, long nx, long ny_override,
                        crd_oracle_problem *p)
{
	if (!p || nx < 2) return -1;
	p->surface = surface;
	p->nx = nx;
	if (surface == CRD_ORACLE_TORUS) {
		double r = surface_width / (2.0 * ORACLE_PI);
		double R = surface_length / (2.0 * ORACLE_PI);
		double radius_ratio = R / r;
		int nx_int = (int)nx;
		p->r = r;
		p->R = R;
		p->ny = (long)(nx_int * (radius_ratio));
		p->xmin = 0.0;
		p->xmax = 2.0 * ORACLE_PI;
		p->ymin = 0.0;
		p->ymax = 2.0 * ORACLE_PI;
	} else if (surface == CRD_ORACLE_FLAT) {
		long ratio = (long)(surface_length / surface_width);
		p->r = 0.0;
		p->R = 0.0;
		p->ny = (int)nx * ratio;
		p->xmin = 0.0;
		p->xmax = surface_width - p->xmin;
		p->ymin = 0.0;
		p->ymax = surface_length - p->ymin;
	} else {
		return -1;
	}
	if (ny_override > 0) p->ny = ny_override;
	if (p->ny < 2) return -1;
	p->dx = (p->xmax - p->xmin) / (1.0 * p->nx - 1.0);
	p->dy = (p->ymax - p->ymin) / (1.0 * p->ny - 1.0);
	p->is = 0;
	p->ie = p->nx - 1;
	p->js = 0;
	p->je = p->ny - 1;
	return 0;
}

/* src/FHNmodel_torus.cpp:750-753 (integer arithmetic on long) */
void crd_oracle_decomp(long nx, long ny, int d0, int d1, int c0, int c1, long *is, long *ie, long *js, long *je)
{
	*is = (nx) * (c0) / (d0);
	*ie = (nx) * (c0 + 1) / (d0)-1;
	*js = (ny) * (c1) / (d1);
	*je = (ny) * (c1 + 1) / (d1)-1;
}

/* src/FHNmodel_torus.cpp:242-244 */
void crd_oracle_fhn_steady(double beta, double *us, double *vs)
{
	*us = -beta;
	*vs = beta * beta * beta - 3 * beta;
}

static double gb_v2(double Z)
{
	/* src/GoldbeterModel_torus.cpp:694 */
	return GB_VM2 * pow(Z, GB_N) / (pow(GB_K2, GB_N) + pow(Z, GB_N));
}

static double gb_v3(double Z, double Y)
{
	/* src/GoldbeterModel_torus.cpp:695 */
	return GB_VM3 * pow(Y, GB_M) * pow(Z, GB_P) / ((pow(GB_KR, GB_M) + pow(Y, GB_M)) * (pow(GB_KA, GB_P) + pow(Z, GB_P)));
}

/* Fixed point of the Goldbeter ODE (src/GoldbeterModel_torus.cpp:715-716 with both left sides zero).  The
 * reference obtains it by running util/GoldbeterModel/SolveGoldbeterODE.py (:254-261); the fixed point itself is
 * Zs = (v0 + v1 beta)/k (sum of the two equations) and Ys the unique positive root of v2(Zs) - v3(Zs,Y) - kf Y. */
int crd_oracle_goldbeter_steady(double beta, double *zs, double *ys)
{
	double Z = (GB_V0 + GB_V1 * beta) / GB_K;
	double lo = 0.0, hi = 1.0;
	int it;
	if (!(Z > 0.0)) return -1;
	while (gb_v2(Z) - gb_v3(Z, hi) - GB_KF * hi > 0.0) {
		hi *= 2.0;
		if (hi > 1e12) return -1;
	}
	for (it = 0; it < 200; it++) {
		double mid = 0.5 * (lo + hi);
		double g = gb_v2(Z) - gb_v3(Z, mid) - GB_KF * mid;
		if (g > 0.0) lo = mid; else hi = mid;
	}
	*zs = Z;
	*ys = 0.5 * (lo + hi);
	return 0;
}

/* Initial conditions.  FHN torus src/FHNmodel_torus.cpp:199-200,285-354; FHN flat src/FHNmodel_flat.cpp:280-319;
 * Goldbeter torus src/GoldbeterModel_torus.cpp:212-213,313-414; Goldbeter flat src/GoldbeterModel_flat.cpp:211-212,
 * 309-379. */
int crd_oracle_initial_conditions(const crd_oracle_problem *p, const crd_oracle_ic *ic, double *y)
{
	long nxl = p->ie - p->is + 1, nyl = p->je - p->js + 1, i, j;
	double wave_length = (p->ymax - p->ymin) * ic->wave_length;
	double wave_width = (p->xmax - p->xmin) * ic->wave_width;
	double mid, wxmin, wxmax;
	int outside = 0;
	if (p->surface == CRD_ORACLE_TORUS) {
		if (ic->wave_inside == 1) {
			mid = ORACLE_PI;
			wxmin = mid - wave_width / 2.0;
			wxmax = mid + wave_width / 2.0;
		} else if (ic->wave_inside == 0) {
			mid = 0.0;
			wxmin = mid - wave_width / 2.0 + (p->xmax - p->xmin);
			wxmax = mid + wave_width / 2.0;
			outside = 1;
		} else {
			return -1; /* the reference prints a message and leaves y uninitialised */
		}
	} else {
		mid = (p->xmax + p->xmin) / 2.0; /* SURFACEWIDTH/2 with XMIN = 0 */
		wxmin = mid - wave_width / 2.0;
		wxmax = mid + wave_width / 2.0;
	}
	if (p->model == CRD_ORACLE_GOLDBETER && p->vary_beta == 1 && ic->ic_type == 2) srand(1); /* fresh process */

	for (j = 0; j < nyl; j++) {
		double yy = p->ymin + (p->js + j) * (p->dy);
		for (i = 0; i < nxl; i++) {
			double xx = p->xmin + (p->is + i) * (p->dx);
			double *u = &y[LIDX(i, j, nxl)], *v = u + 1;
			int in_theta = outside ? (xx >= wxmin || xx <= wxmax) : (xx >= wxmin && xx <= wxmax);
			if (p->model == CRD_ORACLE_FHN) {
				int uniform = (p->surface == CRD_ORACLE_TORUS) ? (p->vary_beta != 0) : (p->vary_beta == 1);
				if (uniform) {
					*u = 1;
					*v = 1;
				} else if (in_theta && yy >= wave_length && yy <= (2.0 * wave_length)) {
					*u = ic->s0 + 2;
					*v = ic->s1 + 1.5;
				} else {
					*u = ic->s0;
					*v = ic->s1;
				}
			} else {
				if (p->vary_beta == 0) {
					double lo = (p->surface == CRD_ORACLE_TORUS) ? 1.0 : 2.0;
					if (in_theta && yy >= lo * wave_length && yy <= ((lo + 1.0) * wave_length)) {
						*u = ic->s0 + 1;
						*v = ic->s1 + 1;
					} else {
						*u = ic->s0;
						*v = ic->s1;
					}
				} else if (p->vary_beta == 1) {
					if (ic->ic_type == 0) {
						*u = 0.4;
						*v = 1.6;
					}
					if (ic->ic_type == 1) {
						/* always the `&&` form, also for an outside-centred torus wave (:389) */
						if (xx >= wxmin && xx <= wxmax && yy >= 2.0 * wave_length && yy <= (3.0 * wave_length)) {
							*u = 1.4;
							*v = 2.6;
						} else {
							*u = 0.4;
							*v = 1.6;
						}
					}
					if (ic->ic_type == 2) {
						*u = (float)rand() / RAND_MAX * 1.4;
						*v = (float)rand() / RAND_MAX * 1.4;
					}
				}
			}
		}
	}
	return 0;
}

/* Torus diffusion of one point, operation order of src/FHNmodel_torus.cpp:535-537. */
static inline double torus_diffusion(double Diff, double r, double R, double dx, double dy, double xx, double uC, double uW,
                                     double uE, double uS, double uN)
{
	return Diff * ((-sin(xx) / (r * (R + r * cos(xx)))) * (uE - uW)) / (2 * dx) +
	       Diff * ((1 / (r * r)) * (uE - 2 * uC + uW)) / (dx * dx) +
	       Diff * ((1 / (((R + r * cos(xx))) * ((R + r * cos(xx))))) * (uN - 2 * uC + uS)) / (dy * dy);
}

/* One RHS evaluation of a subdomain: src/FHNmodel_torus.cpp:504-667, src/FHNmodel_flat.cpp:469-616,
 * src/GoldbeterModel_torus.cpp:547-724, src/GoldbeterModel_flat.cpp:515-689.  The nine regions of the reference
 * (interior, four faces, four corners) differ only in where a neighbour is read from, which the four selects
 * below reproduce. */
int crd_oracle_rhs_subdomain(const crd_oracle_problem *p, double t, const double *y, double *ydot, const double *wrecv,
                             const double *erecv, const double *srecv, const double *nrecv, int nthreads)
{
	const long nxl = p->ie - p->is + 1, nyl = p->je - p->js + 1;
	const double Diff = p->diff, dx = p->dx, dy = p->dy, R = p->R, r = p->r;
	long j;
	if (nthreads < 1) nthreads = 1;
	(void)nthreads;

	/* N_VConst(0.0, ydot) -- :506 */
#pragma omp parallel for num_threads(nthreads) schedule(static)
	for (j = 0; j < nyl; j++) memset(&ydot[LIDX(0, j, nxl)], 0, sizeof(double) * NV * nxl);

	/* diffusion of variable 0 -- :527-615 */
	if (p->surface == CRD_ORACLE_TORUS) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
		for (j = 0; j < nyl; j++) {
			long i;
			for (i = 0; i < nxl; i++) {
				double xx = p->xmin + (p->is + i) * (dx);
				double uC = y[LIDX(i, j, nxl)];
				double uW = (i > 0) ? y[LIDX(i - 1, j, nxl)] : wrecv[NV * j];
				double uE = (i < nxl - 1) ? y[LIDX(i + 1, j, nxl)] : erecv[NV * j];
				double uS = (j > 0) ? y[LIDX(i, j - 1, nxl)] : srecv[NV * i];
				double uN = (j < nyl - 1) ? y[LIDX(i, j + 1, nxl)] : nrecv[NV * i];
				ydot[LIDX(i, j, nxl)] = torus_diffusion(Diff, r, R, dx, dy, xx, uC, uW, uE, uS, uN);
			}
		}
	} else {
		/* src/FHNmodel_flat.cpp:489-491 */
		const double cu1 = Diff / dx / dx;
		const double cu2 = Diff / dy / dy;
		const double cu3 = -2.0 * (cu1 + cu2);
#pragma omp parallel for num_threads(nthreads) schedule(static)
		for (j = 0; j < nyl; j++) {
			long i;
			for (i = 0; i < nxl; i++) {
				double uC = y[LIDX(i, j, nxl)];
				double uW = (i > 0) ? y[LIDX(i - 1, j, nxl)] : wrecv[NV * j];
				double uE = (i < nxl - 1) ? y[LIDX(i + 1, j, nxl)] : erecv[NV * j];
				double uS = (j > 0) ? y[LIDX(i, j - 1, nxl)] : srecv[NV * i];
				double uN = (j < nyl - 1) ? y[LIDX(i, j + 1, nxl)] : nrecv[NV * i];
				ydot[LIDX(i, j, nxl)] = cu1 * (uW + uE) + cu2 * (uS + uN) + cu3 * uC; /* :498-500 */
			}
		}
	}

	/* reaction terms -- FHN :618-664, Goldbeter src/GoldbeterModel_torus.cpp:668-721 */
	if (p->model == CRD_ORACLE_GOLDBETER && p->just_diffusion != 0) return 0;
#pragma omp parallel for num_threads(nthreads) schedule(static)
	for (j = 0; j < nyl; j++) {
		long i;
		double yy = p->ymin + (p->js + j) * (p->dy);
		double b;
		if (p->vary_beta == 0) b = p->beta;
		else b = p->beta_min + yy * (p->beta_max - p->beta_min) / (p->ymax - p->ymin);
		for (i = 0; i < nxl; i++) {
			double a0 = y[LIDX(i, j, nxl)];
			double a1 = y[LIDX(i, j, nxl) + 1];
			double v2 = 0.0, v3 = 0.0;
			if (p->model == CRD_ORACLE_GOLDBETER) {
				v2 = gb_v2(a0);
				v3 = gb_v3(a0, a1);
			}
			if (p->je == p->ny - 1 && t < p->t_boundary && j == nyl - 1) {
				ydot[LIDX(i, j, nxl)] = 0;
				ydot[LIDX(i, j, nxl) + 1] = 0;
			} else if (p->js == 0 && t < p->t_boundary && j == 0) {
				ydot[LIDX(i, j, nxl)] = 0;
				ydot[LIDX(i, j, nxl) + 1] = 0;
			} else if (p->model == CRD_ORACLE_FHN) {
				ydot[LIDX(i, j, nxl)] += 3.0 * a0 - (a0 * a0 * a0) - a1;      /* :657 */
				ydot[LIDX(i, j, nxl) + 1] += ORACLE_EPSILON * (a0 + b);       /* :660 */
			} else {
				ydot[LIDX(i, j, nxl)] += GB_V0 + GB_V1 * b - v2 + v3 + GB_KF * a1 - GB_K * a0; /* :715 */
				ydot[LIDX(i, j, nxl) + 1] += v2 - v3 - GB_KF * a1;                              /* :716 */
			}
		}
	}
	return 0;
}

/* The pack loops of Exchange(): src/FHNmodel_torus.cpp:854-900. */
void crd_oracle_pack_edges(const crd_oracle_problem *p, const double *y, double *wsend, double *esend, double *ssend,
                           double *nsend)
{
	const long nxl = p->ie - p->is + 1, nyl = p->je - p->js + 1;
	long k;
	for (k = 0; k < nyl; k++) {
		wsend[2 * k] = y[LIDX(nxl - 1, k, nxl)];
		wsend[2 * k + 1] = y[LIDX(nxl - 1, k, nxl) + 1];
		esend[2 * k] = y[LIDX(0, k, nxl)];
		esend[2 * k + 1] = y[LIDX(0, k, nxl) + 1];
	}
	for (k = 0; k < nxl; k++) {
		ssend[2 * k] = y[LIDX(k, nyl - 1, nxl)];
		ssend[2 * k + 1] = y[LIDX(k, nyl - 1, nxl) + 1];
		nsend[2 * k] = y[LIDX(k, 0, nxl)];
		nsend[2 * k + 1] = y[LIDX(k, 0, nxl) + 1];
	}
}

/* Whole-domain RHS through a d0 x d1 block decomposition.  Each block receives, as its W strip, the E edge
 * (i = nxl-1 column, i.e. the neighbour's Wsend) of the block at coordinate c0-1 (periodic), and so on: the
 * mathematically periodic halo, which is what src/FHNmodel_torus.cpp:775-950 delivers when every process-grid
 * dimension is <= 2. */
int crd_oracle_rhs_global(const crd_oracle_problem *g, double t, const double *y, double *ydot, int d0, int d1, int nthreads)
{
	const long nx = g->nx, ny = g->ny;
	int nb = d0 * d1, b, rc = 0;
	crd_oracle_problem *sub;
	double **ly, **ld, **ws, **es, **ss, **ns;
	if (d0 < 1 || d1 < 1) return -1;

	if (nb == 1) {
		/* np = 1: every strip is the subdomain's own opposite edge; no copies of y. */
		double *w = (double *)malloc(sizeof(double) * 2 * (size_t)ny), *e = (double *)malloc(sizeof(double) * 2 * (size_t)ny);
		double *s = (double *)malloc(sizeof(double) * 2 * (size_t)nx), *n = (double *)malloc(sizeof(double) * 2 * (size_t)nx);
		if (!w || !e || !s || !n) rc = -1;
		else {
			crd_oracle_pack_edges(g, y, w, e, s, n);
			/* Wrecv <- Wsend of the W neighbour (itself), ... */
			rc = crd_oracle_rhs_subdomain(g, t, y, ydot, w, e, s, n, nthreads);
		}
		free(w); free(e); free(s); free(n);
		return rc;
	}

	sub = (crd_oracle_problem *)calloc((size_t)nb, sizeof(*sub));
	ly = (double **)calloc((size_t)nb * 6, sizeof(double *));
	if (!sub || !ly) { free(sub); free(ly); return -1; }
	ld = ly + nb; ws = ld + nb; es = ws + nb; ss = es + nb; ns = ss + nb;
	for (b = 0; b < nb; b++) {
		int c0 = b / d1, c1 = b % d1; /* MPI_Cart row-major rank order */
		long nxl, nyl, i, j;
		sub[b] = *g;
		crd_oracle_decomp(nx, ny, d0, d1, c0, c1, &sub[b].is, &sub[b].ie, &sub[b].js, &sub[b].je);
		nxl = sub[b].ie - sub[b].is + 1;
		nyl = sub[b].je - sub[b].js + 1;
		ly[b] = (double *)malloc(sizeof(double) * 2 * (size_t)(nxl * nyl));
		ld[b] = (double *)malloc(sizeof(double) * 2 * (size_t)(nxl * nyl));
		ws[b] = (double *)malloc(sizeof(double) * 2 * (size_t)nyl);
		es[b] = (double *)malloc(sizeof(double) * 2 * (size_t)nyl);
		ss[b] = (double *)malloc(sizeof(double) * 2 * (size_t)nxl);
		ns[b] = (double *)malloc(sizeof(double) * 2 * (size_t)nxl);
		if (!ly[b] || !ld[b] || !ws[b] || !es[b] || !ss[b] || !ns[b]) { rc = -1; continue; }
		for (j = 0; j < nyl; j++)
			for (i = 0; i < nxl; i++) {
				long gidx = 2 * (sub[b].is + i) + 2 * (sub[b].js + j) * nx;
				ly[b][LIDX(i, j, nxl)] = y[gidx];
				ly[b][LIDX(i, j, nxl) + 1] = y[gidx + 1];
			}
		crd_oracle_pack_edges(&sub[b], ly[b], ws[b], es[b], ss[b], ns[b]);
	}
	for (b = 0; b < nb && rc == 0; b++) {
		int c0 = b / d1, c1 = b % d1;
		int bw = ((c0 - 1 + d0) % d0) * d1 + c1; /* holds column is-1 */
		int be = ((c0 + 1) % d0) * d1 + c1;      /* holds column ie+1 */
		int bs = c0 * d1 + (c1 - 1 + d1) % d1;   /* holds row js-1 */
		int bn = c0 * d1 + (c1 + 1) % d1;        /* holds row je+1 */
		long nxl = sub[b].ie - sub[b].is + 1, nyl = sub[b].je - sub[b].js + 1, i, j;
		rc = crd_oracle_rhs_subdomain(&sub[b], t, ly[b], ld[b], ws[bw], es[be], ss[bs], ns[bn], nthreads);
		for (j = 0; j < nyl; j++)
			for (i = 0; i < nxl; i++) {
				long gidx = 2 * (sub[b].is + i) + 2 * (sub[b].js + j) * nx;
				ydot[gidx] = ld[b][LIDX(i, j, nxl)];
				ydot[gidx + 1] = ld[b][LIDX(i, j, nxl) + 1];
			}
	}
	for (b = 0; b < nb; b++) { free(ly[b]); free(ld[b]); free(ws[b]); free(es[b]); free(ss[b]); free(ns[b]); }
	free(ly);
	free(sub);
	return rc;
}

/* Classical RK4 around the faithful RHS; replaces the adaptive ARKode loop (src/FHNmodel_torus.cpp:420-435),
 * which cannot be reproduced here (SUNDIALS absent).  Stage times are what f() compares with TBOUNDARY (:643). */
int crd_oracle_rk4(const crd_oracle_problem *g, double *y, double t0, double dt, long nsteps, int nthreads)
{
	const size_t n = 2 * (size_t)g->nx * (size_t)g->ny;
	double *k1 = (double *)malloc(sizeof(double) * n), *k2 = (double *)malloc(sizeof(double) * n);
	double *k3 = (double *)malloc(sizeof(double) * n), *k4 = (double *)malloc(sizeof(double) * n);
	double *ys = (double *)malloc(sizeof(double) * n);
	long s;
	int rc = 0;
	if (!k1 || !k2 || !k3 || !k4 || !ys) rc = -1;
	for (s = 0; s < nsteps && rc == 0; s++) {
		const double t = t0 + (double)s * dt;
		long q;
		rc |= crd_oracle_rhs_global(g, t, y, k1, 1, 1, nthreads);
#pragma omp parallel for num_threads(nthreads) schedule(static)
		for (q = 0; q < (long)n; q++) ys[q] = y[q] + (0.5 * dt) * k1[q];
		rc |= crd_oracle_rhs_global(g, t + 0.5 * dt, ys, k2, 1, 1, nthreads);
#pragma omp parallel for num_threads(nthreads) schedule(static)
		for (q = 0; q < (long)n; q++) ys[q] = y[q] + (0.5 * dt) * k2[q];
		rc |= crd_oracle_rhs_global(g, t + 0.5 * dt, ys, k3, 1, 1, nthreads);
#pragma omp parallel for num_threads(nthreads) schedule(static)
		for (q = 0; q < (long)n; q++) ys[q] = y[q] + dt * k3[q];
		rc |= crd_oracle_rhs_global(g, t + dt, ys, k4, 1, 1, nthreads);
#pragma omp parallel for num_threads(nthreads) schedule(static)
		for (q = 0; q < (long)n; q++) y[q] = y[q] + (dt / 6.0) * (k1[q] + 2.0 * k2[q] + 2.0 * k3[q] + k4[q]);
	}
	free(k1); free(k2); free(k3); free(k4); free(ys);
	return rc;
}
