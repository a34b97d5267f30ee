/*
 * crd_oracle.h -- CPU restatement of CRDModel's RHS hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.  The product
 * (libcrd.so, crdmodel_amd/) never links, loads or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (the four programs under /root/reference/src) needs SUNDIALS 2.x (ARKode +
 * NVECTOR_PARALLEL), Boost.PropertyTree and MPI; SUNDIALS and Boost are absent from this image, so the
 * reference is unbuildable here, and it ships no tests, golden vectors or fixtures.  This restatement is
 * therefore pinned only by (i) an independent numpy restatement (oracle/crd_oracle_np.py), (ii) analytic
 * known answers the reference's own code implies (steady states, index-space eigenfunctions), and
 * (iii) the shipped .ini parameter sets under data/, including the one result they state (the Goldbeter kinetics are
 * "oscillatory when 0.28895 < beta < 0.77427", data/GoldbeterModelArgs.ini:25 -- reproduced to the digits given, as is the
 * FHN Hopf point beta = 1 of util/FHNmodel/plot_FHNmodel_torus.py:90-92; the torus's Gaussian curvature of
 * util/PlotGaussianAndCoupling.py:11-12, which fixes the two theta-dependent coefficients of the diffusion operator) -- not by
 * outputs of the reference itself.
 *
 * Every function cites the reference lines it follows (paths under /root/reference/).
 */
#ifndef CRD_ORACLE_H
#define CRD_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { CRD_ORACLE_FHN = 0, CRD_ORACLE_GOLDBETER = 1 };
enum { CRD_ORACLE_TORUS = 0, CRD_ORACLE_FLAT = 1 };

/* One subdomain's view of the problem: the union of the reference's UserData (src/FHNmodel_torus.cpp:97-122)
 * and its file-scope globals (:80-94) that f() reads. */
typedef struct crd_oracle_problem {
	int model;          /* CRD_ORACLE_FHN | CRD_ORACLE_GOLDBETER */
	int surface;        /* CRD_ORACLE_TORUS | CRD_ORACLE_FLAT */
	long nx, ny;        /* global mesh */
	long is, ie;        /* subdomain extents, inclusive (SetupDecomp :750-755) */
	long js, je;
	double dx, dy;
	double xmin, xmax, ymin, ymax;
	double R, r;        /* torus radii (unused for flat) */
	double diff;        /* DIFF */
	double beta, beta_min, beta_max;
	int vary_beta;
	int just_diffusion; /* Goldbeter only */
	double t_boundary;
} crd_oracle_problem;

/* Geometry scalars.  ny_override > 0 replaces the derived ny (the `phiMesh` extension key). */
int crd_oracle_geometry(int surface, double surface_length, double surface_width, long nx, long ny_override,
                        crd_oracle_problem *p);

/* Block extents of Cartesian coordinate (c0,c1) in a d0 x d1 process grid. */
void crd_oracle_decomp(long nx, long ny, int d0, int d1, int c0, int c1, long *is, long *ie, long *js, long *je);

/* Analytic stable states. */
void crd_oracle_fhn_steady(double beta, double *us, double *vs);
int crd_oracle_goldbeter_steady(double beta, double *zs, double *ys);

/* Initial conditions of the four programs, written into the subdomain-local AoS vector y. */
typedef struct crd_oracle_ic {
	double wave_length;  /* fraction of the phi / y extent */
	double wave_width;   /* fraction of the theta / x extent */
	int wave_inside;     /* torus only */
	int ic_type;         /* Goldbeter, varyBeta = 1: 0 homogeneous, 1 perturbation, 2 rand() */
	double s0, s1;       /* stable state of variable 0 / 1 (Us,Vs or Zs,Ys) */
} crd_oracle_ic;
int crd_oracle_initial_conditions(const crd_oracle_problem *p, const crd_oracle_ic *ic, double *y);

/* Faithful RHS of one subdomain: zero fill, diffusion with halo strips, kinetics.  y / ydot are the local
 * AoS vectors (IDX(i,j) = 2 i + 2 j nxl); the four strips are laid out exactly like the reference's receive
 * buffers (u of row/column k at [2k]).  nthreads > 1 splits the j loops with OpenMP. */
int crd_oracle_rhs_subdomain(const crd_oracle_problem *p, double t, const double *y, double *ydot,
                             const double *wrecv, const double *erecv, const double *srecv, const double *nrecv,
                             int nthreads);

/* The four send strips Exchange() packs from y. */
void crd_oracle_pack_edges(const crd_oracle_problem *p, const double *y, double *wsend, double *esend,
                           double *ssend, double *nsend);

/* RHS of the whole periodic domain evaluated through a d0 x d1 decomposition with mathematically periodic
 * halos (what the reference does at np in {1,2,4}).  y / ydot are global AoS vectors. */
int crd_oracle_rhs_global(const crd_oracle_problem *global, double t, const double *y, double *ydot, int d0, int d1,
                          int nthreads);

/* Classical RK4 on the whole domain (np = 1), nsteps steps of size dt from t0; t_n = t0 + n dt. */
int crd_oracle_rk4(const crd_oracle_problem *global, double *y, double t0, double dt, long nsteps, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
