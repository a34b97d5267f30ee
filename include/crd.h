/*
 * crd.h -- C ABI of libcrd: the MI355X-native replacement for CRDModel's per-timestep RHS path.
 *
 * Everything here is plain C: POD structs, pointers and sizes, `int` status returns (0 = ok, negative =
 * crd_status), no exceptions cross the boundary, no torch/HIP types in any signature.  Each entry point names
 * the reference interface it replaces (paths under the reference tree, e.g. src/FHNmodel_torus.cpp).
 *
 * Field layout at the boundary is the reference's own: state vectors are AoS pairs [var0, var1] per grid point,
 * theta / x (index i) fastest, IDX(i,j) = 2 i + 2 j nxl (src/FHNmodel_torus.cpp:60); var0 is the diffusing
 * variable (FHN u, Goldbeter Z), var1 the local one (FHN v, Goldbeter Y).  A context owns one phi-slab
 * [js, je] x [0, nx-1] of the global grid (the reference's subdomain with dims = {1, G}).
 */
#ifndef CRD_H
#define CRD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRD_ABI_VERSION 6

typedef enum crd_status {
	CRD_OK = 0,
	CRD_EINVAL = -1,   /* bad argument / inconsistent parameters */
	CRD_ENOMEM = -2,   /* host or device allocation failed */
	CRD_EHIP = -3,     /* a HIP runtime call failed (no GPU, launch error, ...) */
	CRD_ERCCL = -4,    /* an RCCL call failed */
	CRD_EIO = -5,      /* file could not be read / written */
	CRD_EPARSE = -6,   /* malformed ini file or missing mandatory key */
	CRD_ESTATE = -7    /* call not valid in the context's current state */
} crd_status;

enum { CRD_MODEL_FHN = 0, CRD_MODEL_GOLDBETER = 1 };
enum { CRD_SURFACE_TORUS = 0, CRD_SURFACE_FLAT = 1 };
enum { CRD_PRECISION_F64 = 0, CRD_PRECISION_F32 = 1 };

/* Which RK4 implementation crd_step_rk4 runs (results agree to round-off). */
enum {
	CRD_STEPPER_AUTO = 0,
	CRD_STEPPER_STAGED = 1, /* four stage kernels per step, one halo row exchanged per stage */
	CRD_STEPPER_FUSED = 2   /* one kernel per step (all four stages on chip); multi-slab: 4 E halo rows every E steps (crd_set_exchange_period); slabs shorter than that use the staged kernels */
};

/* Halo transport between the slabs of one run. */
enum {
	CRD_HALO_SELF = 0,  /* single slab: periodic wrap inside the kernels */
	CRD_HALO_LOCAL = 1, /* several contexts in one process (any devices): device-to-device copies */
	CRD_HALO_RCCL = 2   /* one context per process / GPU: ncclSend / ncclRecv over xGMI */
};

/* ---------------------------------------------------------------------------------------------------------
 * Parameters of the hot path: what f() reads from UserData and from the file-scope globals
 * (src/FHNmodel_torus.cpp:80-94,97-122; src/GoldbeterModel_torus.cpp:91-106).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct crd_params {
	int32_t model;          /* CRD_MODEL_* */
	int32_t surface;        /* CRD_SURFACE_* */
	int64_t nx;             /* thetaMesh / xMesh */
	int64_t ny;             /* 0 = derive as the reference does (:193 / flat :190-192); >0 = phiMesh override */
	double surface_length;  /* surfaceLength: major circumference / flat length */
	double surface_width;   /* surfaceWidth:  minor circumference / flat width */
	double diffusion;       /* DIFF */
	double beta;            /* BETA */
	double beta_min;        /* BETAMIN */
	double beta_max;        /* BETAMAX */
	int32_t vary_beta;      /* VARYBETA */
	int32_t just_diffusion; /* JUST_DIFFUSION (Goldbeter) */
	double t_boundary;      /* TBOUNDARY */
	int32_t precision;      /* CRD_PRECISION_*: arithmetic and storage type on the device */
	int32_t reserved;
} crd_params;

/* Derived geometry (src/FHNmodel_torus.cpp:186-193,233-234; src/FHNmodel_flat.cpp:172-175,190-192,230-231). */
typedef struct crd_grid {
	int64_t nx, ny;
	double dx, dy;
	double xmin, xmax, ymin, ymax;
	double R, r;            /* torus radii; 0 for flat */
} crd_grid;

/* Driver-level configuration: everything main() reads from the ini file (src/FHNmodel_torus.cpp:158-174,
 * src/FHNmodel_flat.cpp:157-170, src/GoldbeterModel_torus.cpp:174-187, src/GoldbeterModel_flat.cpp:169-184),
 * plus this build's extension keys. */
typedef struct crd_run_config {
	crd_params params;
	double wave_length;       /* waveLength */
	double wave_width;        /* waveWidth */
	int32_t wave_inside;      /* waveInside (torus) */
	int32_t output_timestep;  /* outputTimestep = Nt */
	double t_final;           /* tFinal */
	int32_t include_all_vars; /* includeAllVars */
	int32_t ic_type;          /* icType (Goldbeter flat) */
	/* extensions (absent keys take these defaults) */
	double dt;                /* [Solver] dt: fixed RK4 step; 0 = safety * crd_stable_dt */
	double dt_safety;         /* [Solver] dtSafety, default 0.8 */
	int32_t n_gpus;           /* [Solver] gpus, default 1 */
	int32_t stepper;          /* [Solver] stepper: CRD_STEPPER_* */
	int32_t adaptive;         /* [Solver] adaptive: 0 = fixed dt; 1 = error-controlled steps with ARKode's default explicit pair and controller
	                           * (CRD_ADAPT_ARKODE: the reference's integrator); 2 = the RK4(3) pair of earlier rounds (CRD_ADAPT_RK43) */
	int32_t steady_state_decimals; /* [Solver] steadyStateDigits (crd_run --ref-steady-state = 8): 0 = the exact Goldbeter fixed point;
	                                * n > 0 = that fixed point as the reference receives it, through numpy's print of a one-element
	                                * array (n digits behind the decimal point; numpy's default is 8) and fscanf
	                                * (crd_steady_state_as_printed) */
	double rtol, atol;        /* [Solver] rtol / atol, defaults 1e-5 / 1e-10 (src/FHNmodel_torus.cpp:197-198) */
	int32_t exchange_period;  /* [Solver] exchangePeriod: fused steps between two halo exchanges of a multi-slab run (crd_set_exchange_period, 3 .. 16);
	                           * 0 (default) = the driver chooses: 10 where every slab has at least 256 rows, else 8 */
	int32_t reserved;
} crd_run_config;

typedef struct crd_ctx crd_ctx;

/* ---------------------------------------------------------------------------------------------------------
 * Host-side helpers (no GPU needed).
 * --------------------------------------------------------------------------------------------------------- */

int crd_abi_version(void);
const char *crd_status_string(int status);
/* 16 hex digits over the step kernels of THIS build as the assembler printed them (template arguments, registers, occupancy, the
 * steady-state loop's instruction mix: tools/kernel_regs.py writes them into the library when it is built); "" for a build without the
 * table.  The profile tables a run quotes counters from (profiles/pmc_traffic.json, plan_stats.json) carry the digest of the build they
 * were measured on; bench.py quotes them only when it equals this one.  No reference counterpart. */
const char *crd_kernel_table_digest(void);

/* Replaces boost::property_tree::ini_parser::read_ini + the pt.get<T>() block of main()
 * (src/FHNmodel_torus.cpp:158-174 and the three siblings).  The program's own mesh key is preferred
 * (thetaMesh for FHN, xMesh for Goldbeter) but either is accepted; keys the reference program does not read
 * are optional; a missing mandatory key is CRD_EPARSE (the reference aborts on ptree_bad_path).  err (may be
 * NULL) receives a message. */
int crd_config_load_ini(const char *path, int model, int surface, crd_run_config *cfg, char *err, size_t err_len);

/* Geometry scalars a9: r, R, ny, dx, dy, domain bounds. */
int crd_grid_from_params(const crd_params *p, crd_grid *g);

/* phi-slab extents: SetupDecomp (src/FHNmodel_torus.cpp:750-755) with dims = {1, n_slabs}. */
int crd_slab_extents(int64_t ny, int slab, int n_slabs, int64_t *js, int64_t *je);

/* The reference's own 2-D (theta x phi) block decomposition: MPI_Dims_create(nprocs, 2, dims) as SetupDecomp
 * calls it (src/FHNmodel_torus.cpp:724-728: 4 -> {2, 2}, 8 -> {4, 2}, 2 -> {2, 1}), and the extents of block (c0, c1) of a d0 x d1
 * process grid (:750-753: is = nx c0 / d0, ie = nx (c0 + 1) / d0 - 1, likewise js, je).  The rank of a block is MPI's Cartesian
 * rank, c0 d1 + c1.  Blocks with d0 > 1 step with the staged kernels (one ghost row / column strip of var0 per stage, the four
 * strips of the reference's Exchange(), :775-950) in LOCAL groups; the one-launch stepper, the adaptive integrators and the RCCL
 * transport need phi-slabs (d0 = 1), which is the layout to use on one node -- the block layout exists so that `-np 4` of the
 * reference's scripts (util/ShellScripts/runFHNmodelTorus.sh:6) can be reproduced file for file. */
int crd_dims_create(int nprocs, int *d0, int *d1);
int crd_block_extents(int64_t nx, int64_t ny, int c0, int d0, int c1, int d1, int64_t *is, int64_t *ie, int64_t *js, int64_t *je);

/* Stable state used by the initial conditions and the banner: FHN analytic (src/FHNmodel_torus.cpp:242-244);
 * Goldbeter fixed point, computed natively instead of popen("SolveGoldbeterODE.py")
 * (src/GoldbeterModel_torus.cpp:254-261). */
int crd_steady_state(int model, double beta, double *s0, double *s1);

/* The same state the way the reference's Goldbeter programs receive it: `print Z[-1], Y[-1]` of one-element numpy arrays
 * (util/GoldbeterModel/SolveGoldbeterODE.py:111; numpy prints `decimals` = 8 digits behind the decimal point, e.g.
 * "[ 0.392] [ 1.64562147]") read back by fscanf("[%lf] [%lf]", ...) (src/GoldbeterModel_torus.cpp:254-261).  decimals = 0, or
 * the FHN model (computed in C++ by the reference, :242-244): no rounding.  What this cannot reproduce is the error of the
 * script's own BDF integration towards that point (scipy VODE at its default rtol 1e-6 over 50 time units): the reference's
 * printed digits agree with these to about that tolerance, not to the last of the eight. */
int crd_steady_state_as_printed(int model, double beta, int decimals, double *s0, double *s1);

/* Initial conditions of rows [js, je] in the boundary layout (AoS doubles, 2*nx*(je-js+1) values):
 * src/FHNmodel_torus.cpp:285-354, src/FHNmodel_flat.cpp:280-319, src/GoldbeterModel_torus.cpp:313-414,
 * src/GoldbeterModel_flat.cpp:309-379. */
int crd_initial_conditions(const crd_run_config *cfg, int64_t js, int64_t je, double *y_aos);
/* ... of the block [is, ie] x [js, je] (2 (ie-is+1) (je-js+1) values, IDX with nxl = ie-is+1).  The rand() rule draws row by row
 * through the block from the default seed, as every reference rank does. */
int crd_initial_conditions_block(const crd_run_config *cfg, int64_t is, int64_t ie, int64_t js, int64_t je, double *y_aos);

/* Largest step the classical RK4 scheme takes stably on this problem: 2.785 / lambda_max (2.785 = the extent of RK4's stability
 * region on the negative real axis), with
 *   lambda_max = 4 D (1/(r dx)^2 + 1/((R - r) dy)^2) + D / (r (R - r) dx)      torus: the second differences at theta = pi, where
 *                                                                              the phi spacing is smallest, plus the advective term
 *              = 4 D (1/dx^2 + 1/dy^2)                                         flat
 *              + a bound on the reaction Jacobian: 10 for FitzHugh-Nagumo (|3 - 3 u^2| <= 9 for |u| <= 2, which contains the
 *                limit cycle, + 1), 400 for Goldbeter (the spectral radius of the kinetics' Jacobian over 0 <= Z <= 1.5,
 *                0 <= Y <= 3 is 333, the VM3 Hill term's slope in Z reaching 410; rounded up).  On the BASELINE grids the reaction
 *                term is 0.05 % (FHN) / 2 % (Goldbeter at 4096^2) of lambda_max; it decides on coarse grids (the shipped 100 x 400
 *                Goldbeter run).
 * crd_run's default step is dtSafety (0.8) times this, and it is the default cap of crd_integrate_adaptive (h_max = 0).
 * No reference counterpart: ARKode finds its steps by error control (src/FHNmodel_torus.cpp:365). */
double crd_stable_dt(const crd_params *p);

/* Output files of one slab, byte-compatible with src/FHNmodel_torus.cpp:376-410,438-455:
 * <Model>_<surface>_subdomain.%03i.txt, <Model>_<surface>_<var0>.%03i.txt, <..>_<var1>.%03i.txt in `dir`. */
typedef struct crd_writer crd_writer;
int crd_writer_open(const crd_run_config *cfg, const char *dir, int slab, int n_slabs, crd_writer **out);
int crd_writer_open_block(const crd_run_config *cfg, const char *dir, int rank, int c0, int d0, int c1, int d1, crd_writer **out); /* block (c0, c1) of d0 x d1 */
int crd_writer_write_row(crd_writer *w, const double *y_aos); /* one output time: nyl*nxl values per file */
int crd_writer_close(crd_writer *w);

/* Binary side-channel next to the text files (no reference counterpart: src/FHNmodel_torus.cpp:438-455 writes 24 characters per
 * value, 1.6 GB per output time at 8192^2): <Model>_<surface>_<var>.%03i.npy, a NumPy array of shape (frames, nyl, nxl) in the
 * text file's own ordering (frame = output time, then j, then i), value_bytes = 8 or 4 per value, little endian.  `frame` is a
 * contiguous (nyl, nxl) block, e.g. what crd_state_download_rows(ctx, var, 0, nyl, frame) delivers. */
typedef struct crd_npy_writer crd_npy_writer;
int crd_npy_writer_open(const crd_run_config *cfg, const char *dir, int slab, int n_slabs, int var, int value_bytes, crd_npy_writer **out);
int crd_npy_writer_append(crd_npy_writer *w, const void *frame);
int crd_npy_writer_close(crd_npy_writer *w);

/* ---------------------------------------------------------------------------------------------------------
 * Device context: replaces UserData + InitUserData / SetupDecomp / FreeUserData
 * (src/FHNmodel_torus.cpp:97-122,708-772,953-997).  Not thread-safe; calls on one context must be serialised.
 * --------------------------------------------------------------------------------------------------------- */

/* HIP devices visible to this process (0 when there is none or the runtime cannot start). */
int crd_device_count(void);

/* slab / n_slabs: which phi-slab of the global grid this context owns; device: HIP device ordinal. */
int crd_create(const crd_params *p, int slab, int n_slabs, int device, crd_ctx **out);
/* Block (c0, c1) of a d0 x d1 decomposition (crd_create(p, slab, n, ...) = crd_create_block(p, 0, 1, slab, n, ...)); group arrays
 * (crd_comm_attach_local, crd_group_*) hold the blocks in rank order, c0 d1 + c1. */
int crd_create_block(const crd_params *p, int c0, int d0, int c1, int d1, int device, crd_ctx **out);
int crd_get_block(const crd_ctx *ctx, int64_t *is, int64_t *ie, int64_t *js, int64_t *je);
void crd_destroy(crd_ctx *ctx);
const char *crd_last_error(const crd_ctx *ctx); /* never NULL; ctx may be NULL (creation errors) */

int crd_get_grid(const crd_ctx *ctx, crd_grid *g);
int crd_get_slab(const crd_ctx *ctx, int64_t *js, int64_t *je);

/* Multi-slab wiring.  LOCAL: give every context of the run the full array (same process).
 * RCCL: rank r of n_slabs ranks calls crd_comm_init_rccl with the 128-byte id rank 0 obtained from
 * crd_comm_unique_id and distributed by any means (MPI_Bcast, torch.distributed, a file).  The RCCL entry points are bound at
 * first use from librccl.so.1 -- or from the file given to crd_comm_set_rccl_library before that first use (another RCCL
 * build; NULL or "" = the default again; CRD_ESTATE once bound). */
int crd_comm_attach_local(crd_ctx *const *ctxs, int n_slabs);
int crd_comm_set_rccl_library(const char *path);
int crd_comm_unique_id(void *id128);
int crd_comm_init_rccl(crd_ctx *ctx, const void *id128);

/* What the context's transport is and, for RCCL, what the communicator itself reports (ncclCommCount / ncclCommUserRank):
 * halo = CRD_HALO_* (-1 while a multi-slab context is not wired yet), ranks / rank of the ring.  Any pointer may be NULL. */
int crd_comm_info(const crd_ctx *ctx, int *halo, int *ranks, int *rank);

/* ONE halo exchange of the resident state, outside any step: fills ghost rows [-depth, 0) and [nyl, nyl+depth) of both
 * fields from the ring neighbours (the N/S half of Exchange(), src/FHNmodel_torus.cpp:775-950, at the depth the fused
 * stepper uses) and waits for it.  1 <= depth <= 64.  Every rank of an RCCL run must make the same call.  Together with
 * crd_state_download_rows it lets a host program verify the transport (ghost rows == the neighbours' owned rows). */
int crd_halo_exchange(crd_ctx *ctx, int depth);

/* Rows [row_begin, row_begin + row_count) of ONE field (var 0 / 1) of the resident state, ghost rows included
 * (-64 <= row_begin, row_begin + row_count <= nyl + 64), as contiguous rows of nx reals in the DEVICE precision. */
int crd_state_download_rows(crd_ctx *ctx, int var, int64_t row_begin, int64_t row_count, void *rows_host);

/* The ring protocol of one halo exchange, as data: the four point-to-point operations slab `slab` of `n_slabs` issues,
 * in issue order, to fill its ghost rows [-depth, 0) and [nyl, nyl+depth) from its periodic phi neighbours (replaces the
 * N/S half of Exchange(), src/FHNmodel_torus.cpp:775-950; the E/W half disappears because a slab spans all of theta).
 * Rows are local indices (ghost rows are negative or >= nyl).  The order matters when both neighbours are the same
 * rank (n_slabs <= 2): operations between one pair of ranks match in issue order, so the LAST rows are sent first and
 * the LOW ghost rows are received first.  The RCCL transport issues exactly this plan inside one ncclGroup; the CPU
 * tests drive it over gloo. */
typedef struct crd_halo_op {
	int32_t is_send;    /* 1 = send owned rows, 0 = receive into ghost rows */
	int32_t peer;       /* slab index of the other side */
	int64_t row_begin;  /* first local row */
	int64_t row_count;
} crd_halo_op;
int crd_halo_plan(int slab, int n_slabs, int64_t nyl, int depth, crd_halo_op ops[4]);

/* How the ranks of an RCCL ring agree, at the start of a stepping call, where in the deep-halo exchange cycle the ring stands --
 * as data, like crd_halo_plan, so that the rule can be driven over any transport (the CPU tests use gloo).  Each rank votes
 * crd_cycle_vote(pos): pos = steps its resident state has taken since its ghost rows were last exchanged (0 .. E - 1), or -1 for a
 * state whose ghost rows cannot be trusted (new upload, other stepper, failed call).  The element-wise MIN of the votes over
 * the ranks (ncclAllReduce in the library) goes to crd_cycle_agreed: the common position when every rank voted the same
 * non-negative one -- the call carries on there -- else -1: every rank starts with an exchange.  No reference counterpart: the
 * reference exchanges inside every f() (src/FHNmodel_torus.cpp:521). */
int crd_cycle_vote(int pos, double vote[2]);
int crd_cycle_agreed(const double reduced[2]);

/* State transfer in the boundary layout.  host_is_f64 = 1: host buffer holds doubles whatever the device
 * precision (converted on the device); 0: host buffer holds the device precision. */
int crd_state_upload(crd_ctx *ctx, const void *y_aos_host, int host_is_f64);
int crd_state_download(crd_ctx *ctx, void *y_aos_host, int host_is_f64);

/* Page-locked host memory for the vectors handed to crd_rhs_host (e.g. as the data array of N_VMake_Parallel).  With
 * pinned y and ydot a single-slab crd_rhs_host streams the slab band by band -- upload of band k+1, kernel on band k and
 * download of band k-1 overlap, both directions of the host link busy -- instead of copy, compute, copy; pageable
 * vectors work too, at the rate of a staged copy each way.  NULL when the allocation fails. */
void *crd_host_alloc(size_t bytes);
void crd_host_free(void *p);

/* One RHS evaluation, the ARKRhsFn `f(t, y, ydot, user_data)` of src/FHNmodel_torus.cpp:126,504-667 (and
 * siblings): halo exchange + diffusion + kinetics, writes every element of ydot, does not modify y.
 * y / ydot are this slab's AoS vectors in the device precision; *_host takes host pointers (staged through
 * device memory), *_device takes device pointers on the context's device.  Returns 0, or <0 like the
 * reference's f() returns -1 when Exchange fails (:522). */
int crd_rhs_host(crd_ctx *ctx, double t, const void *y_aos, void *ydot_aos);
int crd_rhs_device(crd_ctx *ctx, double t, const void *y_aos_dev, void *ydot_aos_dev);

/* Fast path that never leaves the GPU: nsteps classical RK4 steps of size dt on the context's resident state,
 * stage k of step n evaluated at t0 + n dt + c_k dt (replaces the ARKode(...) call, src/FHNmodel_torus.cpp:423).
 * Asynchronous; crd_synchronize() waits.  With several slabs every context of the run must make the same call: stepping is
 * COLLECTIVE (same t0, dt, nsteps, stepper on every rank, like the reference's ARKode call on every MPI rank).  Calls that only
 * change one rank's state need not be: crd_state_upload on one rank (a stimulus injected by its owner), or a call that failed
 * there.  The one-launch stepper carries its deep-halo exchange cycle across calls, so under RCCL every stepping call begins
 * with a two-value ncclAllReduce in which the ranks agree where in the cycle the ring stands; if any rank holds a new state,
 * all of them start afresh with an exchange (crd_step_timing.agreement_restarts counts those).  The reduction overlaps the
 * call's first step except when that step is the one that exchanges. */
int crd_set_stepper(crd_ctx *ctx, int stepper);
int crd_step_rk4(crd_ctx *ctx, double t0, double dt, int64_t nsteps);

/* The exchange period E of the one-launch stepper on several slabs: E steps between two halo exchanges, each of 4 E ghost rows
 * of both fields; in between every slab recomputes the shrinking ghost region redundantly (same kernel, same inputs: bit-identical
 * to what the owner computes).  3 <= E <= 16; default 10 where every slab of the run has 256 rows or more, else 8.  A property of the RUN: every context of a LOCAL group / every rank of a
 * ring must be given the same value before its next stepping call (which then starts with an exchange).  A longer period halves the
 * per-step share of a cycle's fixed cost (two small launches, two cross-stream waits) and of the exchange's latency for a few per
 * cent more redundant rows; bench.py rehearses 8 and 16 on the machine at hand.  Slabs shorter than 4 E rows step with the staged
 * kernels.  Replaces nothing in the reference: its Exchange() runs inside every f() (src/FHNmodel_torus.cpp:521). */
int crd_set_exchange_period(crd_ctx *ctx, int steps);
int crd_get_exchange_period(const crd_ctx *ctx);
int crd_synchronize(crd_ctx *ctx);

/* Error-controlled integration from t0 to exactly tout on the resident state: replaces what the reference gets from
 * ARKodeSStolerances(rtol, atol) + ARKode(..., tout, ..., ARK_NORMAL) (src/FHNmodel_torus.cpp:365,423).  The propagated
 * solution is classical RK4 (the same one-launch step kernel as crd_step_rk4); a fifth evaluation k5 = f(t+h, y_new) gives
 * the embedded third-order solution y + h (k1/6 + k2/3 + k3/3 + k5/6) and the local error estimate h (k4 - k5)/6, measured
 * in ARKode's WRMS norm with weights 1 / (rtol |y_n| + atol); a step is accepted when bias * norm <= 1 and the next step
 * is safety * h * (bias * norm)^(-1/4), growth-limited (constants in crd_adaptive_options; defaults are ARKode's).
 * Not ARKode's step sequence: its method table and controller are not in the reference tree.
 * Output times: with dense_output = 0 the last step is shortened to land on tout.  With dense_output = 1 the call does what
 * ARK_NORMAL does (src/FHNmodel_torus.cpp:423): steps are never shortened for an output time; the integrator steps until it has
 * reached or passed tout, the state handed back (crd_state_download, the writers) is the cubic Hermite interpolant of the last
 * step at tout, built from y_n, y_{n+1}, f(t_n, y_n), f(t_{n+1}, y_{n+1}) (ARKode's default dense output is of degree 3 too), and
 * the integrator's own state stays at its internal time (stats.t_internal >= tout): the next call with t0 equal to this call's
 * tout continues from there.  Any other call that changes the state (upload, fixed-step stepping, a different t0) drops it.
 * Multi-slab runs exchange 60 ghost rows of the state every twelve
 * accepted steps (each attempt also produces the ghost-region rows its input still covers: the fixed stepper's deep halo, by
 * attempts) and reduce the norm of the OWNED rows over the ring (ncclAllReduce; LOCAL groups add the slabs' sums on the host in
 * slab order), so every rank takes the same decisions.  h0 = 0 starts from the diffusion-stability step; by default steps are
 * also capped at that bound (h_max = 0), which removes the reject / regrow cycle of a stability-limited explicit method. */
/* Which embedded pair and controller crd_integrate_adaptive runs. */
enum {
	CRD_ADAPT_RK43 = 0,   /* rounds 1-2: classical RK4 + k5 = f(t+h, y_new) as third-order embedding, I-controller; steps may be shortened to
	                       * land on tout (dense_output = 0) */
	CRD_ADAPT_ARKODE = 1  /* the integrator the reference uses (src/FHNmodel_torus.cpp:356-372): SUNDIALS ARKode's default explicit
	                       * fourth-order table, Zonneveld 5(3)4 -- whose propagated solution is classical RK4 too -- with ARKode's PID
	                       * controller, its safeguards, its initial-step estimate and ARK_NORMAL output (dense_output is taken as 1).
	                       * Restated from ARKode's published documentation (the library is not in the reference tree): the algorithm
	                       * and every constant are written out in oracle/arkode_erk.py.  The controller's memory (step, error history)
	                       * lives in the context and carries from call to call like ARKode's does, until the state is replaced. */
};

typedef struct crd_adaptive_options {
	double rtol, atol;      /* 1e-5, 1e-10 in the reference (src/FHNmodel_torus.cpp:197-198) */
	double h0;              /* first step; 0 = automatic */
	double safety;          /* 0.96 */
	double bias;            /* 1.5 */
	double growth;          /* 20: largest h_new / h */
	double shrink;          /* 0.1: smallest h_new / h */
	int64_t max_steps;      /* 200000 attempts (ARKodeSetMaxNumSteps, :372) */
	double h_max;           /* largest step: > 0 explicit cap (ARKodeSetMaxStep); 0 = the classical-RK4 stability bound of the
	                         * diffusion operator, crd_stable_dt (what ARKodeSetStabilityFn is for: beyond it the error test
	                         * only finds out by failing); < 0 = no cap, error control alone (ARKode's default) */
	int32_t dense_output;   /* 0: shorten the last step to hit tout; 1: ARK_NORMAL -- overshoot and interpolate back (see above) */
	int32_t method;         /* CRD_ADAPT_*; crd_adaptive_defaults: CRD_ADAPT_ARKODE with dense_output = 1.  Under CRD_ADAPT_ARKODE h0 = 0 means
	                         * ARKode's own estimate (arkHin) on a fresh state, safety / bias / growth / shrink are ARKode's constants of the
	                         * same names (shrink = ETAMIN), and the remaining ones (PID gains 0.58 / 0.21 / 0.1 over the embedding order 3,
	                         * first-step growth 10000, 0.3 after repeated failures, at most 7 failures per step, no change of h for a
	                         * suggested growth within [1, 1.5]) are fixed */
} crd_adaptive_options;
typedef struct crd_adaptive_stats {
	int64_t accepted, rejected;
	double h_last;          /* size of the last accepted step that was not shortened to hit tout */
	double h_next;          /* controller's suggestion for the next call */
	double h_min, h_max;    /* over accepted steps */
	double err_last;        /* bias * WRMS norm of the last attempt */
	double t;               /* time of the state handed back (== tout on success) */
	double t_internal;      /* time the integrator itself has reached: == t without dense output, >= t with it */
	double h_first;         /* size of the first step this call attempted (on a fresh state under CRD_ADAPT_ARKODE: the arkHin estimate) */
	int64_t launched_ahead; /* attempts that were already running when their predecessor's error norm arrived (CRD_ADAPT_ARKODE launches the next
	                         * step ahead of the verdict, assuming "accepted, same size"; a wrong guess costs one discarded launch) */
} crd_adaptive_stats;
int crd_adaptive_defaults(crd_adaptive_options *opt);
int crd_integrate_adaptive(crd_ctx *ctx, double t0, double tout, const crd_adaptive_options *opt, crd_adaptive_stats *stats);
int crd_group_integrate_adaptive(crd_ctx *const *ctxs, int n, double t0, double tout, const crd_adaptive_options *opt,
                                 crd_adaptive_stats *stats); /* LOCAL groups */

/* LOCAL groups (several slabs driven by one host thread): the same two operations on every slab of the run in
 * lockstep; ctxs[k] must be slab k of n.  y[k] / ydot[k] are device pointers on ctxs[k]'s device. */
int crd_group_step_rk4(crd_ctx *const *ctxs, int n, double t0, double dt, int64_t nsteps);
/* ... with the clocks of crd_step_rk4_timed (below) on every slab: device time of the batch and the duration of one full-height
 * launch of the dominant kernel per context; crd_get_step_timing(ctxs[k]) returns slab k's figures.  Returns when every slab's last
 * step is done. */
int crd_group_step_rk4_timed(crd_ctx *const *ctxs, int n, double t0, double dt, int64_t nsteps);
/* Host threads that issue a group's work in crd_group_step_rk4: 0 (default) = one per device the group is spread over, k >= 1 =
 * exactly k (contiguous runs of slabs per thread; k = 1: the calling thread alone).  The threads meet at a host rendezvous around
 * every halo exchange; results do not depend on k. */
int crd_group_set_threads(crd_ctx *const *ctxs, int n, int threads);
int crd_group_rhs_device(crd_ctx *const *ctxs, int n, double t, const void *const *y_aos_dev, void *const *ydot_aos_dev);
int crd_group_rhs_host(crd_ctx *const *ctxs, int n, double t, const void *const *y_aos, void *const *ydot_aos); /* host vectors, device precision */

/* Same as crd_step_rk4 but bracketed by HIP events on the context's compute stream; blocks until done.
 * ms_total: device time of the whole batch; kernel_ms: average duration of one launch of the dominant kernel
 * (the fused step kernel, or the stage-2/3 kernel of the staged stepper), measured by per-launch events on a
 * subset of the steps; launches: how many launches of that kernel one step makes. */
int crd_step_rk4_timed(crd_ctx *ctx, double t0, double dt, int64_t nsteps, double *ms_total, double *kernel_ms,
                       int *launches_per_step);

/* What the last crd_step_rk4_timed call measured, plus -- after crd_set_diagnostics(ctx, 1), on an RCCL context stepping with the
 * one-launch kernel -- two figures per deep-halo exchange of that call: how long the compute stream stood at its wait for the
 * halo (event pair around the wait; a few microseconds of queue latency when the exchange hid under the sweeps it is given, the
 * exposed remainder when it did not), and how long the exchange itself took on the second stream (from "its rows are ready" to
 * "last byte landed").  Replaces nothing in the reference: its Exchange() is eight blocking MPI_Waits inside f()
 * (src/FHNmodel_torus.cpp:904-946), all of it exposed.  Diagnostics put event records between sweeps that otherwise run back to
 * back; leave them off in runs whose rate is being measured. */
typedef struct crd_step_timing {
	double ms_total;            /* device time of the batch */
	double kernel_ms;           /* average duration of the timed launches of the dominant kernel */
	double exposed_halo_ms;     /* sum over halo_waits */
	double exchange_ms;         /* sum over exchanges */
	int64_t steps;
	int32_t halo_slack;         /* crd_set_halo_slack at the time of the call */
	int32_t timed_steps_per_launch; /* RK4 steps one of the timed launches advanced: 1, or 2 (a two-steps-per-launch plan); 0: none timed */
	int32_t halo_waits;         /* waits measured (at most 64 per call) */
	int32_t exchanges;          /* exchanges measured */
	int64_t agreement_restarts; /* stepping calls of this context that started afresh with an exchange because the ring's ranks stood
	                             * at different positions of the exchange cycle (see crd_step_rk4) */
} crd_step_timing;
int crd_set_diagnostics(crd_ctx *ctx, int on);
/* How many sweeps of rows that read owned rows only the one-launch stepper puts between an exchange and its wait for the halo:
 * 1 (default) splits the first step of an exchange cycle, 2 the first two -- one more small launch per cycle for a third sweep
 * of cover, for links on which the exchange does not land within two (exposed_halo_ms says so).  A rank-local choice: it moves
 * the point where THIS rank consumes its halo, not the exchange; results are bit-identical either way. */
int crd_set_halo_slack(crd_ctx *ctx, int sweeps);
int crd_get_step_timing(const crd_ctx *ctx, crd_step_timing *out);

/* Rows of the slab one timed launch of the dominant kernel covers (all of them for a single slab; the interior, i.e. all
 * but the edge rows / bands that are launched separately ahead of the halo exchange, for a multi-slab context). */
int crd_dominant_kernel_rows(const crd_ctx *ctx, int64_t *rows);

/* Name of the dominant kernel as it appears in a rocprofv3 kernel trace (static string). */
const char *crd_dominant_kernel_name(const crd_ctx *ctx);

/* Launch plan of the one-launch step kernel.  How a slab is cut into work items (32-row chunks, or chunks stretched so that
 * every workgroup is resident at once), how the items are dealt to the 8 XCDs, how many columns a lane steps and how the new
 * state is stored is measured on the device at hand, on the
 * context's first full-size step (a handful of extra launches of that step; every plan computes bit-identical results), unless
 * autotuning is off (crd_set_autotune(ctx, 0) or CRD_AUTOTUNE=0 in the environment): launches then take 32-row chunks in dispatch
 * order, plain stores and the default columns per lane of their precision (fp32 on an even nx: two; crd_get_launch_plan reports
 * it).  crd_set_autotune(ctx, 2) / CRD_AUTOTUNE=2 also prints every candidate's timing to stderr.  No reference counterpart: a
 * property of this implementation. */
typedef struct crd_launch_plan {
	int32_t autotune;     /* measuring enabled (2: verbose) */
	int32_t tuned;        /* a measurement has been made (or a plan pinned) */
	int32_t one_round;    /* chunk mode: 0 = 32-row chunks, 1 = stretched so that all workgroups are resident at once, 2 = 64-row chunks */
	int32_t xcd_mapping;  /* 0 theta-first dispatch order, 1 one contiguous band of the slab per XCD, 2 the same with succession in phi */
	int32_t rows;         /* height of the launch it was measured on */
	int32_t columns_per_lane; /* 1: a wavefront steps a strip of 64 columns (56 valid); 2: 128 columns (120 valid), two per lane -- packed
	                           * arithmetic in fp32 */
	int32_t nontemporal_stores; /* 1: the new state is written with the non-temporal hint (it is not read again by the launch, and
	                             * does not displace from L2 what neighbouring work items share) */
	int32_t steps_per_launch; /* 1; 2: one launch advances the state by TWO RK4 steps (the state crosses memory once per two steps, for twice
	                           * the pipeline registers and a 16-row / 16-column apron); crd_step_rk4 then issues pairs; 3 (FHN: in fp64 on a
	                           * single slab, a block's wavefronts running as one strip with one column per lane; in fp32 -- an even nx --
	                           * with two columns per lane and a strip per wavefront, single slabs and the slabs of a run inside their
	                           * exchange cycles; elsewhere taken as 2): THREE steps; triples, then a pair or a single step for what is
	                           * left (fp32: also for triples any stage of which has the absorbing rows on).  What crd_get_launch_plan
	                           * reports -- steps and columns -- is what a launch does. */
	double ms_default, ms_chosen; /* measured times PER STEP: plain plan, chosen plan */
} crd_launch_plan;
int crd_set_autotune(crd_ctx *ctx, int on);
int crd_get_launch_plan(const crd_ctx *ctx, crd_launch_plan *out);
/* What the context's full-height launch of the step kernel looks like under its current plan, and what the assembler printed about
 * the instantiation it runs (read off the device assembly when the library was built: tools/kernel_regs.py) -- enough to price the
 * launch's vector issue without a profiler: loop_valu x (wavefront_iterations / iterations_per_trip) x 4 cycles per wavefront
 * instruction over simds x clock (bench.py: roofline.issue_frac).  No reference counterpart. */
typedef struct crd_launch_geometry {
	int32_t rows;                      /* rows the launch produces (a multi-slab context: the launch the timed calls time) */
	int32_t strips, chunk_rows, chunks; /* work items: strips of columns (one per wavefront) x chunks of rows */
	int32_t workgroups, wavefronts_per_workgroup;
	int32_t fill_iterations;           /* pipeline iterations an item runs beyond its rows (the aprons in phi) */
	int32_t iterations_per_trip;       /* pipeline iterations one trip of the steady-state loop holds (its unroll factor) */
	int32_t lanes, lanes_valid;        /* lanes of a wavefront / lanes whose column is an output (kernels that put one apron around a
	                                    * workgroup's wavefronts -- Goldbeter fp64, two steps per launch -- : the workgroup's average, 60) */
	int32_t vgprs, sgprs, lds_bytes, scratch_bytes, wavefronts_per_simd; /* of the instantiation; 0: the build carries no kernel table */
	int32_t loop_valu, loop_salu, loop_vmem, loop_lds, loop_instructions; /* static instruction mix of one trip of that loop */
	int32_t simds, clock_khz;          /* of the device: 4 x compute units, hipDeviceProp_t::clockRate */
	int32_t exec_skipped_vmem;         /* vector-memory regions of that kernel a wavefront can skip on its execution mask (device assembly,
	                                    * tools/kernel_regs.py).  The multi-step kernels wait for their LDS-DMA row fills with hand-counted
	                                    * s_waitcnt vmcnt(N); N counts the iteration's stores, which therefore must issue whatever the mask:
	                                    * 0 for every such kernel, or the build stops */
	int64_t wavefront_iterations;      /* pipeline iterations all wavefronts of the launch run: strips x (rows + chunks x fill_iterations) */
	int64_t wavefront_iterations_effective; /* ... with an item's filling iterations at what they cost: stage k of the pipeline starts at
	                                    * iteration 2 k, so the fill runs fill_iterations / 2 - 1 iterations' worth of stages */
} crd_launch_geometry;
int crd_get_launch_geometry(crd_ctx *ctx, crd_launch_geometry *out);
/* The plans the measurement chooses among, index 0 .. (first index that returns CRD_EINVAL) - 1: one_round, xcd_mapping,
 * columns_per_lane, nontemporal_stores and steps_per_launch of *out are set, the rest zero.  (tools/plan_sweep.py profiles every one of them;
 * tests/test_profiles.py checks that profiles/pmc_traffic.json has an entry for each.) */
int crd_launch_plan_candidate(int index, crd_launch_plan *out);
/* Use THIS plan (chunk mode 0..2, mapping 0..2, columns per lane 1..2, non-temporal stores 0..1, steps per launch 1..3) for the fixed-step kernel instead of measuring one -- a plan
 * read back from an earlier context of the same shape on the same device (the measurement costs ~0.45 s per context at 8192^2), or
 * a profiling run in which every launch of the kernel should be the plan a previous run chose (`bench.py --launch-plan`).  A pinned
 * plan applies to launches of EVERY size (a measured one only to launches of the height it was measured on).  Where a choice cannot
 * be honoured (two columns per lane on an odd nx, mapping 2 on a launch too short for it) the launch falls back as it would for a
 * measured plan.  crd_get_launch_plan then reports tuned = 1 with both times 0.  CRD_EINVAL outside the ranges. */
int crd_set_launch_plan(crd_ctx *ctx, int chunk_mode, int xcd_mapping, int columns_per_lane, int nontemporal_stores, int steps_per_launch);
/* Measure the plan NOW (a step of the resident state into scratch planes, discarded; the state is not advanced) instead of
 * inside the first crd_step_rk4 -- for callers that time their first steps; also creates the events crd_step_rk4_timed uses.  The
 * measurement is skipped when a plan exists or autotuning is off. */
int crd_plan_launches(crd_ctx *ctx);

/* Named ranges for a profiler's timeline (roctx: `rocprofv3 --marker-trace`), SURVEY section 5.  libcrd puts its own around
 * step batches (crd_step_rk4 ...), halo exchanges, state transfers and output rows; a host program brackets its phases with
 * these two (crd_run: one range per output interval, where the reference prints its progress line, src/FHNmodel_torus.cpp:
 * 457-477).  No-ops unless a profiler is attached (ROCP_TOOL_LIBRARIES in the environment) or CRD_ROCTX=1; the marker library
 * is bound with dlopen on first use, never linked. */
void crd_trace_range_push(const char *name);
void crd_trace_range_pop(void);

/* max |var0| over the slab (blow-up guard; synchronises). */
int crd_state_max_abs(crd_ctx *ctx, double *out);

#ifdef __cplusplus
}
#endif
#endif /* CRD_H */
