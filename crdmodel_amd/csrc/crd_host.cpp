// crd_host.cpp -- host-side pieces of libcrd that need no GPU: geometry, slab extents, stable states, initial
// conditions, coefficient tables.  Each function cites the reference lines whose behaviour it reproduces.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdlib>
#include <cstring>

#include "crd_internal.h"

namespace crd {

bool validate_params(const crd_params &p, std::string *why)
{
	auto fail = [&](const char *m) {
		if (why) *why = m;
		return false;
	};
	if (p.model != CRD_MODEL_FHN && p.model != CRD_MODEL_GOLDBETER) return fail("model must be CRD_MODEL_FHN or CRD_MODEL_GOLDBETER");
	if (p.surface != CRD_SURFACE_TORUS && p.surface != CRD_SURFACE_FLAT) return fail("surface must be CRD_SURFACE_TORUS or CRD_SURFACE_FLAT");
	if (p.precision != CRD_PRECISION_F64 && p.precision != CRD_PRECISION_F32) return fail("precision must be CRD_PRECISION_F64 or CRD_PRECISION_F32");
	if (p.nx < 2 || p.nx > (1 << 24)) return fail("nx (thetaMesh / xMesh) must be in [2, 2^24] (the kernels index columns with 32-bit integers)");
	if (p.ny < 0 || p.ny > INT32_MAX) return fail("ny (phiMesh) must be in [0, 2^31)");
	if (!(p.surface_length > 0.0) || !(p.surface_width > 0.0)) return fail("surfaceLength and surfaceWidth must be positive");
	if (!std::isfinite(p.diffusion) || !std::isfinite(p.beta) || !std::isfinite(p.beta_min) || !std::isfinite(p.beta_max) ||
	    !std::isfinite(p.t_boundary))
		return fail("non-finite parameter");
	if (p.surface == CRD_SURFACE_TORUS && !(p.surface_length > p.surface_width) && p.ny == 0)
		return fail("torus needs surfaceLength > surfaceWidth (R > r) unless phiMesh is given");
	return true;
}

const char *model_name(int model) { return model == CRD_MODEL_FHN ? "FHNmodel" : "GoldbeterModel"; }
const char *surface_name(int surface) { return surface == CRD_SURFACE_TORUS ? "torus" : "flat"; }
const char *var_name(int model, int var)
{
	// src/FHNmodel_torus.cpp:385,388; src/GoldbeterModel_torus.cpp:446,449
	if (model == CRD_MODEL_FHN) return var == 0 ? "u" : "v";
	return var == 0 ? "Z" : "Y";
}

void build_coefficients(const crd_params &p, const crd_grid &g, Coefficients *out)
{
	const double D = p.diffusion;
	out->cA.assign((size_t)g.nx, 0.0);
	out->cP.assign((size_t)g.nx, 0.0);
	if (p.surface == CRD_SURFACE_TORUS) {
		// Factors of src/FHNmodel_torus.cpp:535-537, each folded with its D and mesh divisor; theta_i as in :531.
		out->cX = D * (1 / (g.r * g.r)) / (g.dx * g.dx);
		for (int64_t i = 0; i < g.nx; i++) {
			const double xx = g.xmin + (double)i * g.dx;
			const double rho = g.R + g.r * std::cos(xx);
			out->cA[(size_t)i] = D * (-std::sin(xx) / (g.r * rho)) / (2 * g.dx);
			out->cP[(size_t)i] = D * (1 / (rho * rho)) / (g.dy * g.dy);
		}
	} else {
		// src/FHNmodel_flat.cpp:489-490: cu1 = D/dx/dx, cu2 = D/dy/dy (cu3 = -2(cu1+cu2) is implied by the stencil form).
		out->cX = D / g.dx / g.dx;
		const double cu2 = D / g.dy / g.dy;
		for (int64_t i = 0; i < g.nx; i++) out->cP[(size_t)i] = cu2;
	}
	out->cE.resize((size_t)g.nx);
	out->cWn.resize((size_t)g.nx);
	for (int64_t i = 0; i < g.nx; i++) {
		out->cE[(size_t)i] = out->cX + out->cA[(size_t)i];
		out->cWn[(size_t)i] = out->cA[(size_t)i] - out->cX;
	}
}

void build_beta_rows(const crd_params &p, const crd_grid &g, int64_t j0, int64_t j1, std::vector<double> *out)
{
	out->resize((size_t)(j1 - j0));
	for (int64_t j = j0; j < j1; j++) {
		// Ghost rows wrap periodically; their b is never used for an owned output but keep it well defined.
		const int64_t jj = ((j % g.ny) + g.ny) % g.ny;
		double b = p.beta;
		if (p.vary_beta != 0) {
			const double yy = g.ymin + (double)jj * g.dy;                            // :623
			b = p.beta_min + yy * (p.beta_max - p.beta_min) / (g.ymax - g.ymin);     // :631
		}
		(*out)[(size_t)(j - j0)] = b;
	}
}

}  // namespace crd

using namespace crd;

extern "C" {

int crd_abi_version(void) { return CRD_ABI_VERSION; }

const char *crd_status_string(int status)
{
	switch (status) {
	case CRD_OK: return "ok";
	case CRD_EINVAL: return "invalid argument";
	case CRD_ENOMEM: return "out of memory";
	case CRD_EHIP: return "HIP runtime error";
	case CRD_ERCCL: return "RCCL error";
	case CRD_EIO: return "I/O error";
	case CRD_EPARSE: return "ini parse error";
	case CRD_ESTATE: return "invalid state for this call";
	default: return "unknown status";
	}
}

int crd_grid_from_params(const crd_params *p, crd_grid *g)
{
	if (!p || !g) return CRD_EINVAL;
	if (!validate_params(*p, nullptr)) return CRD_EINVAL;
	std::memset(g, 0, sizeof(*g));
	g->nx = p->nx;
	if (p->surface == CRD_SURFACE_TORUS) {
		// src/FHNmodel_torus.cpp:73-76,188-193: ny = NX*(R/r) is an int*double product truncated to long.
		g->r = p->surface_width / (2.0 * kPi);
		g->R = p->surface_length / (2.0 * kPi);
		const double radius_ratio = g->R / g->r;
		g->ny = (int64_t)((int)p->nx * radius_ratio);
		g->xmin = 0.0;
		g->xmax = 2.0 * kPi;
		g->ymin = 0.0;
		g->ymax = 2.0 * kPi;
	} else {
		// src/FHNmodel_flat.cpp:172-175,190-192: integer length/width ratio.
		const int64_t ratio = (int64_t)(p->surface_length / p->surface_width);
		g->ny = p->nx * ratio;
		g->xmin = 0.0;
		g->xmax = p->surface_width - g->xmin;
		g->ymin = 0.0;
		g->ymax = p->surface_length - g->ymin;
	}
	if (p->ny > 0) g->ny = p->ny;  // phiMesh extension
	if (g->ny < 2) return CRD_EINVAL;
	g->dx = (g->xmax - g->xmin) / (1.0 * (double)g->nx - 1.0);  // :233
	g->dy = (g->ymax - g->ymin) / (1.0 * (double)g->ny - 1.0);  // :234
	return CRD_OK;
}

int crd_slab_extents(int64_t ny, int slab, int n_slabs, int64_t *js, int64_t *je)
{
	if (!js || !je || n_slabs < 1 || slab < 0 || slab >= n_slabs || ny < n_slabs) return CRD_EINVAL;
	*js = ny * slab / n_slabs;            // src/FHNmodel_torus.cpp:752
	*je = ny * (slab + 1) / n_slabs - 1;  // :753
	return CRD_OK;
}

int crd_dims_create(int nprocs, int *d0, int *d1)
{
	if (nprocs < 1 || !d0 || !d1) return CRD_EINVAL;
	// MPI_Dims_create(nprocs, 2, dims) with dims = {0, 0} (src/FHNmodel_torus.cpp:726-728): the factor pair closest to a square, in
	// non-increasing order -- 4 -> {2, 2}, 8 -> {4, 2}, 2 -> {2, 1}, 6 -> {3, 2}
	int b = 1;
	for (int f = 1; (long)f * f <= nprocs; f++)
		if (nprocs % f == 0) b = f;
	*d0 = nprocs / b;
	*d1 = b;
	return CRD_OK;
}

int crd_block_extents(int64_t nx, int64_t ny, int c0, int d0, int c1, int d1, int64_t *is, int64_t *ie, int64_t *js, int64_t *je)
{
	if (!is || !ie || !js || !je || d0 < 1 || d1 < 1 || c0 < 0 || c0 >= d0 || c1 < 0 || c1 >= d1 || nx < d0 || ny < d1) return CRD_EINVAL;
	*is = nx * c0 / d0;            // src/FHNmodel_torus.cpp:750
	*ie = nx * (c0 + 1) / d0 - 1;  // :751
	*js = ny * c1 / d1;            // :752
	*je = ny * (c1 + 1) / d1 - 1;  // :753
	return CRD_OK;
}

int crd_halo_plan(int slab, int n_slabs, int64_t nyl, int depth, crd_halo_op ops[4])
{
	if (!ops || n_slabs < 1 || slab < 0 || slab >= n_slabs || depth < 1 || nyl < depth) return CRD_EINVAL;
	const int prev = (slab + n_slabs - 1) % n_slabs, next = (slab + 1) % n_slabs;
	ops[0] = crd_halo_op{1, next, nyl - depth, depth};  // my last rows are the next slab's low ghosts
	ops[1] = crd_halo_op{1, prev, 0, depth};            // my first rows are the previous slab's high ghosts
	ops[2] = crd_halo_op{0, prev, -(int64_t)depth, depth};
	ops[3] = crd_halo_op{0, next, nyl, depth};
	return CRD_OK;
}

int crd_cycle_vote(int pos, double vote[2])
{
	if (!vote || pos < -1 || pos >= kMaxExchangeEvery) return CRD_EINVAL;
	vote[0] = (double)pos;  // element-wise MIN over the ranks: min(pos) and -max(pos)
	vote[1] = -(double)pos;
	return CRD_OK;
}

int crd_cycle_agreed(const double reduced[2])
{
	if (!reduced) return -1;
	const double lo = reduced[0], hi = -reduced[1];
	return (lo == hi && lo >= 0.0 && lo < (double)kMaxExchangeEvery && lo == (double)(int)lo) ? (int)lo : -1;
}

static double gb_residual_y(double Z, double Y)
{
	// v2 - v3 - kf Y of src/GoldbeterModel_torus.cpp:694-695,716
	const double z2 = Z * Z, z4 = z2 * z2, y2 = Y * Y;
	const double ka4 = kGbKa * kGbKa * kGbKa * kGbKa;
	const double v2 = kGbVm2 * z2 / (kGbK2 * kGbK2 + z2);
	const double v3 = kGbVm3 * y2 * z4 / ((kGbKr * kGbKr + y2) * (ka4 + z4));
	return v2 - v3 - kGbKf * Y;
}

int crd_steady_state(int model, double beta, double *s0, double *s1)
{
	if (!s0 || !s1 || !std::isfinite(beta)) return CRD_EINVAL;
	if (model == CRD_MODEL_FHN) {
		*s0 = -beta;                            // src/FHNmodel_torus.cpp:243
		*s1 = beta * beta * beta - 3 * beta;    // :244
		return CRD_OK;
	}
	if (model != CRD_MODEL_GOLDBETER) return CRD_EINVAL;
	// Summing the two Goldbeter equations at rest gives v0 + v1 beta - k Z = 0; Y then is the root of a function
	// that decreases monotonically for Y > 0.  Safeguarded Newton (falls back to bisection steps).
	const double Z = (kGbV0 + kGbV1 * beta) / kGbK;
	if (!(Z > 0.0)) return CRD_EINVAL;
	double lo = 0.0, hi = 1.0;
	while (gb_residual_y(Z, hi) > 0.0) {
		hi *= 2.0;
		if (hi > 1e12) return CRD_EINVAL;
	}
	double Y = 0.5 * (lo + hi);
	for (int it = 0; it < 200; it++) {
		const double g = gb_residual_y(Z, Y);
		if (g > 0.0) lo = Y; else hi = Y;
		const double h = 1e-7 * (1.0 + std::fabs(Y));
		const double dg = (gb_residual_y(Z, Y + h) - gb_residual_y(Z, Y - h)) / (2.0 * h);
		double Yn = (dg < 0.0) ? Y - g / dg : 0.5 * (lo + hi);
		if (!(Yn > lo && Yn < hi)) Yn = 0.5 * (lo + hi);
		if (std::fabs(Yn - Y) <= 4e-16 * std::fabs(Y)) {
			Y = Yn;
			break;
		}
		Y = Yn;
	}
	*s0 = Z;
	*s1 = Y;
	return CRD_OK;
}

int crd_steady_state_as_printed(int model, double beta, int decimals, double *s0, double *s1)
{
	if (decimals < 0 || decimals > 17) return CRD_EINVAL;
	const int rc = crd_steady_state(model, beta, s0, s1);
	if (rc != CRD_OK || decimals == 0 || model != CRD_MODEL_GOLDBETER) return rc;
	// print Z[-1], Y[-1] of one-element numpy arrays (util/GoldbeterModel/SolveGoldbeterODE.py:111): positional notation with
	// `precision` = 8 digits BEHIND THE DECIMAL POINT for values in [1e-4, 1e8) -- both numbers are O(0.1 .. 10) for any beta the
	// model is run with -- trailing zeros dropped; fscanf("[%lf] [%lf]") reads the text back (src/GoldbeterModel_torus.cpp:258).
	for (double *v : {s0, s1}) {
		char text[64];
		std::snprintf(text, sizeof text, "%.*f", decimals, *v);
		*v = std::strtod(text, nullptr);
	}
	return CRD_OK;
}

int crd_initial_conditions(const crd_run_config *cfg, int64_t js, int64_t je, double *y_aos)
{
	if (!cfg) return CRD_EINVAL;
	crd_grid g;
	const int rc = crd_grid_from_params(&cfg->params, &g);
	if (rc != CRD_OK) return rc;
	return crd_initial_conditions_block(cfg, 0, g.nx - 1, js, je, y_aos);
}

int crd_initial_conditions_block(const crd_run_config *cfg, int64_t is, int64_t ie, int64_t js, int64_t je, double *y_aos)
{
	if (!cfg || !y_aos) return CRD_EINVAL;
	const crd_params &p = cfg->params;
	crd_grid g;
	int rc = crd_grid_from_params(&p, &g);
	if (rc != CRD_OK) return rc;
	if (js < 0 || je < js || je >= g.ny || is < 0 || ie < is || ie >= g.nx) return CRD_EINVAL;
	// The stable state enters only the rules that start from it (the reference computes Us, Vs / reads Zs, Ys for these
	// and never touches them under the uniform and varyBeta rules, so an unused `beta` without a fixed point must not matter).
	const bool needs_steady = (p.model == CRD_MODEL_FHN) ? !((p.surface == CRD_SURFACE_TORUS) ? (p.vary_beta != 0) : (p.vary_beta == 1))
	                                                     : (p.vary_beta != 1);
	double s0 = 0.0, s1 = 0.0;
	if (needs_steady) {
		rc = crd_steady_state_as_printed(p.model, p.beta, cfg->steady_state_decimals, &s0, &s1);
		if (rc != CRD_OK) return rc;
	}

	// Rectangle limits: src/FHNmodel_torus.cpp:199-200,285-300; flat src/FHNmodel_flat.cpp:280-282.
	const double wave_length = (g.ymax - g.ymin) * cfg->wave_length;
	const double wave_width = (g.xmax - g.xmin) * cfg->wave_width;
	double x_lo, x_hi;
	bool wraps = false;  // outside-centred torus wave: theta >= x_lo OR theta <= x_hi
	if (p.surface == CRD_SURFACE_TORUS) {
		if (cfg->wave_inside == 1) {
			x_lo = kPi - wave_width / 2.0;
			x_hi = kPi + wave_width / 2.0;
		} else if (cfg->wave_inside == 0) {
			x_lo = 0.0 - wave_width / 2.0 + (g.xmax - g.xmin);
			x_hi = 0.0 + wave_width / 2.0;
			wraps = true;
		} else {
			return CRD_EINVAL;  // the reference only prints "WaveInside must be 0 or 1"
		}
	} else {
		const double mid = p.surface_width / 2.0;
		x_lo = mid - wave_width / 2.0;
		x_hi = mid + wave_width / 2.0;
	}

	// Which perturbation rule applies (see the four IC blocks cited in crd.h).
	enum { UNIFORM, RECT, RAND } rule = RECT;
	double base0 = s0, base1 = s1, pert0, pert1, phi_lo_mult;
	bool theta_and_only = false;
	if (p.model == CRD_MODEL_FHN) {
		const bool uniform = (p.surface == CRD_SURFACE_TORUS) ? (p.vary_beta != 0) : (p.vary_beta == 1);
		if (uniform) {
			rule = UNIFORM;
			base0 = 1;
			base1 = 1;
		}
		pert0 = s0 + 2;    // src/FHNmodel_torus.cpp:319-320
		pert1 = s1 + 1.5;
		phi_lo_mult = 1.0;
	} else {
		pert0 = s0 + 1;    // src/GoldbeterModel_torus.cpp:349-350
		pert1 = s1 + 1;
		phi_lo_mult = (p.surface == CRD_SURFACE_TORUS) ? 1.0 : 2.0;
		if (p.vary_beta == 1) {
			base0 = 0.4;       // src/GoldbeterModel_torus.cpp:383-384,392-393
			base1 = 1.6;
			pert0 = 1.4;
			pert1 = 2.6;
			phi_lo_mult = 2.0;
			theta_and_only = true;  // src/GoldbeterModel_torus.cpp:389 uses && even for an outside wave
			if (cfg->ic_type == 0) rule = UNIFORM;
			else if (cfg->ic_type == 1) rule = RECT;
			else if (cfg->ic_type == 2) rule = RAND;
			else return CRD_EINVAL;
		} else if (p.vary_beta != 0) {
			return CRD_EINVAL;  // the reference leaves y uninitialised
		}
	}

	const int64_t nx = g.nx;
	std::vector<unsigned char> theta_in((size_t)nx, 0);
	for (int64_t i = 0; i < nx; i++) {
		const double xx = g.xmin + (double)i * g.dx;
		theta_in[(size_t)i] = (wraps && !theta_and_only) ? (xx >= x_lo || xx <= x_hi) : (xx >= x_lo && xx <= x_hi);
	}
	if (rule == RAND) srand(1);  // every reference rank starts from the default seed
	const int64_t nxl = ie - is + 1;
	for (int64_t j = js; j <= je; j++) {
		const double yy = g.ymin + (double)j * g.dy;
		const bool phi_in = yy >= phi_lo_mult * wave_length && yy <= (phi_lo_mult + 1.0) * wave_length;
		double *row = y_aos + 2 * nxl * (j - js) - 2 * is;  // (indexed by the global column below)
		for (int64_t i = is; i <= ie; i++) {
			double a = base0, b = base1;
			if (rule == RECT && phi_in && theta_in[(size_t)i]) {
				a = pert0;
				b = pert1;
			} else if (rule == RAND) {
				a = (float)rand() / (float)RAND_MAX * 1.4;  // src/GoldbeterModel_torus.cpp:409-410 (int -> float as C converts it)
				b = (float)rand() / (float)RAND_MAX * 1.4;
			}
			row[2 * i] = a;
			row[2 * i + 1] = b;
		}
	}
	return CRD_OK;
}

double crd_stable_dt(const crd_params *p)
{
	crd_grid g;
	if (!p || crd_grid_from_params(p, &g) != CRD_OK) return 0.0;
	const double D = std::fabs(p->diffusion);
	double lam;
	if (p->surface == CRD_SURFACE_TORUS) {
		const double rin = g.R - g.r;  // smallest distance from the axis (theta = pi)
		const double adv = 1.0 / (g.r * std::fabs(rin) * g.dx);
		lam = 4.0 * D * (1.0 / (g.r * g.dx * g.r * g.dx) + 1.0 / (rin * g.dy * rin * g.dy)) + D * adv;
	} else {
		lam = 4.0 * D * (1.0 / (g.dx * g.dx) + 1.0 / (g.dy * g.dy));
	}
	// Reaction Jacobian bound (documented in crd.h): FHN |3 - 3u^2| <= ~9 on the limit cycle; Goldbeter's Hill terms are far stiffer.
	lam += (p->model == CRD_MODEL_FHN) ? kFhnReactionRate : kGoldbeterReactionRate;
	return 2.785 / lam;
}

}  // extern "C"
