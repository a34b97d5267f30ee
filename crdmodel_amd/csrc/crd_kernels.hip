// crd_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, wave64): the RHS of CRDModel's f()
// (src/FHNmodel_torus.cpp:504-667 and its three siblings) as (a) a bare RHS on the reference's AoS vectors and
// (b) RK4 stage kernels on SoA planes with the stage update fused into the epilogue.
//
// Shape of every kernel here: one thread per grid point, 64(theta) x 4 thread blocks sweeping a 64 x 16 tile;
// the activator tile plus a one-point halo ring is staged in LDS from coalesced row reads (theta is the contiguous
// axis, so a wavefront reads one 512-B fp64 row segment); theta wrap is index arithmetic at tile load, phi
// neighbours of the slab's first / last row come from ghost rows (or wrap inside a single slab); the curvature
// coefficients are per-column tables in HBM (L2-resident); no MFMA -- there is no contraction in this path.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <type_traits>

#include "crd_device.h"

namespace crd {

namespace {

// Results that are written once and not read again by the launch go out with the non-temporal hint when the slab is large (NT =
// true, chosen per launch by stores_nontemporal() below): they do not stay in L2 at the expense of what neighbouring tiles share
// and stream to memory instead of being evicted later -- staged step -2.3 % at 8192^2 fp64, -4.1 % at 4096^2, -4 % on the 8192 x
// 1024 share (profiles/r03/nt_stores.txt).  Small slabs keep plain stores: there the next stage finds its input in L2.
template <bool NT, typename T>
__device__ __forceinline__ void result_store(T *p, T v)
{
	if constexpr (NT) __builtin_nontemporal_store(v, p);
	else *p = v;
}
template <bool NT, typename Real, typename P>
__device__ __forceinline__ void pair_store(P *p, const P &k)
{
	if constexpr (NT) {
		typedef Real vec2 __attribute__((ext_vector_type(2)));
		const vec2 v = {k.x, k.y};
		__builtin_nontemporal_store(v, reinterpret_cast<vec2 *>(p));
	} else {
		*p = k;
	}
}
inline bool stores_nontemporal(const SlabDesc &d, size_t real_bytes) { return (size_t)d.nx * (size_t)d.nyl * real_bytes >= ((size_t)32 << 20); }

using namespace dev;

constexpr int kTX = 64;             // tile width = one wavefront along theta
constexpr int kBY = 4;              // waves per block
constexpr int kRPT = 4;             // rows per thread
constexpr int kTY = kBY * kRPT;     // tile height
template <typename Real>
struct StageArgs {
	const Real *in_u, *in_v;    // stage input, pointers to local row 0
	const Real *y0_u, *y0_v;    // step-start state
	Real *acc_u, *acc_v;        // running combination
	Real *out_u, *out_v;        // stage output
	Real h_out, h_acc;          // yout = y0 + h_out k ; acc (+)= h_acc k
	int absorb;
	const Real *gw, *ge;        // theta-blocks: the input's columns west / east of the block, by local row
};

// RK4 stage kernel on SoA planes.  STAGE 0 writes k = f(yin); stages 1-4 follow SURVEY 8(d)'s scheme:
//   1: y1 = y0 + dt/2 k1, acc  = y0 + dt/6 k1      2: y2 = y0 + dt/2 k2, acc += dt/3 k2
//   3: y3 = y0 + dt k3,   acc += dt/3 k3           4: y  = acc + dt/6 k4
// i.e. 6 + 10 + 10 + 6 = 32 reals of HBM traffic per grid-point-step.
template <typename Real, int MODEL, int STAGE, bool NT>
__global__ void __launch_bounds__(kTX *kBY) crd_rk4_stage_kernel(Slab<Real> s, StageArgs<Real> a, int row_begin, int row_end, int nbx, int nblocks)
{
	__shared__ Real tile[kTY + 2][kTX + 2];

	const int tid = xcd_remap((int)blockIdx.x, nblocks);
	const int by = tid / nbx, bx = tid - by * nbx;
	const int tx = threadIdx.x, ty = threadIdx.y;
	const int nx = s.nx;
	const int i0 = bx * kTX, i = i0 + tx;
	const int j0 = row_begin + by * kTY;
	const bool col_ok = i < nx;
	const int iw = (i0 == 0) ? nx - 1 : i0 - 1;          // theta wrap: column -1 is column nx-1 (Appendix A.3)
	const bool last_col = col_ok && (tx == kTX - 1 || i == nx - 1);
	const int ie = (i == nx - 1) ? 0 : i + 1;

	Real uC[kRPT], vC[kRPT], p0u[kRPT], p0v[kRPT], pau[kRPT], pav[kRPT];

	// Own rows: one coalesced row read per wave, value kept in a register and mirrored into LDS for the neighbours.
#pragma unroll
	for (int r = 0; r < kRPT; r++) {
		const int j = j0 + ty + kBY * r;
		uC[r] = vC[r] = p0u[r] = p0v[r] = pau[r] = pav[r] = (Real)0;
		if (j < row_end && col_ok) {
			const size_t off = (size_t)j * nx;
			const Real *row = a.in_u + off;
			uC[r] = row[i];
			vC[r] = a.in_v[off + i];
			// (theta-blocks: beyond the block's first / last column lie the neighbour blocks' columns, exchanged into strips)
			if (tx == 0) tile[ty + kBY * r + 1][0] = (i0 == 0 && !s.wrap_x) ? a.gw[j] : row[iw];
			if (last_col) tile[ty + kBY * r + 1][tx + 2] = (i == nx - 1 && !s.wrap_x) ? a.ge[j] : row[ie];
			if (STAGE == 2 || STAGE == 3) {
				p0u[r] = a.y0_u[off + i];
				p0v[r] = a.y0_v[off + i];
			}
			if (STAGE >= 2) {
				pau[r] = a.acc_u[off + i];
				pav[r] = a.acc_v[off + i];
			}
		}
		if (j < row_end && col_ok) tile[ty + kBY * r + 1][tx + 1] = uC[r];  // rows past the range belong to the halo loader, columns past nx to the wrap writer
	}
	// Halo rows j0-1 (wave 0) and min(j0+kTY, row_end) (wave 1): ghost rows of the slab, or the wrapped row.
	if (ty < 2 && col_ok) {
		const int jt = (ty == 0) ? j0 - 1 : ((j0 + kTY < row_end) ? j0 + kTY : row_end);
		const int tr = (ty == 0) ? 0 : jt - j0 + 1;
		const Real *row = a.in_u + (ptrdiff_t)wrap_row(s, jt) * nx;
		tile[tr][tx + 1] = row[i];
	}
	const Real cE = col_ok ? s.cE[i] : (Real)0;
	const Real cWn = col_ok ? s.cWn[i] : (Real)0;
	const Real cP = col_ok ? s.cP[i] : (Real)0;
	__syncthreads();

#pragma unroll
	for (int r = 0; r < kRPT; r++) {
		const int j = j0 + ty + kBY * r;
		if (!(j < row_end && col_ok)) continue;
		const int tr = ty + kBY * r + 1;
		const bool zero = a.absorb && ((s.has_row0 && j == 0) || (s.has_rowN && j == s.nyl - 1));
		Real du, dv;
		rhs_point_values<Real, MODEL>(uC[r], tile[tr][tx], tile[tr][tx + 2], tile[tr - 1][tx + 1], tile[tr + 1][tx + 1], vC[r], cE, cWn, cP,
		                       s.brow[j], s.ka4, zero, du, dv);
		const size_t o = (size_t)j * nx + i;
		if (STAGE == 0) {
			result_store<NT>(&a.out_u[o], du);
			result_store<NT>(&a.out_v[o], dv);
		} else if (STAGE == 1) {
			result_store<NT>(&a.out_u[o], fmadd(a.h_out, du, uC[r]));
			result_store<NT>(&a.out_v[o], fmadd(a.h_out, dv, vC[r]));
			result_store<NT>(&a.acc_u[o], fmadd(a.h_acc, du, uC[r]));
			result_store<NT>(&a.acc_v[o], fmadd(a.h_acc, dv, vC[r]));
		} else if (STAGE == 2 || STAGE == 3) {
			result_store<NT>(&a.out_u[o], fmadd(a.h_out, du, p0u[r]));
			result_store<NT>(&a.out_v[o], fmadd(a.h_out, dv, p0v[r]));
			result_store<NT>(&a.acc_u[o], fmadd(a.h_acc, du, pau[r]));
			result_store<NT>(&a.acc_v[o], fmadd(a.h_acc, dv, pav[r]));
		} else {
			result_store<NT>(&a.out_u[o], fmadd(a.h_acc, du, pau[r]));
			result_store<NT>(&a.out_v[o], fmadd(a.h_acc, dv, pav[r]));
		}
	}
}

// Bare RHS on the reference's AoS vectors: y[j][i] = (var0, var1) pairs, one 16-B (fp64) load per point.
template <typename Real, int MODEL, bool NT>
__global__ void __launch_bounds__(kTX *kBY) crd_rhs_aos_kernel(Slab<Real> s, const typename Pair<Real>::type *__restrict__ y,
                                                               typename Pair<Real>::type *__restrict__ ydot, const Real *__restrict__ ghost_lo,
                                                               const Real *__restrict__ ghost_hi, const Real *__restrict__ gcol_w,
                                                               const Real *__restrict__ gcol_e, int absorb, int row_begin, int row_end, int nbx, int nblocks)
{
	using P = typename Pair<Real>::type;
	__shared__ Real tile[kTY + 2][kTX + 2];

	const int tid = xcd_remap((int)blockIdx.x, nblocks);
	const int by = tid / nbx, bx = tid - by * nbx;
	const int tx = threadIdx.x, ty = threadIdx.y;
	const int nx = s.nx, nyl = s.nyl;
	const int i0 = bx * kTX, i = i0 + tx;
	const int j0 = row_begin + by * kTY;  // rows [row_begin, row_end) of the slab; their phi neighbours must be resident too
	const bool col_ok = i < nx;
	const int iw = (i0 == 0) ? nx - 1 : i0 - 1;
	const bool last_col = col_ok && (tx == kTX - 1 || i == nx - 1);
	const int ie = (i == nx - 1) ? 0 : i + 1;

	P own[kRPT];
#pragma unroll
	for (int r = 0; r < kRPT; r++) {
		const int j = j0 + ty + kBY * r;
		own[r].x = own[r].y = (Real)0;
		if (j < row_end && col_ok) {
			const P *row = y + (size_t)j * nx;
			own[r] = row[i];
			if (tx == 0) tile[ty + kBY * r + 1][0] = (i0 == 0 && !s.wrap_x) ? gcol_w[j] : row[iw].x;
			if (last_col) tile[ty + kBY * r + 1][tx + 2] = (i == nx - 1 && !s.wrap_x) ? gcol_e[j] : row[ie].x;
		}
		if (j < row_end && col_ok) tile[ty + kBY * r + 1][tx + 1] = own[r].x;
	}
	if (ty < 2 && col_ok) {
		const int jt = (ty == 0) ? j0 - 1 : ((j0 + kTY < row_end) ? j0 + kTY : row_end);
		const int tr = (ty == 0) ? 0 : jt - j0 + 1;
		Real val;
		if (jt >= 0 && jt < nyl) val = y[(size_t)jt * nx + i].x;
		else if (s.wrap) val = y[(size_t)(jt < 0 ? nyl - 1 : 0) * nx + i].x;
		else val = (jt < 0) ? ghost_lo[i] : ghost_hi[i];
		tile[tr][tx + 1] = val;
	}
	const Real cE = col_ok ? s.cE[i] : (Real)0;
	const Real cWn = col_ok ? s.cWn[i] : (Real)0;
	const Real cP = col_ok ? s.cP[i] : (Real)0;
	__syncthreads();

#pragma unroll
	for (int r = 0; r < kRPT; r++) {
		const int j = j0 + ty + kBY * r;
		if (!(j < row_end && col_ok)) continue;
		const int tr = ty + kBY * r + 1;
		const bool zero = absorb && ((s.has_row0 && j == 0) || (s.has_rowN && j == nyl - 1));
		P k;
		rhs_point_values<Real, MODEL>(own[r].x, tile[tr][tx], tile[tr][tx + 2], tile[tr - 1][tx + 1], tile[tr + 1][tx + 1], own[r].y, cE, cWn, cP,
		                       s.brow[j], s.ka4, zero, k.x, k.y);
		pair_store<NT, Real>(&ydot[(size_t)j * nx + i], k);
	}
}

// ---- layout adaptors -------------------------------------------------------------------------------------------
template <typename Src, typename Real>
__global__ void __launch_bounds__(256) crd_aos_to_planes_kernel(const Src *__restrict__ aos, Real *__restrict__ u, Real *__restrict__ v, size_t n)
{
	for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
		u[q] = (Real)aos[2 * q];
		v[q] = (Real)aos[2 * q + 1];
	}
}

template <typename Dst, typename Real>
__global__ void __launch_bounds__(256) crd_planes_to_aos_kernel(const Real *__restrict__ u, const Real *__restrict__ v, Dst *__restrict__ aos, size_t n)
{
	for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
		aos[2 * q] = (Dst)u[q];
		aos[2 * q + 1] = (Dst)v[q];
	}
}

template <typename Real>
__global__ void __launch_bounds__(256) crd_aos_row_extract_kernel(const Real *__restrict__ aos_row, Real *__restrict__ row, int nx)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nx) row[i] = aos_row[2 * (size_t)i];
}

// column 0 and column nx-1 of rows 0 .. nyl-1 (element stride `stride` reals: 1 for a plane, 2 for var0 of an AoS vector)
template <typename Real>
__global__ void __launch_bounds__(256) crd_cols_extract_kernel(const Real *__restrict__ base, Real *__restrict__ col_w, Real *__restrict__ col_e, int nx, int nyl, int stride)
{
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < nyl) {
		const size_t row = (size_t)j * (size_t)nx * (size_t)stride;
		col_w[j] = base[row];
		col_e[j] = base[row + (size_t)(nx - 1) * (size_t)stride];
	}
}

template <typename Real>
__global__ void __launch_bounds__(256) crd_max_abs_kernel(const Real *__restrict__ u, size_t n, double *out)
{
	__shared__ double part[4];
	double m = 0.0;
	for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
		const double a = fabs((double)u[q]);
		m = (a > m || a != a) ? a : m;  // NaN propagates
	}
	for (int off = 32; off > 0; off >>= 1) {
		const double o = __shfl_down(m, off, 64);
		m = (o > m || o != o) ? o : m;
	}
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < 4; w++) m = (part[w] > m || part[w] != part[w]) ? part[w] : m;
		// non-negative doubles order like their bit patterns; NaN (0x7ff8...) sorts above every finite value
		atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long)__double_as_longlong(m));
	}
}

inline int grid_for(size_t n) { return (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048); }

template <typename Real, int MODEL>
hipError_t launch_stage_t(const SlabDesc &d, const StageCall &c, int row_begin, int row_end, hipStream_t st)
{
	clear_launch_status();
	const Slab<Real> s = typed<Real>(d);
	StageArgs<Real> a;
	a.in_u = row0<Real>(c.yin.u, d.nx);
	a.in_v = row0<Real>(c.yin.v, d.nx);
	a.y0_u = row0<Real>(c.y0.u, d.nx);
	a.y0_v = row0<Real>(c.y0.v, d.nx);
	a.acc_u = row0<Real>(c.acc.u, d.nx);
	a.acc_v = row0<Real>(c.acc.v, d.nx);
	a.out_u = row0<Real>(c.yout.u, d.nx);
	a.out_v = row0<Real>(c.yout.v, d.nx);
	a.absorb = c.absorb;
	a.gw = static_cast<const Real *>(c.gcol_w);
	a.ge = static_cast<const Real *>(c.gcol_e);
	if (!d.wrap_x && (!a.gw || !a.ge)) return hipErrorInvalidValue;
	switch (c.stage) {
	case 1: a.h_out = (Real)(0.5 * c.dt); a.h_acc = (Real)(c.dt / 6.0); break;
	case 2: a.h_out = (Real)(0.5 * c.dt); a.h_acc = (Real)(c.dt / 3.0); break;
	case 3: a.h_out = (Real)c.dt; a.h_acc = (Real)(c.dt / 3.0); break;
	case 4: a.h_out = (Real)0; a.h_acc = (Real)(c.dt / 6.0); break;
	default: a.h_out = a.h_acc = (Real)0; break;
	}
	if (row_end <= row_begin) return hipSuccess;
	const int nbx = (d.nx + kTX - 1) / kTX, nby = (row_end - row_begin + kTY - 1) / kTY;
	const int nblocks = nbx * nby;
	const dim3 block(kTX, kBY);
	auto fire = [&](auto nt_c) -> bool {
		constexpr bool NT = decltype(nt_c)::value;
		switch (c.stage) {
		case 0: crd_rk4_stage_kernel<Real, MODEL, 0, NT><<<nblocks, block, 0, st>>>(s, a, row_begin, row_end, nbx, nblocks); break;
		case 1: crd_rk4_stage_kernel<Real, MODEL, 1, NT><<<nblocks, block, 0, st>>>(s, a, row_begin, row_end, nbx, nblocks); break;
		case 2: crd_rk4_stage_kernel<Real, MODEL, 2, NT><<<nblocks, block, 0, st>>>(s, a, row_begin, row_end, nbx, nblocks); break;
		case 3: crd_rk4_stage_kernel<Real, MODEL, 3, NT><<<nblocks, block, 0, st>>>(s, a, row_begin, row_end, nbx, nblocks); break;
		case 4: crd_rk4_stage_kernel<Real, MODEL, 4, NT><<<nblocks, block, 0, st>>>(s, a, row_begin, row_end, nbx, nblocks); break;
		default: return false;
		}
		return true;
	};
	if (!(stores_nontemporal(d, sizeof(Real)) ? fire(std::true_type{}) : fire(std::false_type{}))) return hipErrorInvalidValue;
	return launch_status();
}

template <typename Real, int MODEL>
hipError_t launch_rhs_aos_t(const SlabDesc &d, int absorb, const void *y, void *ydot, const void *glo, const void *ghi, int row_begin, int row_end,
                            hipStream_t st, const void *gw, const void *ge)
{
	clear_launch_status();
	if (row_end <= row_begin) return hipSuccess;
	if (!d.wrap_x && (!gw || !ge)) return hipErrorInvalidValue;
	using P = typename Pair<Real>::type;
	const Slab<Real> s = typed<Real>(d);
	const int nbx = (d.nx + kTX - 1) / kTX, nby = (row_end - row_begin + kTY - 1) / kTY;
	const int nblocks = nbx * nby;
	auto fire = [&](auto nt_c) {
		crd_rhs_aos_kernel<Real, MODEL, decltype(nt_c)::value><<<nblocks, dim3(kTX, kBY), 0, st>>>(
		    s, static_cast<const P *>(y), static_cast<P *>(ydot), static_cast<const Real *>(glo), static_cast<const Real *>(ghi), static_cast<const Real *>(gw),
		    static_cast<const Real *>(ge), absorb, row_begin, row_end, nbx, nblocks);
	};
	if (stores_nontemporal(d, sizeof(Real))) fire(std::true_type{});
	else fire(std::false_type{});
	return launch_status();
}

}  // namespace

hipError_t launch_stage(int precision, const SlabDesc &d, const StageCall &c, int row_begin, int row_end, hipStream_t s)
{
	const int model = kernel_model(d);
#define CRD_STAGE_DISPATCH(REAL)                                                                            \
	switch (model) {                                                                                        \
	case CRD_MODEL_FHN: return launch_stage_t<REAL, CRD_MODEL_FHN>(d, c, row_begin, row_end, s);            \
	case CRD_MODEL_GOLDBETER: return launch_stage_t<REAL, CRD_MODEL_GOLDBETER>(d, c, row_begin, row_end, s); \
	default: return launch_stage_t<REAL, kModelDiffusionOnly>(d, c, row_begin, row_end, s);                 \
	}
	if (precision == CRD_PRECISION_F64) CRD_STAGE_DISPATCH(double)
	CRD_STAGE_DISPATCH(float)
#undef CRD_STAGE_DISPATCH
}

const char *stage_kernel_name(int, int) { return "crd_rk4_stage_kernel"; }

hipError_t launch_rhs_aos(int precision, const SlabDesc &d, int absorb, const void *y, void *ydot, const void *ghost_lo, const void *ghost_hi,
                          int row_begin, int row_end, hipStream_t s, const void *gcol_w, const void *gcol_e)
{
	const int model = kernel_model(d);
#define CRD_AOS_DISPATCH(REAL)                                                                                                               \
	switch (model) {                                                                                                                         \
	case CRD_MODEL_FHN: return launch_rhs_aos_t<REAL, CRD_MODEL_FHN>(d, absorb, y, ydot, ghost_lo, ghost_hi, row_begin, row_end, s, gcol_w, gcol_e);         \
	case CRD_MODEL_GOLDBETER: return launch_rhs_aos_t<REAL, CRD_MODEL_GOLDBETER>(d, absorb, y, ydot, ghost_lo, ghost_hi, row_begin, row_end, s, gcol_w, gcol_e); \
	default: return launch_rhs_aos_t<REAL, kModelDiffusionOnly>(d, absorb, y, ydot, ghost_lo, ghost_hi, row_begin, row_end, s, gcol_w, gcol_e);              \
	}
	if (precision == CRD_PRECISION_F64) CRD_AOS_DISPATCH(double)
	CRD_AOS_DISPATCH(float)
#undef CRD_AOS_DISPATCH
}

hipError_t launch_aos_to_planes(int precision, int src_is_f64, const void *aos, Planes dst, int nx, int nyl, hipStream_t s)
{
	clear_launch_status();
	const size_t n = (size_t)nx * (size_t)nyl;
	if (n == 0) return hipSuccess;
	const int g = grid_for(n);
	if (precision == CRD_PRECISION_F64) {
		if (!src_is_f64) return hipErrorInvalidValue;
		crd_aos_to_planes_kernel<double, double><<<g, 256, 0, s>>>(static_cast<const double *>(aos), row0<double>(dst.u, nx), row0<double>(dst.v, nx), n);
	} else if (src_is_f64) {
		crd_aos_to_planes_kernel<double, float><<<g, 256, 0, s>>>(static_cast<const double *>(aos), row0<float>(dst.u, nx), row0<float>(dst.v, nx), n);
	} else {
		crd_aos_to_planes_kernel<float, float><<<g, 256, 0, s>>>(static_cast<const float *>(aos), row0<float>(dst.u, nx), row0<float>(dst.v, nx), n);
	}
	return launch_status();
}

hipError_t launch_planes_to_aos(int precision, int dst_is_f64, Planes src, void *aos, int nx, int nyl, hipStream_t s)
{
	clear_launch_status();
	const size_t n = (size_t)nx * (size_t)nyl;
	if (n == 0) return hipSuccess;
	const int g = grid_for(n);
	if (precision == CRD_PRECISION_F64) {
		if (!dst_is_f64) return hipErrorInvalidValue;
		crd_planes_to_aos_kernel<double, double><<<g, 256, 0, s>>>(row0<double>(src.u, nx), row0<double>(src.v, nx), static_cast<double *>(aos), n);
	} else if (dst_is_f64) {
		crd_planes_to_aos_kernel<double, float><<<g, 256, 0, s>>>(row0<float>(src.u, nx), row0<float>(src.v, nx), static_cast<double *>(aos), n);
	} else {
		crd_planes_to_aos_kernel<float, float><<<g, 256, 0, s>>>(row0<float>(src.u, nx), row0<float>(src.v, nx), static_cast<float *>(aos), n);
	}
	return launch_status();
}

hipError_t launch_aos_row_extract(int precision, const void *aos, void *row, int nx, int j, hipStream_t s)
{
	clear_launch_status();
	const int g = (nx + 255) / 256;
	if (precision == CRD_PRECISION_F64)
		crd_aos_row_extract_kernel<double><<<g, 256, 0, s>>>(static_cast<const double *>(aos) + 2 * (size_t)j * nx, static_cast<double *>(row), nx);
	else
		crd_aos_row_extract_kernel<float><<<g, 256, 0, s>>>(static_cast<const float *>(aos) + 2 * (size_t)j * nx, static_cast<float *>(row), nx);
	return launch_status();
}

// Cubic Hermite interpolant of a step [t_n, t_n + h] at theta = (t - t_n) / h from y_n, y_{n+1}, f_n, f_{n+1} (the degree-3
// dense output ARKode's ARK_NORMAL mode returns, src/FHNmodel_torus.cpp:423): whole planes, ghost rows excluded.
template <typename Real>
__global__ void __launch_bounds__(256) crd_hermite_kernel(const Real *__restrict__ yn, const Real *__restrict__ yp, const Real *__restrict__ fn,
                                                          const Real *__restrict__ fp, Real *__restrict__ out, size_t n, Real h00, Real h10, Real h01, Real h11)
{
	for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x)
		out[q] = fmadd(h00, yn[q], fmadd(h10, fn[q], fmadd(h01, yp[q], h11 * fp[q])));
}

hipError_t launch_hermite(int precision, Planes yn, Planes yp, Planes fn, Planes fp, Planes out, int nx, int nyl, double theta, double h, hipStream_t s)
{
	clear_launch_status();
	const size_t n = (size_t)nx * (size_t)nyl;
	if (n == 0) return hipSuccess;
	const double t2 = theta * theta, t3 = t2 * theta;
	const double h00 = 2.0 * t3 - 3.0 * t2 + 1.0, h10 = (t3 - 2.0 * t2 + theta) * h, h01 = -2.0 * t3 + 3.0 * t2, h11 = (t3 - t2) * h;
	const int g = grid_for(n);
	void *a[5][2] = {{yn.u, yn.v}, {yp.u, yp.v}, {fn.u, fn.v}, {fp.u, fp.v}, {out.u, out.v}};
	for (int f = 0; f < 2; f++) {
		if (precision == CRD_PRECISION_F64)
			crd_hermite_kernel<double><<<g, 256, 0, s>>>(row0<double>(a[0][f], nx), row0<double>(a[1][f], nx), row0<double>(a[2][f], nx), row0<double>(a[3][f], nx),
			                                           row0<double>(a[4][f], nx), n, h00, h10, h01, h11);
		else
			crd_hermite_kernel<float><<<g, 256, 0, s>>>(row0<float>(a[0][f], nx), row0<float>(a[1][f], nx), row0<float>(a[2][f], nx), row0<float>(a[3][f], nx),
			                                          row0<float>(a[4][f], nx), n, (float)h00, (float)h10, (float)h01, (float)h11);
	}
	return launch_status();
}

// ---- the three vector operations of ARKode's initial-step estimate (arkHin / arkUpperBoundH0 / arkYddNorm, restated in
// oracle/arkode_erk.py; the reference gets them from ARKodeInit + the first ARKode call, src/FHNmodel_torus.cpp:362,423) ----

// max_i |f_i| / (0.1 |y_i| + rtol |y_i| + atol) over one field, folded into *out with an atomic max on the bit pattern
template <typename Real>
__global__ void __launch_bounds__(256) crd_hin_bound_kernel(const Real *__restrict__ y, const Real *__restrict__ f, size_t n, double rtol, double atol, double *out)
{
	__shared__ double part[4];
	double m = 0.0;
	for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
		const double ay = fabs((double)y[q]);
		const double r = fabs((double)f[q]) / (0.1 * ay + (rtol * ay + atol));
		m = (r > m || r != r) ? r : m;
	}
	for (int off = 32; off > 0; off >>= 1) {
		const double o = __shfl_down(m, off, 64);
		m = (o > m || o != o) ? o : m;
	}
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < 4; w++) m = (part[w] > m || part[w] != part[w]) ? part[w] : m;
		atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long)__double_as_longlong(m));
	}
}

// out = y + h f
template <typename Real>
__global__ void __launch_bounds__(256) crd_axpy_kernel(const Real *__restrict__ y, const Real *__restrict__ f, Real h, Real *__restrict__ out, size_t n)
{
	for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) out[q] = fmadd(h, f[q], y[q]);
}

// partials[block] (+)= sum over the block's elements of (((f2 - f0) / h) / (rtol |y| + atol))^2: one partial per block, blocks
// stride the field in a fixed pattern and the partials are added in block order afterwards, so the norm is reproducible
constexpr int kNormBlocks = 256;
template <typename Real>
__global__ void __launch_bounds__(256) crd_ydd_sumsq_kernel(const Real *__restrict__ f2, const Real *__restrict__ f0, const Real *__restrict__ y, size_t n, double inv_h,
                                                            double rtol, double atol, double *__restrict__ partials, int accumulate)
{
	__shared__ double part[256];
	double sum = 0.0;
	for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
		const double e = ((double)f2[q] - (double)f0[q]) * inv_h / (rtol * fabs((double)y[q]) + atol);
		sum = fma(e, e, sum);
	}
	part[threadIdx.x] = sum;
	__syncthreads();
	for (int w = 128; w > 0; w >>= 1) {
		if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
		__syncthreads();
	}
	if (threadIdx.x == 0) partials[blockIdx.x] = (accumulate ? partials[blockIdx.x] : 0.0) + part[0];
}

__global__ void __launch_bounds__(kNormBlocks) crd_sum_blocks_kernel(const double *__restrict__ partials, double *__restrict__ out)
{
	__shared__ double part[kNormBlocks];
	part[threadIdx.x] = partials[threadIdx.x];
	__syncthreads();
	for (int w = kNormBlocks / 2; w > 0; w >>= 1) {
		if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
		__syncthreads();
	}
	if (threadIdx.x == 0) *out = part[0];
}

template <typename Real>
hipError_t launch_hin_ops_t(int op, Planes y, Planes a, Planes b, int nx, int nyl, double h, double rtol, double atol, double *partials, double *out, hipStream_t st)
{
	const size_t n = (size_t)nx * (size_t)nyl;
	const int g = grid_for(n);
	void *yy[2] = {y.u, y.v}, *aa[2] = {a.u, a.v}, *bb[2] = {b.u, b.v};
	for (int f = 0; f < 2; f++) {
		const Real *yp = row0<Real>(yy[f], nx), *ap = row0<Real>(aa[f], nx);
		Real *bp = row0<Real>(bb[f], nx);
		if (op == 0) crd_hin_bound_kernel<Real><<<g, 256, 0, st>>>(yp, ap, n, rtol, atol, out);
		else if (op == 1) crd_axpy_kernel<Real><<<g, 256, 0, st>>>(yp, ap, (Real)h, bp, n);
		else crd_ydd_sumsq_kernel<Real><<<kNormBlocks, 256, 0, st>>>(bp, ap, yp, n, 1.0 / h, rtol, atol, partials, f);
	}
	if (op == 2) crd_sum_blocks_kernel<<<1, kNormBlocks, 0, st>>>(partials, out);
	return launch_status();
}

hipError_t launch_hin_bound(int precision, Planes y, Planes f, int nx, int nyl, double rtol, double atol, double *out_dev, hipStream_t s)
{
	clear_launch_status();
	hipError_t e = hipMemsetAsync(out_dev, 0, sizeof(double), s);
	if (e != hipSuccess) return e;
	return precision == CRD_PRECISION_F64 ? launch_hin_ops_t<double>(0, y, f, Planes{nullptr, nullptr}, nx, nyl, 0.0, rtol, atol, nullptr, out_dev, s)
	                                      : launch_hin_ops_t<float>(0, y, f, Planes{nullptr, nullptr}, nx, nyl, 0.0, rtol, atol, nullptr, out_dev, s);
}

hipError_t launch_axpy_planes(int precision, Planes y, Planes f, double h, Planes out, int nx, int nyl, hipStream_t s)
{
	clear_launch_status();
	return precision == CRD_PRECISION_F64 ? launch_hin_ops_t<double>(1, y, f, out, nx, nyl, h, 0.0, 0.0, nullptr, nullptr, s)
	                                      : launch_hin_ops_t<float>(1, y, f, out, nx, nyl, h, 0.0, 0.0, nullptr, nullptr, s);
}

hipError_t launch_ydd_sumsq(int precision, Planes y, Planes f0, Planes f2, double h, double rtol, double atol, int nx, int nyl, double *partials_dev, double *out_dev,
                            hipStream_t s)
{
	clear_launch_status();
	return precision == CRD_PRECISION_F64 ? launch_hin_ops_t<double>(2, y, f0, f2, nx, nyl, h, rtol, atol, partials_dev, out_dev, s)
	                                      : launch_hin_ops_t<float>(2, y, f0, f2, nx, nyl, h, rtol, atol, partials_dev, out_dev, s);
}

hipError_t launch_aos_cols_extract(int precision, const void *aos, void *col_w, void *col_e, int nx, int nyl, hipStream_t s)
{
	clear_launch_status();
	if (nyl <= 0) return hipSuccess;
	const int g = (nyl + 255) / 256;
	if (precision == CRD_PRECISION_F64) crd_cols_extract_kernel<double><<<g, 256, 0, s>>>(static_cast<const double *>(aos), static_cast<double *>(col_w), static_cast<double *>(col_e), nx, nyl, 2);
	else crd_cols_extract_kernel<float><<<g, 256, 0, s>>>(static_cast<const float *>(aos), static_cast<float *>(col_w), static_cast<float *>(col_e), nx, nyl, 2);
	return launch_status();
}

hipError_t launch_plane_cols_extract(int precision, const void *u_plane, void *col_w, void *col_e, int nx, int nyl, hipStream_t s)
{
	clear_launch_status();
	if (nyl <= 0) return hipSuccess;
	const int g = (nyl + 255) / 256;
	if (precision == CRD_PRECISION_F64)
		crd_cols_extract_kernel<double><<<g, 256, 0, s>>>(row0<double>(const_cast<void *>(u_plane), nx), static_cast<double *>(col_w), static_cast<double *>(col_e), nx, nyl, 1);
	else crd_cols_extract_kernel<float><<<g, 256, 0, s>>>(row0<float>(const_cast<void *>(u_plane), nx), static_cast<float *>(col_w), static_cast<float *>(col_e), nx, nyl, 1);
	return launch_status();
}

__global__ void crd_scalar_to_host_kernel(const double *__restrict__ src, double *__restrict__ dst) { *dst = *src; }

hipError_t launch_scalar_to_host(const double *src_dev, double *dst_host_mapped, hipEvent_t done, hipStream_t s)
{
	clear_launch_status();
	hipExtLaunchKernelGGL(crd_scalar_to_host_kernel, dim3(1), dim3(1), 0, s, nullptr, done, 0, src_dev, dst_host_mapped);
	return launch_status();
}

hipError_t launch_max_abs(int precision, const void *u_plane, int nx, int nyl, double *out_dev, hipStream_t s)
{
	clear_launch_status();
	const size_t n = (size_t)nx * (size_t)nyl;
	hipError_t e = hipMemsetAsync(out_dev, 0, sizeof(double), s);
	if (e != hipSuccess || n == 0) return e;
	const int g = grid_for(n);
	if (precision == CRD_PRECISION_F64)
		crd_max_abs_kernel<double><<<g, 256, 0, s>>>(row0<double>(const_cast<void *>(u_plane), nx), n, out_dev);
	else
		crd_max_abs_kernel<float><<<g, 256, 0, s>>>(row0<float>(const_cast<void *>(u_plane), nx), n, out_dev);
	return launch_status();
}

}  // namespace crd
