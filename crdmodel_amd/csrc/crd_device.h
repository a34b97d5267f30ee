// crd_device.h -- device-side helpers shared by the kernel translation units (crd_kernels.hip, crd_fused.hip):
// the typed slab descriptor, the point function (diffusion + kinetics of one grid point) and the XCD tile remap.
#pragma once

#include <hip/hip_runtime.h>

#include "crd_internal.h"
#include "crd_kernels.h"

namespace crd {
namespace dev {

constexpr int kNumXcd = 8;

// Status of the launch just made.  HIP keeps the last error of ANY earlier call on this thread until it is read, so the
// launchers clear it before launching (clear_launch_status) and read it after: a stale error from an unrelated, already
// handled failure (e.g. a refused crd_create) must not be pinned on a later, successful launch.
inline void clear_launch_status() { (void)hipGetLastError(); }
inline hipError_t launch_status() { return hipGetLastError(); }

template <typename Real> struct Pair;
template <> struct Pair<double> { using type = double2; };
template <> struct Pair<float> { using type = float2; };

template <typename Real>
struct Slab {
	const Real *cE, *cWn, *cP, *brow;
	Real ka4;
	int nx, nyl, wrap, has_row0, has_rowN, just_diffusion, wrap_x;
};

template <typename Real>
inline Slab<Real> typed(const SlabDesc &d)
{
	Slab<Real> s;
	s.cE = static_cast<const Real *>(d.cE);
	s.cWn = static_cast<const Real *>(d.cWn);
	s.cP = static_cast<const Real *>(d.cP);
	s.brow = static_cast<const Real *>(d.brow) + kGhost;  // index by local row
	s.ka4 = (Real)d.ka4;
	s.nx = d.nx;
	s.nyl = d.nyl;
	s.wrap = d.wrap;
	s.has_row0 = d.has_row0;
	s.has_rowN = d.has_rowN;
	s.just_diffusion = d.just_diffusion;
	s.wrap_x = d.wrap_x;
	return s;
}

// Pointer to local row 0 of a plane (ghost rows sit at negative row offsets).
template <typename Real>
inline Real *row0(void *plane, int nx)
{
	return plane ? static_cast<Real *>(plane) + (size_t)kGhost * (size_t)nx : nullptr;
}

// The point function: diffusion + kinetics of one grid point.
//   diffusion  src/FHNmodel_torus.cpp:535-537 with the theta-only factors folded into the tables cE / cWn / cP
//   FHN        src/FHNmodel_torus.cpp:657,660
//   Goldbeter  src/GoldbeterModel_torus.cpp:694-695,715-716 (pow(x,2), pow(x,4) as multiplies)
//   absorbing  src/FHNmodel_torus.cpp:643-653 (zero = row is a global phi boundary row and t < TBOUNDARY)
// Fused multiply-add, spelled out.  Implicit contraction is switched off for the kernels (pragma below): left to the
// compiler, `a*b + c*d` may become fma(a,b,c*d) in one inlined copy of the point function and fma(c,d,a*b) in another, and
// the same grid point then rounds differently depending on which pipeline slot or which slab computes it.  With every fma
// explicit, a point's arithmetic is one fixed sequence: results are bit-identical across steppers' redundant recomputation,
// slab decompositions and compiler versions.
#pragma clang fp contract(off)
__device__ __forceinline__ double fmadd(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fmadd(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Two grid columns per lane (crd_fused.hip, COLS = 2): the lane's values are a two-element vector.  In fp32 the element-wise
// operations on it are the packed instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two columns per issue slot, which is
// where gfx950's fp32 vector rate beyond its fp64 rate comes from); in fp64 they are two independent instruction streams in
// one wavefront.  Element by element the arithmetic is the scalar sequence exactly (IEEE fma / mul / add, contraction off), so
// one and two columns per lane give the same bits.
typedef float float2v __attribute__((ext_vector_type(2)));
typedef double double2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2v fmadd(float2v a, float2v b, float2v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ double2v fmadd(double2v a, double2v b, double2v c) { return __builtin_elementwise_fma(a, b, c); }

template <typename Real, int COLS> struct LaneValue { using type = Real; };
template <> struct LaneValue<float, 2> { using type = float2v; };
template <> struct LaneValue<double, 2> { using type = double2v; };
template <typename V> struct ScalarOf { using type = V; };
template <> struct ScalarOf<float2v> { using type = float; };
template <> struct ScalarOf<double2v> { using type = double; };
// the constant c in every element of a V
template <typename V>
__device__ __forceinline__ V splat(double c)
{
	return (V)((typename ScalarOf<V>::type)c);
}

// A constant held in vector registers.  A VOP3 instruction reads at most ONE scalar operand: fma(constant, x, row parameter) with both
// in scalar registers costs a v_mov of one of them in front of every use -- one vector instruction in 17.5 per FHN stage-point.  The
// empty asm makes the value opaque (it cannot be rematerialised as a scalar) and, not being volatile, is hoisted out of the loops: one
// register pair for the whole kernel.
template <typename V>
__device__ __forceinline__ V in_vector_registers(V x)
{
#ifndef CRD_NO_VECTOR_CONSTANTS  // (experiment switch)
	asm("" : "+v"(x));
#endif
	return x;
}

// 1 / x for a positive, normal x well inside the exponent range (here x >= K2^2 KR^2 KA^4 / VM3 = 5.2e-3): the hardware estimate
// (v_rcp_f64: relative error 2^-24.4, and 3.3 times the issue cost of an fp64 FMA -- tools/rcp_probe.hip) refined by ONE Newton
// step: relative error <= 2.2e-15 = 2^-48.7 measured over [2.6, 1e6] (relative: the scale of the range does not matter), three orders of magnitude inside the 1e-12 parity
// tolerance of a right-hand side, for 3 instructions against the 11 of the compiler's correctly rounded division (a second
// step, CRD_RCP_NEWTON=2, brings 2^-53 for two more).
#ifndef CRD_RCP_NEWTON
#define CRD_RCP_NEWTON 1
#endif
__device__ __forceinline__ double reciprocal(double x)
{
	double r = __builtin_amdgcn_rcp(x);
	r = fmadd(r, fmadd(-x, r, 1.0), r);
	if (CRD_RCP_NEWTON >= 2) r = fmadd(r, fmadd(-x, r, 1.0), r);
	return r;
}
__device__ __forceinline__ float reciprocal(float x)
{
	float r = __builtin_amdgcn_rcpf(x);
	return fmadd(r, fmadd(-x, r, 1.0f), r);
}
__device__ __forceinline__ float2v reciprocal(float2v x)
{
	float2v r;
	r.x = __builtin_amdgcn_rcpf(x.x);
	r.y = __builtin_amdgcn_rcpf(x.y);
	return fmadd(r, fmadd(-x, r, splat<float2v>(1.0)), r);
}
__device__ __forceinline__ double2v reciprocal(double2v x)
{
	double2v r;
	r.x = reciprocal(x.x);
	r.y = reciprocal(x.y);
	return r;
}

// Third point-function variant besides CRD_MODEL_FHN / CRD_MODEL_GOLDBETER: Goldbeter with justDiffusion = 1, where the whole
// reaction block, absorbing rows included, is skipped (src/GoldbeterModel_torus.cpp:668).  A compile-time variant: as a
// run-time flag it cost every Goldbeter stage four conditional moves.
constexpr int kModelDiffusionOnly = 2;
inline int kernel_model(const SlabDesc &d) { return (d.model == CRD_MODEL_GOLDBETER && d.just_diffusion) ? kModelDiffusionOnly : d.model; }

// rowp is the per-row parameter of the kinetics, formed once on the host (crd_create): FHN EPSILON b(j) (src/FHNmodel_torus.cpp:623-632,
// 660); Goldbeter v0 + v1 b(j), the row-constant source term of src/GoldbeterModel_torus.cpp:715.
// V: the lane's value type -- a Real, or two of them (two columns per lane); rowp, ka4 are the same for both columns.
//
// Round 5: the theta part of the operator works on FIRST differences.  With gE = uE - uC, gW = uC - uW,
//   cA (uE - uW) + cX (uE - 2 uC + uW)  =  (cX + cA) gE - (cX - cA) gW  =  cE gE + cWn gW
// (cE = cX + cA and cWn = cA - cX are tables, build_coefficients): two fused multiply-adds on one difference per point -- the
// neighbour's gE IS this point's gW, so the one-launch steppers form one difference per point and fetch the other with the lane
// shift they need anyway -- where round 4 spent five instructions (uE - uW, two for the second difference, two multiply-adds).
// Like the second difference, a first difference of neighbouring values is exact or correct to half an ulp of u, and a uniform
// field still diffuses to exactly zero (gE = gW = 0, d2y = 0).  The reaction terms ride the same chain: FHN's "- v" is the
// chain's first addend, 3 u - u^3 = u (3 - u^2) is two multiply-adds, EPSILON (u + b) one with EPSILON b(j) as the row parameter:
// 9 arithmetic instructions per FHN point (round 4: 15), Goldbeter 25 (31) -- these kernels are bound by vector issue.
// `west(X)` = fmadd(cWn, gW, X), the western difference's term, LAST in the diffusion's chain: how it is formed is the caller's
// (rhs_point below: from gW as a value; the packed fp32 marching kernels: without ever assembling gW, crd_fused_impl.h: rhs_lane).
template <typename V, int MODEL, typename West>
__device__ __forceinline__ void rhs_point_west(V uC, V gE, V uS, V uN, V v, V cE, V cP, typename ScalarOf<V>::type rowp,
                                               typename ScalarOf<V>::type ka4, bool zero, V &du, V &dv, West &&west)
{
	// the phi second difference as the reference writes it, (uN - 2 uC + uS)
	const V d2y = fmadd(splat<V>(-2.0), uC, uN) + uS;
	if (MODEL == kModelDiffusionOnly) {
		du = west(fmadd(cE, gE, cP * d2y));
		dv = splat<V>(0.0);
		return;
	} else if (MODEL == CRD_MODEL_FHN) {
		const V r = west(fmadd(cE, gE, fmadd(cP, d2y, -v)));  // diffusion - v
		du = fmadd(uC, fmadd(-uC, uC, splat<V>(3.0)), r);               // + u (3 - u^2)
		dv = fmadd(in_vector_registers(splat<V>(kFhnEpsilon)), uC, (V)rowp);  // rowp = EPSILON b
	} else {
		// v2 = VM2 z^2 / (K2^2 + z^2), v3 = VM3 y^2 z^4 / ((KR^2 + y^2)(KA^4 + z^4)) enter both equations only through
		// w = v2 - v3 (src/GoldbeterModel_torus.cpp:715-716: dZ = v0 + v1 b - w + kf Y - k Z, dY = w - kf Y): one quotient
		// w = (VM2 z^2 dB - y^2 z^4 dA) / (dA dB) with dA = K2^2 + z^2, dB = (KR^2 + y^2)(KA^4 + z^4) / VM3.
		// Numerator and denominator both divided by VM3 -- it enters as the constants of ONE fused multiply-add, (KR^2 + y^2) / VM3 --
		// which saves the product VM3 (y^2 z^4 dA): 25 arithmetic instructions per point.
		const V z2 = uC * uC, z4 = z2 * z2, y2 = v * v;
		const V dA = splat<V>(kGbK2 * kGbK2) + z2, dB = fmadd(splat<V>(1.0 / kGbVm3), y2, splat<V>(kGbKr * kGbKr / kGbVm3)) * fmadd(z2, z2, (V)ka4);
		const V w = fmadd(splat<V>(kGbVm2), z2 * dB, -((y2 * z4) * dA)) * reciprocal(dA * dB);
		dv = fmadd(splat<V>(-kGbKf), v, w);
		const V r = fmadd(splat<V>(-kGbK), uC, (V)rowp - dv);  // rowp = v0 + v1 b
		du = west(fmadd(cE, gE, fmadd(cP, d2y, r)));
	}
	if (zero) {
		du = splat<V>(0.0);
		dv = splat<V>(0.0);
	}
}
template <typename V, int MODEL>
__device__ __forceinline__ void rhs_point(V uC, V gW, V gE, V uS, V uN, V v, V cE, V cWn, V cP, typename ScalarOf<V>::type rowp,
                                          typename ScalarOf<V>::type ka4, bool zero, V &du, V &dv)
{
	rhs_point_west<V, MODEL>(uC, gE, uS, uN, v, cE, cP, rowp, ka4, zero, du, dv, [&](V X) { return fmadd(cWn, gW, X); });
}

// ... from the neighbour VALUES (the tiled kernels, which have them in LDS): the same differences, the same bits.
template <typename V, int MODEL>
__device__ __forceinline__ void rhs_point_values(V uC, V uW, V uE, V uS, V uN, V v, V cE, V cWn, V cP, typename ScalarOf<V>::type rowp,
                                                 typename ScalarOf<V>::type ka4, bool zero, V &du, V &dv)
{
	rhs_point<V, MODEL>(uC, uC - uW, uE - uC, uS, uN, v, cE, cWn, cP, rowp, ka4, zero, du, dv);
}

// blockIdx -> tile id.  Workgroups are dealt round-robin over the 8 XCDs (each with a private L2), so tile ids that
// are consecutive in phi would land on eight different L2s and every halo row would be fetched twice from beyond
// L2.  This bijection hands each XCD one contiguous run of tiles instead.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks)
{
	const int q = nblocks / kNumXcd, rem = nblocks - q * kNumXcd;
	const int x = bid % kNumXcd, l = bid / kNumXcd;
	return x * q + (x < rem ? x : rem) + l;
}

template <typename Real>
__device__ __forceinline__ int wrap_row(const Slab<Real> &s, int j)
{
	if (s.wrap) {
		if (j < 0) j += s.nyl;
		else if (j >= s.nyl) j -= s.nyl;
	}
	return j;
}

}  // namespace dev
}  // namespace crd
