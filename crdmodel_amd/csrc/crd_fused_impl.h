#pragma once
// crd_fused_impl.h -- (included by crd_fused.hip, the fp64 instantiations, and crd_fused_f32.hip, the fp32 ones: two translation
// units so that the two halves of the instantiation matrix compile side by side)
// One classical RK4 step of the whole slab in ONE kernel launch: all four RHS evaluations and the
// stage updates happen on chip, so a grid-point-step costs one read and one write of the state (32 B in fp64) instead
// of the 256 B the four stage kernels of crd_kernels.hip move.  Same arithmetic per point as the staged stepper.
//
// Structure (no LDS, no MFMA; no data passes between wavefronts -- the one barrier per iteration only keeps the four
// wavefronts of a block in step): every WAVEFRONT is an independent work item.  It owns a strip of 64
// consecutive theta columns -- one column per lane, 56 valid outputs in the middle and a 4-column apron on each side
// that is recomputed redundantly -- and marches along phi through a chunk of rows as a 4-deep software pipeline:
//   iteration m:  take row p        (state y0, fetched four iterations earlier)
//                 stage 1 on row p-1 (needs y0 rows p-2..p)          -> y1 row p-1, acc row p-1
//                 stage 2 on row p-2 (needs y1 rows p-3..p-1)        -> y2 row p-2
//                 stage 3 on row p-3 (needs y2 rows p-4..p-2)        -> y3 row p-3
//                 stage 4 on row p-4 (needs y3 rows p-5..p-3)        -> new state row p-4, stored
// The phi neighbours of a row are the lane's own registers from neighbouring iterations; the theta neighbours are the
// adjacent lanes' registers, fetched with DPP wavefront shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1).  Each stage
// invalidates one more apron column per side, hence 4 + 4 of 64; chunks start 4 rows early and end 4 rows late for the
// same reason in phi (rows come from the slab's ghost rows, or wrap for a single slab).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "crd_device.h"
#include "crd_tuning.h"

namespace crd {

namespace {

using namespace dev;

constexpr int kApron = 4;                    // RK4 stages = halo depth
constexpr int kLanes = 64;
constexpr int kValid = kLanes - 2 * kApron;  // 56 output columns per wavefront
constexpr int kWavesPerBlock = 4;     // default; the launch may use 1 .. kMaxWavesPerBlock (tuning knob)
constexpr int kMaxWavesPerBlock = 4;  // (8 strips per workgroup never measured faster than 4)
// Rows in flight per wavefront (a divisor of the unroll factor, so slots stay static).  Tuning builds override per model.
#ifndef CRD_PREFETCH_FHN
#define CRD_PREFETCH_FHN 4
#endif
#ifndef CRD_PREFETCH_GB
#define CRD_PREFETCH_GB 4
#endif
#ifndef CRD_EMBED_SLOTS
#define CRD_EMBED_SLOTS 6
#endif
#ifndef CRD_PREFETCH_EMBED
#define CRD_PREFETCH_EMBED 3
#endif

// Value held by lane-1 / lane+1 of this wavefront (the edge lane gets 0: it is apron garbage by design).  `old` = 0 with
// bound_ctrl lets the DPP move write its destination without a tied input, i.e. without a copy in front of it.
__device__ __forceinline__ double from_lane_below(double x)
{
	int lo = __double2loint(x), hi = __double2hiint(x);
	lo = __builtin_amdgcn_update_dpp(0, lo, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
	hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_lane_above(double x)
{
	int lo = __double2loint(x), hi = __double2hiint(x);
	lo = __builtin_amdgcn_update_dpp(0, lo, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
	hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float from_lane_below(float x)
{
	const int v = __float_as_int(x);
	return __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_lane_above(float x)
{
	const int v = __float_as_int(x);
	return __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true));
}
// Two columns per lane, (x, y) = columns (2 lane, 2 lane + 1): the western neighbours of the pair are (lane-1's y, own x), the
// eastern ones (own y, lane+1's x) -- one DPP move per direction for two columns instead of one per column.
template <typename V2>
__device__ __forceinline__ V2 pair_from_below(V2 v)
{
	V2 r;
	r.x = from_lane_below(v.y);
	r.y = v.x;
	return r;
}
template <typename V2>
__device__ __forceinline__ V2 pair_from_above(V2 v)
{
	V2 r;
	r.x = v.y;
	r.y = from_lane_above(v.x);
	return r;
}
__device__ __forceinline__ float2v from_lane_below(float2v v) { return pair_from_below(v); }
__device__ __forceinline__ float2v from_lane_above(float2v v) { return pair_from_above(v); }
__device__ __forceinline__ double2v from_lane_below(double2v v) { return pair_from_below(v); }
__device__ __forceinline__ double2v from_lane_above(double2v v) { return pair_from_above(v); }

// The point function on a lane of the marching wavefront: the theta neighbours are the adjacent lanes.  ONE first difference per
// point, gE = u(lane + 1) - u(lane); the other one, gW = u(lane) - u(lane - 1), is the gE of the lane below, fetched with the second
// shift (crd_device.h: rhs_point).  Edge lanes get apron garbage by design.
template <typename V, int MODEL>
__device__ __forceinline__ void rhs_lane(V uC, V uS, V uN, V v, V cE, V cWn, V cP, typename ScalarOf<V>::type rowp, typename ScalarOf<V>::type ka4, bool zero,
                                         V &du, V &dv)
{
#ifndef CRD_NO_PAIR_SHIFTS_FOLDED  // (experiment switch)
	if constexpr (std::is_same<V, float2v>::value) {
		// Packed fp32, two columns (x, y) per lane (round 6): the shifted PAIRS are never assembled.  pair_from_above / pair_from_below cost a
		// DPP move AND a plain move each (the packed instructions want both halves in one register pair): 100 plain moves per trip of the
		// three-step kernel's 797 vector instructions.  Written per column, the lane shift folds into the instruction that consumes it
		// (VOP2 with a DPP operand): gE.y = shl(u.x) - u.y is one v_sub_f32_dpp, the western term of column x one v_fmac_f32_dpp on gE.y
		// of the lane below, that of column y a plain multiply-add on gE.x.  Same operations on the same operands: same bits.
		V gE;
		gE.x = uC.y - uC.x;
		gE.y = from_lane_above(uC.x) - uC.y;
		rhs_point_west<V, MODEL>(uC, gE, uS, uN, v, cE, cP, rowp, ka4, zero, du, dv, [&](V X) {
			V r;
			r.x = fmadd(cWn.x, from_lane_below(gE.y), X.x);
			r.y = fmadd(cWn.y, gE.x, X.y);
			return r;
		});
		return;
	}
#endif
	const V gE = from_lane_above(uC) - uC;
	rhs_point<V, MODEL>(uC, from_lane_below(gE), gE, uS, uN, v, cE, cWn, cP, rowp, ka4, zero, du, dv);
}

// ---- the BLOCK as the strip (two-step Goldbeter pipeline, fp64; round 5) ----------------------------------------------------------
// The wavefronts of a block hold adjacent runs of 64 columns with ONE apron around all of them -- 240 valid columns of 256 instead of
// 4 x 48 of 4 x 64 -- and a wavefront's edge lanes take their theta neighbour from the wavefront next door through LDS.  Every value
// that is the centre of a later stage is published when it is formed (the lanes' 8 bytes at a per-lane address: lane 0 into its
// wavefront's western slot, lane 63 into the eastern one, the 62 in between into a dump area -- switching them off costs two writes
// of exec per value and measured more than the full-width write), the iteration ends with the block's barrier, and behind it the
// reads of what the neighbours published go out -- they land while the next iteration waits for its row.  Slots are double-buffered
// by the iteration's parity (a wavefront can be one barrier ahead of its neighbour, not two).  The neighbour's value enters the DPP
// shift as its `old` operand (bound_ctrl off: a lane without a source keeps `old`), so no instruction merges it; the western first
// difference of lane 0 is one subtraction more per stage-point.  Same arithmetic per point, same bits.
// What it buys is instructions: (603 / 563) x (48 / 60) = 0.86 of the vector work per useful column; what it costs is the barrier and
// eight LDS writes per iteration, which measured 12 - 15 % of an FHN iteration (150 vector instructions) and half that of a Goldbeter
// one (280).  So: Goldbeter, bound by issue, runs 6 - 8 % faster (4096^2: 0.111 -> 0.102 ms per step); FHN, bound by its memory
// traffic, does not (8192^2 0.2200 -> 0.2195; 4096^2 2 % slower) and keeps a strip per wavefront.
// profiles/r05/block_strip_ab.txt; round 4 tried the same at 173 VGPRs -- two wavefronts per SIMD -- and lost 13 %.
// THREE steps per launch (round 6) run the block as the strip in FHN too: with a strip per wavefront the apron would be 12 columns a
// side, 40 valid lanes of 64; around a block it is 232 of 256.
#if defined(CRD_COOP_ALL)  // (experiment switches: every fp64 one-column multi-step pipeline / none)
template <typename Real, int MODEL, int COLS, int STEPS = 2>
constexpr bool kCoop = sizeof(Real) == 8 && COLS == 1;
#elif defined(CRD_NO_COOP)
template <typename Real, int MODEL, int COLS, int STEPS = 2>
constexpr bool kCoop = false;
#else
template <typename Real, int MODEL, int COLS, int STEPS = 2>
constexpr bool kCoop = sizeof(Real) == 8 && COLS == 1 && (MODEL == CRD_MODEL_GOLDBETER || STEPS == 3);
#endif
// An edge slot holds the 4 STEPS quantities of an iteration (each step's input row and its three stage values): 64 B at two steps,
// 96 of 128 at three.
// (Round 6 tried slots placed on the LDS banks lanes 0 / 63 would have had in the dump area -- lane 63's slot shares its banks with lane
// 48's dump address, SQ_LDS_BANK_CONFLICT 13.5 % of the LDS cycles; the eastern slot is then 8- but not 16-byte aligned and the readers
// need ds_read2_b64, eight LDS cycles each instead of four: the three-step launch 3.8 % SLOWER (0.2040 -> 0.2118 ms per step).  A store's
// cost is the transfer of its operands to the LDS, six cycles whatever the banks do; profiles/r06/three_step_ab.txt.)
template <int STEPS>
constexpr int kEdgeSlotBytes = STEPS == 3 ? 128 : 64;
template <int STEPS>
constexpr int kEdgeParityBytes = 4 /* kMaxWavesPerBlock */ * 2 * kEdgeSlotBytes<STEPS>;  // [wavefront][side]
template <int STEPS>
constexpr int kEdgeBytes = 2 * kEdgeParityBytes<STEPS>;                                  // [parity]
template <int STEPS>
constexpr int kEdgeDumpBytes = 4 /* kMaxWavesPerBlock */ * 64 * 8 + kEdgeBytes<STEPS>;   // where the lanes in between drop their values (+ the largest slot offset)
__device__ __forceinline__ double from_lane_below_old(double x, double old)
{
	int lo = __double2loint(x), hi = __double2hiint(x);
	lo = __builtin_amdgcn_update_dpp(__double2loint(old), lo, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
	hi = __builtin_amdgcn_update_dpp(__double2hiint(old), hi, 0x138, 0xf, 0xf, false);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_lane_above_old(double x, double old)
{
	int lo = __double2loint(x), hi = __double2hiint(x);
	lo = __builtin_amdgcn_update_dpp(__double2loint(old), lo, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
	hi = __builtin_amdgcn_update_dpp(__double2hiint(old), hi, 0x130, 0xf, 0xf, false);
	return __hiloint2double(hi, lo);
}
// X: lane 63 holds the eastern neighbour wavefront's value of lane 0, lane 0 the western one's value of lane 63 (other lanes: anything)
template <int MODEL>
__device__ __forceinline__ void rhs_lane_edge(double uC, double X, double uS, double uN, double v, double cE, double cWn, double cP, double rowp, double ka4, bool zero,
                                              double &du, double &dv)
{
	const double t = uC - X;  // lane 0: its western first difference
	const double gE = from_lane_above_old(uC, X) - uC;
	rhs_point<double, MODEL>(uC, from_lane_below_old(gE, t), gE, uS, uN, v, cE, cWn, cP, rowp, ka4, zero, du, dv);
}
template <int OFF>
__device__ __forceinline__ void edge_publish(unsigned pub, double val)
{
#ifndef CRD_PROBE_NOPUBLISH  // (probe build: what the publishes cost -- results are wrong without them)
	asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(pub), "v"(val), "n"(OFF) : "memory");
#endif
}
typedef double double2r __attribute__((ext_vector_type(2)));
#ifdef CRD_PROBE_NOBARRIER  // (probe build: what the iteration's barrier costs -- results are wrong without it)
#define CRD_EDGE_BARRIER "s_nop 0"
#else
#define CRD_EDGE_BARRIER "s_barrier"
#endif
constexpr unsigned long kEdgeLanes = 0x8000000000000001ul;  // exec mask: lanes 0 and 63
// The barrier of an iteration (this wavefront's own publishes have landed: lgkmcnt(0)), and behind it the reads -- issued, not waited
// for: ring_read_with_edges does that -- of the 4 STEPS quantities the neighbours published during it: 16 N bytes from OFF on, into
// lanes 0 and 63.
template <int OFF>
__device__ __forceinline__ void edge_exchange(unsigned con, double2r (&e)[4])
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\t" CRD_EDGE_BARRIER "\n\ts_mov_b64 exec, %5\n\tds_read_b128 %0, %4 offset:%6\n\tds_read_b128 %1, %4 offset:%7\n\t"
	             "ds_read_b128 %2, %4 offset:%8\n\tds_read_b128 %3, %4 offset:%9\n\ts_mov_b64 exec, -1"
	             : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3]) : "v"(con), "s"(kEdgeLanes), "n"(OFF), "n"(OFF + 16), "n"(OFF + 32), "n"(OFF + 48) : "memory");
}
template <int OFF>
__device__ __forceinline__ void edge_exchange(unsigned con, double2r (&e)[6])
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\t" CRD_EDGE_BARRIER "\n\ts_mov_b64 exec, %7\n\tds_read_b128 %0, %6 offset:%8\n\tds_read_b128 %1, %6 offset:%9\n\t"
	             "ds_read_b128 %2, %6 offset:%10\n\tds_read_b128 %3, %6 offset:%11\n\tds_read_b128 %4, %6 offset:%12\n\tds_read_b128 %5, %6 offset:%13\n\t"
	             "s_mov_b64 exec, -1"
	             : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3]), "=&v"(e[4]), "=&v"(e[5])
	             : "v"(con), "s"(kEdgeLanes), "n"(OFF), "n"(OFF + 16), "n"(OFF + 32), "n"(OFF + 48), "n"(OFF + 64), "n"(OFF + 80)
	             : "memory");
}

// f(integral_constant<int, 0>{}), f(integral_constant<int, 1>{}), ... in order: a compile-time unrolled loop.
template <typename F, int... Is>
__device__ __forceinline__ void for_sequence(F &&f, std::integer_sequence<int, Is...>)
{
	(f(std::integral_constant<int, Is>{}), ...);
}

// row base (uniform) + this lane's byte offset
template <typename T>
__device__ __forceinline__ T *at_lane(T *row, unsigned byte_offset)
{
	using Bytes = std::conditional_t<std::is_const<T>::value, const char, char>;
	// (the empty asm keeps the compiler from hoisting `constant pointer + lane offset` out of the row loop as a 64-bit vector
	// base, which would cost a 64-bit vector add per access to put the row offset back in)
	asm volatile("" : "+v"(byte_offset));
	return reinterpret_cast<T *>(reinterpret_cast<Bytes *>(row) + byte_offset);
}

// ... the same address as a pointer to the lane's value (one Real, or two adjacent ones: an 8- or 16-byte access)
template <typename V, typename T>
__device__ __forceinline__ auto at_lane_as(T *row, unsigned byte_offset)
{
	using P = std::conditional_t<std::is_const<T>::value, const V, V>;
	return reinterpret_cast<P *>(at_lane(row, byte_offset));
}

// a lane's value(s) added up in double
__device__ __forceinline__ double lane_total(double x) { return x; }
__device__ __forceinline__ double lane_total(float x) { return (double)x; }
template <typename V2>
__device__ __forceinline__ double lane_total(V2 v)
{
	return (double)v.x + (double)v.y;
}

// A value known to be identical in every lane, moved to scalar registers.
__device__ __forceinline__ double uniform(double x)
{
	return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
__device__ __forceinline__ float uniform(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

// (non-temporal row loads measured 19 % slower, fp64 and fp32 alike: the apron rows and columns two items share come from cache)
#define CRD_ROW_LOAD(p) (*(p))
// The new state is written once and not read again by this launch.  NT = true gives its stores the non-temporal hint
// (`global_store ... nt`): the lines do not stay in L2 at the expense of the apron rows and columns neighbouring items share --
// a launch-plan choice (FusedPlan::nt), measured like the others: -6.5 % on 8192^2 fp64 under the plain mapping, -8 % at 4096^2,
// +1 % on some two-column plans (profiles/r03/nt_stores.txt).
template <bool NT, typename T>
__device__ __forceinline__ void row_store(T *p, T v)
{
	if constexpr (NT) __builtin_nontemporal_store(v, p);
	else *p = v;
}

template <typename Real>
struct FusedArgs {
	const Real *in_u, *in_v;  // y0, pointers to local row 0 (ghost rows at negative offsets)
	Real *out_u, *out_v;      // new state, local row 0
	Real h2, h3, h6, h1;      // dt/2, dt/3, dt/6, dt
	int absorb[5];            // t_stage < tBoundary for the four stages (+ the embedded pair's fifth)
	int absorb2[4];           // two steps per launch: the second step's stages
	int absorb3[4];           // three: the third's
	int js, ny;               // global index of local row 0, global row count (absorbing rule is by global row)
	// Rows this launch produces: one or two ranges, cut into work items ("chunks") of `chunk` rows; chunk ids run through the
	// ranges in order (range 1 starts at id `first2`).  One range: an ordinary sweep.  Two: the rows of a step that read ghost
	// rows, below and above the slab, or the two edge bands of a cycle's last step.
	int r_begin[2], r_end[2], chunk, first2;
	int nstrips, nitems, nblocks, remap;
	int xs_lanes;             // remap 2: phi-lanes of chunk sequences per strip block and XCD
	int nchunks;              // chunks of both ranges
	int sw;                   // wavefronts per block = adjacent strips a block covers
	double *err_partials;     // EMBED: one weighted square sum per work item
	Real rtol, atol;          // EMBED: error weights 1 / (rtol |y_n| + atol)
	int err_lo, err_hi;       // EMBED: rows whose error counts (a multi-slab attempt also produces ghost-region rows: the owner counts those)
};

// EMBED adds a fifth pipeline stage and with it a local error estimate whose weighted square sum
//   sum_i (err_i / (rtol |y_n,i| + atol))^2
// over this work item's outputs is written to err_partials[item] (ARKode's WRMS norm, src/FHNmodel_torus.cpp:365, is
// sqrt(sum / N)).  The propagated solution is classical RK4 either way.  The pipeline is then five rows / columns deep
// (apron 5, 54 valid lanes) and uses 6 register slots per array, the loop being unrolled 6 times.
//   EMBED = 1  the RK4(3) pair of rounds 1-2: k5 = f(t + dt, y_new), yhat = y + dt (k1/6 + k2/3 + k3/3 + k5/6), err = dt (k4 - k5)/6.
//   EMBED = 2  ARKode's default fourth-order explicit table, Zonneveld 5(3)4 (what the reference integrates with,
//              src/FHNmodel_torus.cpp:356-372; table and order conditions in oracle/arkode_erk.py): the fifth stage is
//              k5 = f(t + 3/4 dt, y + dt (5/32 k1 + 7/32 k2 + 13/32 k3 - 1/32 k4)) and
//              err = y_new - yhat = dt (2/3 k1 - 2 k2 - 2 k3 - 2 k4 + 16/3 k5).
//              No accumulators in this variant: when stage 4 of a row runs, the row's stage values y1 = y + dt/2 k1,
//              y2 = y + dt/2 k2, y3 = y + dt k3 are still in their register slots (six rows deep), so with
//              d_i = y_i - y the three combinations the row needs come out of d1, d2, d3 and dt k4 there and then,
//                y_new = y + (d1 + 2 d2 + d3)/3 + dt k4/6,   z5 = y + 5/16 d1 + 7/16 d2 + 13/32 d3 - dt k4/32,
//                e4 = 4/3 d1 - 4 d2 - 2 d3 - 2 dt k4        (err = e4 + 16/3 dt k5 one iteration later);
//              the subtractions are exact (y_i is within a factor 2 of y wherever it matters), so what this costs against running
//              sums is a rounding of y_i itself, 1e-16 |y| in quantities that are compared with rtol |y| + atol.
// COLS = 2: two adjacent grid columns per lane -- a wavefront's strip is 128 columns, 120 valid (the apron is two whole lanes a
// side), rows are read and written with one 8-byte (fp32) or 16-byte (fp64) access per lane, a stage needs ONE DPP move per
// direction for two columns, and the fp32 arithmetic is the packed instructions (v_pk_fma_f32 ...).  Needs an even nx (the
// pair must not straddle the periodic seam); results are the one-column kernel's bit for bit.
// NT = true: the new state is stored with the non-temporal hint (row_store above).
// One work item: strip `strip` (a wavefront's columns) of chunk `chunk` (its rows).  ABSORB as a template argument of the ITEM: a
// launch some stage of which has t < tBoundary runs the selects only in the items whose rows (aprons included) contain a global
// phi boundary row -- two chunks per boundary of a launch; every other item runs the body without them (the kernel below decides
// per chunk, uniformly for the block).  With the selects in every item the absorbing-rows run cost 5.4 % at 8192^2 (round 3).
template <typename Real, int MODEL, bool ABSORB, int EMBED, int COLS, bool NT>
__device__ __forceinline__ void fused_item(const Slab<Real> &s, const FusedArgs<Real> &a, const int strip, const int chunk)
{
	static_assert(COLS == 1 || (COLS == 2 && EMBED == 0), "the embedded pairs run one column per lane");
	using V = typename LaneValue<Real, COLS>::type;
	constexpr bool ZONN = EMBED == 2;
	constexpr int APRON = EMBED != 0 ? kApron + 1 : kApron;
	static_assert(APRON % COLS == 0, "the apron is whole lanes");
	constexpr int VALID = COLS * kLanes - 2 * APRON;
	// Register slots per pipeline array = unroll factor: the rows alive at once (4, or 6 with the fifth stage) -- even, because
	// the two-deep arrays below are addressed with the slot's parity.
	constexpr int M = EMBED != 0 ? CRD_EMBED_SLOTS : 4;
	static_assert(M % 2 == 0 && M >= (EMBED != 0 ? 6 : 4) && (EMBED != 2 || M % 3 == 0), "slot count");
	constexpr int kPrefetch = EMBED != 0 ? ((MODEL == CRD_MODEL_GOLDBETER && sizeof(Real) == 8) ? 2 : CRD_PREFETCH_EMBED) : (MODEL == CRD_MODEL_GOLDBETER) ? CRD_PREFETCH_GB : CRD_PREFETCH_FHN;
	static_assert(M % kPrefetch == 0, "prefetch slots are addressed with the unrolled iteration index");
	const int lane = threadIdx.x & (kLanes - 1);
	const int item = chunk * a.nstrips + strip;
	const int nx = s.nx;

	int x = strip * VALID - APRON + COLS * lane;  // this lane's (first) column, wrapped periodically (nx may be smaller than a strip)
	x %= nx;
	if (x < 0) x += nx;
	// lane offsets in bytes, unsigned 32-bit: with a scalar row base the accesses take the `global_load v, v_off, s[base]` form and
	// no 64-bit vector add is spent per access
	const unsigned xb = (unsigned)x * (unsigned)sizeof(Real);  // (a lane that stores has x == out_col: its column is inside [0, nx) unwrapped, so xb serves the stores too)
	const int out_col = strip * VALID + (COLS * lane - APRON);
	const bool lane_stores = COLS * lane >= APRON && COLS * lane < COLS * kLanes - APRON && out_col < nx;  // (two columns: nx is even, so is out_col)

	const int range = chunk >= a.first2 ? 1 : 0;  // (an unused second range starts at nchunks)
	const int range_end = a.r_end[range];
	const int j0 = a.r_begin[range] + (chunk - (range ? a.first2 : 0)) * a.chunk;
	const int j1 = (j0 + a.chunk < range_end) ? j0 + a.chunk : range_end;
	const int jbase = j0 - APRON;
	const int niter = (j1 - j0) + 2 * APRON;
	const int jlast = j1 + APRON - 1;  // last row the pipeline consumes

	const V cE = *reinterpret_cast<const V *>(s.cE + x), cWn = *reinterpret_cast<const V *>(s.cWn + x), cP = *reinterpret_cast<const V *>(s.cP + x);
	const Real ka4 = s.ka4;
	const V h1 = (V)a.h1, h2 = (V)a.h2, h3 = (V)a.h3, h6 = (V)a.h6;
	// b(j) is read-only for the whole launch and its index is uniform: through the constant address space the reads become
	// scalar-cache loads into SGPRs (s_load_dwordx2), no vector registers and no vector-memory instruction
	const __attribute__((address_space(4))) Real *const brow = (const __attribute__((address_space(4))) Real *)(s.brow);

	// Row base pointers are scalar; a single slab wraps rows outside [0, nyl).  Branch-free: this runs once per pipeline
	// iteration in every wavefront's instruction stream (scalar work is not free: ~40 of the ~140 instructions of an iteration
	// were scalar before the row bookkeeping was pared down).
	const int wrap_nyl = s.wrap ? s.nyl : 0;
	auto row_base = [&](int j) -> ptrdiff_t {
		j += wrap_nyl & (j >> 31);
		j -= (j >= s.nyl) ? wrap_nyl : 0;
		return (ptrdiff_t)j * nx;
	};
	// Global phi boundary rows (src/FHNmodel_torus.cpp:643-653), also when they are recomputed as another slab's ghost rows.
	auto boundary_row = [&](int j) -> bool {
		int gj = a.js + j;
		if (gj < 0) gj += a.ny;
		else if (gj >= a.ny) gj -= a.ny;
		return gj == 0 || gj == a.ny - 1;
	};

	// Pipeline registers.  Row jbase+m of an array lives in slot m mod M (m & 1 for the two-deep v arrays), so with the
	// loop unrolled M times every access has a compile-time slot and no value is ever moved between registers.
	// (Zonneveld variant: the local field's stage values stay until the row's stage 4 has used them -- y1 four rows, y2 three)
	constexpr int NV1 = ZONN ? M : 2, NV2 = ZONN ? 3 : 2;
	V u0[M], v0[M], U1[M], U2[M], U3[M], V1[NV1], V2[NV2], V3[2], aU[M], aV[M];
	// (Keeping aU / aV in LDS instead -- 90 VGPRs, five wavefronts per SIMD -- measured 3-5 % SLOWER on every grid: the twelve LDS
	// accesses per iteration cost more than the fifth wavefront brings.)
#define ACC_U(S) aU[S]
#define ACC_V(S) aV[S]
	V U4[M], V4[2], K4U[2], K4V[2];  // EMBED 1: y_new window and k4 of the last two rows; EMBED 2: z5 window (three rows deep) and e4
	const V zero_v = splat<V>(0.0);
	V err2 = zero_v;
#pragma unroll
	for (int k = 0; k < M; k++) u0[k] = v0[k] = U1[k] = U2[k] = U3[k] = aU[k] = aV[k] = U4[k] = zero_v;
#pragma unroll
	for (int k = 0; k < NV1; k++) V1[k] = zero_v;
#pragma unroll
	for (int k = 0; k < NV2; k++) V2[k] = zero_v;
	V3[0] = V3[1] = V4[0] = V4[1] = K4U[0] = K4U[1] = K4V[0] = K4V[1] = zero_v;

	// Rows are fetched kPrefetch iterations before they enter the pipeline: with ~16 wavefronts per CU one row in flight
	// per wavefront is far too little to cover HBM latency (Little's law), four rows (8 loads, 4 KiB per wavefront) is enough.
	// The per-row reaction parameter b(j) rides along: a plain `s.brow[c]` at the point of use is a VECTOR load whose
	// full latency the stage then waits for (four exposed L2 round trips per iteration, 60 % of the wave's lifetime when
	// measured); fetched with the row and moved to scalar registers on arrival it costs nothing.
	V pu[kPrefetch], pv[kPrefetch];
	Real pb[kPrefetch], bq[M];
#pragma unroll
	for (int k = 0; k < kPrefetch; k++) {
		const int jr = (jbase + k < jlast) ? jbase + k : jlast;
		const ptrdiff_t rb = row_base(jr);
		pu[k] = CRD_ROW_LOAD(at_lane_as<V>(a.in_u + rb, xb));
		pv[k] = CRD_ROW_LOAD(at_lane_as<V>(a.in_v + rb, xb));
		pb[k] = brow[jr];
	}
#pragma unroll
	for (int k = 0; k < M; k++) bq[k] = (Real)0;
	// running scalars of the row loop: the row the next prefetch takes (the tail re-reads the last valid row instead of running
	// past the plane) and the output rows of stage 4 (row jbase + m - 4 at iteration m)
	int jn = (jbase + kPrefetch < jlast) ? jbase + kPrefetch : jlast;
	Real *out_row_u = a.out_u + (ptrdiff_t)(jbase - 4) * nx, *out_row_v = a.out_v + (ptrdiff_t)(jbase - 4) * nx;

	// One pipeline iteration at m == K (mod M).  GUARDED: the first iterations of a chunk, where stage k's inputs exist only
	// from iteration 2k on.
	auto iteration = [&](int m, auto kk, auto guarded) {
		constexpr int K = decltype(kk)::value;
		constexpr bool GUARDED = decltype(guarded)::value;
#ifndef CRD_NO_LOCKSTEP  // (an experiment switch: -DCRD_NO_LOCKSTEP lets a block's wavefronts drift)
		__builtin_amdgcn_s_barrier();  // lockstep; uniform over the block: its wavefronts share the chunk, hence niter
#endif
		// slots of rows p, p-1, ... p-6 (with M = 4, row p-4 shares its slot with row p)
		constexpr int S0 = K % M, S1 = (K + M - 1) % M, S2 = (K + M - 2) % M, S3 = (K + M - 3) % M, S4 = (K + 2 * M - 4) % M;
		constexpr int S5 = (K + 2 * M - 5) % M, S6 = (K + 2 * M - 6) % M;
		// slots of the local field's stage values (two deep; in the Zonneveld variant as deep as they have to live) and of the z5 window
		constexpr int A1 = ZONN ? S1 : (S1 & 1), A2 = ZONN ? S2 : (S2 & 1), A4 = ZONN ? S4 : (S4 & 1);              // y1 rows p-1, p-2, p-4
		constexpr int B2 = ZONN ? S2 % 3 : (S2 & 1), B3 = ZONN ? S3 % 3 : (S3 & 1), B4 = ZONN ? S4 % 3 : (S4 & 1);  // y2 rows p-2, p-3, p-4
		constexpr int Z4 = ZONN ? S4 % 3 : S4, Z5 = ZONN ? S5 % 3 : S5, Z6 = ZONN ? S6 % 3 : S6;                    // z5 / y_new rows p-4, p-5, p-6
		constexpr int P = K % kPrefetch;
		const int p = jbase + m;
		const Real b4 = bq[S4];  // b of row p-4 (stage 4), read before row p takes over the slot when M = 4
		u0[S0] = pu[P];
		v0[S0] = pv[P];
		bq[S0] = uniform(pb[P]);
		{
			const ptrdiff_t rb = row_base(jn);
			pu[P] = CRD_ROW_LOAD(at_lane_as<V>(a.in_u + rb, xb));
			pv[P] = CRD_ROW_LOAD(at_lane_as<V>(a.in_v + rb, xb));
			pb[P] = brow[jn];
			jn = (jn < jlast) ? jn + 1 : jlast;
		}
		V du, dv;
		// ---- stage 1, centre row p-1: y0 rows p-2, p-1, p -----------------------------------------------------
		if (!GUARDED || m >= 2) {
			const int c = p - 1;
			rhs_lane<V, MODEL>(u0[S1], u0[S2], u0[S0], v0[S1], cE, cWn, cP, bq[S1], ka4,
			                       ABSORB && a.absorb[0] && boundary_row(c), du, dv);
			U1[S1] = fmadd(h2, du, u0[S1]);
			V1[A1] = fmadd(h2, dv, v0[S1]);
			if (!ZONN) {
				ACC_U(S1) = fmadd(h6, du, u0[S1]);
				ACC_V(S1) = fmadd(h6, dv, v0[S1]);
			}
		}
		// ---- stage 2, centre row p-2: y1 rows p-3, p-2, p-1 ---------------------------------------------------
		if (!GUARDED || m >= 4) {
			const int c = p - 2;
			rhs_lane<V, MODEL>(U1[S2], U1[S3], U1[S1], V1[A2], cE, cWn, cP, bq[S2], ka4,
			                       ABSORB && a.absorb[1] && boundary_row(c), du, dv);
			U2[S2] = fmadd(h2, du, u0[S2]);
			V2[B2] = fmadd(h2, dv, v0[S2]);
			if (!ZONN) {
				ACC_U(S2) = fmadd(h3, du, ACC_U(S2));
				ACC_V(S2) = fmadd(h3, dv, ACC_V(S2));
			}
		}
		// ---- stage 3, centre row p-3: y2 rows p-4, p-3, p-2 ---------------------------------------------------
		if (!GUARDED || m >= 6) {
			const int c = p - 3;
			rhs_lane<V, MODEL>(U2[S3], U2[S4], U2[S2], V2[B3], cE, cWn, cP, bq[S3], ka4,
			                       ABSORB && a.absorb[2] && boundary_row(c), du, dv);
			U3[S3] = fmadd(h1, du, u0[S3]);
			V3[S3 & 1] = fmadd(h1, dv, v0[S3]);
			if (!ZONN) {
				ACC_U(S3) = fmadd(h3, du, ACC_U(S3));
				ACC_V(S3) = fmadd(h3, dv, ACC_V(S3));
			}
		}
		// ---- stage 4, centre row p-4: y3 rows p-5, p-4, p-3 -> the new state ----------------------------------
		if (!GUARDED || m >= 8) {
			const int c = p - 4;
			rhs_lane<V, MODEL>(U3[S4], U3[S5], U3[S3], V3[S4 & 1], cE, cWn, cP, b4, ka4,
			                       ABSORB && a.absorb[3] && boundary_row(c), du, dv);
			V nu, nv;
			if (ZONN) {
				// everything the row still needs, from its stage values (see the kernel's header comment)
				const V yu = u0[S4], yv = v0[S4];
				const V d1u = U1[S4] - yu, d2u = U2[S4] - yu, d3u = U3[S4] - yu, d1v = V1[A4] - yv, d2v = V2[B4] - yv, d3v = V3[S4 & 1] - yv;
				nu = fmadd(splat<V>(1.0 / 3.0), fmadd(splat<V>(2.0), d2u, d1u + d3u), fmadd(h6, du, yu));
				nv = fmadd(splat<V>(1.0 / 3.0), fmadd(splat<V>(2.0), d2v, d1v + d3v), fmadd(h6, dv, yv));
				const V h32 = splat<V>(-1.0 / 32.0) * h1, h2m = splat<V>(-2.0) * h1;
				U4[Z4] = fmadd(splat<V>(5.0 / 16.0), d1u, fmadd(splat<V>(7.0 / 16.0), d2u, fmadd(splat<V>(13.0 / 32.0), d3u, fmadd(h32, du, yu))));
				V4[S4 & 1] = fmadd(splat<V>(5.0 / 16.0), d1v, fmadd(splat<V>(7.0 / 16.0), d2v, fmadd(splat<V>(13.0 / 32.0), d3v, fmadd(h32, dv, yv))));
				K4U[S4 & 1] = fmadd(splat<V>(4.0 / 3.0), d1u, fmadd(splat<V>(-4.0), d2u, fmadd(splat<V>(-2.0), d3u, h2m * du)));
				K4V[S4 & 1] = fmadd(splat<V>(4.0 / 3.0), d1v, fmadd(splat<V>(-4.0), d2v, fmadd(splat<V>(-2.0), d3v, h2m * dv)));
			} else {
				nu = fmadd(h6, du, ACC_U(S4));
				nv = fmadd(h6, dv, ACC_V(S4));
			}
			// Without the fifth stage rows j0 <= c < j1 are exactly iterations 8 .. niter-1; with it (one more apron row each side)
			// the first and the last iteration of the range fall outside.
			if ((EMBED == 0 || (c >= j0 && c < j1)) && lane_stores) {
				row_store<NT>(at_lane_as<V>(out_row_u, xb), nu);
				row_store<NT>(at_lane_as<V>(out_row_v, xb), nv);
			}
			if (EMBED == 1) {
				U4[S4] = nu;
				V4[S4 & 1] = nv;
				K4U[S4 & 1] = du;
				K4V[S4 & 1] = dv;
			}
		}
		// ---- stage 5 (EMBED), centre row p-5: y_new rows p-6, p-5, p-4 -> k5 and the error of row p-5 ----------
		if (EMBED != 0 && (!GUARDED || m >= 10)) {
			const int c = p - 5;
			rhs_lane<V, MODEL>(U4[Z5], U4[Z6], U4[Z4], V4[S5 & 1], cE, cWn, cP, bq[S5], ka4,
			                       ABSORB && a.absorb[4] && boundary_row(c), du, dv);  // k5 at t + dt like k4 (EMBED 1) or at t + 3/4 dt (Zonneveld)
			if (lane_stores && c >= a.err_lo && c < a.err_hi) {  // rows j0 .. j1-1 exactly (stage 5 starts at iteration 10, row j0, and the loop ends at row j1-1), owned rows only
				const V au = __builtin_elementwise_abs(u0[S5]), av = __builtin_elementwise_abs(v0[S5]);
				const V wu = fmadd((V)a.rtol, au, (V)a.atol), wv = fmadd((V)a.rtol, av, (V)a.atol);
				V eu, ev;
				if (ZONN) {
					const V h163 = splat<V>(16.0 / 3.0) * h1;
					eu = fmadd(h163, du, K4U[S5 & 1]) / wu;
					ev = fmadd(h163, dv, K4V[S5 & 1]) / wv;
				} else {
					eu = h6 * (K4U[S5 & 1] - du) / wu;
					ev = h6 * (K4V[S5 & 1] - dv) / wv;
				}
				err2 = fmadd(eu, eu, fmadd(ev, ev, err2));
			}
		}
		out_row_u += nx;
		out_row_v += nx;
	};
	using std::integral_constant;
	// guarded prologue: up to the first multiple of M at or beyond 2 * APRON, so the steady-state loop starts at slot 0
	constexpr int PRO = ((2 * APRON + M - 1) / M) * M;
	for_sequence(
	    [&](auto k) {
		    constexpr int I = decltype(k)::value;
		    if (I < niter) iteration(I, integral_constant<int, I % M>{}, std::true_type{});
	    },
	    std::make_integer_sequence<int, PRO>{});
	int m = PRO;
	for (; m + M - 1 < niter; m += M)  // steady state: M iterations per trip, every register slot a compile-time constant
		for_sequence([&](auto k) { iteration(m + decltype(k)::value, k, std::false_type{}); }, std::make_integer_sequence<int, M>{});
	for_sequence(
	    [&](auto k) {
		    if (m + decltype(k)::value < niter) iteration(m + decltype(k)::value, k, std::false_type{});
	    },
	    std::make_integer_sequence<int, M - 1>{});
	if constexpr (EMBED != 0) {
		// wavefront sum in a fixed order (butterfly over lane distances 32 .. 1), one partial per work item: the host-side
		// reduction adds them in item order, so the norm is reproducible run to run
		double sum = lane_total(err2);
		for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
		if (lane == 0) a.err_partials[item] = sum;
	}
}

// y + h k of the stage updates as the three-address v_fma_f64, spelled out: left to itself the compiler forms a third of them as
// v_mov_b64 + v_fmac_f64 (the addend is still needed, and its two-address form wants it in the destination) -- ten moves per iteration
// of the two-step pipeline, 7 % of its vector instructions.  Same operation, same rounding.
#ifndef CRD_NO_ASM_FMA
__device__ __forceinline__ double stage_fma(double h, double k, double y)
{
	double d;
	asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "s"(h), "v"(k), "v"(y));
	return d;
}
#else
__device__ __forceinline__ double stage_fma(double h, double k, double y) { return fmadd(h, k, y); }
#endif
template <typename V>
__device__ __forceinline__ V stage_fma(V h, V k, V y)
{
	return fmadd(h, k, y);
}

// ---- the two-step pipeline's memory path (round 5) ----------------------------------------------------------------------------
// Rows enter through LDS, fetched by LDS-DMA (buffer_load_dword ... lds) kRingRows iterations ahead of their use: the data of a
// load in flight needs no vector register, so the depth of the prefetch is a matter of LDS (a wavefront's ring: kRingRows x
// 1 KiB in fp64), not of the 168-register line the pipeline sits at.  (Round 4 held two rows per field in registers; under the
// counter's in-order rule -- vmcnt counts loads and stores together -- the wait in front of a row's first use also waited for
// the previous iteration's stores and loads: a wavefront took 2400 cycles per iteration of which it issued 590, and three
// wavefronts per SIMD could not cover that.)  A buffer resource per row (scalar base) + a lane offset: no 64-bit vector
// address arithmetic, and the stores take the same form.
#ifndef CRD_RING_ROWS
#define CRD_RING_ROWS 4
#endif
constexpr int kRingRowsWanted = CRD_RING_ROWS;  // a multiple of the unroll factor 4; rows in flight per wavefront
// ... as far as the six bits of vmcnt allow (a row of 1 KiB per field takes eight LDS-DMA instructions)
#ifndef CRD_RING_ROWS_THREE
#define CRD_RING_ROWS_THREE 8
#endif
// (three steps per launch: two wavefronts per SIMD -- eight rows in flight per wavefront for the bytes in flight twelve wavefronts of four had)
template <typename V, int STEPS = 2>
constexpr int kRingRowsOf = (kLanes * (int)sizeof(V) / 256 >= 4 && (STEPS == 3 ? CRD_RING_ROWS_THREE : kRingRowsWanted) > 4) ? 4 : (STEPS == 3 ? CRD_RING_ROWS_THREE : kRingRowsWanted);
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
constexpr unsigned kRowResourceBytes = 0x7fffffffu;  // num_records of a row's buffer resource
constexpr unsigned kStoreNowhere = 0x80000000u;      // a byte offset out of its range: the hardware drops the write
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_resource(const void *row)
{
	// raw buffer (stride 0) over the bytes from `row` on: offsets are a lane's byte offset within its row (< 2 GiB)
	return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(row), 0, (int)kRowResourceBytes, 0x00020000);
}
template <bool NT, typename V>
__device__ __forceinline__ void buffer_row_store(__amdgpu_buffer_rsrc_t r, unsigned byte_offset, V v)
{
	constexpr int aux = NT ? 2 : 0;  // (2 = nt)
	if constexpr (sizeof(V) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byte_offset, 0, aux);
	else if constexpr (sizeof(V) == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, byte_offset, 0, aux);
	else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_offset, 0, aux);
}
// Both fields of a ring slot into registers: waits until at most VMCNT vector-memory operations of this wavefront are outstanding
// (the slot's LDS-DMA has landed then: the counter retires in issue order), reads, waits for the reads.  One asm statement, so no
// instruction of the compiler's can come between a read and its wait.
template <int VMCNT, int OFF_U, int OFF_V, typename V>
__device__ __forceinline__ void ring_read(unsigned lds_lane, V &u, V &v)
{
	if constexpr (sizeof(V) == 4)
		asm volatile("s_waitcnt vmcnt(%3)\n\tds_read_b32 %0, %2 offset:%4\n\tds_read_b32 %1, %2 offset:%5\n\ts_waitcnt lgkmcnt(0)"
		             : "=&v"(u), "=&v"(v) : "v"(lds_lane), "n"(VMCNT), "n"(OFF_U), "n"(OFF_V) : "memory");
	else if constexpr (sizeof(V) == 8)
		asm volatile("s_waitcnt vmcnt(%3)\n\tds_read_b64 %0, %2 offset:%4\n\tds_read_b64 %1, %2 offset:%5\n\ts_waitcnt lgkmcnt(0)"
		             : "=&v"(u), "=&v"(v) : "v"(lds_lane), "n"(VMCNT), "n"(OFF_U), "n"(OFF_V) : "memory");
	else
		asm volatile("s_waitcnt vmcnt(%3)\n\tds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %2 offset:%5\n\ts_waitcnt lgkmcnt(0)"
		             : "=&v"(u), "=&v"(v) : "v"(lds_lane), "n"(VMCNT), "n"(OFF_U), "n"(OFF_V) : "memory");
}
// ... and the wait covers the edge reads in flight too (edge_exchange): they pass through as operands, so that nothing uses them before it
template <int VMCNT, int OFF_U, int OFF_V>
__device__ __forceinline__ void ring_read_with_edges(unsigned lds_lane, double &u, double &v, double2r (&e)[4])
{
	asm volatile("s_waitcnt vmcnt(%7)\n\tds_read_b64 %0, %6 offset:%8\n\tds_read_b64 %1, %6 offset:%9\n\ts_waitcnt lgkmcnt(0)"
	             : "=&v"(u), "=&v"(v), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]) : "v"(lds_lane), "n"(VMCNT), "n"(OFF_U), "n"(OFF_V) : "memory");
}
template <int VMCNT, int OFF_U, int OFF_V>
__device__ __forceinline__ void ring_read_with_edges(unsigned lds_lane, double &u, double &v, double2r (&e)[6])
{
	asm volatile("s_waitcnt vmcnt(%9)\n\tds_read_b64 %0, %8 offset:%10\n\tds_read_b64 %1, %8 offset:%11\n\ts_waitcnt lgkmcnt(0)"
	             : "=&v"(u), "=&v"(v), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]) : "v"(lds_lane), "n"(VMCNT), "n"(OFF_U), "n"(OFF_V) : "memory");
}
template <int VMCNT, int OFF_U, int OFF_V, typename V, int N>
__device__ __forceinline__ void ring_read_with_edges(unsigned, V &, V &, double2r (&)[N]) {}  // (COOP is fp64, one column per lane)
// LDS bytes of a block's rings
template <typename Real, int COLS, int STEPS = 2>
constexpr int kRingBytes = kMaxWavesPerBlock * kRingRowsOf<typename LaneValue<Real, COLS>::type, STEPS> * 2 * kLanes * COLS * (int)sizeof(Real);

// TWO classical RK4 steps of the item's rows in one pass over memory (STEPS = 2): the pipeline of fused_item twice over, eight stages
// deep -- iteration m takes row p (out of its LDS ring slot), runs stages 1..4 of step n on rows p-1 .. p-4, hands the new row p-4
// to a second, identical pipeline as ITS input row, which runs stages 1..4 of step n+1 on rows p-5 .. p-8 and stores row p-8.
// The state crosses memory once per TWO steps: 8 B (fp32) / 16 B (fp64) per grid-point-step instead of 16 / 32.  What it costs:
// the apron is 8 columns and 8 rows a side (48 valid columns of 64; 112 of 128 with two columns per lane), 16 filling iterations per
// chunk -- which run only the stages that have inputs, straight-line code in front of the loop -- and twice the pipeline registers
// (three wavefronts per SIMD instead of four).  What bounds the launch is its memory traffic, with vector issue close behind
// (profiles/r05/two_step_memory_path_ab.txt).
// (Goldbeter in fp64 runs the BLOCK as the strip -- kCoop above: one apron around four wavefronts' 256 lanes, 240 valid columns.)
// Slot arithmetic: row r of either pipeline lives in slot r mod 4; the second pipeline's rows are the first one's shifted by 4,
// i.e. the SAME slots -- the stage code is one lambda applied to two sets of arrays.  Per point the arithmetic is the sequence of
// two single steps exactly (same fused multiply-adds, same constants), so the result is theirs bit for bit.
template <typename Real, int MODEL, bool ABSORB, int COLS, bool NT, int STEPS = 2>
__device__ __forceinline__ void fused_item_multi_step(const Slab<Real> &s, const FusedArgs<Real> &a, const int strip, const int chunk, lds_char *const block_rings,
                                                      const int sblk = 0, lds_char *const block_edges = nullptr)
{
	static_assert(STEPS == 2 || STEPS == 3, "two or three steps per launch");
	constexpr bool COOP = kCoop<Real, MODEL, COLS, STEPS>;
#ifdef CRD_NO_SETPRIO
	constexpr bool PRIO = false;
#else
#ifdef CRD_PRIO_THREE  // (experiment switch: the three-step pipeline's memory operations at raised priority too)
	constexpr bool PRIO = sizeof(Real) == 8 && COLS == 1 && MODEL == CRD_MODEL_FHN;
#else
	constexpr bool PRIO = sizeof(Real) == 8 && COLS == 1 && MODEL == CRD_MODEL_FHN && STEPS == 2;
#endif
#endif
	using V = typename LaneValue<Real, COLS>::type;
	constexpr int APRON = STEPS * kApron;
	constexpr int FILL = 2 * APRON;  // iterations of a chunk before rows come out: stage k of step n has inputs from iteration 8 n + 2 k on
	static_assert(APRON % COLS == 0, "the apron is whole lanes");
	constexpr int VALID = COLS * kLanes - 2 * APRON;
	constexpr int M = 4;
	// the ring: kRingRows slots of [u row segment | v row segment], RB bytes each, filled 256 B (64 lanes x one dword) per LDS-DMA
	constexpr int RB = kLanes * (int)sizeof(V), SLOT = 2 * RB, G1 = RB / 256, kRingRows = kRingRowsOf<V, STEPS>;
	static_assert(kRingRows % M == 0 && kRingRows >= M && (kRingRows & (kRingRows - 1)) == 0, "ring slots are addressed with the unrolled iteration index");
	// vector-memory operations a wavefront issues per iteration: 2 G1 LDS-DMA loads, and 2 stores once rows come out.  When the slot
	// of iteration m is read, the operations issued after its fill (at iteration m - kRingRows) are the fills of kRingRows - 1
	// iterations and the stores of kRingRows iterations -- none of the latter while the pipeline still fills.
	constexpr int kWaitFill = (kRingRows - 1) * 2 * G1, kWaitSteady = kWaitFill + 2 * kRingRows;
	static_assert(kWaitSteady <= 63, "vmcnt is six bits");
	const int lane = threadIdx.x & (kLanes - 1);
	const int nx = s.nx;
	// column of lane 0 (before wrapping), and this lane's place in what the apron surrounds: the wavefront's 64 lanes -- or the block's
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int x0 = COOP ? sblk * (a.sw * kLanes - 2 * APRON) - APRON + wave * kLanes : strip * VALID - APRON;
	const int place = COOP ? wave * kLanes + lane : COLS * lane, span = COOP ? a.sw * kLanes : COLS * kLanes;
	int x = x0 + COLS * lane;
	x %= nx;
	if (x < 0) x += nx;
	const unsigned xb = (unsigned)x * (unsigned)sizeof(Real);  // (a lane that stores has x == out_col: its column is inside [0, nx) unwrapped, so xb serves the stores too)
	const int out_col = x0 + COLS * lane;
	const bool lane_stores = place >= APRON && place < span - APRON && out_col < nx;
	// ... and where its stores go: its column's bytes, or beyond the range of the row's buffer resource (row_resource: 0x7fffffff bytes),
	// where a write is dropped -- the store instructions themselves are issued by every lane (see the stores below)
	const unsigned xb_store = lane_stores ? xb : kStoreNowhere;

	const int range = chunk >= a.first2 ? 1 : 0;
	const int range_end = a.r_end[range];
	const int j0 = a.r_begin[range] + (chunk - (range ? a.first2 : 0)) * a.chunk;
	const int j1 = (j0 + a.chunk < range_end) ? j0 + a.chunk : range_end;
	const int jbase = j0 - APRON;
	const int niter = (j1 - j0) + 2 * APRON;
	const int jlast = j1 + APRON - 1;

	const V cE = *reinterpret_cast<const V *>(s.cE + x), cWn = *reinterpret_cast<const V *>(s.cWn + x), cP = *reinterpret_cast<const V *>(s.cP + x);
	const Real ka4 = s.ka4;
	const V h1 = (V)a.h1, h2 = (V)a.h2, h3 = (V)a.h3, h6 = (V)a.h6;
	const __attribute__((address_space(4))) Real *const brow = (const __attribute__((address_space(4))) Real *)(s.brow);
	const int wrap_nyl = s.wrap ? s.nyl : 0;
	auto row_base = [&](int j) -> ptrdiff_t {
		j += wrap_nyl & (j >> 31);
		j -= (j >= s.nyl) ? wrap_nyl : 0;
		return (ptrdiff_t)j * nx;
	};
	// Global phi boundary rows (src/FHNmodel_torus.cpp:643-653).  Rows ny - 1 and 0 are neighbours on the periodic grid: within the rows
	// this item's pipeline touches (fewer than 2 ny of them) they are local rows jb, jb + 1 and possibly jb + ny, jb + ny + 1.  Two
	// scalars and a bit mask of the 4 STEPS stage flags instead of js, ny and as many flag words: the body with the selects must not need
	// more registers than the one without (168 VGPRs = three wavefronts per SIMD), or the launch's kernel, which holds both, runs
	// every item at two.
	int jb = 0, amask = 0;
	if (ABSORB) {
		int g0 = (a.js + jbase) % a.ny;  // global row of the first row the pipeline takes
		if (g0 < 0) g0 += a.ny;
		jb = jbase + (a.ny - 1 - g0);
#pragma unroll
		for (int k = 0; k < 4; k++) amask |= (a.absorb[k] ? 1 << k : 0) | (a.absorb2[k] ? 16 << k : 0) | ((STEPS == 3 && a.absorb3[k]) ? 256 << k : 0);
		jb = __builtin_amdgcn_readfirstlane(jb);
		amask = __builtin_amdgcn_readfirstlane(amask);
	}
	const int ny_rows = a.ny;
	auto boundary_row = [&](int j) -> bool { return (unsigned)(j - jb) <= 1u || (unsigned)(j - jb - ny_rows) <= 1u; };

	struct Pipe {
		V u0[M], v0[M], U1[M], U2[M], U3[M], V1[2], V2[2], V3[2], aU[M], aV[M];
		Real bq[M];
	};
	Pipe P[STEPS];  // P[n]: the pipeline of step n of the launch; its input rows are P[n - 1]'s output rows
	const V zero_v = splat<V>(0.0);
#pragma unroll
	for (int n = 0; n < STEPS; n++) {
#pragma unroll
		for (int k = 0; k < M; k++) {
			P[n].u0[k] = P[n].v0[k] = P[n].U1[k] = P[n].U2[k] = P[n].U3[k] = P[n].aU[k] = P[n].aV[k] = zero_v;
			P[n].bq[k] = (Real)0;
		}
		P[n].V1[0] = P[n].V1[1] = P[n].V2[0] = P[n].V2[1] = P[n].V3[0] = P[n].V3[1] = zero_v;
	}

	// This wavefront's ring, and where its lanes read: lane l takes the l-th value (of sizeof(V) bytes) of a row segment.
	lds_char *const ring = block_rings + wave * (kRingRows * SLOT);
	// (COOP) where lanes 0 / 63 publish their values -- the wavefront's western / eastern slot -- and where they find their neighbours':
	// lane 0 the eastern slot of the wavefront to the west, lane 63 the western slot of the one to the east (the block's outer edges:
	// any slot, they are apron)
	unsigned edge_pub = 0, edge_con = 0;
	if constexpr (COOP) {
		const unsigned base = (unsigned)(uintptr_t)block_edges;
		edge_pub = (lane == 0 || lane == kLanes - 1) ? base + (unsigned)((wave * 2 + (lane == 0 ? 0 : 1)) * kEdgeSlotBytes<STEPS>)
		                                             : base + (unsigned)(kEdgeBytes<STEPS> + (wave * kLanes + lane) * 8);
		edge_con = base + (unsigned)((lane == 0 ? (wave > 0 ? wave - 1 : wave) * 2 + 1 : (wave + 1 < a.sw ? wave + 1 : wave) * 2) * kEdgeSlotBytes<STEPS>);
	}
	const unsigned ring_lane = (unsigned)(uintptr_t)ring + (unsigned)lane * (unsigned)sizeof(V);
	// ... and what they fetch: LDS-DMA instruction h of a row moves dwords 64 h + lane of the segment; the column of a dword's
	// element wraps periodically like x above
	unsigned doff[G1];
#pragma unroll
	for (int h = 0; h < G1; h++) {
		const int q = 4 * (kLanes * h + lane), e = q / (int)sizeof(Real);
		int col = (x0 + e) % nx;
		if (col < 0) col += nx;
		doff[h] = (unsigned)col * (unsigned)sizeof(Real) + (unsigned)(q % (int)sizeof(Real));
	}
	auto fill = [&](int jrow, int slot_byte) {  // row jrow of both fields -> the ring slot at byte `slot_byte` (uniform)
		const ptrdiff_t rb = row_base(jrow);
		const __amdgpu_buffer_rsrc_t ru = row_resource(a.in_u + rb), rv = row_resource(a.in_v + rb);
#pragma unroll
		for (int h = 0; h < G1; h++) __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, ring + (slot_byte + 256 * h), 4, doff[h], 0, 0, 0);
#pragma unroll
		for (int h = 0; h < G1; h++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, ring + (slot_byte + RB + 256 * h), 4, doff[h], 0, 0, 0);
	};
#pragma unroll
	for (int k = 0; k < kRingRows; k++) fill((jbase + k < jlast) ? jbase + k : jlast, k * SLOT);
	int jn = (jbase + kRingRows < jlast) ? jbase + kRingRows : jlast;  // the row the next fill takes (the tail re-reads the last valid row)
	Real pb = brow[jbase];                                            // b(j) of the row the next iteration takes
	int trip_byte = 0;                                                // ring byte offset of the slots of this trip of four iterations
	unsigned ring_trip = ring_lane;
	// (the rows that come out of the last pipeline at iteration m: row jbase + m - 4 STEPS)
	Real *out_row_u = a.out_u + (ptrdiff_t)(jbase - APRON) * nx, *out_row_v = a.out_v + (ptrdiff_t)(jbase - APRON) * nx;

	// Stages 1..4 of one step on the pipeline P whose newest row is `p` (slot S0): new state of row p - 4 in (nu, nv).
	// flag_bit: where the step's four stage flags start in amask; b4: b(j) of row p - 4 (read by the caller before row p took over its slot).
	// `live`: the iterations this pipeline has been fed for (m for the first, m - 8 n for the n-th; 8 and more: all stages run).  In the
	// chunk's first 8 STEPS iterations stage k has complete inputs only from the pipeline's iteration 2 k on -- the stages before that are
	// skipped (their rows lie outside what the chunk's outputs depend on: nothing downstream reads what they would have written).  Half
	// the work of those iterations, which is a tenth of a 61-row item's -- one rank's share of an 8-GPU run -- and a third of an edge band's.
	[[maybe_unused]] double2r edge[2 * STEPS];  // (COOP) the neighbours' edge values, lanes 0 and 63: quantity q of an iteration in edge[q / 2]
#pragma unroll
	for (int q = 0; q < 2 * STEPS; q++) edge[q] = double2r{0.0, 0.0};
	auto point = [&](V uC, V uS, V uN, V v, Real rowp, bool zero, [[maybe_unused]] V X, V &du, V &dv) {
		if constexpr (COOP) rhs_lane_edge<MODEL>(uC, X, uS, uN, v, cE, cWn, cP, rowp, ka4, zero, du, dv);
		else rhs_lane<V, MODEL>(uC, uS, uN, v, cE, cWn, cP, rowp, ka4, zero, du, dv);
	};
	auto stages = [&](Pipe &Q, const int p, auto kk, auto live_c, const int flag_bit, const Real b4, V &nu, V &nv, [[maybe_unused]] const V (&E)[4], auto qbase_c) {
		constexpr int K = decltype(kk)::value;
		[[maybe_unused]] constexpr int PUB = (K & 1) * kEdgeParityBytes<STEPS> + decltype(qbase_c)::value * 8;
		constexpr int live = decltype(live_c)::value;  // (compile-time: the filling iterations are so many pieces of straight-line code)
		constexpr int S0 = K % M, S1 = (K + M - 1) % M, S2 = (K + M - 2) % M, S3 = (K + M - 3) % M, S4 = (K + 2 * M - 4) % M, S5 = (K + 2 * M - 5) % M;
		V du, dv;
		if constexpr (live >= 2) {
			point(Q.u0[S1], Q.u0[S2], Q.u0[S0], Q.v0[S1], Q.bq[S1], ABSORB && ((amask >> (flag_bit + 0)) & 1) && boundary_row(p - 1), E[0], du, dv);
			Q.U1[S1] = stage_fma(h2, du, Q.u0[S1]);
			if constexpr (COOP) edge_publish<PUB + 8>(edge_pub, Q.U1[S1]);
			Q.V1[S1 & 1] = stage_fma(h2, dv, Q.v0[S1]);
			Q.aU[S1] = stage_fma(h6, du, Q.u0[S1]);
			Q.aV[S1] = stage_fma(h6, dv, Q.v0[S1]);
		}
		if constexpr (live >= 4) {
			point(Q.U1[S2], Q.U1[S3], Q.U1[S1], Q.V1[S2 & 1], Q.bq[S2], ABSORB && ((amask >> (flag_bit + 1)) & 1) && boundary_row(p - 2), E[1], du, dv);
			Q.U2[S2] = stage_fma(h2, du, Q.u0[S2]);
			if constexpr (COOP) edge_publish<PUB + 16>(edge_pub, Q.U2[S2]);
			Q.V2[S2 & 1] = stage_fma(h2, dv, Q.v0[S2]);
			Q.aU[S2] = stage_fma(h3, du, Q.aU[S2]);
			Q.aV[S2] = stage_fma(h3, dv, Q.aV[S2]);
		}
		if constexpr (live >= 6) {
			point(Q.U2[S3], Q.U2[S4], Q.U2[S2], Q.V2[S3 & 1], Q.bq[S3], ABSORB && ((amask >> (flag_bit + 2)) & 1) && boundary_row(p - 3), E[2], du, dv);
			Q.U3[S3] = stage_fma(h1, du, Q.u0[S3]);
			if constexpr (COOP) edge_publish<PUB + 24>(edge_pub, Q.U3[S3]);
			Q.V3[S3 & 1] = stage_fma(h1, dv, Q.v0[S3]);
			Q.aU[S3] = stage_fma(h3, du, Q.aU[S3]);
			Q.aV[S3] = stage_fma(h3, dv, Q.aV[S3]);
		}
		nu = nv = zero_v;
		if constexpr (live >= 8) {
			point(Q.U3[S4], Q.U3[S5], Q.U3[S3], Q.V3[S4 & 1], b4, ABSORB && ((amask >> (flag_bit + 3)) & 1) && boundary_row(p - 4), E[3], du, dv);
			nu = stage_fma(h6, du, Q.aU[S4]);
			nv = stage_fma(h6, dv, Q.aV[S4]);
		}
	};
	auto iteration = [&](int m, auto kk, auto fed_c) {
		constexpr int FED = decltype(fed_c)::value;  // iterations before this one, if fewer than FILL
		constexpr int K = decltype(kk)::value;
		constexpr int S0 = K % M, S4 = (K + 2 * M - 4) % M;
#if !defined(CRD_NO_LOCKSTEP) && !defined(CRD_NO_LOCKSTEP_TWO)
		// Lockstep of the block's wavefronts: fp32 only.  With the rows coming through the rings, the fp64 pipelines run 1 - 5 % faster
		// when their wavefronts drift (8192^2 FHN 0.2292 -> 0.2251 ms per step, 4096^2 0.0683 -> 0.0663, Goldbeter 4096^2 0.1212 -> 0.1148),
		// the packed fp32 one 2 % slower (16384^2 0.4230 -> 0.4308; profiles/r05/two_step_memory_path_ab.txt).
		// (round 6: ... and of the fp32 kernels only the two-step one: the three-step launch, bound by issue at two wavefronts per SIMD, runs
		// 2 % faster without -- C5 0.3689 -> 0.3625 ms per step, 8192^2 0.0998 -> 0.0975; the two-step one 3 % slower, as before)
		if constexpr (sizeof(Real) == 4 && STEPS == 2) __builtin_amdgcn_s_barrier();
#endif
		const int p = jbase + m;
		Real b4[STEPS];  // b(j) of the rows stage 4 of each step works on, read before row p takes over the slot
#pragma unroll
		for (int n = 0; n < STEPS; n++) b4[n] = P[n].bq[S4];
		// row p out of its ring slot (filled kRingRows iterations ago) ...
		if (m < FILL + kRingRows) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWaitFill) : "memory");  // (fewer operations in flight while no rows come out yet)
		V E[STEPS][4];
		if constexpr (COOP) {
			// ... together with the neighbours' edge values, whose reads the previous iteration issued behind its barrier
			ring_read_with_edges<kWaitSteady, S0 * SLOT, S0 * SLOT + RB>(ring_trip, P[0].u0[S0], P[0].v0[S0], edge);
#pragma unroll
			for (int n = 0; n < STEPS; n++) E[n][0] = edge[2 * n].x, E[n][1] = edge[2 * n].y, E[n][2] = edge[2 * n + 1].x, E[n][3] = edge[2 * n + 1].y;
			edge_publish<(K & 1) * kEdgeParityBytes<STEPS>>(edge_pub, P[0].u0[S0]);
		} else {
			ring_read<kWaitSteady, S0 * SLOT, S0 * SLOT + RB>(ring_trip, P[0].u0[S0], P[0].v0[S0]);
#pragma unroll
			for (int n = 0; n < STEPS; n++) E[n][0] = E[n][1] = E[n][2] = E[n][3] = zero_v;
		}
		P[0].bq[S0] = uniform(pb);
#ifndef CRD_PROBE_NOLOAD  // (probe builds, tools/build_variant.sh: what a launch costs without its row reads / its later steps / its stores)
		fill(jn, trip_byte + S0 * SLOT);  // ... and row p + kRingRows into it
#endif
		jn = (jn < jlast) ? jn + 1 : jlast;
		pb = brow[(p < jlast) ? p + 1 : jlast];
		V nu, nv;
		if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);  // (see the stores below)
#ifdef CRD_PROBE_NOMATH  // (probe build: the launch as a copy -- its memory traffic alone)
		nu = P[0].u0[S0] + (V)b4[0];
		nv = P[0].v0[S0] + (V)b4[STEPS - 1];
#else
		// step n of the launch: the new row p - 4 (n + 1), which is the next pipeline's newest row (same slot: rows shifted by 4)
		for_sequence(
		    [&](auto nn) {
			    constexpr int N = decltype(nn)::value;
#ifdef CRD_PROBE_HALFMATH
			    if constexpr (N > 0) return;
#endif
			    if constexpr (N > 0) {
				    P[N].u0[S0] = nu;
				    P[N].v0[S0] = nv;
				    P[N].bq[S0] = b4[N - 1];
				    if constexpr (COOP) edge_publish<(K & 1) * kEdgeParityBytes<STEPS> + 32 * N>(edge_pub, nu);
			    }
			    constexpr int live = FED - 8 * N < 0 ? 0 : (FED - 8 * N > 8 ? 8 : FED - 8 * N);
			    stages(P[N], p - kApron * N, kk, std::integral_constant<int, live>{}, 4 * N, b4[N], nu, nv, E[N], std::integral_constant<int, 4 * N>{});
		    },
		    std::make_integer_sequence<int, STEPS>{});
#endif
		// (FHN fp64) From its stores to the next row's fill a wavefront issues ahead of its SIMD's other wavefronts (which are in their
		// arithmetic): its memory operations go out when it reaches them instead of waiting their turn among vector instructions.
		// Nothing on 8192^2, -1 % on 4096^2, -2 % on a rank's 8192 x 1024 share; 1 - 2 % SLOWER in fp32 and 0 - 1 % in Goldbeter, which
		// do without (profiles/r05/two_step_memory_path_ab.txt, K).  Non-temporal row LOADS, tried beside it, cost 15 %: the rows' reuse
		// by the neighbouring items is what keeps the traffic at 1.06 x compulsory.
		if constexpr (PRIO) __builtin_amdgcn_s_setprio(2);
		// Rows j0 .. j1 - 1 exactly: every iteration behind the filling ones.  BOTH store instructions go out in every such iteration of
		// every wavefront, with all lanes on -- the waits of ring_read COUNT them (kWaitSteady).  A lane that has nothing to store (apron,
		// or a column beyond nx) is parked on an offset beyond the buffer resource's range, where the hardware drops its write; under an
		// `if (lane_stores)` the compiler puts a skip branch (s_cbranch_execz) in front of the stores, a wavefront without a storing lane
		// then has fewer operations in flight than the wait assumes, and its ring read can overtake the LDS-DMA fill of its slot.
		if constexpr (FED >= FILL) {
#if defined(CRD_PROBE_BRANCHY_STORES)  // (probe build: round 5's form, stores under the lanes' condition -- what the parked lanes cost; NOT safe, see above)
			if (lane_stores) {
				buffer_row_store<NT>(row_resource(out_row_u), xb, nu);
				buffer_row_store<NT>(row_resource(out_row_v), xb, nv);
			}
#elif !defined(CRD_PROBE_NOSTORE)
			buffer_row_store<NT>(row_resource(out_row_u), xb_store, nu);
			buffer_row_store<NT>(row_resource(out_row_v), xb_store, nv);
#else
			if (a.nchunks < 0) {  // (never: keeps the arithmetic alive)
				buffer_row_store<NT>(row_resource(out_row_u), xb_store, nu);
				buffer_row_store<NT>(row_resource(out_row_v), xb_store, nv);
			}
#endif
		}
		out_row_u += nx;
		out_row_v += nx;
		// (COOP) what the block's wavefronts published during this iteration is complete beyond the barrier; the reads of what the next
		// iteration needs of it go out at once and land while that iteration waits for its row
		if constexpr (COOP) edge_exchange<(K & 1) * kEdgeParityBytes<STEPS>>(edge_con, edge);
	};
	auto next_trip = [&]() {
		if constexpr (kRingRows > M) {
			trip_byte = (trip_byte + M * SLOT) & (kRingRows * SLOT - 1);
			ring_trip = ring_lane + (unsigned)trip_byte;
		}
	};
	int m = 0;
	for_sequence(  // the pipeline fills (a chunk has at least one row: FILL iterations and more)
	    [&](auto i) {
		    constexpr int I = decltype(i)::value;
		    iteration(I, std::integral_constant<int, I % M>{}, i);
		    if constexpr (I % M == M - 1) next_trip();
	    },
	    std::make_integer_sequence<int, FILL>{});
	m = FILL;
	for (; m + M - 1 < niter; m += M) {
		for_sequence([&](auto k) { iteration(m + decltype(k)::value, k, std::integral_constant<int, FILL>{}); }, std::make_integer_sequence<int, M>{});
		next_trip();
	}
	for_sequence(
	    [&](auto k) {
		    if (m + decltype(k)::value < niter) iteration(m + decltype(k)::value, k, std::integral_constant<int, FILL>{});
	    },
	    std::make_integer_sequence<int, M - 1>{});
	// the fills still in flight write LDS: they must have landed before the wavefront ends and its LDS goes to the next workgroup
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ABSORB = false: no stage of the step has t < tBoundary (every launch after the switch-off time, every launch of a run with
// tBoundary = 0) -- the absorbing-row selects are compiled out.  ABSORB = true: the items that can meet a global phi boundary row
// (src/FHNmodel_torus.cpp:643-653) run the body with the selects, all others the body without (see fused_item).
// EMBED, COLS, NT: see FusedArgs / fused_item above.
// STEPS = 2: two steps per launch (fused_item_two_steps).
// Wavefronts per SIMD the register allocator is held to: the two-step pipelines live at the 168-register line (three wavefronts).
template <typename Real, int MODEL, int COLS, int STEPS, bool ABSORB>
constexpr int kMinWaves = (STEPS == 2 && COLS * (int)sizeof(Real) == 8 && MODEL == CRD_MODEL_FHN && !ABSORB) ? 3 : (STEPS == 3 ? 2 : 1);
template <typename Real, int MODEL, bool ABSORB, int EMBED, int COLS, bool NT = false, int STEPS = 1>
__global__ void __launch_bounds__(kLanes *kMaxWavesPerBlock) __attribute__((amdgpu_waves_per_eu(kMinWaves<Real, MODEL, COLS, STEPS, ABSORB>))) crd_rk4_fused_step_kernel(Slab<Real> s, FusedArgs<Real> a)
{
	static_assert(STEPS == 1 || ((STEPS == 2 || STEPS == 3) && EMBED == 0), "several steps per launch: the plain step only");
	// The work item is a property of the wavefront: keep it (and everything derived from it: rows, trip counts, the
	// per-row table reads, the boundary-row tests) in scalar registers.
	// Optional remap: blocks are dealt round-robin over the 8 XCDs; the remap gives each XCD one contiguous run of items.
	// A block's wavefronts take adjacent strips of ONE chunk, blocks walk theta first.  The wavefronts of a block therefore
	// run the same trip counts, and in lockstep (one barrier per pipeline iteration) their row reads reach the memory system
	// together as one contiguous, overlapping run of a.sw x 448 B per row instead of drifting apart.
	const int nsb = (a.nstrips + a.sw - 1) / a.sw;
	int sblk, cblk;
	{
		const int b0 = (int)blockIdx.x;
		const int blk = a.remap == 1 ? xcd_remap(b0, a.nblocks) : b0;
		sblk = blk % nsb;
		cblk = blk / nsb;
		if (a.remap == 2) {
			// Succession in phi: XCD x owns a contiguous run of chunks; its resident workgroups form `xs_lanes` lanes per strip
			// block, and the workgroup that takes a finished one's place (ids are dispatched in order, 8 apart on one XCD) continues
			// that lane with the NEXT chunk in phi -- whose first rows are the rows its predecessor has just read into this L2.
			const int x = blk % kNumXcd, p = blk / kNumXcd, width = nsb * a.xs_lanes;
			const int d = p / width, sl = p - d * width;
			const int c0 = (int)((long)a.nchunks * x / kNumXcd), c1 = (int)((long)a.nchunks * (x + 1) / kNumXcd);
			const int depth = (c1 - c0 + a.xs_lanes - 1) / a.xs_lanes, lane_id = sl / nsb;
			sblk = sl - lane_id * nsb;
			cblk = (d < depth && c0 + lane_id * depth + d < c1) ? c0 + lane_id * depth + d : a.nchunks;  // nchunks: nothing to do
		}
	}
	const int strip = __builtin_amdgcn_readfirstlane(sblk * a.sw + (int)(threadIdx.x >> 6));
	const int chunk = __builtin_amdgcn_readfirstlane(cblk);
	if (strip >= a.nstrips || chunk >= a.nchunks) return;  // (a barrier waits for the surviving wavefronts of the workgroup only)
	// the row rings of a two-step launch's wavefronts (fused_item_two_steps)
	__shared__ __attribute__((aligned(16))) char rings[STEPS >= 2 ? kRingBytes<Real, COLS, STEPS> : 16];
	lds_char *const block_rings = (lds_char *)rings;
	__shared__ __attribute__((aligned(16))) char edges[(STEPS >= 2 && kCoop<Real, MODEL, COLS, STEPS>) ? kEdgeBytes<STEPS> + kEdgeDumpBytes<STEPS> : 16];
	lds_char *const block_edges = (lds_char *)edges;
	if constexpr (ABSORB) {
		// Does any row this chunk's pipeline touches -- [j0 - APRON, j1 + APRON) -- map to global row 0 or ny - 1?  The two are
		// neighbours on the periodic grid: the rows contain one of them exactly when [lo, hi + 1] contains a multiple of ny.
		constexpr int APRON = STEPS * (EMBED != 0 ? kApron + 1 : kApron);
		const int range = chunk >= a.first2 ? 1 : 0;
		const int j0 = a.r_begin[range] + (chunk - (range ? a.first2 : 0)) * a.chunk;
		const int j1 = (j0 + a.chunk < a.r_end[range]) ? j0 + a.chunk : a.r_end[range];
		const int lo = a.js + j0 - APRON, hi1 = a.js + j1 + APRON;  // (lo > -ny and hi1 < 3 ny: a slab is at most the grid, ghost rows at most a slab)
		const bool touches = (lo <= 0 && 0 <= hi1) || (lo <= a.ny && a.ny <= hi1) || (lo <= 2 * a.ny && 2 * a.ny <= hi1);
		if constexpr (STEPS >= 2) {
			if (touches) fused_item_multi_step<Real, MODEL, true, COLS, NT, STEPS>(s, a, strip, chunk, block_rings, sblk, block_edges);
			else fused_item_multi_step<Real, MODEL, false, COLS, NT, STEPS>(s, a, strip, chunk, block_rings, sblk, block_edges);
		} else {
			if (touches) fused_item<Real, MODEL, true, EMBED, COLS, NT>(s, a, strip, chunk);
			else fused_item<Real, MODEL, false, EMBED, COLS, NT>(s, a, strip, chunk);
		}
	} else if constexpr (STEPS >= 2) {
		fused_item_multi_step<Real, MODEL, false, COLS, NT, STEPS>(s, a, strip, chunk, block_rings, sblk, block_edges);
	} else {
		fused_item<Real, MODEL, false, EMBED, COLS, NT>(s, a, strip, chunk);
	}
}

// Adds the per-item error sums in a fixed order (thread t takes items t, t+256, ...; then a fixed LDS tree), so the error
// norm, and with it every accept / reject decision of the adaptive stepper, is reproducible from run to run.
__global__ void __launch_bounds__(256) crd_sum_partials_kernel(const double *__restrict__ partials, int n, double *__restrict__ out)
{
	__shared__ double part[256];
	double sum = 0.0;
	for (int q = threadIdx.x; q < n; q += 256) sum += partials[q];
	part[threadIdx.x] = sum;
	__syncthreads();
	for (int w = 128; w > 0; w >>= 1) {
		if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
		__syncthreads();
	}
	if (threadIdx.x == 0) *out = part[0];
}

// Rows per work item.  Every item pays 8 apron rows, which argues for long chunks; but the wavefronts of a launch run in
// "rounds" of (resident wavefront slots) items, a partly filled last round idles most of the chip, unequal wavefront
// speeds cost about half a round at the end whatever the count, and short chunks keep the rows two phi-neighbouring items
// share in L2.  Measured on 8192^2 fp64 (147 strips, 4096 slots; 200-step medians, one process, tools/tune_fused.py):
// chunk 32 0.434 ms, 24 0.451, 50 0.463, 60 0.477, 75 0.494, 128 0.51, 1024 0.62 -- many short items win.  So: 32 rows,
// halved while the launch would not fill every slot once (an 8192 x 1024 slab, one rank's share of 8 GPUs, sweeps in
// 58.4 us with 32-row chunks and 60.5 with 16; the edge-band launches of a multi-slab step end up with 8-row chunks).
// `one_round` (a launch-plan choice, see FusedPlan): a launch that needs more than one round of resident blocks but would
// fit into one with chunks of up to 96 rows gets those longer chunks -- no tail round on an almost idle chip.
int device_cus()
{
	// compute units of a device of this node (the GPUs of a node are alike); initialised once, also when several issuing threads of a
	// LOCAL group arrive together
	static const int cus = [] {
		int dev = 0;
		hipDeviceProp_t prop;
		const int n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
		(void)hipGetLastError();
		return n;
	}();
	return cus;
}

template <typename Real, int MODEL, int COLS, int STEPS>
int resident_wavefronts()
{
	// resident wavefronts of this kernel on a device of this node (initialised once, thread-safely: the issuing threads of a LOCAL group
	// may arrive together)
	static const int slots = [] {
		int blocks_per_cu = 4;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, crd_rk4_fused_step_kernel<Real, MODEL, false, 0, COLS, false, STEPS>, kLanes * kWavesPerBlock, 0) != hipSuccess ||
		    blocks_per_cu < 1)
			blocks_per_cu = 4;
		(void)hipGetLastError();
		return device_cus() * blocks_per_cu * kWavesPerBlock;
	}();
	return slots;
}

// Three steps per launch (round 6), for the kernels whose two-step launch is bound by its memory path rather than by issue (DESIGN.md
// 4c, 4d): FHN in fp64 with one column per lane and the block as the strip (232 valid lanes of 256), and FHN in fp32 with two columns
// per lane and a strip per wavefront (104 valid columns of 128 -- no exchange between wavefronts).  kThreeStepCols: the columns per
// lane of the model's three-step kernel, 0 where there is none.
#ifdef CRD_THREE_STEPS_GOLDBETER  // (probe build: the three-step pipeline for Goldbeter fp64 too -- profiles/r06/goldbeter_floor.txt)
template <typename Real, int MODEL>
constexpr int kThreeStepCols = (MODEL == CRD_MODEL_FHN || (MODEL == CRD_MODEL_GOLDBETER && sizeof(Real) == 8)) ? (sizeof(Real) == 8 ? 1 : 2) : 0;
#else
template <typename Real, int MODEL>
constexpr int kThreeStepCols = MODEL == CRD_MODEL_FHN ? (sizeof(Real) == 8 ? 1 : 2) : 0;
#endif
template <typename Real, int MODEL>
constexpr bool kCanThreeSteps = kThreeStepCols<Real, MODEL> != 0;

template <typename Real, int MODEL>
int resident_wavefronts(int cols, int steps = 1)
{
	if constexpr (kCanThreeSteps<Real, MODEL>)
		if (steps == 3) return resident_wavefronts<Real, MODEL, kThreeStepCols<Real, MODEL>, 3>();
	if constexpr (MODEL != kModelDiffusionOnly)  // (the diffusion-only variant has no two-step instantiation)
		if (steps == 2) return cols == 2 ? resident_wavefronts<Real, MODEL, 2, 2>() : resident_wavefronts<Real, MODEL, 1, 2>();
	return cols == 2 ? resident_wavefronts<Real, MODEL, 2, 1>() : resident_wavefronts<Real, MODEL, 1, 1>();
}

template <typename Real, int MODEL>
int fused_chunk_rows(int nstrips, int rows, int chunk_mode, int cols, int steps = 1)  // 0: 32 rows, 1: one round, 2: 64 rows (two steps per launch: twice that)
{
	bool one_round = chunk_mode == 1;
	const int slots = resident_wavefronts<Real, MODEL>(cols, steps);
	// Two steps per launch: 16 fill rows per chunk instead of 8, and a pipeline bound by issue, not by the memory system: 128-row
	// chunks (8192^2 fp64: 48 rows 0.301 ms per step, 64 0.289, 96 0.278, 128 0.267, 192 0.276, 256 0.276; fp32 alike,
	// profiles/r04/two_step_tune.txt); chunk mode 2 is the 64-row alternative there.
	int chunk = steps == 3 ? (chunk_mode == 2 ? 96 : 192) : steps == 2 ? (chunk_mode == 2 ? 64 : 128) : 32;  // (three steps: 24 fill rows per chunk)
	while (chunk > 8 && (long)nstrips * ((rows + chunk - 1) / chunk) < (long)slots) chunk /= 2;
	// Tiny launches: where even 8-row chunks make fewer blocks than half the CUs, 4-row chunks put twice as many CUs to work (256^2:
	// 7.3 -> 6.4 us per step; the reference's 100 x 400 Goldbeter grid: 8.2 -> 6.5).  With more blocks than that the extra apron rows
	// cost more than they bring (512^2: 8.9 -> 9.8 us, 400 x 1600: 11.1 -> 12.9; the edge bands of a ring share: no change).
	if (chunk == 8 && (long)((nstrips + kWavesPerBlock - 1) / kWavesPerBlock) * ((rows + 7) / 8) < device_cus() / 2) chunk = 4;
	if (const char *e = tuning::knob("CRD_FUSED_ONEROUND")) one_round = std::atoi(e) != 0;  // tuning knob
	if (steps == 1 && chunk_mode == 2 && chunk == 32 && (long)nstrips * ((rows + 63) / 64) >= 2L * slots) chunk = 64;  // fewer apron rows recomputed: pays where fp64 issue binds (Goldbeter)
	if (one_round && steps == 1) {
		const long strip_blocks = (nstrips + kWavesPerBlock - 1) / kWavesPerBlock, fit = (slots / kWavesPerBlock) / strip_blocks;
		const long need = fit >= 1 ? (rows + fit - 1) / fit : 0;
		if (need > chunk && need <= 96) chunk = (int)need;
	}
	if (one_round && steps >= 2) {
		// Two steps per launch are bound by issue, and what a launch loses is its last, partly filled round of resident blocks: chunks
		// such that the launch is just under a WHOLE NUMBER of rounds -- the fewest rounds whose chunks stay within 288 rows (longer
		// ones have fewer fill rows per row; 8192^2 fp64: 128 rows = 3.6 rounds 0.2667 ms per step, 155 = 2.97 rounds 0.2617, 235 =
		// 1.96 rounds 0.2618, but 161 = 2.86 rounds 0.2669 and 241 = 1.90 rounds 0.2697; 16384^2 fp32: 128 rows 0.4948, 274 rows 0.4830;
		// a rank's share of 1024 rows: one round of 61 rows; profiles/r04/whole_rounds.txt).
		const long strip_blocks = (nstrips + kWavesPerBlock - 1) / kWavesPerBlock, resident_blocks = slots / kWavesPerBlock;
		for (long k = 1; k <= 8; k++) {
			const long chunks = k * resident_blocks / strip_blocks;
			if (chunks < 1) continue;
			const long need = (rows + chunks - 1) / chunks;
			if (need <= (steps == 3 ? 432 : 288)) {
				if (need >= 16) chunk = (int)need;
				break;
			}
		}
	}
	if (const char *e = tuning::knob("CRD_FUSED_CHUNK")) {  // tuning knob
		const int v = std::atoi(e);
		if (v >= 1) chunk = v;
	}
	return chunk < rows ? chunk : rows;
}

// Launch-plan candidates the autotuner times: (chunks stretched to one round?, block -> item mapping).  Workgroups are dealt
// round-robin to the 8 XCDs, each with its own L2.  Mapping 0 walks the items theta-first in dispatch order: neighbouring
// items land on different XCDs and every apron column and row is fetched from beyond L2 by both items that need it (PMC:
// 39.5 B per point, reads 1.47 x the plane).  Mapping 1 gives each XCD one contiguous run of items, i.e. a contiguous band of
// the slab: theta-neighbours share an L2 and the apron columns are fetched once (36.6 B per point, reads 1.29 x).  Mapping 2
// adds succession in phi -- the workgroup that takes a finished one's place continues with the next chunk in phi, whose first
// rows its predecessor has just pulled into that L2 (34.9 B per point, reads 1.18 x).  Which plan is fastest depends on the
// grid shape AND on the device: on 8192^2 fp64 mapping 1 measured -6.3 %, -0.7 % and +1.7 % against mapping 0 on three
// MI355X of the same pool, mapping 2 -4.4 % on a fourth; one-round chunks -10 % (4096 x 1024) to +2 % (16384 x 2048 fp32) --
// hence measured at run time, on the device and the shape at hand.  Every candidate computes bit-identical results.
// Third dimension (round 3): columns per lane.  Two columns per lane halve the DPP moves and the apron share of a strip (8 of 128
// columns instead of 8 of 64) and, in fp32, use the packed arithmetic; they also halve the wavefronts in flight for the same
// bytes.  Which wins is again a matter of the kernel (fp32 / Goldbeter are issue-bound, FHN fp64 is not) and of the device.
// Fourth dimension (round 3, late): non-temporal stores of the new state (row_store).  They keep L2 for what items share; which
// mapping is fastest changes with them (the plain mapping gains most), so they are timed in combination.
struct PlanCandidate {
	int one_round, remap, cols, nt;  // one_round: the chunk mode -- 0 = 32 rows, 1 = stretched to one round, 2 = 64 rows
	int steps = 1;                   // RK4 steps per launch (2: fused_item_two_steps)
};
// (round 4: 64-row chunks also under mappings 1 / 2 and with two columns per lane -- where fp64 issue binds, Goldbeter, the recompute
// factor of the apron rows is what is left to cut: (64 + 8) / 64 x 128 / 120 = 1.20 against (32 + 8) / 32 x 64 / 56 = 1.43)
constexpr PlanCandidate kPlanCandidates[] = {
    {0, 0, 1, 0}, {0, 1, 1, 0}, {0, 2, 1, 0}, {1, 0, 1, 0}, {1, 1, 1, 0}, {2, 0, 1, 0}, {2, 1, 1, 0}, {2, 2, 1, 0},
    {0, 0, 2, 0}, {0, 1, 2, 0}, {0, 2, 2, 0}, {1, 0, 2, 0}, {1, 1, 2, 0}, {2, 0, 2, 0}, {2, 1, 2, 0}, {2, 2, 2, 0},
    {0, 0, 1, 1}, {0, 1, 1, 1}, {0, 2, 1, 1}, {1, 0, 1, 1}, {1, 1, 1, 1}, {2, 0, 1, 1}, {2, 1, 1, 1}, {2, 2, 1, 1},
    {0, 0, 2, 1}, {0, 1, 2, 1}, {0, 2, 2, 1}, {1, 0, 2, 1}, {1, 1, 2, 1}, {2, 0, 2, 1}, {2, 1, 2, 1}, {2, 2, 2, 1},
    // fifth dimension (round 4): two steps per launch (128-row chunks, or 64), non-temporal stores
    {0, 0, 1, 1, 2}, {0, 1, 1, 1, 2}, {0, 2, 1, 1, 2}, {1, 0, 1, 1, 2}, {1, 1, 1, 1, 2}, {2, 1, 1, 1, 2},
    {0, 0, 2, 1, 2}, {0, 1, 2, 1, 2}, {0, 2, 2, 1, 2}, {1, 0, 2, 1, 2}, {1, 1, 2, 1, 2}, {2, 1, 2, 1, 2},
    // sixth (round 6): three steps per launch, the block as the strip (FHN fp64: kCanThreeSteps), non-temporal stores
    // (whole-rounds chunks win by 7 % -- 293 rows = two rounds of resident blocks on 8192^2 against 192 rows = three and a bit --, plain
    // stores by half a per cent on some boxes)
    {0, 0, 1, 1, 3}, {0, 1, 1, 1, 3}, {0, 2, 1, 1, 3}, {1, 0, 1, 1, 3}, {1, 1, 1, 1, 3}, {1, 0, 1, 0, 3}, {1, 1, 1, 0, 3},
    // ... and in fp32: two columns per lane, a strip per wavefront (kThreeStepCols)
    {0, 0, 2, 1, 3}, {0, 1, 2, 1, 3}, {0, 2, 2, 1, 3}, {1, 0, 2, 1, 3}, {1, 1, 2, 1, 3}, {2, 1, 2, 1, 3}};
constexpr int kNumPlanCandidates = (int)(sizeof kPlanCandidates / sizeof kPlanCandidates[0]);

template <typename Real, int MODEL>
hipError_t launch_fused_t(const SlabDesc &d, const FusedCall &c, int row_begin, int row_end, int row_begin2, int row_end2, int js, int ny,
                          hipStream_t st)
{
	clear_launch_status();
	if (row_end <= row_begin) {
		// nothing to launch: the events a caller bound to the launch are recorded on the stream instead, so that a wait on them sees
		// this call and not an earlier one
		if (c.geometry) return hipSuccess;
		if (c.start_event)
			if (hipError_t e = hipEventRecord(c.start_event, st); e != hipSuccess) return e;
		if (c.done_event)
			if (hipError_t e = hipEventRecord(c.done_event, st); e != hipSuccess) return e;
		return hipSuccess;
	}
	if (row_end2 < row_begin2) row_end2 = row_begin2;
	// Two steps per launch: plain steps only, and not the diffusion-only variant (no instantiation: that model is a plumbing case).
	constexpr bool kCanTwoSteps = MODEL != kModelDiffusionOnly;
	if (c.steps != 1 && (c.embed || !((c.steps == 2 && kCanTwoSteps) || (c.steps == 3 && kCanThreeSteps<Real, MODEL> && (kThreeStepCols<Real, MODEL> == 1 || d.nx % 2 == 0)))))
		return hipErrorInvalidValue;
	// rows may extend into the ghost region (deep-halo steps), but the pipeline reads kStepHalo rows per step beyond them
	if (!d.wrap && (row_begin < -(kGhost - c.steps * kStepHalo) || row_end > d.nyl + (kGhost - c.steps * kStepHalo))) return hipErrorInvalidValue;
	const Slab<Real> s = typed<Real>(d);
	FusedArgs<Real> a;
	a.in_u = row0<Real>(c.y0.u, d.nx);
	a.in_v = row0<Real>(c.y0.v, d.nx);
	a.out_u = row0<Real>(c.yout.u, d.nx);
	a.out_v = row0<Real>(c.yout.v, d.nx);
	a.h2 = (Real)(0.5 * c.dt);
	a.h3 = (Real)(c.dt / 3.0);
	a.h6 = (Real)(c.dt / 6.0);
	a.h1 = (Real)c.dt;
	for (int k = 0; k < 5; k++) a.absorb[k] = c.absorb[k];
	for (int k = 0; k < 4; k++) a.absorb2[k] = c.absorb2[k];
	for (int k = 0; k < 4; k++) a.absorb3[k] = c.absorb3[k];
	a.js = js;
	a.ny = ny;
	const int rows = row_end - row_begin, rows2 = row_end2 - row_begin2;
	// two columns per lane need an even nx (a pair must not straddle the periodic seam; rows then are 8- / 16-byte aligned too)
	const bool cols2_ok = !c.embed && d.nx % 2 == 0;
	// (where nothing has been measured: the packed arithmetic for fp32, one column for fp64 -- fused_default_columns)
	int cols_default = (cols2_ok && sizeof(Real) == 4) ? 2 : 1;
	if (const char *e = tuning::knob("CRD_FUSED_COLS")) cols_default = (std::atoi(e) == 2 && cols2_ok) ? 2 : 1;
	// Four adjacent strips per block marching in lockstep: 0.417 ms on 8192^2 fp64 against 0.441 without the barriers and
	// 0.4205 with one barrier per four iterations (tools/tune_fused.py, interleaved in one process; fp32 0.219 vs 0.232,
	// Goldbeter -- instruction-bound -- unchanged); 2 or 8 strips per block lose half of the gain, 3 / 5 / 6 more.
	int sw = kWavesPerBlock;
	if (const char *e = tuning::knob("CRD_FUSED_STRIPS")) {
		const int v = std::atoi(e);
		if (v >= 1 && v <= kMaxWavesPerBlock) sw = v;
	}
	a.sw = sw;
	a.err_partials = c.err_partials ? c.err_partials + c.err_offset : nullptr;
	a.err_lo = d.wrap ? INT32_MIN : 0;
	a.err_hi = d.wrap ? INT32_MAX : d.nyl;
	a.rtol = (Real)c.rtol;
	a.atol = (Real)c.atol;
	// the reaction block of a diffusion-only run is skipped, absorbing rows included (src/GoldbeterModel_torus.cpp:668)
	constexpr bool kCanAbsorb = MODEL != kModelDiffusionOnly;
	const bool absorb1 = kCanAbsorb && (c.absorb[0] || c.absorb[1] || c.absorb[2] || c.absorb[3] || (c.embed && c.absorb[4]));
	const bool absorb12 = absorb1 || (kCanAbsorb && (c.absorb2[0] || c.absorb2[1] || c.absorb2[2] || c.absorb2[3])) ||
	                      (kCanAbsorb && c.steps == 3 && (c.absorb3[0] || c.absorb3[1] || c.absorb3[2] || c.absorb3[3]));  // (any stage of any step of the launch)
	// (fp32: the three-step pipeline with the absorbing-row selects does not fit the registers -- 256 and scratch; the steppers step
	// such triples as a pair and a single step, run_steps: triple_absorbs)
	if (c.steps == 3 && sizeof(Real) == 4 && absorb12) return hipErrorInvalidValue;
	const dim3 block(kLanes * sw);

	int cols = cols_default, steps = 1;
	bool nt = false;
	// The launch's geometry in two layers: configure() fixes the plan's choices, layout() cuts the rows in R -- the caller's, or
	// a part of them (fire() below) -- into items accordingly.
	int R[4] = {row_begin, row_end, row_begin2, row_end2}, plan_mode = 0, plan_remap = 0, chunk_override = 0;
	auto layout = [&]() {
		const int rows_a = R[1] - R[0], rows_b = R[3] - R[2];
		const int nsb = (a.nstrips + sw - 1) / sw;
		a.chunk = chunk_override > 0 ? std::min(chunk_override, rows_a + rows_b) : fused_chunk_rows<Real, MODEL>(a.nstrips, rows_a + rows_b, plan_mode, cols, steps);
		const int n1 = (rows_a + a.chunk - 1) / a.chunk, n2 = (rows_b + a.chunk - 1) / a.chunk;
		a.nchunks = n1 + n2;
		a.first2 = n2 > 0 ? n1 : a.nchunks;
		a.r_begin[0] = R[0];
		a.r_end[0] = R[1];
		a.r_begin[1] = R[2];
		a.r_end[1] = R[3];
		a.nitems = a.nstrips * a.nchunks;
		a.nblocks = nsb * a.nchunks;
		a.remap = plan_remap;
		if (const char *e = tuning::knob("CRD_FUSED_REMAP")) a.remap = std::atoi(e);
		a.xs_lanes = 1;
		if (a.remap == 2) {
			const int per_xcd = resident_wavefronts<Real, MODEL>(cols, steps) / sw / kNumXcd;
			if (rows_b > 0 || a.nchunks < 2 * kNumXcd || per_xcd < nsb) {
				a.remap = 0;  // two row ranges, or too few chunks / slots for lanes: plain order
			} else {
				a.xs_lanes = per_xcd / nsb;
				const int most = (a.nchunks + kNumXcd - 1) / kNumXcd;  // chunks of the best-served XCD
				a.nblocks = kNumXcd * ((most + a.xs_lanes - 1) / a.xs_lanes) * nsb * a.xs_lanes;
			}
		}
	};
	auto configure = [&](int one_round, int remap, int want_cols, int want_nt = 0, int want_steps = 1) {
		steps = (want_steps == 3 && kCanThreeSteps<Real, MODEL> && (kThreeStepCols<Real, MODEL> == 1 || cols2_ok)) ? 3 : (want_steps >= 2 && kCanTwoSteps) ? 2 : 1;
		nt = want_nt != 0;
		if (const char *e = tuning::knob("CRD_FUSED_NT")) nt = std::atoi(e) != 0;
		cols = (want_cols == 2 && cols2_ok) ? 2 : 1;
		if (const char *e = tuning::knob("CRD_FUSED_COLS")) cols = (std::atoi(e) == 2 && cols2_ok) ? 2 : 1;
		if (steps == 3) cols = kThreeStepCols<Real, MODEL>;  // (the three-step pipeline has ONE form per model and precision)
		const int valid = cols * kLanes - 2 * steps * (c.embed ? kApron + 1 : kApron);  // (the embedded estimators' fifth stage costs one more apron column per side)
		a.nstrips = (d.nx + valid - 1) / valid;
		if ((steps == 2 && cols == 1 && kCoop<Real, MODEL, 1, 2>) || (steps == 3 && kCoop<Real, MODEL, 1, 3>)) {  // the block as the strip: one apron around its sw wavefronts
			const int block_valid = sw * kLanes - 2 * steps * kApron;
			a.nstrips = sw * ((d.nx + block_valid - 1) / block_valid);
		}
		plan_mode = one_round;
		plan_remap = remap;
		layout();
	};
	auto fire = [&]() -> hipError_t {
		if (c.embed) {
			if (!c.err_partials || c.err_capacity < c.err_offset + a.nitems || !c.err_sum) return hipErrorInvalidValue;
			auto with = [&](auto absorb_c, auto embed_c, auto nt_c) {
				auto kernel = crd_rk4_fused_step_kernel<Real, MODEL, decltype(absorb_c)::value && kCanAbsorb, decltype(embed_c)::value, 1, decltype(nt_c)::value>;
				// (the sum deferred to the caller -- another stream, launch_sum_partials: the event is this kernel's own completion)
				if (c.err_defer_sum && c.done_event) hipExtLaunchKernelGGL(kernel, dim3(a.nblocks), block, 0, st, nullptr, c.done_event, 0, s, a);
				else kernel<<<a.nblocks, block, 0, st>>>(s, a);
			};
			auto with_embed = [&](auto absorb_c, auto nt_c) {
				if (c.embed == 2) with(absorb_c, std::integral_constant<int, 2>{}, nt_c);
				else with(absorb_c, std::integral_constant<int, 1>{}, nt_c);
			};
			auto with_nt = [&](auto absorb_c) {
				if (nt) with_embed(absorb_c, std::true_type{});
				else with_embed(absorb_c, std::false_type{});
			};
			if (absorb1) with_nt(std::true_type{});
			else with_nt(std::false_type{});
			if (c.err_items_out) *c.err_items_out = a.nitems;
			if (c.err_defer_sum) return launch_status();  // (the second launch of a cut attempt sums both launches' partials)
			if (c.done_event) hipExtLaunchKernelGGL(crd_sum_partials_kernel, dim3(1), dim3(256), 0, st, nullptr, c.done_event, 0, (const double *)c.err_partials, c.err_offset + a.nitems, c.err_sum);
			else crd_sum_partials_kernel<<<1, 256, 0, st>>>(c.err_partials, c.err_offset + a.nitems, c.err_sum);
		} else {
			// plain step: absorbing rows x columns per lane x store hint x steps per launch, all compile-time
			bool last_launch = true, first_launch = true;  // (of this call: the ones a done_event / a start_event is bound to)
			auto with = [&](auto absorb_c, auto cols_c, auto nt_c, auto steps_c) {
				constexpr int kSteps = decltype(steps_c)::value;
				auto kernel = crd_rk4_fused_step_kernel<Real, MODEL, decltype(absorb_c)::value && kCanAbsorb, 0, decltype(cols_c)::value, decltype(nt_c)::value,
				                                        (kSteps == 3 ? kCanThreeSteps<Real, MODEL> && decltype(cols_c)::value == kThreeStepCols<Real, MODEL> &&
				                                                           !(sizeof(Real) == 4 && decltype(absorb_c)::value)  // (never launched: see above)
				                                                     : kCanTwoSteps) ? kSteps : 1>;
				hipEvent_t e0 = first_launch ? c.start_event : nullptr, e1 = last_launch ? c.done_event : nullptr;
				if (e0 || e1) hipExtLaunchKernelGGL(kernel, dim3(a.nblocks), block, 0, st, e0, e1, 0, s, a);
				else kernel<<<a.nblocks, block, 0, st>>>(s, a);
				first_launch = false;
			};
			auto with_steps = [&](auto absorb_c, auto cols_c, auto nt_c) {
				if (steps == 3) with(absorb_c, cols_c, nt_c, std::integral_constant<int, 3>{});
				else if (steps == 2) with(absorb_c, cols_c, nt_c, std::integral_constant<int, 2>{});
				else with(absorb_c, cols_c, nt_c, std::integral_constant<int, 1>{});
			};
			auto with_cols = [&](auto absorb_c, auto nt_c) {
				if (cols == 2) with_steps(absorb_c, std::integral_constant<int, 2>{}, nt_c);
				else with_steps(absorb_c, std::integral_constant<int, 1>{}, nt_c);
			};
			auto with_nt = [&](auto absorb_c) {
				if (nt) with_cols(absorb_c, std::true_type{});
				else with_cols(absorb_c, std::false_type{});
			};
			auto launch = [&](bool with_selects) {
				if (with_selects) with_nt(std::true_type{});
				else with_nt(std::false_type{});
			};
			if (steps >= 2 && absorb12) {
				// (Three steps per launch alike: ONE launch of the kernel with the selects -- 246 registers against 244, two wavefronts per
				// SIMD either way -- measured 0.2579 ms per step against 0.2180 with the cut and 0.2049 without absorbing rows: it is
				// the kernel that holds both bodies that is slow, not its occupancy.)
				// Two steps per launch with absorbing rows on.  The ABSORB kernel holds the body with the selects AND the one without
				// (it decides per chunk), and the former's scalar registers spill into two vector registers of the whole kernel: 170
				// VGPRs, two wavefronts per SIMD instead of three for EVERY item of the launch (+20 ... 38 % measured).  So the rows are
				// cut: those whose pipeline can meet a global boundary row -- within 2 kApron rows of rows ny - 1 / 0, a band of 18 --
				// go out first as a launch of their own (ABSORB kernel, 3-row items: it is the items' length, not their number, that
				// sets such a launch's duration), the rest as launches of the select-free kernel.  Same arithmetic, same bits.
				const int want[4] = {R[0], R[1], R[2], R[3]};
				int with_sel[4][2], without[6][2], n_with = 0, n_without = 0;
				bool fits = true;
				for (int r = 0; r < 2 && fits; r++) {
					int cursor = want[2 * r];
					const int end = want[2 * r + 1];
					for (int g = -1; g <= 1 && cursor < end; g++) {  // local rows of global rows ny - 1 and 0, one period down / here / one up
						const int jb = (ny - 1 - js) + g * ny, lo = std::max(cursor, jb - steps * kApron), hi = std::min(end, jb + 2 + steps * kApron);
						if (lo >= hi) continue;
						if (cursor < lo) {
							if (n_without == 6) fits = false;
							else without[n_without][0] = cursor, without[n_without++][1] = lo;
						}
						if (n_with == 4) fits = false;
						else with_sel[n_with][0] = lo, with_sel[n_with++][1] = hi;
						cursor = hi;
					}
					if (cursor < end) {
						if (n_without == 6) fits = false;
						else without[n_without][0] = cursor, without[n_without++][1] = end;
					}
				}
				if (!fits || n_with == 0) {
					launch(true);
				} else {
					auto issue = [&](int (*piece)[2], int count, bool selects, int item_rows, bool final_pieces) {
						for (int q = 0; q < count; q += 2) {
							last_launch = final_pieces && q + 2 >= count;
							R[0] = piece[q][0];
							R[1] = piece[q][1];
							R[2] = q + 1 < count ? piece[q + 1][0] : 0;
							R[3] = q + 1 < count ? piece[q + 1][1] : 0;
							chunk_override = item_rows;
							layout();
							launch(selects);
						}
					};
					issue(with_sel, n_with, true, 3, n_without == 0);
					issue(without, n_without, false, 0, true);
					last_launch = true;
					for (int q = 0; q < 4; q++) R[q] = want[q];
					chunk_override = 0;
					layout();
				}
			} else {
				launch(steps >= 2 ? absorb12 : absorb1);
			}
		}
		return launch_status();
	};

	// Launch plan: measured once per context on the first full-size launch (a launch reads one plane set and writes another, so
	// repeating it is harmless: every candidate writes the same values), then reused for every launch of similar height.
	FusedPlan *plan = c.plan;
	const bool plannable = plan && rows2 == 0 && (long)rows * d.nx >= (1L << 20) && !tuning::enabled();  // (under CRD_TUNING the knobs decide)
	if (plannable && !plan->tuned && plan->autotune && !c.geometry) {
		hipEvent_t e0 = nullptr, e1 = nullptr;
		// (nothing else may run on the device while candidates are timed: a halo exchange still in flight on the second stream
		// made a ring context pick a different -- worse -- plan than a plain slab of the same shape)
		hipError_t err = hipDeviceSynchronize();
		if (err == hipSuccess) err = hipEventCreate(&e0);
		if (err == hipSuccess) err = hipEventCreate(&e1);
		// What a candidate is timed on: with a scratch plane set, launches that step yout -> scratch -> yout -> ... (each reads what
		// its predecessor wrote, as consecutive steps do); without, repetitions of y0 -> yout.  The difference matters on slabs
		// small enough for the memory-side cache: an input that is never overwritten stays there, and a plan that keeps its output
		// out of the caches (non-temporal stores) then looks better than it steps.
		const bool pingpong = c.tune_scratch.u != nullptr && c.tune_scratch.v != nullptr;
		auto set_io = [&](const Planes &in, const Planes &out) {
			a.in_u = row0<Real>(in.u, d.nx);
			a.in_v = row0<Real>(in.v, d.nx);
			a.out_u = row0<Real>(out.u, d.nx);
			a.out_v = row0<Real>(out.v, d.nx);
		};
		int flip = 0;
		auto fire_timed = [&]() -> hipError_t {
			if (pingpong) {
				if (flip) set_io(c.tune_scratch, c.yout);
				else set_io(c.yout, c.tune_scratch);
				flip ^= 1;
			}
			return fire();
		};
		if (pingpong && err == hipSuccess) {
			configure(0, 0, cols_default, 0);
			err = fire();  // yout holds a state to step on from
		}
		// Candidates are timed round-robin, kRounds times, and each keeps its best round: a device's clock drifts while the
		// measurement runs (a Goldbeter launch sequence lost 15 % over five candidates timed one after the other), and a
		// candidate must not win or lose by its place in the queue.
		constexpr int kCandidates = kNumPlanCandidates, kRounds = 3;
		// (a candidate that takes more steps per launch than the call reads more rows beyond the launch's: they must exist -- a launch of
		// a multi-slab cycle that reaches far into the ghost region is measured with the candidates of its own step count only)
		const bool room_for_two = d.wrap || (row_begin >= -(kGhost - 2 * kStepHalo) && row_end <= d.nyl + (kGhost - 2 * kStepHalo));
		const bool two_steps_ok = kCanTwoSteps && !c.embed && c.steps == 1 && d.nyl >= 4 * kStepHalo && room_for_two;  // (the caller steps pairs once the plan says so)
		const bool room_for_three = d.wrap || (row_begin >= -(kGhost - 3 * kStepHalo) && row_end <= d.nyl + (kGhost - 3 * kStepHalo));
		const bool three_steps_ok = two_steps_ok && kCanThreeSteps<Real, MODEL> && (kThreeStepCols<Real, MODEL> == 1 || cols2_ok) && d.nyl >= 6 * kStepHalo &&
		                            (d.wrap || sizeof(Real) == 4) && room_for_three &&
		                            !(sizeof(Real) == 4 && absorb12);  // (... or triples: single slabs, and in fp32 slabs of a run; fp32 without absorbing rows on)
		float t_best[kCandidates];
		bool live[kCandidates];
		int reps = 3;
		for (int k = 0; k < kCandidates; k++) {
			t_best[k] = 0.f;
			configure(kPlanCandidates[k].one_round, kPlanCandidates[k].remap, kPlanCandidates[k].cols, kPlanCandidates[k].nt, kPlanCandidates[k].steps);
			live[k] = k == 0 || !(kPlanCandidates[k].one_round && a.chunk == fused_chunk_rows<Real, MODEL>(a.nstrips, rows, 0, cols, steps));  // (same as a 32-row plan)
			if (live[k] && kPlanCandidates[k].steps != ((kPlanCandidates[k].steps == 3 ? three_steps_ok : two_steps_ok) ? steps : 1)) live[k] = false;  // (several steps per launch: plain steps)
			if (live[k] && steps == 2 && cols == 2 && sizeof(Real) == 8) live[k] = false;  // (256 VGPRs, one wavefront per SIMD: measured 0.347 against 0.312 ms)
			if (live[k] && kPlanCandidates[k].remap != a.remap) live[k] = false;  // (the mapping fell back to dispatch order)
			if (live[k] && kPlanCandidates[k].cols != cols) live[k] = false;      // (two columns per lane not possible here, or pinned by a knob)
			if (live[k] && (kPlanCandidates[k].nt != 0) != nt) live[k] = false;   // (pinned by a knob)
		}
		for (int round = 0; round < kRounds && err == hipSuccess; round++)
			for (int k = 0; err == hipSuccess && k < kCandidates; k++) {
				if (!live[k]) continue;
				configure(kPlanCandidates[k].one_round, kPlanCandidates[k].remap, kPlanCandidates[k].cols, kPlanCandidates[k].nt, kPlanCandidates[k].steps);
				float ms = 0.f;
				for (int pass = 0; pass < 2 && err == hipSuccess; pass++) {
					err = fire_timed();  // warm-up of this variant
					if (err == hipSuccess) err = hipEventRecord(e0, st);
					for (int r = 0; err == hipSuccess && r < reps; r++) err = fire_timed();
					if (err == hipSuccess) err = hipEventRecord(e1, st);
					if (err == hipSuccess) err = hipEventSynchronize(e1);
					if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
					if (round > 0 || k > 0 || reps > 3 || ms >= 4.0f || ms <= 0.f) break;
					reps = (int)(12.0f / ms) + 1 < 40 ? (int)(12.0f / ms) + 1 : 40;  // time about four milliseconds' worth per candidate and round, then again
				}
				if (err != hipSuccess) break;
				ms /= (float)(reps * steps);  // per STEP: a two-step launch does twice the work
				if ((plan->autotune >= 2 || tuning::verbose()))
					std::fprintf(stderr, "libcrd autotune: %d x %d rows, round %d, chunk mode %d (%d rows), mapping %d, %d column(s) per lane, %s stores, %d step(s) per launch: %.4f ms per step (%d launches timed)\n",
					             d.nx, rows, round, kPlanCandidates[k].one_round, a.chunk, a.remap, cols, nt ? "non-temporal" : "plain", steps, ms, reps);
				if (t_best[k] == 0.f || ms < t_best[k]) t_best[k] = ms;
			}
		// Final: with two dozen candidates a few per cent apart, the fastest of the short bursts above is as often the luckiest as
		// the best, and a burst runs at clocks a sustained run does not keep.  The plain plan and the three fastest candidates are
		// therefore timed again, ~16 ms each and twice round, and the final alone decides between them.
		constexpr int kFinalists = 4, kFinalRounds = 2;
		int finalist[kFinalists] = {0, -1, -1, -1};
		for (int f = 1; f < kFinalists; f++)
			for (int k = 1; k < kCandidates; k++) {
				if (!live[k] || t_best[k] <= 0.f || k == finalist[1] || k == finalist[2]) continue;
				if (finalist[f] < 0 || t_best[k] < t_best[finalist[f]]) finalist[f] = k;
			}
		float t_final[kFinalists] = {0.f, 0.f, 0.f, 0.f};
		for (int round = 0; round < kFinalRounds && err == hipSuccess; round++)
			for (int f = 0; err == hipSuccess && f < kFinalists; f++) {
				const int k = finalist[f];
				if (k < 0 || t_best[k] <= 0.f) continue;
				configure(kPlanCandidates[k].one_round, kPlanCandidates[k].remap, kPlanCandidates[k].cols, kPlanCandidates[k].nt, kPlanCandidates[k].steps);
				const int reps2 = (int)(16.0f / t_best[k]) + 1 < 400 ? (int)(16.0f / t_best[k]) + 1 : 400;
				float ms = 0.f;
				err = fire_timed();
				if (err == hipSuccess) err = hipEventRecord(e0, st);
				for (int r = 0; err == hipSuccess && r < reps2; r++) err = fire_timed();
				if (err == hipSuccess) err = hipEventRecord(e1, st);
				if (err == hipSuccess) err = hipEventSynchronize(e1);
				if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
				if (err != hipSuccess) break;
				ms /= (float)(reps2 * steps);
				if ((plan->autotune >= 2 || tuning::verbose()))
					std::fprintf(stderr, "libcrd autotune: %d x %d rows, final %d, chunk mode %d (%d rows), mapping %d, %d column(s) per lane, %s stores, %d step(s) per launch: %.4f ms per step (%d launches timed)\n",
					             d.nx, rows, round, kPlanCandidates[k].one_round, a.chunk, a.remap, cols, nt ? "non-temporal" : "plain", steps, ms, reps2);
				if (t_final[f] == 0.f || ms < t_final[f]) t_final[f] = ms;
			}
		int best_k = 0;
		float base = t_best[0], best = t_best[0];
		if (t_final[0] > 0.f) {
			base = best = t_final[0];
			for (int f = 1; f < kFinalists; f++)
				if (finalist[f] >= 0 && t_final[f] > 0.f && t_final[f] < best) {
					best = t_final[f];
					best_k = finalist[f];
				}
		}
		if (e0) (void)hipEventDestroy(e0);
		if (e1) (void)hipEventDestroy(e1);
		set_io(c.y0, c.yout);  // (the launch this call was made for follows below)
		if (err != hipSuccess) return err;
		if (best > 0.985f * base) best_k = 0;  // a candidate has to beat the plain plan by more than timing noise
		plan->tuned = 1;
		plan->one_round = kPlanCandidates[best_k].one_round;
		plan->remap = kPlanCandidates[best_k].remap;
		plan->cols = kPlanCandidates[best_k].cols;
		plan->nt = kPlanCandidates[best_k].nt;
		plan->steps = kPlanCandidates[best_k].steps;
		plan->rows = rows;
		plan->ms_default = base;
		plan->ms_best = best_k ? best : base;
	}
	// A measured plan applies to the launches it was measured on (heights within a tenth of it: the sweeps of a deep-halo cycle are).
	// Launches it was not measured on -- edge bands, short ranges -- still take its columns per lane: that choice is about the
	// kernel's arithmetic, not about the launch's shape.  A PINNED plan (crd_set_launch_plan) is an instruction, not a measurement:
	// every single-range launch of the context takes all of it, of whatever size (chunk mode and mapping fall back inside configure
	// where the launch is too small for them), two-range launches its columns per lane and store hint.
	const bool pinned = plan && plan->tuned && plan->pinned && !tuning::enabled();
	const bool use_plan = (plannable && plan->tuned && 10L * rows >= 9L * plan->rows && 10L * rows <= 11L * plan->rows) || (pinned && rows2 == 0);
	configure(use_plan ? plan->one_round : 0, use_plan ? plan->remap : 0, (plan && plan->tuned) ? plan->cols : cols_default, (use_plan || pinned) ? plan->nt : 0, c.steps);
	if (c.geometry) {
		FusedGeometry &g = *c.geometry;
		const int apron = steps * (c.embed ? kApron + 1 : kApron);
		g.real_bytes = (int)sizeof(Real);
		g.model = MODEL;
		g.absorb = steps >= 2 ? 0 : (absorb1 ? 1 : 0);  // (several steps with absorbing rows on: the bulk goes out as the select-free kernel)
		g.embed = c.embed;
		g.cols = cols;
		g.nt = nt ? 1 : 0;
		g.steps = steps;
		g.strips = a.nstrips;
		g.chunk_rows = a.chunk;
		g.chunks = a.nchunks;
		g.blocks = a.nblocks;
		g.waves_per_block = sw;
		g.fill_iterations = 2 * apron;
		g.iterations_per_trip = c.embed ? CRD_EMBED_SLOTS : 4;
		g.lanes = kLanes;
		g.lanes_valid = (cols * kLanes - 2 * apron) / cols;
		if ((steps == 2 && cols == 1 && kCoop<Real, MODEL, 1, 2>) || (steps == 3 && kCoop<Real, MODEL, 1, 3>)) g.lanes_valid = (sw * kLanes - 2 * apron) / sw;  // the block as the strip: 60 (58) of 64 on average
		g.mapping = a.remap;
		g.wave_iterations = (long)a.nstrips * ((long)(rows + rows2) + (long)a.nchunks * 2 * apron);
		return hipSuccess;
	}
	return fire();
}

}  // namespace

// the fp32 half of launch_fused_step (crd_fused_f32.hip)
hipError_t launch_fused_step_f32(const SlabDesc &d, const FusedCall &c, int row_begin, int row_end, int row_begin2, int row_end2, hipStream_t s);

}  // namespace crd
