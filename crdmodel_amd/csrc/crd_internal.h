// crd_internal.h -- declarations shared by the translation units of libcrd (not part of the ABI).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "crd.h"

namespace crd {

// Constants of the reference programs.
constexpr double kPi = 3.1415926535897932;  // src/FHNmodel_torus.cpp:63
constexpr double kFhnEpsilon = 0.36;        // src/FHNmodel_torus.cpp:68
// src/GoldbeterModel_torus.cpp:67-78
constexpr double kGbV0 = 1.0, kGbK = 10.0, kGbKf = 1.0, kGbV1 = 7.3, kGbVm2 = 65.0, kGbVm3 = 500.0;
constexpr double kGbK2 = 1.0, kGbKr = 2.0, kGbKa = 0.9;

// Bounds on the spectral radius of the reaction Jacobian that crd_stable_dt adds to the diffusion operator's (crd.h has the
// derivation): FHN |d(3u - u^3)/du| = |3 - 3u^2| <= 9 for |u| <= 2 (the limit cycle stays inside), +1 for the v coupling;
// Goldbeter: spectral radius of the kinetics' Jacobian over 0 <= Z <= 1.5, 0 <= Y <= 3 is 333 (|dv3/dZ| reaches 410 there), rounded up.
constexpr double kFhnReactionRate = 10.0, kGoldbeterReactionRate = 400.0;

// Rows one fused RK4 step consumes on each side of the rows it produces (one per stage).
constexpr int kStepHalo = 4;
// Fused steps between two halo exchanges of a multi-slab run (the exchange period E): the exchange moves kStepHalo * E ghost rows
// and each slab recomputes the shrinking ghost region redundantly in between (communication-avoiding deep halo).  A property of
// the run, chosen at run time (crd_set_exchange_period; bench.py rehearses 8, 10 and 16 on the machine at hand): 8 against 4 measured
// 0.9-1.6 % faster on 1024- to 4096-row slabs of an 8192-column grid on a world-size-1 ring; 16 halves the fixed cost per step of
// a cycle again (two small launches, two cross-stream waits) for 3 % more redundant rows on a 1024-row share.
constexpr int kMinExchangeEvery = 3;  // the multi-slab fused stepper splits the first and the last step of a cycle
constexpr int kDefaultExchangeEvery = 8;
constexpr int kMaxExchangeEvery = 16;
constexpr int kTallSlabExchangeEvery = 10;  // the default where every slab has 256 rows or more (crd_create)
// Ghost rows kept above and below every slab plane: what the longest cycle exchanges.
constexpr int kGhost = kStepHalo * kMaxExchangeEvery;

// Host-side coefficient tables of the diffusion operator written as
//   du = cA[i] (uE - uW) + cX (uE - 2 uC + uW) + cP[i] (uN - 2 uC + uS)
// (torus: src/FHNmodel_torus.cpp:535-537; flat: cA = 0, cX = D/dx^2, cP = D/dy^2, src/FHNmodel_flat.cpp:489-491).
// The kernels take the theta part on first differences, cA (uE - uW) + cX (uE - 2 uC + uW) = cE (uE - uC) + cWn (uC - uW) with
// cE[i] = cX + cA[i], cWn[i] = cA[i] - cX (crd_device.h: rhs_point).
struct Coefficients {
	std::vector<double> cA;   // nx
	std::vector<double> cP;   // nx
	double cX = 0.0;
	std::vector<double> cE;   // nx
	std::vector<double> cWn;  // nx
};
void build_coefficients(const crd_params &p, const crd_grid &g, Coefficients *out);

// b(j) of src/FHNmodel_torus.cpp:623-632 for global rows [j0, j1).
void build_beta_rows(const crd_params &p, const crd_grid &g, int64_t j0, int64_t j1, std::vector<double> *out);

bool validate_params(const crd_params &p, std::string *why);

// crd_trace.cpp: roctx ranges (no-ops unless a profiler listens)
void trace_push(const char *name);
void trace_pop();
struct TraceRange {
	explicit TraceRange(const char *name) { trace_push(name); }
	~TraceRange() { trace_pop(); }
	TraceRange(const TraceRange &) = delete;
	TraceRange &operator=(const TraceRange &) = delete;
};

const char *model_name(int model);      // "FHNmodel" / "GoldbeterModel"
const char *surface_name(int surface);  // "torus" / "flat"
const char *var_name(int model, int var);

}  // namespace crd
