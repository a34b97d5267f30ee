// crd_steppers.cpp -- the time-stepping drivers behind the C ABI: classical RK4 with the staged or the fused kernels on one
// slab or on the slabs of a run (deep-halo exchange cycles), and the error-controlled RK4(3) integrator.  Host code only.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "crd_ctx.h"

namespace crd {

namespace {

struct StagePlan {
	int in, out;
	double c;  // stage time = t + c dt
};
const StagePlan kStages[4] = {{crd_ctx::Y, crd_ctx::SA, 0.0}, {crd_ctx::SA, crd_ctx::SB, 0.5}, {crd_ctx::SB, crd_ctx::SA, 0.5}, {crd_ctx::SA, crd_ctx::Y, 1.0}};

StageCall make_stage_call(const crd_ctx *c, int stage, double t, double dt)
{
	const StagePlan &sp = kStages[stage - 1];
	StageCall call{};
	call.stage = stage;
	call.dt = dt;
	call.absorb = absorbing(c, t + sp.c * dt) ? 1 : 0;
	call.yin = c->planes(sp.in);
	call.y0 = c->planes(crd_ctx::Y);
	call.acc = c->planes(crd_ctx::ACC);
	call.yout = c->planes(sp.out);
	return call;
}

// Single slab: four launches per step, phi wrap inside the kernel.
int staged_step_self(crd_ctx *c, double t, double dt, hipEvent_t *k_begin, hipEvent_t *k_end)
{
	for (int stage = 1; stage <= 4; stage++) {
		const StageCall call = make_stage_call(c, stage, t, dt);
		if (stage == 2 && k_begin) HIP_TRY(c, hipEventRecord(*k_begin, c->compute));
		HIP_TRY(c, launch_stage(c->p.precision, c->desc, call, 0, c->nyl, c->compute));
		if (stage == 2 && k_end) HIP_TRY(c, hipEventRecord(*k_end, c->compute));
	}
	return CRD_OK;
}

int fused_step_self(crd_ctx *c, double t, double dt, int src, int dst, hipEvent_t *k_begin, hipEvent_t *k_end, int steps = 1)
{
	FusedCall call = make_fused_call(c, t, dt, src, dst);
	call.steps = steps;  // 2: this one launch takes the state from t to t + 2 dt
	// a sampled launch is timed by events bound to the kernel itself (start / completion), not recorded around it: a record is a barrier
	// packet, and ten of them in a 20-step timed region were 2 % of it
	if (k_begin) call.start_event = *k_begin;
	if (k_end) call.done_event = *k_end;
	HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, 0, c->nyl, 0, 0, c->compute));
	return CRD_OK;
}

// Several slabs (LOCAL group driven by one thread, or this rank's slab under RCCL).  Per stage:
//   compute: [wait halo(in)] boundary rows 0 and nyl-1 -> record edges(out) -> interior rows
//   comm:    wait edges(out) -> exchange one ghost row of out.u -> record halo(out)
// so the exchange of stage s+1's input overlaps stage s's interior sweep.  The step's first input (Y) has its
// halo exchanged at the end of the previous step's stage 4 (or by prime_halo before the first step).
int staged_step_multi(crd_ctx *const *cs, int n, double t, double dt, bool timed_step)
{
	for (int stage = 1; stage <= 4; stage++) {
		const StagePlan &sp = kStages[stage - 1];
		for (int k = 0; k < n; k++) {
			crd_ctx *c = cs[k];
			if (int rc = set_device(c)) return rc;
			const StageCall call = make_stage_call(c, stage, t, dt);
			HIP_TRY(c, hipStreamWaitEvent(c->compute, c->ev_halo, 0));
			HIP_TRY(c, launch_stage(c->p.precision, c->desc, call, 0, 1, c->compute));
			HIP_TRY(c, launch_stage(c->p.precision, c->desc, call, c->nyl - 1, c->nyl, c->compute));
			HIP_TRY(c, hipEventRecord(c->ev_edges, c->compute));
		}
		// start moving the edge rows of `out` while the interiors run
		if (int rc = exchange_stage_input(cs, n, sp.out, 1, false)) return rc;
		for (int k = 0; k < n; k++) {
			crd_ctx *c = cs[k];
			if (int rc = set_device(c)) return rc;
			const StageCall call = make_stage_call(c, stage, t, dt);
			const bool timed = timed_step && stage == 2 && !c->ev_k.empty();
			if (timed) c->timed_steps = 1;
			if (timed) HIP_TRY(c, hipEventRecord(c->ev_k[0], c->compute));
			HIP_TRY(c, launch_stage(c->p.precision, c->desc, call, 1, c->nyl - 1, c->compute));
			if (timed) HIP_TRY(c, hipEventRecord(c->ev_k[1], c->compute));
		}
	}
	return CRD_OK;
}

// 2-D blocks (theta split: a LOCAL group driven by one thread).  Per stage: [wait halo(in)] all rows of the block -> pack the
// output's edge columns -> record edges; then every block pulls one ghost row / column strip of the output from its neighbours.
// No boundary / interior overlap here: the layout exists to reproduce the reference's `-np 4` file for file, the fast layout on
// one node is phi-slabs.
int staged_step_blocks(crd_ctx *const *cs, int n, double t, double dt)
{
	for (int stage = 1; stage <= 4; stage++) {
		const StagePlan &sp = kStages[stage - 1];
		for (int k = 0; k < n; k++) {
			crd_ctx *c = cs[k];
			if (int rc = set_device(c)) return rc;
			StageCall call = make_stage_call(c, stage, t, dt);
			call.gcol_w = c->gcol[sp.in][0];
			call.gcol_e = c->gcol[sp.in][1];
			HIP_TRY(c, hipStreamWaitEvent(c->compute, c->ev_halo, 0));
			HIP_TRY(c, launch_stage(c->p.precision, c->desc, call, 0, c->nyl, c->compute));
			if (c->d0 > 1)
				HIP_TRY(c, launch_plane_cols_extract(c->p.precision, c->plane[sp.out][0], c->ecol[sp.out][0], c->ecol[sp.out][1], c->nx, c->nyl, c->compute));
			HIP_TRY(c, hipEventRecord(c->ev_edges, c->compute));
		}
		if (int rc = exchange_block_input(cs, n, sp.out)) return rc;
	}
	return CRD_OK;
}

// Fused stepper on several slabs: ONE exchange every E = exchange_every steps (crd_set_exchange_period; default 8), G = 4 E ghost
// rows of both fields.  Step q of a cycle (q = 0 right after an exchange) produces rows [-e, nyl + e) with e = 4 (E-1-q): the
// still-valid part of the ghost region is recomputed redundantly (same kernel, same inputs, so bit-identical to what the owning
// slab computes) instead of being communicated.  The last step of a cycle (e = 0) is split
//   compute: edge bands [0, B) and [nyl-B, nyl) in one small launch -> record edges        (B = max(32, G): every row the exchange sends)
//   comm:    wait edges -> exchange G rows of u and v with the ring neighbours -> record halo
//   compute: interior [B, nyl-B) (needs neither ghost rows nor the bands)
// and so is the first step of the next cycle
//   compute: rows [4, nyl-4), which read owned rows only, straight after the interior sweep
//   compute: [wait halo] rows [-e, 4) and [nyl-4, nyl+e), the ones that read ghost rows, in one small launch (e = G - 4)
// so the exchange has two sweeps to hide under (three with crd_set_halo_slack(ctx, 2)).  Per step that is 1 + 2 / E launches and
// 1 / E of an RCCL group on the host, against 3 launches + 1 group for a per-step exchange.
inline int cycle_steps(const crd_ctx *c) { return c->exchange_every; }
inline int cycle_ghost(const crd_ctx *c) { return kStepHalo * c->exchange_every; }
inline int cycle_band(const crd_ctx *c) { return std::max(32, cycle_ghost(c)); }

// The compute stream's wait for the halo of the exchange just made (and, in a LOCAL group, for the neighbours to have pulled
// theirs out of this context's planes), with the diagnostics' event pair around it.
int wait_for_halo(crd_ctx *c)
{
	// (diagnostics: how long does the compute stream stand at this wait?  Zero when the exchange hid under the sweeps)
	const bool diag = c->diag_active && 4 * c->diag_waits + 1 < (int)c->ev_diag.size();
	if (diag) HIP_TRY(c, hipEventRecord(c->ev_diag[(size_t)(4 * c->diag_waits)], c->compute));
	HIP_TRY(c, hipStreamWaitEvent(c->compute, c->ev_halo, 0));
	if (diag) HIP_TRY(c, hipEventRecord(c->ev_diag[(size_t)(4 * c->diag_waits++ + 1)], c->compute));
	if (c->halo == CRD_HALO_LOCAL) {
		// LOCAL halos are PULLED by the neighbours from this context's planes: the next launch that overwrites those rows must not
		// start before both neighbours have finished copying them
		HIP_TRY(c, hipStreamWaitEvent(c->compute, c->group[(size_t)((c->slab + c->n_slabs - 1) % c->n_slabs)]->ev_halo, 0));
		HIP_TRY(c, hipStreamWaitEvent(c->compute, c->group[(size_t)((c->slab + 1) % c->n_slabs)]->ev_halo, 0));
	}
	return CRD_OK;
}

// One launch of the cycle on every context of the call: `nsub` steps (1, or 2 with a two-steps-per-launch plan: the pair must not
// straddle an exchange, q + nsub <= E) from cycle position q.  H = 4 nsub rows are consumed beyond the rows produced.
int fused_launch_multi(crd_ctx *const *cs, int n, double t, double dt, int src, int dst, int q, int nsub, bool timed_step, bool last_launch_of_call, int max_nsub = 2)
{
	const int E = cycle_steps(cs[0]), G = cycle_ghost(cs[0]), B = cycle_band(cs[0]);
	const int H = kStepHalo * nsub;
	const int ext = kStepHalo * (E - q - nsub);  // ghost rows still valid once this launch has run
	auto call_of = [&](crd_ctx *c, double tt, int from, int to, int steps) {
		FusedCall call = make_fused_call(c, tt, dt, from, to);
		call.steps = steps;
		return call;
	};
	if (q + nsub < E) {
		for (int k = 0; k < n; k++) {
			crd_ctx *c = cs[k];
			if (int rc = set_device(c)) return rc;
			const FusedCall call = call_of(c, t, src, dst, nsub);
			const bool timed = timed_step && !c->ev_k.empty();
			if (c->ghost_deferred) {
				// Halo slack 2 (crd_set_halo_slack): the exchange gets a THIRD sweep to land under.  The cycle's first launch has only
				// produced its rows that read owned rows; this one does the same -- rows [B, nyl - B) only, B = the band the
				// exchange sends from / the neighbours pull from, which nobody may overwrite before the exchange is through -- and
				// only then the compute stream waits, finishes the first launch (the rows that read ghost rows) and this one (its edges).
				const int n0 = c->deferred_nsub, H0 = kStepHalo * n0, ext0 = kStepHalo * (E - n0);
				const FusedCall call0 = call_of(c, c->deferred_t, dst, src, n0);  // the first launch read this launch's output plane and wrote its input plane
				HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, B, c->nyl - B, 0, 0, c->compute));
				if (int rc = wait_for_halo(c)) return rc;
				HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call0, -ext0, H0, c->nyl - H0, c->nyl + ext0, c->compute));
				HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, -ext, B, c->nyl - B, c->nyl + ext, c->compute));
				c->ghost_deferred = false;
				continue;
			}
			if (q > 0) {
				// (nothing to record: the next launch runs on the same stream)
				if (timed) HIP_TRY(c, hipEventRecord(c->ev_k[0], c->compute));
				HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, -ext, c->nyl + ext, 0, 0, c->compute));
				if (timed) HIP_TRY(c, hipEventRecord(c->ev_k[1], c->compute));
				if (timed) c->timed_rows = c->nyl + 2 * ext;  // what crd_dominant_kernel_rows reports for this launch
				if (timed) c->timed_steps = nsub;
				continue;
			}
			// First launch after an exchange.  Output rows [H, nyl - H) read owned rows only, so they are launched straight behind
			// the interior sweep of the previous launch: the exchange gets this sweep as extra time to land.
			const bool split = c->nyl >= 4 * B;
			if (split) HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, H, c->nyl - H, 0, 0, c->compute));
			if (split && c->halo_slack >= 2 && !last_launch_of_call && q + nsub + max_nsub < E) {  // (whatever the next launch takes -- up to max_nsub steps --, it is not the cycle's last)
				c->ghost_deferred = true;  // the wait and the rows that read ghost rows follow behind the NEXT launch's owned-only rows
				c->deferred_t = t;
				c->deferred_nsub = nsub;
				continue;
			}
			if (int rc = wait_for_halo(c)) return rc;
			// the rows that read ghost rows: [-ext, H) and [nyl - H, nyl + ext) in one launch
			if (split) HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, -ext, H, c->nyl - H, c->nyl + ext, c->compute));
			else HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, -ext, c->nyl + ext, 0, 0, c->compute));
		}
		return CRD_OK;
	}
	// last launch of the cycle: edge bands, exchange released behind them, interior under the exchange
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (int rc = set_device(c)) return rc;
		FusedCall call = call_of(c, t, src, dst, nsub);
		call.done_event = c->ev_edges;  // (set by the launch's own completion: no record, no bubble in front of the interior launch)
		if (c->nyl >= 4 * B) HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, 0, B, c->nyl - B, c->nyl, c->compute));
		else HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, 0, c->nyl, 0, 0, c->compute));
	}
	if (int rc = exchange_stage_input(cs, n, dst, G, true)) return rc;
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (int rc = set_device(c)) return rc;
		if (c->nyl >= 4 * B) {
			const FusedCall call = call_of(c, t, src, dst, nsub);
			HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, B, c->nyl - B, 0, 0, c->compute));
		}
	}
	return CRD_OK;
}

// May steps s and s + 1 of a call go out as ONE two-step launch?  The kernel takes the second step's absorbing-row flags from
// t + dt (make_fused_call); the stepping loops form that step's time as t0 + (s + 1) dt, which may differ by a rounding -- which
// matters only where a stage time lands within an ulp of tBoundary.  Such a pair is stepped singly, so that paired and unpaired
// stepping take the same decisions.
bool pair_is_exact(const crd_ctx *c, double t0, int64_t s, double dt)
{
	const double t = t0 + (double)s * dt, t2 = t0 + (double)(s + 1) * dt, cs4[4] = {0.0, 0.5, 0.5, 1.0};
	for (int k = 0; k < 4; k++)
		if (absorbing(c, t2 + cs4[k] * dt) != absorbing(c, (t + dt) + cs4[k] * dt)) return false;
	return true;
}
// Does any stage of the three steps from t on have the absorbing rows on?  (The fp32 three-step kernel has no instantiation with the
// selects -- it would not fit the registers: 256 + scratch --; such triples are stepped as a pair and a single step.)
bool triple_absorbs(const crd_ctx *c, double t, double dt)
{
	const double ts[3] = {t, t + dt, (t + dt) + dt}, cs4[4] = {0.0, 0.5, 0.5, 1.0};
	for (double tk : ts)
		for (int k = 0; k < 4; k++)
			if (absorbing(c, tk + cs4[k] * dt)) return true;
	return false;
}
// ... and steps s, s + 1, s + 2 as one three-step launch: the third step's flags come from (t + dt) + dt.
bool triple_is_exact(const crd_ctx *c, double t0, int64_t s, double dt)
{
	if (!pair_is_exact(c, t0, s, dt)) return false;
	const double t = t0 + (double)s * dt, t3 = t0 + (double)(s + 2) * dt, cs4[4] = {0.0, 0.5, 0.5, 1.0};
	for (int k = 0; k < 4; k++)
		if (absorbing(c, t3 + cs4[k] * dt) != absorbing(c, ((t + dt) + dt) + cs4[k] * dt)) return false;
	return true;
}

// RCCL runs: one decision for the whole ring on where in the exchange cycle the call starts (see crd_ctx::agree_dev).  begin_
// enqueues the reduction on the comm stream and returns; finish_ waits for it.  Every rank of the run makes both calls in every
// fused stepping call, before any other RCCL operation of that call.
int begin_cycle_agreement(crd_ctx *c, int mine)
{
	if (!c->agree_dev) {
		HIP_TRY(c, hipMalloc((void **)&c->agree_dev, 2 * sizeof(double)));
		HIP_TRY(c, hipHostMalloc((void **)&c->agree_host, 4 * sizeof(double), hipHostMallocDefault));
		HIP_TRY(c, hipEventCreateWithFlags(&c->ev_agree, hipEventDisableTiming));
	}
	if (crd_cycle_vote(mine, c->agree_host) != CRD_OK) return fail(c, CRD_ESTATE, "exchange-cycle position out of range");
	HIP_TRY(c, hipMemcpyAsync(c->agree_dev, c->agree_host, 2 * sizeof(double), hipMemcpyHostToDevice, c->comm));
	NCCL_TRY(c, g_rccl.AllReduce(c->agree_dev, c->agree_dev, 2, ncclDouble, ncclMin, c->nccl, c->comm));
	HIP_TRY(c, hipMemcpyAsync(c->agree_host + 2, c->agree_dev, 2 * sizeof(double), hipMemcpyDeviceToHost, c->comm));
	HIP_TRY(c, hipEventRecord(c->ev_agree, c->comm));
	return CRD_OK;
}

int finish_cycle_agreement(crd_ctx *c, int *agreed)
{
	HIP_TRY(c, hipEventSynchronize(c->ev_agree));
	*agreed = crd_cycle_agreed(c->agree_host + 2);
	return CRD_OK;
}

constexpr int kMaxTimedLaunches = 64;
// Step of an exchange cycle whose single full-slab launch is the one timed in a multi-slab fused run (step 0 is split in two).
constexpr int kTimedCycleStep = 1;

int ensure_timing_events(crd_ctx *c)
{
	while ((int)c->ev_k.size() < 2 * kMaxTimedLaunches) {
		hipEvent_t e;
		HIP_TRY(c, hipEventCreate(&e));
		c->ev_k.push_back(e);
	}
	return CRD_OK;
}

}  // namespace

void decide_cycle_start(crd_ctx *const *all, int n_all)
{
	int q0 = all[0]->cycle_pos;
	for (int k = 0; k < n_all; k++)
		if (all[k]->cycle_pos != q0 || all[k]->cycle_pos < 0) q0 = -1;
	for (int k = 0; k < n_all; k++) all[k]->cycle_start = q0;
}

FusedCall make_fused_call(const crd_ctx *c, double t, double dt, int src, int dst)
{
	FusedCall call{};
	call.dt = dt;
	const double cs[4] = {0.0, 0.5, 0.5, 1.0};
	for (int k = 0; k < 4; k++) call.absorb[k] = absorbing(c, t + cs[k] * dt) ? 1 : 0;
	for (int k = 0; k < 4; k++) call.absorb2[k] = absorbing(c, (t + dt) + cs[k] * dt) ? 1 : 0;  // the step after this one, as the stepping loop forms its time
	for (int k = 0; k < 4; k++) call.absorb3[k] = absorbing(c, ((t + dt) + dt) + cs[k] * dt) ? 1 : 0;  // ... and the one after that (three-step launches)
	call.absorb[4] = call.absorb[3];  // (the embedded pairs' fifth stage: set by the adaptive integrator)
	call.y0 = c->planes(src);
	call.yout = c->planes(dst);
	call.plan = const_cast<FusedPlan *>(&c->plan);
	// (ACC is the staged stepper's accumulator and a temporary of arkHin / the Hermite output: free whenever a one-launch step runs)
	if (src != crd_ctx::ACC && dst != crd_ctx::ACC) call.tune_scratch = c->planes(crd_ctx::ACC);
	return call;
}

// The stepping loop shared by crd_step_rk4 / crd_step_rk4_timed / the group call.
int run_steps(crd_ctx *const *cs, int n, double t0, double dt, int64_t nsteps, int *timed_launches)
{
	crd_ctx *lead = cs[0];
	if (nsteps < 0 || !(dt > 0.0) || !std::isfinite(t0)) return fail(lead, CRD_EINVAL, "bad t0 / dt / nsteps");
	const int stepper = resolve_stepper(lead);
	if (stepper < 0) return fail(lead, CRD_EINVAL, "fused stepper not available for this configuration");
	for (int k = 0; k < n; k++) cs[k]->dense.pending = cs[k]->ark.live = false;  // stepping on from the state handed back, not from the integrator's internal one
	if (nsteps > 0)
		for (int k = 0; k < n; k++) cs[k]->last_step_t = t0 + (double)(nsteps - 1) * dt, cs[k]->last_step_dt = dt;
	if (stepper != CRD_STEPPER_FUSED)
		for (int k = 0; k < n; k++) cs[k]->cycle_pos = -1;  // the staged stepper keeps one ghost row of one field current, not the deep halo
	for (int k = 0; k < n; k++)
		if (resolve_stepper(cs[k]) != stepper) return fail(lead, CRD_EINVAL, "contexts of one run disagree on the stepper");
	int timed = 0;
	const bool single = (lead->halo == CRD_HALO_SELF);
	if (lead->d0 > 1) {
		// theta-blocks: staged kernels, every block of the run in this one call
		if (lead->halo != CRD_HALO_LOCAL || n != lead->n_slabs) return fail(lead, CRD_ESTATE, "theta-blocks step as a LOCAL group holding every block of the run");
		for (int k = 0; k < n; k++) cs[k]->cycle_pos = -1;
		if (nsteps > 0)
			if (int rc = prime_block_halo(cs, n, crd_ctx::Y)) return rc;
		for (int64_t s = 0; s < nsteps; s++)
			if (int rc = staged_step_blocks(cs, n, t0 + (double)s * dt, dt)) return rc;
		for (int k = 0; k < n; k++) {
			if (int rc = set_device(cs[k])) return rc;
			HIP_TRY(cs[k], hipStreamWaitEvent(cs[k]->compute, cs[k]->ev_halo, 0));
		}
		if (timed_launches) *timed_launches = 0;
		return CRD_OK;
	}
	if (single) {
		crd_ctx *c = lead;
		if (int rc = set_device(c)) return rc;
		int cur = crd_ctx::Y;
		for (int64_t s = 0; s < nsteps;) {
			const double t = t0 + (double)s * dt;
			hipEvent_t *kb = nullptr, *ke = nullptr;
			// The plan may say "two steps per launch" (measured, or pinned): pairs while two steps are left, a single-step launch
			// for an odd last one (pair_is_exact: see there).
			const int per_launch = (stepper == CRD_STEPPER_FUSED && c->plan.tuned) ? fused_steps_supported(c->p.precision, c->desc, c->plan.steps) : 1;
			const bool pairing = per_launch >= 2;
			// (a three-step plan: triples while three steps are left, then a pair or a single step)
			const bool triple = per_launch == 3 && s + 3 <= nsteps && triple_is_exact(c, t0, s, dt) && (c->p.precision == CRD_PRECISION_F64 || !triple_absorbs(c, t, dt));
			int rc, took = triple ? 3 : (pairing && s + 2 <= nsteps && pair_is_exact(c, t0, s, dt)) ? 2 : 1;
			// every fourth step at most: an event pair around EVERY launch of a short run would sit inside the region being timed.  Only
			// launches of the plan's own kind are timed (a pairing plan's odd last step, or a pair stepped singly, goes out as a one-step
			// launch: another kernel, which must not enter the average; runs too short to hold a pair time what there is).
			const int64_t want = std::min<int64_t>(kMaxTimedLaunches, std::max<int64_t>(1, nsteps / 4));
			const bool triples_here = per_launch == 3 && nsteps >= 3 && (c->p.precision == CRD_PRECISION_F64 || !triple_absorbs(c, t0, dt));  // (the kind of launch the call starts with)
			const int kind = triples_here ? 3 : (pairing && nsteps >= 2) ? 2 : 1;
			if (timed_launches && timed < want && took == kind && (s * want / std::max<int64_t>(nsteps, 1)) >= timed && ((s % 4) >= 1 || nsteps < 4)) {
				kb = &c->ev_k[(size_t)(2 * timed)];
				ke = &c->ev_k[(size_t)(2 * timed + 1)];
				timed++;
				c->timed_steps = took;
			}
			if (stepper == CRD_STEPPER_STAGED) {
				rc = staged_step_self(c, t, dt, kb, ke);
			} else {
				const int dst = (cur == crd_ctx::Y) ? crd_ctx::SA : crd_ctx::Y;
				rc = fused_step_self(c, t, dt, cur, dst, kb, ke, took);
				cur = dst;
			}
			if (rc) return rc;
			s += took;
		}
		if (cur != crd_ctx::Y) {  // odd number of fused steps: the result sits in SA; swap the plane pointers
			std::swap(c->plane[crd_ctx::Y][0], c->plane[crd_ctx::SA][0]);
			std::swap(c->plane[crd_ctx::Y][1], c->plane[crd_ctx::SA][1]);
		}
	} else {
		const bool fused = (stepper == CRD_STEPPER_FUSED);
		const int E = cycle_steps(lead);
		for (int k = 0; k < n; k++)
			if (cs[k]->exchange_every != E) return fail(lead, CRD_EINVAL, "contexts of one run disagree on the exchange period");
		// Where in the exchange cycle does the resident state stand?  If the previous call left it mid-cycle (or right after an
		// exchange) the ghost rows are as good as they need to be and this call carries on from there; otherwise -- new state,
		// staged stepper, slabs that disagree -- it starts with an exchange.  (A 20-step call on an 8192 x 1024 slab is 1.2 ms:
		// an exposed exchange in front of it is several per cent.)
		int q0 = fused ? lead->cycle_start : -1;  // (decide_cycle_start: one decision for every slab of the run)
		// Under RCCL that decision has to be the RING's, not this rank's: a rank that alone holds a new state (an upload on that
		// rank only, a failed call) would prime its halo while its neighbours carry on, and the ring's send / receive sequences
		// would no longer pair.  So the ranks reduce their positions first.  The reduction (and the host's wait for it) hides
		// under the call's first step wherever that step involves no exchange of its own (q0 = 0 .. E - 4: the step is issued
		// speculatively into the scratch planes and simply issued again, behind an exchange, if the ring turns out to disagree).
		const bool ring = fused && n == 1 && lead->halo == CRD_HALO_RCCL;
		bool agreement_pending = false;
		if (nsteps > 0) {
			for (int k = 0; k < n; k++) cs[k]->cycle_pos = -1;  // until this call has gone through
			for (int k = 0; k < n; k++) cs[k]->ghost_deferred = false;  // (a call always finishes what its last step deferred)
			if (ring) {
				if (int rc = begin_cycle_agreement(lead, q0)) return rc;
				agreement_pending = true;
				if (q0 < 0 || q0 >= E - 3) {  // nothing to issue ahead of the answer (the call's first launch may be the cycle's last: a pair, or -- fp32 -- a triple)
					if (int rc = finish_cycle_agreement(lead, &q0)) return rc;
					agreement_pending = false;
				}
			}
			if (q0 < 0) {
				if (ring && lead->cycle_start >= 0) lead->agreement_restarts++;
				if (int rc = prime_halo(cs, n, crd_ctx::Y, fused ? cycle_ghost(lead) : 1, fused)) return rc;
				q0 = 0;
			}
		}
		int cur = crd_ctx::Y;
		// Two steps per launch where the lead context's plan says so (measured, or pinned): pairs that do not straddle an exchange.
		// Where the exchanges fall in the step sequence does not depend on the pairing, so the ranks of a ring / the threads of a group
		// may pair differently (each by its own plan) and still meet at the same collectives.
		// Three where the three-step kernel takes slabs (fp32: a strip per wavefront; fp64's block strip measured nothing on a rank's
		// share and keeps pairs: fused_steps_supported), by the same rule -- a triple never straddles an exchange either.
		bool pairs = fused && lead->plan.tuned && lead->plan.steps >= 2;
		for (int k = 0; k < n; k++) pairs = pairs && fused_two_steps_supported(cs[k]->desc);
		bool triples = pairs && lead->plan.steps == 3;
		for (int k = 0; k < n; k++) triples = triples && fused_steps_supported(cs[k]->p.precision, cs[k]->desc, 3) == 3;
		for (int64_t s = 0; s < nsteps;) {
			const int q = (int)((s + q0) % E);
			const double t_s = t0 + (double)s * dt;
			// (a launch is the cycle's first -- which waits for the halo -- or its last -- which sends the next --, never both: at E = 3 a
			// triple from q = 0 would be; found by tests/long_oracle_sweep.py)
			const int nsub = (triples && s + 3 <= nsteps && q + 3 <= E && (q > 0 || E > 3) && triple_is_exact(lead, t0, s, dt) && !triple_absorbs(lead, t_s, dt))
			                     ? 3
			                     : (pairs && s + 2 <= nsteps && q + 2 <= E && pair_is_exact(lead, t0, s, dt)) ? 2 : 1;
			// time one launch of the dominant kernel mid-run (fused: a launch of the cycle that is one full-height sweep, i.e. neither
			// the split first nor the split last one, nor the one that finishes a first whose ghost readers were deferred -- whatever
			// the two launches take, one step or two)
			bool finishes_deferred = false;
			for (int k = 0; k < n; k++) finishes_deferred = finishes_deferred || cs[k]->ghost_deferred;
			const bool middle = q >= 1 && q + nsub < E && !finishes_deferred;
			const bool timed_step = timed_launches && !timed && (fused ? (middle && (s >= nsteps / 2 || s + E >= nsteps)) : s >= nsteps / 2);
			const double t = t0 + (double)s * dt;
			if (fused) {
				const int dst = (cur == crd_ctx::Y) ? crd_ctx::SA : crd_ctx::Y;
				if (int rc = fused_launch_multi(cs, n, t, dt, cur, dst, q, nsub, timed_step, s + nsub == nsteps, triples ? 3 : 2)) return rc;
				cur = dst;
			} else if (int rc = staged_step_multi(cs, n, t, dt, timed_step)) {
				return rc;
			}
			if (timed_step) timed = 1;
			s += nsub;
			if (agreement_pending) {  // the first launch is on its way: now hear what the ring says
				int agreed = -1;
				if (int rc = finish_cycle_agreement(lead, &agreed)) return rc;
				agreement_pending = false;
				if (agreed != q0) {
					// Some rank holds a new state: every rank starts afresh.  The launch just issued wrote the scratch planes only (plane Y
					// is untouched) and is overwritten by the one issued again below, in stream order.
					lead->agreement_restarts++;
					for (int k = 0; k < n; k++) cs[k]->ghost_deferred = false;  // (the launch issued ahead of the answer is void, and so is what it deferred)
					if (int rc = prime_halo(cs, n, crd_ctx::Y, cycle_ghost(lead), true)) return rc;
					q0 = 0;
					cur = crd_ctx::Y;
					timed = 0;
					s = 0;
				}
			}
		}
		// (several issuing threads: the neighbours read this thread's plane pointers while they enqueue their pulls)
		if (lead->bar && !lead->bar->wait()) return fail(lead, CRD_ESTATE, "another slab's issuing thread failed");
		if (cur != crd_ctx::Y)
			for (int k = 0; k < n; k++) {
				std::swap(cs[k]->plane[crd_ctx::Y][0], cs[k]->plane[crd_ctx::SA][0]);
				std::swap(cs[k]->plane[crd_ctx::Y][1], cs[k]->plane[crd_ctx::SA][1]);
			}
		// leave the compute stream of every context ordered behind its last band launch and exchange
		for (int k = 0; k < n; k++) {
			if (int rc = set_device(cs[k])) return rc;
			HIP_TRY(cs[k], hipStreamWaitEvent(cs[k]->compute, cs[k]->ev_edges, 0));
			HIP_TRY(cs[k], hipStreamWaitEvent(cs[k]->compute, cs[k]->ev_halo, 0));
		}
		if (fused && nsteps > 0)
			for (int k = 0; k < n; k++) cs[k]->cycle_pos = (int)((q0 + nsteps) % E);
	}
	if (timed_launches) *timed_launches = timed;
	return CRD_OK;
}

}  // namespace crd

using namespace crd;

extern "C" {

int crd_set_stepper(crd_ctx *c, int stepper)
{
	if (!c) return CRD_EINVAL;
	if (stepper != CRD_STEPPER_AUTO && stepper != CRD_STEPPER_STAGED && stepper != CRD_STEPPER_FUSED) return fail(c, CRD_EINVAL, "unknown stepper");
	const int before = c->stepper;
	c->stepper = stepper;
	if (stepper == CRD_STEPPER_FUSED && resolve_stepper(c) < 0) {
		c->stepper = before;
		return fail(c, CRD_EINVAL, "fused stepper not available for this configuration (slab too short)");
	}
	return CRD_OK;
}

int crd_step_rk4(crd_ctx *c, double t0, double dt, int64_t nsteps)
{
	if (!c) return CRD_EINVAL;
	if (c->halo < 0) return fail(c, CRD_ESTATE, "multi-slab context is not wired (crd_comm_attach_local / crd_comm_init_rccl)");
	if (c->halo == CRD_HALO_LOCAL) return fail(c, CRD_ESTATE, "LOCAL groups step through crd_group_step_rk4");
	crd_ctx *one[1] = {c};
	decide_cycle_start(one, 1);
	TraceRange range("crd_step_rk4");
	return run_steps(one, 1, t0, dt, nsteps, nullptr);
}

// LOCAL groups spread over several devices get one issuing thread per device: a single thread that enqueues every launch, event
// and peer copy of every GPU becomes the bottleneck once a GPU's share of a step is short (an 8192 x 1024 slab steps in ~60 us,
// a thread needs a good part of that to issue one GPU's work).  The threads run the same step loop on their own contexts and meet
// at the group's rendezvous around every halo exchange.  crd_group_set_threads(ctxs, n, k) forces k threads (also on one device:
// how the tests exercise the rendezvous on a one-GPU box), k = 1 the single-thread path.
static int group_step(crd_ctx *const *ctxs, int n, double t0, double dt, int64_t nsteps, bool timed)
{
	if (int rc = check_group(ctxs, n)) return rc;
	if (n > 1)
		for (int k = 0; k < n; k++)
			if (ctxs[k]->halo != CRD_HALO_LOCAL) return fail(ctxs[0], CRD_ESTATE, "group is not attached");
	// contiguous runs of slabs per thread: by device, or by the knob
	std::vector<int> first{0};
	const int forced = ctxs[0]->group_threads;  // crd_group_set_threads: 0 = one per device
	if (forced > 1) {
		const int t = std::min(forced, n);
		for (int q = 1; q < t; q++) first.push_back((int)((long)n * q / t));
	} else if (forced == 0) {
		for (int k = 1; k < n; k++)
			if (ctxs[k]->device != ctxs[k - 1]->device) first.push_back(k);
	}
	if (ctxs[0]->d0 > 1) first.assign(1, 0);  // theta-blocks: one issuing thread
	const int nthreads = (int)first.size();
	decide_cycle_start(ctxs, n);
	TraceRange range("crd_group_step_rk4");
	std::vector<int> timed_of((size_t)nthreads, 0);
	if (nthreads <= 1 || nsteps <= 0) return run_steps(ctxs, n, t0, dt, nsteps, timed ? &timed_of[0] : nullptr);
	first.push_back(n);
	GroupBarrier bar;
	bar.members = nthreads;
	for (int k = 0; k < n; k++) ctxs[k]->bar = &bar;
	std::vector<int> rcs((size_t)nthreads, CRD_OK);
	auto work = [&](int q) {
		rcs[(size_t)q] = run_steps(ctxs + first[(size_t)q], first[(size_t)q + 1] - first[(size_t)q], t0, dt, nsteps, timed ? &timed_of[(size_t)q] : nullptr);
		if (rcs[(size_t)q] != CRD_OK) bar.leave_failed();
	};
	std::vector<std::thread> pool;
	for (int q = 1; q < nthreads; q++) pool.emplace_back(work, q);
	work(0);
	for (auto &th : pool) th.join();
	for (int k = 0; k < n; k++) ctxs[k]->bar = nullptr;
	for (int q = 0; q < nthreads; q++)
		if (rcs[(size_t)q] != CRD_OK) {
			if (q > 0 && ctxs[0]->err.empty()) ctxs[0]->err = ctxs[first[(size_t)q]]->err;  // the caller asks the first context for the text
			return rcs[(size_t)q];
		}
	return CRD_OK;
}

int crd_adaptive_defaults(crd_adaptive_options *o)
{
	if (!o) return CRD_EINVAL;
	o->rtol = 1.e-5;   // src/FHNmodel_torus.cpp:197
	o->atol = 1.e-10;  // :198
	o->h0 = 0.0;
	o->safety = 0.96;
	o->bias = 1.5;
	o->growth = 20.0;
	o->shrink = 0.1;
	o->max_steps = 200000;  // :372
	o->h_max = 0.0;         // automatic: the diffusion-stability bound
	o->dense_output = 1;    // ARK_NORMAL, :423
	o->method = CRD_ADAPT_ARKODE;  // the reference's integrator: ARKodeInit(mem, f, NULL, ...) = default explicit table, order 4 (:362)
	return CRD_OK;
}

// f(t, plane src) -> plane dst on every slab (bare RHS with the stage kernel; multi-slab: one ghost row of var0 first).
static int rhs_on_planes(crd_ctx *const *cs, int n, double t, int src, int dst)
{
	const bool multi = cs[0]->halo != CRD_HALO_SELF;
	if (multi)
		if (int rc = prime_halo(cs, n, src, 1, false)) return rc;
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (int rc = set_device(c)) return rc;
		StageCall call{};
		call.stage = 0;
		call.absorb = absorbing(c, t) ? 1 : 0;
		call.yin = c->planes(src);
		call.yout = c->planes(dst);
		if (multi) HIP_TRY(c, hipStreamWaitEvent(c->compute, c->ev_halo, 0));
		HIP_TRY(c, launch_stage(c->p.precision, c->desc, call, 0, c->nyl, c->compute));
	}
	return CRD_OK;
}

// One scalar per slab (left in c->scalar_dev by work already enqueued on the compute streams) combined over the run: the sum or
// the maximum, identical on every rank (RCCL: ncclAllReduce; LOCAL groups: added / compared on the host in slab order).
static int collect_scalar(crd_ctx *const *cs, int n, bool take_max, double *out)
{
	// The producers wrote to scalar_sink(): host memory, or -- RCCL runs -- device memory that is reduced over the ranks first and
	// copied to the page-locked buffer (a copy to pageable memory goes through the runtime's staging buffer: ~10 us per attempt of an
	// error-controlled run on a small grid).  Every slab's work is enqueued before the first one is waited for.
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (c->scalar_sink() == c->scalar_host) continue;
		if (int rc = set_device(c)) return rc;
		if (c->halo == CRD_HALO_RCCL) NCCL_TRY(c, g_rccl.AllReduce(c->scalar_dev, c->scalar_dev, 1, ncclDouble, take_max ? ncclMax : ncclSum, c->nccl, c->compute));
		HIP_TRY(c, hipMemcpyAsync(c->scalar_host, c->scalar_dev, sizeof(double), hipMemcpyDeviceToHost, c->compute));
	}
	double acc = 0.0;
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (int rc = set_device(c)) return rc;
		HIP_TRY(c, hipStreamSynchronize(c->compute));
		const double part = c->scalar_host[0];
		if (take_max) acc = (part > acc || part != part) ? part : acc;
		else acc += part;
	}
	*out = acc;
	return CRD_OK;
}

// ARKode's step-size controller (CRD_ADAPT_ARKODE), restated from its documentation; oracle/arkode_erk.py has the same rules
// with the same names, and the tests compare the two attempt by attempt.
namespace arkode {
constexpr double kK1 = 0.58, kK2 = 0.21, kK3 = 0.1;  // ARK_ADAPT_PID gains
constexpr int kEmbeddingOrder = 3;                   // Zonneveld 5(3)4; "pq = 0": the embedding's order enters the exponents
constexpr double kEtamx1 = 10000.0, kEtamxf = 0.3;   // growth bound of the very first step; bound from the second failure of a step on
constexpr int kSmallNef = 2, kMaxNef = 7;
constexpr double kLbound = 1.0, kUbound = 1.5, kOnePsm = 1.000001, kOneMsm = 0.999999;  // no change of h for lbound <= eta <= ubound
constexpr double kTiny = 1.0e-10, kUround = 2.220446049250313e-16;
constexpr double kH0LbFactor = 100.0, kH0UbFactor = 0.1, kH0Bias = 0.5;
constexpr int kH0Iters = 4;

// arkAdapt with the PID method and no explicit-stability function: eta = h_new / h.  e = (this step's biased error, the previous
// accepted step's, the one before that).
double pid_eta(double h, const double e[3], double etamax, double safety, double etamin, double h_cap)
{
	const double e1 = std::fmax(e[0], kTiny), e2 = std::fmax(e[1], kTiny), e3 = std::fmax(e[2], kTiny);
	double h_acc = h * std::pow(e1, -kK1 / kEmbeddingOrder) * std::pow(e2, kK2 / kEmbeddingOrder) * std::pow(e3, -kK3 / kEmbeddingOrder);
	h_acc *= safety;
	h_acc = std::fmin(std::fabs(h_acc), std::fabs(etamax * h));
	h_acc = std::fmax(std::fabs(h_acc), std::fabs(etamin * h));
	if (std::fabs(h_acc) > std::fabs(h * kLbound * kOneMsm) && std::fabs(h_acc) < std::fabs(h * kUbound * kOnePsm)) h_acc = h;
	double eta = h_acc / h;
	if (std::isfinite(h_cap)) eta /= std::fmax(1.0, std::fabs(h) * eta / h_cap);  // hmax_inv
	return eta;
}
}  // namespace arkode

// arkHin: ARKode's estimate of the first step from y'' along a forward Euler trial step, on the planes: f0 -> SB, trial state ->
// SA, f(trial) -> ACC (all three are scratch on a fresh state).
static int arkode_initial_step(crd_ctx *const *cs, int n, double t0, double tout, int cur, const crd_adaptive_options &o, double n_components, double *h0)
{
	using namespace arkode;
	crd_ctx *lead = cs[0];
	const double tdist = std::fabs(tout - t0), tround = kUround * std::fmax(std::fabs(t0), std::fabs(tout));
	if (tdist < 2.0 * tround) return fail(lead, CRD_EINVAL, "adaptive integration: tout too close to t0 to estimate a first step");
	const double hlb = kH0LbFactor * tround;
	if (int rc = rhs_on_planes(cs, n, t0, cur, crd_ctx::SB)) return rc;
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (int rc = set_device(c)) return rc;
		HIP_TRY(c, launch_hin_bound(c->p.precision, c->planes(cur), c->planes(crd_ctx::SB), c->nx, c->nyl, o.rtol, o.atol, c->scalar_sink(), c->compute));
	}
	double hub_inv = 0.0;
	if (int rc = collect_scalar(cs, n, true, &hub_inv)) return rc;
	double hub = kH0UbFactor * tdist;
	if (hub * hub_inv > 1.0) hub = 1.0 / hub_inv;
	double hg = std::sqrt(hlb * hub);
	if (hub < hlb) {
		*h0 = hg;
		return CRD_OK;
	}
	bool hnew_ok = false;
	double hnew = hg;
	for (int count = 1; count <= kH0Iters; count++) {
		for (int k = 0; k < n; k++) {
			crd_ctx *c = cs[k];
			if (int rc = set_device(c)) return rc;
			HIP_TRY(c, launch_axpy_planes(c->p.precision, c->planes(cur), c->planes(crd_ctx::SB), hg, c->planes(crd_ctx::SA), c->nx, c->nyl, c->compute));
		}
		if (int rc = rhs_on_planes(cs, n, t0 + hg, crd_ctx::SA, crd_ctx::ACC)) return rc;
		for (int k = 0; k < n; k++) {
			crd_ctx *c = cs[k];
			if (int rc = set_device(c)) return rc;
			HIP_TRY(c, launch_ydd_sumsq(c->p.precision, c->planes(cur), c->planes(crd_ctx::SB), c->planes(crd_ctx::ACC), hg, o.rtol, o.atol, c->nx, c->nyl, c->err_partials,
			                            c->scalar_sink(), c->compute));
		}
		double sum = 0.0;
		if (int rc = collect_scalar(cs, n, false, &sum)) return rc;
		const double yddnrm = std::sqrt(sum / n_components);
		if (hnew_ok || count == kH0Iters) {
			hnew = hg;
			break;
		}
		hnew = (yddnrm * hub * hub > 2.0) ? std::sqrt(2.0 / yddnrm) : std::sqrt(hg * hub);
		const double hrat = hnew / hg;
		if (hrat > 0.5 && hrat < 2.0) hnew_ok = true;
		if (count > 1 && hrat > 2.0) {
			hnew = hg;
			hnew_ok = true;
		}
		hg = hnew;
	}
	*h0 = std::fmin(std::fmax(kH0Bias * hnew, hlb), hub);
	return CRD_OK;
}

// Error-controlled integration of all slabs of a run (n = 1: a single-slab or RCCL context; n > 1: a LOCAL group).
static int integrate_adaptive_impl(crd_ctx *const *cs, int n, double t0, double tout, const crd_adaptive_options *opt_in, crd_adaptive_stats *stats)
{
	crd_ctx *lead = cs[0];
	TraceRange range("crd_integrate_adaptive");
	crd_adaptive_options o;
	crd_adaptive_defaults(&o);
	if (opt_in) o = *opt_in;
	if (!(o.rtol >= 0.0) || !(o.atol >= 0.0) || !(o.rtol + o.atol > 0.0) || !(o.safety > 0.0) || !(o.bias > 0.0) || !(o.growth >= 1.0) ||
	    !(o.shrink > 0.0 && o.shrink < 1.0) || o.max_steps < 1 || !(o.h0 >= 0.0) || std::isnan(o.h_max) || !std::isfinite(t0) || !std::isfinite(tout) || tout < t0 ||
	    (o.method != CRD_ADAPT_RK43 && o.method != CRD_ADAPT_ARKODE))
		return fail(lead, CRD_EINVAL, "bad adaptive options / time interval");
	if (lead->d0 > 1) return fail(lead, CRD_EINVAL, "the error-controlled integrators need phi-slabs (theta-blocks step with the staged RK4 only)");
	const bool multi = lead->halo != CRD_HALO_SELF;
	const bool arkode_method = o.method == CRD_ADAPT_ARKODE;
	const bool dense = o.dense_output != 0 || arkode_method;  // ARKode's ARK_NORMAL never shortens a step for an output time
	for (int k = 0; k < n; k++) {
		crd_ctx *c = cs[k];
		if (!fused_step_supported(c->p.precision, c->desc)) return fail(lead, CRD_EINVAL, "slab too small for the fused step kernel");
		if (int rc = set_device(c)) return rc;
		if (!c->err_partials) {  // two slots: an attempt and the one launched ahead of its verdict
			c->err_capacity = std::max(fused_max_items(c->desc), 256);  // (256: the block partials of the initial-step norm)
			HIP_TRY(c, hipMalloc((void **)&c->err_partials, 2 * sizeof(double) * (size_t)c->err_capacity));
			for (int q = 0; q < 2; q++) {
				HIP_TRY(c, hipEventCreateWithFlags(&c->ev_attempt[q], hipEventDisableTiming));
				HIP_TRY(c, hipEventCreateWithFlags(&c->ev_norm[q], hipEventDisableTiming));
			}
		}
		if (dense)
			for (int f = 0; f < 2; f++)
				if (!c->plane[crd_ctx::OUT][f])
					if (int rc = alloc_plane(c, crd_ctx::OUT, f)) return rc;
	}
	for (int k = 0; k < n; k++) cs[k]->cycle_pos = -1;  // the integrator exchanges as it needs; a fixed-step call afterwards starts afresh
	crd_adaptive_stats st{};
	const double h_cap = o.h_max > 0.0 ? o.h_max : (o.h_max == 0.0 ? crd_stable_dt(&lead->p) : INFINITY);
	const double n_components = 2.0 * (double)lead->g.nx * (double)lead->g.ny;  // WRMS norm over the whole grid
	constexpr int kEmbedHalo = kStepHalo + 1;                                     // the fifth stage reads one more row

	// Three state planes take turns as y_n, y_{n+1} and scratch: Y and SA (and OUT with dense output).
	double t = t0;
	int cur = crd_ctx::Y, spare = crd_ctx::SA, third = dense ? crd_ctx::OUT : -1, prev = -1;
	double t_prev = t0;
	// A call continues the previous one when it starts at the output time that one handed back and nothing has replaced the state
	// since (ARKode keeps its memory between ARKode() calls the same way); the ARKode-style controller also needs its own memory.
	bool resume = dense && t0 == lead->dense.t_out && (!arkode_method || lead->ark.live);
	for (int k = 0; k < n; k++)  // (every slab of a group must still hold the step: an upload to one of them alone ends the carry-over for all)
		resume = resume && cs[k]->dense.pending && cs[k]->dense.t_out == lead->dense.t_out && cs[k]->dense.t_np1 == lead->dense.t_np1;
	// Under RCCL that decision has to be the RING's (round-3 advice): crd_state_upload on one rank only is allowed between calls, and a
	// rank that alone started afresh would run arkHin's exchanges and reductions while its neighbours skip them -- the ring's
	// collectives would no longer pair.  One ncclAllReduce(min) of the ranks' votes: every rank resumes, or none does.
	if (lead->halo == CRD_HALO_RCCL && n == 1) {
		int all = 0;
		if (int rc = begin_cycle_agreement(lead, resume ? 0 : -1)) return rc;
		if (int rc = finish_cycle_agreement(lead, &all)) return rc;
		if (resume && all != 0) lead->agreement_restarts++;
		resume = resume && all == 0;
	}
	for (int k = 0; k < n; k++)
		if (!resume) cs[k]->dense.pending = false;
	auto hand_back = [&](double theta, double hstep) -> int {  // interpolant of step prev -> cur at t_prev + theta hstep into plane Y, planes re-labelled
		// roles now: prev = y_n, cur = y_{n+1}, SB = f_n, ACC = f_{n+1}; the interpolant goes to the remaining state plane
		const int out = crd_ctx::Y + crd_ctx::SA + crd_ctx::OUT - prev - cur;
		for (int k = 0; k < n; k++) {
			crd_ctx *c = cs[k];
			if (int rc = set_device(c)) return rc;
			HIP_TRY(c, launch_hermite(c->p.precision, c->planes(prev), c->planes(cur), c->planes(crd_ctx::SB), c->planes(crd_ctx::ACC), c->planes(out), c->nx, c->nyl,
			                          theta, hstep, c->compute));
			void *b_out[2] = {c->plane[out][0], c->plane[out][1]}, *b_n[2] = {c->plane[prev][0], c->plane[prev][1]},
			     *b_np1[2] = {c->plane[cur][0], c->plane[cur][1]};
			for (int f = 0; f < 2; f++) {
				c->plane[crd_ctx::Y][f] = b_out[f];
				c->plane[crd_ctx::SA][f] = b_np1[f];
				c->plane[crd_ctx::OUT][f] = b_n[f];
			}
		}
		return CRD_OK;
	};
	auto &A = lead->ark;  // (every context of the run gets a copy at the end)
	double h = 0.0;
	if (resume) {
		st.t_internal = lead->dense.t_np1;
		if (tout <= lead->dense.t_np1) {  // still inside the step the integrator has already taken: interpolate again
			prev = crd_ctx::OUT;
			cur = crd_ctx::SA;
			const double hstep = lead->dense.t_np1 - lead->dense.t_n;
			if (int rc = hand_back((tout - lead->dense.t_n) / hstep, hstep)) return rc;
			for (int k = 0; k < n; k++) cs[k]->dense.t_out = tout;
			st.t = tout;
			st.h_next = arkode_method ? A.hprime : std::fmin(o.h0 > 0.0 ? o.h0 : 0.8 * crd_stable_dt(&lead->p), h_cap);
			if (stats) *stats = st;
			return CRD_OK;
		}
		t = lead->dense.t_np1;
		cur = crd_ctx::SA;   // the integrator's own state
		spare = crd_ctx::OUT;  // y_n of the finished step: free
		third = crd_ctx::Y;    // the interpolant handed back last time: free
	}
	if (arkode_method) {
		if (!resume) {  // a fresh state: ARKodeInit
			A.live = false;
			A.nst = 0;
			A.tn = t0;
			A.eta = 1.0;
			A.etamax = arkode::kEtamx1;
			A.ehist[0] = A.ehist[1] = A.ehist[2] = 1.0;
			A.h = o.h0;
			if (!(A.h > 0.0) && tout > t0)
				if (int rc = arkode_initial_step(cs, n, t0, tout, cur, o, n_components, &A.h)) return rc;
			A.h = std::fmin(A.h, h_cap);
			A.hprime = A.h;
		}
		h = A.h;
	} else {
		h = std::fmin(o.h0 > 0.0 ? o.h0 : 0.8 * crd_stable_dt(&lead->p), h_cap);
	}
	st.h_first = (arkode_method && A.nst > 0 && A.hprime != A.h) ? A.h * A.eta : h;

	// Ghost rows of each plane's present content that are still valid (multi-slab; a single slab wraps in the kernel).  The
	// integrator exchanges kAdaptGhost rows of the state it is about to step from when fewer than the attempt's five are left, and
	// every attempt produces its rows AND the ghost-region rows its input still covers -- the deep halo of the fixed stepper, by
	// attempts: one exchange per kAdaptGhost / 5 accepted steps instead of one per attempt (a rejected attempt re-runs on the same
	// input).  Only owned rows count towards the error norm (FusedArgs::err_lo / err_hi), so every rank still sees the same sum.
	int shortest = lead->nyl;
	for (int k = 0; k < lead->d1; k++) {
		int64_t a0, a1;
		if (crd_slab_extents(lead->g.ny, k, lead->d1, &a0, &a1) == CRD_OK) shortest = (int)std::min<int64_t>(shortest, a1 - a0 + 1);
	}
	const int kAdaptGhost = kEmbedHalo * std::max(1, std::min((kGhost - kStepHalo) / kEmbedHalo, shortest / kEmbedHalo));  // 60 rows: 12 attempts
	int ext_of[crd_ctx::NPLANES];
	for (int &e : ext_of) e = multi ? 0 : (1 << 20);
	// One attempt of size hh at time tt from plane `src` into plane `dst`, enqueued only: the kernel, the fixed-order sum of its
	// partials, and -- on the second stream, behind an event -- the reduction over the ranks and the copy of the scalar to
	// page-locked memory.  `slot` (0 / 1) names the buffers; finish_attempt(slot) waits for the scalar.
	// The two attempt slots (an attempt and the one launched ahead of its verdict) have scalars of their own -- elements 2 and 3; 0 is
	// collect_scalar's / crd_state_max_abs's, which run on the compute stream with no ordering against a slot's second-stream work.
	constexpr int kAttemptScalar = 2;
	bool in_flight[2] = {false, false};  // launched and not yet waited for
	auto launch_attempt = [&](double tt, double hh, int src, int dst, int slot) -> int {
		// The attempt behind an exchange is cut in two (round 5): its rows that read owned rows only, [5, nyl - 5), are launched straight
		// behind the exchange's release and run UNDER it; the rest -- the edge rows and the ghost-region rows -- follow behind the halo
		// event as one small launch, and the one sum kernel behind that adds both launches' partials.  (Uncut, the 2 x 7.9 MB of an
		// 8192-column slab's 60 ghost rows travelled with nothing to hide under: 167 us per exchange, 13 us per attempt.)
		bool under_exchange = false;
		if (ext_of[src] < kEmbedHalo) {
			if (int rc = prime_halo(cs, n, src, kAdaptGhost, true)) return rc;
			under_exchange = true;
			for (int k = 0; k < n; k++) under_exchange = under_exchange && cs[k]->nyl >= 8 * kEmbedHalo;
			if (!under_exchange)
				for (int k = 0; k < n; k++) {
					if (int rc = set_device(cs[k])) return rc;
					HIP_TRY(cs[k], hipStreamWaitEvent(cs[k]->compute, cs[k]->ev_halo, 0));
				}
			ext_of[src] = kAdaptGhost;
		}
		const int e = multi ? ext_of[src] - kEmbedHalo : 0;
		for (int k = 0; k < n; k++) {
			crd_ctx *c = cs[k];
			if (int rc = set_device(c)) return rc;
			FusedCall call = make_fused_call(c, tt, hh, src, dst);
			call.embed = arkode_method ? 2 : 1;
			call.absorb[4] = arkode_method ? (absorbing(c, tt + 0.75 * hh) ? 1 : 0) : call.absorb[3];  // the fifth stage's time: t + 3/4 h (Zonneveld) / t + h
			call.plan = arkode_method ? &c->plan_arkode : &c->plan_embed;
			call.rtol = o.rtol;
			call.atol = o.atol;
			call.err_partials = c->err_partials + (size_t)slot * (size_t)c->err_capacity;
			call.err_capacity = c->err_capacity;
			const bool reduce_on_device = c->halo == CRD_HALO_RCCL;
			// (the slot's previous user may have been launched ahead and never waited for: its sum and reduction on the second stream must be
			// through with the slot's partials and scalar before this attempt writes them -- only then: a slot whose attempt the host has
			// waited for (finish_attempt) is quiet, and a cross-stream wait in front of every attempt is a bubble of its own)
			if (in_flight[slot]) HIP_TRY(c, hipStreamWaitEvent(c->compute, c->ev_norm[slot], 0));
			// The sum of the partials runs on the SECOND stream behind the step kernel's own completion signal, so that the compute stream
			// carries step kernels only, back to back: a one-block kernel between two sweeps cost its 8 us and 5 more of dispatch latency
			// it is too short to hide (kernel-trace timelines, profiles/r05/adaptive_attempt_timeline_kernel_trace.txt).
			// (The RK4(3) pair launches nothing ahead of a verdict: there the hop to the second stream is latency added to every attempt --
			// 76 -> 86 us measured -- and the sum stays behind its sweep on the compute stream.)
			const bool sum_off_stream = arkode_method;
			double *const sum_to = reduce_on_device ? c->scalar_dev + kAttemptScalar + slot : c->scalar_host + kAttemptScalar + slot;
			call.err_sum = sum_to;
			call.err_defer_sum = sum_off_stream;
			call.done_event = sum_off_stream || reduce_on_device ? c->ev_attempt[slot] : c->ev_norm[slot];
			int items = 0, inner_items = 0;
			call.err_items_out = &items;
			if (under_exchange) {
				FusedCall inner = call;
				inner.done_event = nullptr;
				inner.err_defer_sum = true;  // (the second launch, or the second stream, sums both launches' partials)
				inner.err_items_out = &inner_items;
				HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, inner, kEmbedHalo, c->nyl - kEmbedHalo, 0, 0, c->compute));
				HIP_TRY(c, hipStreamWaitEvent(c->compute, c->ev_halo, 0));
				call.err_offset = inner_items;
				HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, -e, kEmbedHalo, c->nyl - kEmbedHalo, c->nyl + e, c->compute));
			} else {
				HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, -e, c->nyl + e, 0, 0, c->compute));
			}
			if (sum_off_stream || reduce_on_device) HIP_TRY(c, hipStreamWaitEvent(c->comm, c->ev_attempt[slot], 0));
			if (reduce_on_device) {
				if (sum_off_stream) HIP_TRY(c, launch_sum_partials(call.err_partials, inner_items + items, sum_to, nullptr, c->comm));
				NCCL_TRY(c, g_rccl.AllReduce(c->scalar_dev + kAttemptScalar + slot, c->scalar_dev + kAttemptScalar + slot, 1, ncclDouble, ncclSum, c->nccl, c->comm));  // (every rank gets the same bits, hence takes the same decision)
				HIP_TRY(c, launch_scalar_to_host(c->scalar_dev + kAttemptScalar + slot, c->scalar_host + kAttemptScalar + slot, c->ev_norm[slot], c->comm));
			} else if (sum_off_stream) {
				HIP_TRY(c, launch_sum_partials(call.err_partials, inner_items + items, sum_to, c->ev_norm[slot], c->comm));  // (straight into page-locked memory)
			}
		}
		ext_of[dst] = multi ? e : (1 << 20);
		in_flight[slot] = true;
		return CRD_OK;
	};
	auto finish_attempt = [&](int slot, double *sum) -> int {
		double acc = 0.0;
		for (int k = 0; k < n; k++) {  // (LOCAL groups: the slabs' sums added on the host in slab order)
			crd_ctx *c = cs[k];
			if (int rc = set_device(c)) return rc;
			HIP_TRY(c, hipEventSynchronize(c->ev_norm[slot]));
			acc += c->scalar_host[kAttemptScalar + slot];
		}
		in_flight[slot] = false;
		*sum = acc;
		return CRD_OK;
	};
	auto attempt = [&](double hh, int dst, double *sum) -> int {  // launch and wait (the RK4(3) pair's loop)
		if (int rc = launch_attempt(t, hh, cur, dst, 0)) return rc;
		return finish_attempt(0, sum);
	};
	auto accept_planes = [&](int dst) {  // rotate: the old state becomes y_n (kept for the interpolant), the old y_n / scratch becomes the next target
		const int old_cur = cur;
		cur = dst;
		if (dense) {
			spare = (prev >= 0) ? prev : third;
			if (prev < 0) third = -1;
			prev = old_cur;
		} else {
			spare = old_cur;
		}
	};

	bool after_reject = false;
	int rc = CRD_OK;
	int64_t steps_this_call = 0;
	struct {
		bool live;
		double h;
		int src, dst, slot;
	} ahead = {false, 0.0, -1, -1, 0};  // the attempt launched ahead of its predecessor's verdict (CRD_ADAPT_ARKODE)
	int next_slot = 0;
	while (t < tout && rc == CRD_OK && arkode_method) {
		// ---- CRD_ADAPT_ARKODE: one step = attempts until the error test passes (arkStep), then arkPrepareNextStep / arkCompleteStep ----
		if (steps_this_call >= o.max_steps) {
			rc = fail(lead, CRD_ESTATE, "adaptive integration: max_steps steps taken before reaching tout (ARK_TOO_MUCH_WORK)");
			break;
		}
		if (A.nst > 0 && A.hprime != A.h) A.h *= A.eta;
		if (!(A.h > 1e-14 * std::fmax(std::fabs(t), 1e-300)) && !(t == 0.0 && A.h > 0.0)) {
			rc = fail(lead, CRD_ESTATE, "adaptive integration: step size underflow");
			break;
		}
		const int dst = spare;
		double dsm = 0.0;
		for (int nef = 0;;) {
			// This attempt -- unless it is already in flight: the previous step launched it ahead of its own verdict (below).
			int my_slot;
			if (ahead.live && ahead.h == A.h && ahead.src == cur && ahead.dst == dst) {
				my_slot = ahead.slot;
				st.launched_ahead++;
			} else {
				my_slot = next_slot;
				next_slot ^= 1;
				if ((rc = launch_attempt(t, A.h, cur, dst, my_slot))) break;
			}
			ahead.live = false;
			// The step after this one, launched BEFORE this one's error norm is known, on the assumption the integrator makes most of
			// the time on these grids: the attempt passes and the step size stays (the controller's dead band, or the cap).  If that
			// turns out wrong the launch is void -- it wrote a plane nobody needs (the one that becomes scratch once this step is
			// accepted; if this step is rejected, y_{n-1} in it is not needed either: the interpolant is built on the LAST step) --
			// and the next attempt is simply issued afresh.  The host's wait for the norm (reduction over the ranks, copy, wake-up)
			// then hides under a kernel that is already running.  No exchange is ever issued ahead.
			const int after = prev >= 0 ? prev : third;
			if (after >= 0 && t + A.h < tout && steps_this_call + 1 < o.max_steps && ext_of[dst] >= kEmbedHalo) {
				if ((rc = launch_attempt(t + A.h, A.h, dst, after, next_slot))) break;
				ahead = {true, A.h, dst, after, next_slot};
				next_slot ^= 1;
			}
			double sum = 0.0;
			if ((rc = finish_attempt(my_slot, &sum))) break;
			dsm = std::sqrt(sum / n_components);
			st.err_last = dsm;
#ifdef CRD_TUNING_BUILD
			if (std::getenv("CRD_ADAPT_TRACE")) std::fprintf(stderr, "libcrd attempt: t %.17g h %.17g sum %.17g\n", t, A.h, sum);
#endif
			if (dsm <= 1.0) break;  // (a NaN fails the test)
			ahead.live = false;     // a failed step: what was launched ahead of it is void
			nef++;
			st.rejected++;
			if (nef == arkode::kMaxNef) {
				rc = fail(lead, CRD_ESTATE, "adaptive integration: the error test failed 7 times on one step (ARK_ERR_FAILURE)");
				break;
			}
			A.etamax = 1.0;  // no growth for the rest of this step, and none after it
			const double e[3] = {std::isfinite(dsm) ? dsm * o.bias : 1e300, A.ehist[0], A.ehist[1]};
			double eta = arkode::pid_eta(A.h, e, A.etamax, o.safety, o.shrink, h_cap);
			if (nef >= arkode::kSmallNef) eta = std::fmin(eta, arkode::kEtamxf);
			A.h *= eta;
		}
		if (rc != CRD_OK) break;
		A.ehist[2] = A.ehist[1];
		A.ehist[1] = A.ehist[0];
		A.ehist[0] = dsm * o.bias;
		if (A.etamax == 1.0) {  // the step failed its test at least once: keep its size for the next one
			A.hprime = A.h;
			A.eta = 1.0;
		} else {
			A.eta = arkode::pid_eta(A.h, A.ehist, A.etamax, o.safety, o.shrink, h_cap);
			A.hprime = A.h * A.eta;
		}
		A.etamax = o.growth;
		t_prev = t;
		t += A.h;
		A.tn = t;
		A.nst++;
		steps_this_call++;
		accept_planes(dst);
		st.accepted++;
		st.h_last = A.h;
		st.h_min = (st.h_min == 0.0) ? A.h : std::fmin(st.h_min, A.h);
		st.h_max = std::fmax(st.h_max, A.h);
	}
	while (t < tout && rc == CRD_OK && !arkode_method) {
		// ---- CRD_ADAPT_RK43: the embedded pair and I-controller of rounds 1-2 ----
		if (st.accepted + st.rejected >= o.max_steps) {
			rc = fail(lead, CRD_ESTATE, "adaptive integration: max_steps attempts taken before reaching tout");
			break;
		}
		double hh = h;
		bool clipped = false;
		if (!dense && (t + hh >= tout || tout - (t + hh) < 1e-12 * std::fabs(tout))) {  // land on tout exactly; absorb a sliver of a last step
			hh = tout - t;
			clipped = true;
		}
		if (!(hh > 1e-14 * std::fmax(std::fabs(t), 1e-300)) && !(t == 0.0 && hh > 0.0)) {
			rc = fail(lead, CRD_ESTATE, "adaptive integration: step size underflow");
			break;
		}
		const int dst = spare;
		double sum = 0.0;
		if ((rc = attempt(hh, dst, &sum))) break;
		const double err = o.bias * std::sqrt(sum / n_components);
		st.err_last = err;
		double eta;
		if (!(err == err) || std::isinf(err)) eta = o.shrink;  // NaN / inf: the step blew up
		else if (err <= 0.0) eta = o.growth;
		else eta = std::fmin(o.growth, std::fmax(o.shrink, o.safety * std::pow(err, -0.25)));
		if (err <= 1.0) {
			t_prev = t;
			t = clipped ? tout : t + hh;
			accept_planes(dst);
			st.accepted++;
			if (after_reject) eta = std::fmin(eta, 1.0);  // no growth right after a rejection
			after_reject = false;
			if (!clipped || st.accepted == 1) {
				st.h_last = hh;
				st.h_min = (st.h_min == 0.0) ? hh : std::fmin(st.h_min, hh);
				st.h_max = std::fmax(st.h_max, hh);
			}
			if (!clipped) h = hh * eta;
			else h = std::fmax(h, hh * eta);  // a step shortened to hit tout says nothing against the step it replaced
			h = std::fmin(h, h_cap);
		} else {
			st.rejected++;
			after_reject = true;
			h = hh * std::fmin(eta, 0.9);
		}
	}
	// No launch outlives the call -- neither a void one (issued ahead of a verdict that then went the other way) nor, on an error exit,
	// the attempt that was being waited for: their reductions and copies on the second stream still write the slots' scalars.
	for (int slot = 0; slot < 2; slot++)
		if (in_flight[slot]) {
			double ignored;
			(void)finish_attempt(slot, &ignored);
		}
	st.t_internal = t;
	if (rc == CRD_OK && dense && prev >= 0 && t >= tout) {
		// ARK_NORMAL: the step t_prev -> t has reached or passed tout.  f at both ends, then the cubic Hermite interpolant at tout.
		const double hstep = t - t_prev;
		if ((rc = rhs_on_planes(cs, n, t_prev, prev, crd_ctx::SB)) == CRD_OK && (rc = rhs_on_planes(cs, n, t, cur, crd_ctx::ACC)) == CRD_OK &&
		    (rc = hand_back((tout - t_prev) / hstep, hstep)) == CRD_OK) {
			for (int k = 0; k < n; k++) {
				cs[k]->dense.pending = true;
				cs[k]->dense.t_out = tout;
				cs[k]->dense.t_n = t_prev;
				cs[k]->dense.t_np1 = t;
			}
			t = tout;
		}
	} else {  // no dense output (or a zero-length interval, or an error): hand back the state reached
		for (int k = 0; k < n; k++) {
			if (cur != crd_ctx::Y) {
				std::swap(cs[k]->plane[crd_ctx::Y][0], cs[k]->plane[cur][0]);
				std::swap(cs[k]->plane[crd_ctx::Y][1], cs[k]->plane[cur][1]);
			}
			cs[k]->dense.pending = false;
		}
	}
	if (arkode_method) {
		A.live = rc == CRD_OK && lead->dense.pending;
		for (int k = 1; k < n; k++) cs[k]->ark = A;
		h = A.hprime;
	}
	st.t = t;
	st.h_next = h;
	if (stats) *stats = st;
	return rc;
}

int crd_integrate_adaptive(crd_ctx *c, double t0, double tout, const crd_adaptive_options *opt, crd_adaptive_stats *stats)
{
	if (!c) return CRD_EINVAL;
	if (c->halo < 0) return fail(c, CRD_ESTATE, "multi-slab context is not wired (crd_comm_attach_local / crd_comm_init_rccl)");
	if (c->halo == CRD_HALO_LOCAL) return fail(c, CRD_ESTATE, "LOCAL groups integrate through crd_group_integrate_adaptive");
	crd_ctx *one[1] = {c};
	return integrate_adaptive_impl(one, 1, t0, tout, opt, stats);
}

int crd_group_integrate_adaptive(crd_ctx *const *ctxs, int n, double t0, double tout, const crd_adaptive_options *opt, crd_adaptive_stats *stats)
{
	if (int rc = check_group(ctxs, n)) return rc;
	if (n > 1)
		for (int k = 0; k < n; k++)
			if (ctxs[k]->halo != CRD_HALO_LOCAL) return fail(ctxs[0], CRD_ESTATE, "group is not attached");
	return integrate_adaptive_impl(ctxs, n, t0, tout, opt, stats);
}

int crd_plan_launches(crd_ctx *c)
{
	if (!c) return CRD_EINVAL;
	if (int rc = set_device(c)) return rc;
	if (int rc = ensure_timing_events(c)) return rc;  // (a first crd_step_rk4_timed would otherwise create its 128 events inside the region it times)
	if (resolve_stepper(c) != CRD_STEPPER_FUSED || c->plan.tuned || !c->plan.autotune) return CRD_OK;
	// After a dense-output call the "scratch" plane SA holds the integrator's own state y_{n+1}, which a resumed
	// crd_integrate_adaptive continues from: leave it alone (the plan is then measured by the first fixed-step launch instead).
	if (c->dense.pending) return CRD_OK;
	// One step of the resident state into the scratch planes, discarded: its first launch is where the plan is measured.  The
	// state itself (plane Y) is only read; ghost rows may be stale, which only matters to results nobody keeps.
	FusedCall call = make_fused_call(c, 0.0, 1e-9 * crd_stable_dt(&c->p), crd_ctx::Y, crd_ctx::SA);
	for (int k = 0; k < 5; k++) call.absorb[k] = 0;
	const int lo = c->halo == CRD_HALO_SELF ? 0 : kStepHalo, hi = c->halo == CRD_HALO_SELF ? c->nyl : c->nyl - kStepHalo;
	HIP_TRY(c, launch_fused_step(c->p.precision, c->desc, call, lo, hi, 0, 0, c->compute));
	HIP_TRY(c, hipStreamSynchronize(c->compute));
	return CRD_OK;
}

int crd_group_set_threads(crd_ctx *const *ctxs, int n, int threads)
{
	if (int rc = check_group(ctxs, n)) return rc;
	if (threads < 0) return fail(ctxs[0], CRD_EINVAL, "issuing threads: 0 = one per device, k >= 1 = exactly k");
	ctxs[0]->group_threads = threads;
	return CRD_OK;
}

int crd_set_exchange_period(crd_ctx *c, int steps)
{
	if (!c) return CRD_EINVAL;
	if (steps < kMinExchangeEvery || steps > kMaxExchangeEvery) return fail(c, CRD_EINVAL, "exchange period: 3 .. 16 steps");
	if (c->exchange_every != steps) c->cycle_pos = -1;  // the ghost rows were exchanged for another cycle length
	c->exchange_every = steps;
	return CRD_OK;
}

int crd_get_exchange_period(const crd_ctx *c) { return c ? c->exchange_every : CRD_EINVAL; }

int crd_set_halo_slack(crd_ctx *c, int sweeps)
{
	if (!c) return CRD_EINVAL;
	if (sweeps != 1 && sweeps != 2) return fail(c, CRD_EINVAL, "halo slack is 1 or 2 sweeps");
	c->halo_slack = sweeps;
	return CRD_OK;
}

int crd_set_diagnostics(crd_ctx *c, int on)
{
	if (!c) return CRD_EINVAL;
	c->diagnostics = on != 0;
	return CRD_OK;
}

int crd_get_step_timing(const crd_ctx *c, crd_step_timing *out)
{
	if (!c || !out) return CRD_EINVAL;
	*out = c->timing;
	out->agreement_restarts = c->agreement_restarts;
	return CRD_OK;
}

int crd_group_step_rk4(crd_ctx *const *ctxs, int n, double t0, double dt, int64_t nsteps) { return group_step(ctxs, n, t0, dt, nsteps, false); }

// The group call with the clocks of crd_step_rk4_timed on every slab: an event pair around the batch on each context's compute stream
// and one around ONE full-height launch of the dominant kernel per context, mid-run (a launch of the exchange cycle that is neither
// its split first nor its split last).  Returns when every slab's last step is done; crd_get_step_timing(ctxs[k]) has slab k's figures.
int crd_group_step_rk4_timed(crd_ctx *const *ctxs, int n, double t0, double dt, int64_t nsteps)
{
	if (int rc = check_group(ctxs, n)) return rc;
	for (int k = 0; k < n; k++) {
		crd_ctx *c = ctxs[k];
		if (int rc = set_device(c)) return rc;
		if (int rc = ensure_timing_events(c)) return rc;
		c->timed_rows = c->timed_steps = 0;
		c->timing = crd_step_timing{};
		HIP_TRY(c, hipEventRecord(c->ev_t0, c->compute));
	}
	if (int rc = group_step(ctxs, n, t0, dt, nsteps, true)) return rc;
	for (int k = 0; k < n; k++) {
		crd_ctx *c = ctxs[k];
		if (int rc = set_device(c)) return rc;
		HIP_TRY(c, hipEventRecord(c->ev_t1, c->compute));
	}
	for (int k = 0; k < n; k++) {
		crd_ctx *c = ctxs[k];
		if (int rc = set_device(c)) return rc;
		HIP_TRY(c, hipEventSynchronize(c->ev_t1));
		HIP_TRY(c, hipStreamSynchronize(c->comm));
		float ms = 0.f, km = 0.f;
		HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
		if (c->timed_steps > 0) HIP_TRY(c, hipEventElapsedTime(&km, c->ev_k[0], c->ev_k[1]));  // (set where a launch of this context was timed)
		c->timing.ms_total = ms;
		c->timing.kernel_ms = km;
		c->timing.steps = nsteps;
		c->timing.halo_slack = c->halo_slack;
		c->timing.timed_steps_per_launch = c->timed_steps;
	}
	return CRD_OK;
}

int crd_step_rk4_timed(crd_ctx *c, double t0, double dt, int64_t nsteps, double *ms_total, double *kernel_ms, int *launches_per_step)
{
	if (!c) return CRD_EINVAL;
	if (c->halo < 0 || c->halo == CRD_HALO_LOCAL) return fail(c, CRD_ESTATE, "timed stepping needs a single-slab or RCCL context");
	if (int rc = set_device(c)) return rc;
	if (int rc = ensure_timing_events(c)) return rc;
	constexpr int kMaxDiagExchanges = 64;
	if (c->diagnostics)
		while ((int)c->ev_diag.size() < 4 * kMaxDiagExchanges) {
			hipEvent_t e;
			HIP_TRY(c, hipEventCreate(&e));
			c->ev_diag.push_back(e);
		}
	crd_ctx *one[1] = {c};
	decide_cycle_start(one, 1);
	int timed = 0;
	c->timed_rows = c->timed_steps = 0;
	c->diag_waits = c->diag_exchanges = 0;
	c->diag_active = c->diagnostics && c->halo == CRD_HALO_RCCL;
	c->timing = crd_step_timing{};
	HIP_TRY(c, hipEventRecord(c->ev_t0, c->compute));
	trace_push("crd_step_rk4_timed");
	const int rc_steps = run_steps(one, 1, t0, dt, nsteps, &timed);
	trace_pop();
	c->diag_active = false;
	if (rc_steps) return rc_steps;
	HIP_TRY(c, hipEventRecord(c->ev_t1, c->compute));
	HIP_TRY(c, hipEventSynchronize(c->ev_t1));
	HIP_TRY(c, hipStreamSynchronize(c->comm));
	float ms = 0.f;
	HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
	if (ms_total) *ms_total = ms;
	double sum = 0.0;
	for (int k = 0; k < timed; k++) {
		float m = 0.f;
		HIP_TRY(c, hipEventElapsedTime(&m, c->ev_k[(size_t)(2 * k)], c->ev_k[(size_t)(2 * k + 1)]));
		sum += m;
	}
	if (kernel_ms) *kernel_ms = timed ? sum / timed : 0.0;
	if (launches_per_step) *launches_per_step = (resolve_stepper(c) == CRD_STEPPER_FUSED) ? 1 : 2;
	crd_step_timing &tm = c->timing;
	tm.ms_total = ms;
	tm.kernel_ms = timed ? sum / timed : 0.0;
	tm.steps = nsteps;
	tm.halo_slack = c->halo_slack;
	tm.timed_steps_per_launch = timed ? c->timed_steps : 0;
	tm.halo_waits = c->diag_waits;
	tm.exchanges = c->diag_exchanges;
	for (int k = 0; k < c->diag_waits; k++) {
		float m = 0.f;
		HIP_TRY(c, hipEventElapsedTime(&m, c->ev_diag[(size_t)(4 * k)], c->ev_diag[(size_t)(4 * k + 1)]));
		tm.exposed_halo_ms += m;
	}
	for (int k = 0; k < c->diag_exchanges; k++) {
		float m = 0.f;
		HIP_TRY(c, hipEventElapsedTime(&m, c->ev_diag[(size_t)(4 * k + 2)], c->ev_diag[(size_t)(4 * k + 3)]));
		tm.exchange_ms += m;
	}
	return CRD_OK;
}

int crd_dominant_kernel_rows(const crd_ctx *c, int64_t *rows)
{
	if (!c || !rows) return CRD_EINVAL;
	const int stepper = resolve_stepper(c);
	if (c->halo == CRD_HALO_SELF) *rows = c->nyl;
	else if (stepper == CRD_STEPPER_FUSED)  // the launch crd_step_rk4_timed last timed (before any: the second step of an exchange cycle)
		*rows = c->timed_rows > 0 ? c->timed_rows : c->nyl + 2 * kStepHalo * (c->exchange_every - 1 - kTimedCycleStep);
	else *rows = c->nyl - 2;
	return CRD_OK;
}

const char *crd_dominant_kernel_name(const crd_ctx *c)
{
	if (!c) return "";
	return resolve_stepper(c) == CRD_STEPPER_FUSED ? fused_kernel_name(c->p.precision, c->p.model) : stage_kernel_name(c->p.precision, c->p.model);
}

}  // extern "C"
