// persist_probe.hip -- would several RK4 steps in ONE launch pay?  The memory side of the fused step kernel (as tools/march_probe.hip:
// strips, chunks, aprons, four-row prefetch, lockstep) with a radius-4 data dependence between consecutive "steps":
//     out[r][x] = (in[r-4][x] + in[r+4][x] + in[r][x-4] + in[r][x+4]) / 4 + 1        (planes ping-pong, periodic rows)
// so that after S steps from zero every element equals S exactly and ANY stale read (a value two steps old) shows.
//   baseline   : S launches, one per step (kernel boundary = the dependence)
//   persistent : ONE launch of S x chunks x strip-blocks work items; a workgroup takes its item from a ticket counter (so that
//                every item it depends on is held by a workgroup that has already started: no deadlock whatever the dispatch
//                order), waits -- a BOUNDED spin -- until the three chunks of the previous step it reads are complete
//                (per-(step, chunk) counters, agent-scope release by the producers / acquire by the consumer), runs, and
//                publishes its own completion.
// hipcc --offload-arch=gfx950 -O3 tools/persist_probe.hip -o tools/persist_probe && tools/persist_probe [nx ny steps]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kApron = 4, kPrefetch = 4, kChunk = 32;

struct Args {
	double *plane[2][2];  // [parity][field]
	int nx, ny, nstrips, nchunks, nsb, steps;
	int *ticket;          // persistent: next item
	int *done;            // persistent: [steps][nchunks] completed strip-blocks
	int *error;           // persistent: a spin ran out
};

__device__ __forceinline__ void sweep_item(const Args &a, int step, int chunk, int strip, int lane)
{
	constexpr int VALID = 64 - 2 * kApron;
	const double *in_u = a.plane[step & 1][0], *in_v = a.plane[step & 1][1];
	double *out_u = a.plane[(step + 1) & 1][0], *out_v = a.plane[(step + 1) & 1][1];
	const int nx = a.nx, ny = a.ny;
	int x = strip * VALID - kApron + lane;
	x %= nx;
	if (x < 0) x += nx;
	const int oc = strip * VALID + lane - kApron;
	const bool stores = lane >= kApron && lane < 64 - kApron && oc < nx;
	const int j0 = chunk * kChunk, j1 = std::min(j0 + kChunk, ny), jbase = j0 - kApron, niter = (j1 - j0) + 2 * kApron, jlast = j1 + kApron - 1;
	auto row = [&](int j) { return (size_t)((j + ny) % ny) * nx; };
	double pu[kPrefetch], pv[kPrefetch], wu[9], wv[9];
#pragma unroll
	for (int k = 0; k < 9; k++) wu[k] = wv[k] = 0.0;
#pragma unroll
	for (int k = 0; k < kPrefetch; k++) {
		const size_t rb = row(std::min(jbase + k, jlast));
		pu[k] = in_u[rb + x];
		pv[k] = in_v[rb + x];
	}
	for (int m0 = 0; m0 < niter; m0 += kPrefetch) {
#pragma unroll
		for (int k = 0; k < kPrefetch; k++) {
			const int m = m0 + k;
			if (m >= niter) break;
			__builtin_amdgcn_s_barrier();
#pragma unroll
			for (int q = 0; q < 8; q++) wu[q] = wu[q + 1], wv[q] = wv[q + 1];  // window of rows p-8 .. p
			wu[8] = pu[k];
			wv[8] = pv[k];
			const size_t rb = row(std::min(jbase + m + kPrefetch, jlast));
			pu[k] = in_u[rb + x];
			pv[k] = in_v[rb + x];
			const int r = jbase + m - kApron;  // centre row p-4: rows p-8 and p are its radius-4 neighbours
			const double lu = __shfl(wu[4], lane - 4, 64), ru = __shfl(wu[4], lane + 4, 64);
			const double lv = __shfl(wv[4], lane - 4, 64), rv = __shfl(wv[4], lane + 4, 64);
			if (m >= 2 * kApron && r < j1 && stores) {
#ifdef NT  // (-DNT: non-temporal stores -- is the producers' release still a write-back of a whole dirty L2 then?)
				__builtin_nontemporal_store((wu[0] + wu[8] + lu + ru) * 0.25 + 1.0, out_u + (size_t)r * nx + oc);
				__builtin_nontemporal_store((wv[0] + wv[8] + lv + rv) * 0.25 + 1.0, out_v + (size_t)r * nx + oc);
#else
				out_u[(size_t)r * nx + oc] = (wu[0] + wu[8] + lu + ru) * 0.25 + 1.0;
				out_v[(size_t)r * nx + oc] = (wv[0] + wv[8] + lv + rv) * 0.25 + 1.0;
#endif
			}
		}
	}
}

__global__ void __launch_bounds__(256) step_kernel(Args a, int step)
{
	const int blk = blockIdx.x, sblk = blk % a.nsb, chunk = blk / a.nsb;
	const int strip = __builtin_amdgcn_readfirstlane(sblk * 4 + (int)(threadIdx.x >> 6));
	if (strip >= a.nstrips) return;
	sweep_item(a, step, chunk, strip, threadIdx.x & 63);
}

__global__ void __launch_bounds__(256) persistent_kernel(Args a)
{
	__shared__ int sh_item, sh_ok;
	if (threadIdx.x == 0) {
		const int item = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const int per_step = a.nchunks * a.nsb, step = item / per_step, chunk = (item - step * per_step) / a.nsb;
		int ok = 1;
		if (step > 0) {
			for (int dc = -1; dc <= 1 && ok; dc++) {
				const int c = (chunk + dc + a.nchunks) % a.nchunks;
				const int *flag = a.done + (size_t)(step - 1) * a.nchunks + c;
				int spins = 0;
				while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.nsb) {
					__builtin_amdgcn_s_sleep(8);
					if (++spins > (1 << 22)) {  // ~ a second: give up instead of hanging the device
						ok = 0;
						__hip_atomic_store(a.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						break;
					}
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		}
		sh_item = item;
		sh_ok = ok;
	}
	__syncthreads();
	const int item = sh_item;
	if (!sh_ok) return;
	const int per_step = a.nchunks * a.nsb, step = item / per_step, r = item - step * per_step, chunk = r / a.nsb, sblk = r - chunk * a.nsb;
	if (step >= a.steps) return;
	const int strip = __builtin_amdgcn_readfirstlane(sblk * 4 + (int)(threadIdx.x >> 6));
	if (strip < a.nstrips) sweep_item(a, step, chunk, strip, threadIdx.x & 63);
	// publish: every wavefront's stores have left it, then one release for the workgroup, then the counter
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__hip_atomic_fetch_add(a.done + (size_t)step * a.nchunks + chunk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

int main(int argc, char **argv)
{
	const int nx = argc > 1 ? std::atoi(argv[1]) : 8192, ny = argc > 2 ? std::atoi(argv[2]) : 1024, steps = argc > 3 ? std::atoi(argv[3]) : 8;
	const size_t n = (size_t)nx * ny;
	Args a{};
	for (auto &pp : a.plane)
		for (auto &p : pp)
			if (hipMalloc(&p, n * 8) != hipSuccess) return 1;
	a.nx = nx;
	a.ny = ny;
	a.nstrips = (nx + 55) / 56;
	a.nchunks = (ny + kChunk - 1) / kChunk;
	a.nsb = (a.nstrips + 3) / 4;
	a.steps = steps;
	(void)hipMalloc(&a.ticket, 4);
	(void)hipMalloc(&a.error, 4);
	(void)hipMalloc(&a.done, sizeof(int) * steps * a.nchunks);
	const int per_step = a.nchunks * a.nsb;
	std::vector<double> h(n);
	auto reset = [&]() {
		for (auto &pp : a.plane)
			for (auto &p : pp) (void)hipMemset(p, 0, n * 8);
		(void)hipMemset(a.ticket, 0, 4);
		(void)hipMemset(a.error, 0, 4);
		(void)hipMemset(a.done, 0, sizeof(int) * steps * a.nchunks);
	};
	auto check = [&](const char *what) {
		long bad = 0;
		for (int f = 0; f < 2; f++) {
			(void)hipMemcpy(h.data(), a.plane[steps & 1][f], n * 8, hipMemcpyDeviceToHost);
			for (size_t q = 0; q < n; q++) bad += h[q] != (double)steps;
		}
		int err = 0;
		(void)hipMemcpy(&err, a.error, 4, hipMemcpyDeviceToHost);
		std::printf("%s: %ld wrong values of %zu%s\n", what, bad, 2 * n, err ? "  (a spin ran out!)" : "");
		return bad == 0 && !err;
	};
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	auto timeit = [&](auto &&launch, int reps) {
		std::vector<float> t;
		for (int r = 0; r < reps; r++) {
			(void)hipMemset(a.ticket, 0, 4);
			(void)hipMemset(a.done, 0, sizeof(int) * steps * a.nchunks);
			(void)hipDeviceSynchronize();
			(void)hipEventRecord(e0);
			launch();
			(void)hipEventRecord(e1);
			(void)hipEventSynchronize(e1);
			float ms;
			(void)hipEventElapsedTime(&ms, e0, e1);
			t.push_back(ms);
		}
		std::sort(t.begin(), t.end());
		return t[t.size() / 2];
	};
	auto base = [&]() {
		for (int s = 0; s < steps; s++) step_kernel<<<per_step, 256>>>(a, s);
	};
	auto pers = [&]() { persistent_kernel<<<per_step * steps, 256>>>(a); };
	std::printf("%d x %d, %d steps, %d work items per step\n", nx, ny, steps, per_step);
	reset();
	base();
	(void)hipDeviceSynchronize();
	bool ok = check("one launch per step");
	reset();
	pers();
	(void)hipDeviceSynchronize();
	ok = check("persistent launch  ") && ok;
	int wrong_runs = 0;
	for (int r = 0; r < 20; r++) {  // staleness is a race: look more than once
		reset();
		pers();
		(void)hipDeviceSynchronize();
		long bad = 0;
		(void)hipMemcpy(h.data(), a.plane[steps & 1][0], n * 8, hipMemcpyDeviceToHost);
		for (size_t q = 0; q < n; q++) bad += h[q] != (double)steps;
		wrong_runs += bad != 0;
	}
	std::printf("persistent launch, 20 more runs: %d with wrong values\n", wrong_runs);
	const float tb = timeit(base, 15), tp = timeit(pers, 15);
	std::printf("one launch per step: %.2f us per step;  persistent: %.2f us per step  (%.1f %%)\n", tb * 1e3 / steps, tp * 1e3 / steps, 100.0 * (tp / tb - 1.0));
	return ok && wrong_runs == 0 ? 0 : 1;
}
