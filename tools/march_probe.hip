// march_probe.hip -- the memory side of the fused step kernel without its arithmetic: every wavefront marches through a chunk of
// rows (+ 2 x 4 apron rows) of a strip of columns (+ 2 x 4 apron columns), reads two planes with a four-row prefetch and writes two
// planes, four wavefronts per workgroup in lockstep, work items dealt to the XCDs as crd_fused.hip deals them.  What this probe
// achieves is the ceiling of that ACCESS PATTERN on the device at hand; crd_rk4_fused_step_kernel's distance from it is what its
// arithmetic (and register pressure) costs.  Variants: 8 bytes per lane and row (what the step kernel does; 64 columns per
// wavefront) or 16 bytes (128 columns per wavefront: would wider accesses pay?).
// hipcc --offload-arch=gfx950 -O3 tools/march_probe.hip -o tools/march_probe && tools/march_probe [nx ny]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr int kApron = 4;
#ifndef PF
#define PF 4
#endif
constexpr int kPrefetch = PF;
static int g_lds_bytes = 0;  // dynamic LDS per workgroup: throttles the workgroups resident per CU (160 KiB / g_lds_bytes)

template <int W>  // doubles per lane
struct Vec;
template <> struct Vec<1> { using type = double; };
template <> struct Vec<2> { using type = double2; };

__device__ inline double sum(double v) { return v; }
__device__ inline double sum(double2 v) { return v.x + v.y; }
__device__ inline void set(double &o, double a) { o = a; }
__device__ inline void set(double2 &o, double a) { o.x = a; o.y = a + 1.0; }

template <int W>
__global__ void __launch_bounds__(256) march(const double *__restrict__ in_u, const double *__restrict__ in_v, double *__restrict__ out_u, double *__restrict__ out_v,
                                             int nx, int ny, int chunk, int nstrips, int nchunks, int mapping, int xs_lanes, int lockstep)
{
	using V = typename Vec<W>::type;
	constexpr int COLS = 64 * W, VALID = COLS - 2 * kApron;
	const int lane = threadIdx.x & 63;
	const int nsb = (nstrips + 3) / 4;
	int blk = blockIdx.x, sblk = blk % nsb, cblk = blk / nsb;
	if (mapping == 1) {  // one contiguous run of items per XCD
		const int nb = gridDim.x, q = nb / 8, rem = nb - q * 8, xx = blk % 8, l = blk / 8;
		blk = xx * q + (xx < rem ? xx : rem) + l;
		sblk = blk % nsb;
		cblk = blk / nsb;
	}
	if (mapping == 2) {
		const int x = blk % 8, p = blk / 8, width = nsb * xs_lanes;
		const int d = p / width, sl = p - d * width;
		const int c0 = (int)((long)nchunks * x / 8), c1 = (int)((long)nchunks * (x + 1) / 8);
		const int depth = (c1 - c0 + xs_lanes - 1) / xs_lanes, lane_id = sl / nsb;
		sblk = sl - lane_id * nsb;
		cblk = (d < depth && c0 + lane_id * depth + d < c1) ? c0 + lane_id * depth + d : nchunks;
	}
	const int strip = __builtin_amdgcn_readfirstlane(sblk * 4 + (int)(threadIdx.x >> 6));
	const int c = __builtin_amdgcn_readfirstlane(cblk);
	if (strip >= nstrips || c >= nchunks) return;
	int x = strip * VALID - kApron + lane * W;
	x %= nx;
	if (x < 0) x += nx;
	const int oc = strip * VALID + lane * W - kApron;
	const bool stores = lane * W >= kApron && lane * W + W <= COLS - kApron && oc + W <= nx;  // (W = 2: the apron is 4 = 2 lanes)
	const int j0 = c * chunk, j1 = std::min(j0 + chunk, ny), jbase = j0 - kApron, niter = (j1 - j0) + 2 * kApron, jlast = j1 + kApron - 1;
	auto row = [&](int j) { return (size_t)((j + ny) % ny) * nx; };
	V pu[kPrefetch], pv[kPrefetch];
#pragma unroll
	for (int k = 0; k < kPrefetch; k++) {
		const size_t rb = row(std::min(jbase + k, jlast));
		pu[k] = *reinterpret_cast<const V *>(in_u + rb + x);
		pv[k] = *reinterpret_cast<const V *>(in_v + rb + x);
	}
	double acc = 0.0;
	for (int m0 = 0; m0 < niter; m0 += kPrefetch) {
#pragma unroll
		for (int k = 0; k < kPrefetch; k++) {
			const int m = m0 + k;
			if (m >= niter) break;
			if (lockstep) __builtin_amdgcn_s_barrier();
			const V u = pu[k], v = pv[k];
			const size_t rb = row(std::min(jbase + m + kPrefetch, jlast));
			pu[k] = *reinterpret_cast<const V *>(in_u + rb + x);
			pv[k] = *reinterpret_cast<const V *>(in_v + rb + x);
			acc += sum(u) * 0.5 + sum(v);
			const int r = jbase + m - kApron;  // the row four iterations back is "done"
			if (m >= 2 * kApron && r < j1 && stores) {
				V a, b;
				set(a, acc);
				set(b, acc * 0.25);
#ifdef NT  // (-DNT: the output with the non-temporal hint, as the step kernel's `row_store<true>`)
				if constexpr (W == 1) {
					__builtin_nontemporal_store(a, reinterpret_cast<V *>(out_u + (size_t)r * nx + oc));
					__builtin_nontemporal_store(b, reinterpret_cast<V *>(out_v + (size_t)r * nx + oc));
				} else {
					typedef double native2 __attribute__((ext_vector_type(2)));
					const native2 aa = {a.x, a.y}, bb = {b.x, b.y};
					__builtin_nontemporal_store(aa, reinterpret_cast<native2 *>(out_u + (size_t)r * nx + oc));
					__builtin_nontemporal_store(bb, reinterpret_cast<native2 *>(out_v + (size_t)r * nx + oc));
				}
#else
				*reinterpret_cast<V *>(out_u + (size_t)r * nx + oc) = a;
				*reinterpret_cast<V *>(out_v + (size_t)r * nx + oc) = b;
#endif
			}
		}
	}
}

template <int W>
double run(const double *iu, const double *iv, double *ou, double *ov, int nx, int ny, int chunk, int mapping, int lockstep, int reps, int lanes = 0)
{
	constexpr int VALID = 64 * W - 2 * kApron;
	const int nstrips = (nx + VALID - 1) / VALID, nchunks = (ny + chunk - 1) / chunk, nsb = (nstrips + 3) / 4;
	int nblocks = nsb * nchunks, xs_lanes = 1;
	if (mapping == 2) {
		xs_lanes = lanes > 0 ? lanes : std::max(1, 128 / nsb);
		const int most = (nchunks + 7) / 8;
		nblocks = 8 * ((most + xs_lanes - 1) / xs_lanes) * nsb * xs_lanes;
	}
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	std::vector<float> t;
	for (int r = 0; r < reps + 3; r++) {
		(void)hipEventRecord(e0);
		for (int k = 0; k < 10; k++) march<W><<<nblocks, 256, g_lds_bytes>>>(iu, iv, ou, ov, nx, ny, chunk, nstrips, nchunks, mapping, xs_lanes, lockstep);
		(void)hipEventRecord(e1);
		(void)hipEventSynchronize(e1);
		float ms;
		(void)hipEventElapsedTime(&ms, e0, e1);
		if (r >= 3) t.push_back(ms / 10);
	}
	std::sort(t.begin(), t.end());
	return t[t.size() / 2];
}

int main(int argc, char **argv)
{
	const int nx = argc > 1 ? std::atoi(argv[1]) : 8192, ny = argc > 2 ? std::atoi(argv[2]) : 8192;
	const size_t n = (size_t)nx * ny;
	double *b[4];
	for (auto &p : b) {
		if (hipMalloc(&p, n * 8) != hipSuccess) return 1;
		(void)hipMemset(p, 0, n * 8);
	}
	const double gb = 32.0 * n / 1e9;  // compulsory bytes: read two planes, write two planes
	std::printf("%d x %d, compulsory %.3f GB per sweep\n", nx, ny, gb);
	if (const char *e = std::getenv("PROBE_BLOCKS_PER_CU")) {
		g_lds_bytes = (160 * 1024) / std::atoi(e);
		(void)hipFuncSetAttribute((const void *)march<1>, hipFuncAttributeMaxDynamicSharedMemorySize, g_lds_bytes);
		std::printf("-- %s workgroups per CU (dynamic LDS %d B per workgroup), prefetch %d rows\n", e, g_lds_bytes, kPrefetch);
	}
	if (!std::getenv("PROBE_QUIET"))
	for (int W : {1, 2})
		for (int mapping : {0, 2})
			for (int chunk : {32, 64})
				for (int lockstep : {1, 0}) {
					const double ms = W == 1 ? run<1>(b[0], b[1], b[2], b[3], nx, ny, chunk, mapping, lockstep, 7) : run<2>(b[0], b[1], b[2], b[3], nx, ny, chunk, mapping, lockstep, 7);
					std::printf("%2d B/lane  mapping %d  chunk %3d  lockstep %d : %.4f ms  -> %.0f GB/s of compulsory bytes (%.3f of 8 TB/s)\n", 8 * W, mapping, chunk, lockstep, ms,
					            gb / (ms * 1e-3), gb / (ms * 1e-3) / 8000.0);
				}
	if (argc > 3) {  // extra sweep: mapping / chunk / lanes combinations "m:chunk:lanes,..."
		std::printf("-- extra combinations (8 B/lane, lockstep 1)\n");
		char *tok = std::strtok(argv[3], ",");
		while (tok) {
			int m = 0, ch = 32, ln = 0;
			std::sscanf(tok, "%d:%d:%d", &m, &ch, &ln);
			const double ms = run<1>(b[0], b[1], b[2], b[3], nx, ny, ch, m, 1, 7, ln);
			std::printf("mapping %d  chunk %4d  lanes %d : %.4f ms  -> %.0f GB/s (%.3f of 8 TB/s)\n", m, ch, ln, ms, gb / (ms * 1e-3), gb / (ms * 1e-3) / 8000.0);
			tok = std::strtok(nullptr, ",");
		}
	}
	return 0;
}
