// rcp_probe.hip -- accuracy of v_rcp_f64 and of its Newton refinements on gfx950, and the issue cost of v_rcp_f64.
// Answers: how many Newton steps does the Goldbeter Hill-term reciprocal need (crd_device.h: reciprocal)?
// hipcc --offload-arch=gfx950 -O3 tools/rcp_probe.hip -o tools/rcp_probe && tools/rcp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void acc(const double *x, double *r0, double *r1, double *r2, int n)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double v = x[i];
	double r = __builtin_amdgcn_rcp(v);
	r0[i] = r;
	r = __builtin_fma(r, __builtin_fma(-v, r, 1.0), r);
	r1[i] = r;
	r = __builtin_fma(r, __builtin_fma(-v, r, 1.0), r);
	r2[i] = r;
}

template <int MODE>
__global__ void __launch_bounds__(256) rate(double *out, unsigned long long *cyc, int iters)
{
	double x[8];
#pragma unroll
	for (int k = 0; k < 8; k++) x[k] = 1.5 + threadIdx.x + k;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int k = 0; k < 8; k++) x[k] = MODE == 0 ? __builtin_amdgcn_rcp(x[k]) : __builtin_fma(x[k], 1.0000001, 1e-9);
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	double s = 0;
#pragma unroll
	for (int k = 0; k < 8; k++) s += x[k];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

int main()
{
	const int n = 1 << 22;
	std::vector<double> h(n);
	unsigned long long s = 88172645463325252ull;
	for (int i = 0; i < n; i++) {  // log-uniform over [2.6, 1e6]: the range of (K2^2+z^2)(KR^2+y^2)(KA^4+z^4) and beyond
		s ^= s << 13; s ^= s >> 7; s ^= s << 17;
		h[i] = 2.6 * std::exp((double)(s >> 11) / 9007199254740992.0 * std::log(1e6 / 2.6));
	}
	double *x, *r[3];
	(void)hipMalloc(&x, n * 8);
	for (auto &p : r) (void)hipMalloc(&p, n * 8);
	(void)hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
	acc<<<n / 256, 256>>>(x, r[0], r[1], r[2], n);
	std::vector<double> g(n);
	for (int k = 0; k < 3; k++) {
		(void)hipMemcpy(g.data(), r[k], n * 8, hipMemcpyDeviceToHost);
		double worst = 0;
		for (int i = 0; i < n; i++) worst = std::fmax(worst, std::fabs(g[i] * h[i] - 1.0));  // |r x - 1| in double is good to ~1e-16
		long double worst_l = 0;
		for (int i = 0; i < n; i++) worst_l = fmaxl(worst_l, fabsl((long double)g[i] * (long double)h[i] - 1.0L));
		std::printf("v_rcp_f64 + %d Newton step(s): max relative error %.3Le = 2^%.1Lf\n", k, worst_l, log2l(worst_l));
	}
	unsigned long long *cyc;
	double *out;
	(void)hipMalloc(&cyc, 8 * 256 * 4 * 4);
	(void)hipMalloc(&out, 8 * 256 * 4 * 256);
	for (int mode = 0; mode < 2; mode++)
		for (int bpc : {1, 4}) {
			const int nb = 256 * bpc, iters = 5000;
			if (mode == 0) rate<0><<<nb, 256>>>(out, cyc, iters); else rate<1><<<nb, 256>>>(out, cyc, iters);
			(void)hipDeviceSynchronize();
			std::vector<unsigned long long> c(nb * 4);
			(void)hipMemcpy(c.data(), cyc, 8 * c.size(), hipMemcpyDeviceToHost);
			double sum = 0;
			for (auto v : c) sum += (double)v;
			std::printf("%s, %d wave(s)/SIMD: %.2f cycles of SIMD time per wave-instruction\n", mode == 0 ? "v_rcp_f64" : "v_fma_f64", bpc, sum / c.size() / (iters * 8.0) / bpc);
		}
	return 0;
}
