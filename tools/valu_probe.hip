// valu_probe.hip -- issue rate of fp64 / fp32 VALU instructions on gfx950: cycles per wave-instruction for independent FMA
// chains at 1, 2 and 4 wavefronts per SIMD (s_memtime counts shader cycles).  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <typename T, int CHAINS>
__global__ void __launch_bounds__(256) probe(T *out, unsigned long long *cyc, int iters, T a, T b)
{
	T x[CHAINS];
#pragma unroll
	for (int k = 0; k < CHAINS; k++) x[k] = (T)(threadIdx.x + k);
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int k = 0; k < CHAINS; k++) x[k] = x[k] * a + b;
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	T s = 0;
#pragma unroll
	for (int k = 0; k < CHAINS; k++) s += x[k];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <typename T>
void run(const char *name)
{
	const int iters = 20000;
	constexpr int CH = 8;
	for (int blocks_per_cu : {1, 2, 4}) {  // 256 threads = 4 waves = 1 per SIMD
		const int nblocks = 256 * blocks_per_cu;
		T *out;
		unsigned long long *cyc;
		(void)hipMalloc(&out, sizeof(T) * nblocks * 256);
		(void)hipMalloc(&cyc, 8 * nblocks * 4);
		hipEvent_t e0, e1;
		(void)hipEventCreate(&e0);
		(void)hipEventCreate(&e1);
		probe<T, CH><<<nblocks, 256>>>(out, cyc, 100, (T)1.0000001, (T)1e-9);
		(void)hipEventRecord(e0);
		probe<T, CH><<<nblocks, 256>>>(out, cyc, iters, (T)1.0000001, (T)1e-9);
		(void)hipEventRecord(e1);
		(void)hipEventSynchronize(e1);
		float ms;
		(void)hipEventElapsedTime(&ms, e0, e1);
		std::vector<unsigned long long> h(nblocks * 4);
		(void)hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
		std::sort(h.begin(), h.end());
		const double med = (double)h[h.size() / 2];
		const double per_wave_instr = med / ((double)iters * CH);
		std::printf("%s %d wave(s)/SIMD: %.2f cycles per wave-instruction seen by one wave -> %.2f cycles of SIMD time per instruction; %.3f ms, %.1f TFLOP/s\n", name,
		            blocks_per_cu, per_wave_instr, per_wave_instr / blocks_per_cu, ms, 2.0 * 64 * 4 * nblocks * (double)iters * CH / (ms * 1e-3) / 1e12);
		(void)hipFree(out);
		(void)hipFree(cyc);
	}
}

int main()
{
	run<double>("f64 fma");
	run<float>("f32 fma");
	return 0;
}
