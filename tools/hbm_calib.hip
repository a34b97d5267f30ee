// hbm_calib.hip -- known-byte-count streaming kernels used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// for the access widths libcrd uses (8 B per lane fp64 rows; the guide calibrates only 16 B per lane).
// Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_calib.hip -o tools/hbm_calib ; run under rocprofv3 --pmc ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

__global__ void __launch_bounds__(256) calib_read8(const double *__restrict__ a, size_t n, double *out)
{
	double s = 0;
	for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) s += a[q];
	if (s == 1.2345e300) out[0] = s;
}
__global__ void __launch_bounds__(256) calib_write8(double *__restrict__ a, size_t n)
{
	for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) a[q] = (double)q;
}
__global__ void __launch_bounds__(256) calib_copy8(const double *__restrict__ a, double *__restrict__ b, size_t n)
{
	for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) b[q] = a[q];
}
__global__ void __launch_bounds__(256) calib_copy8_nt(const double *__restrict__ a, double *__restrict__ b, size_t n)  // stores with the non-temporal hint
{
	for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) __builtin_nontemporal_store(a[q], b + q);
}
__global__ void __launch_bounds__(256) calib_copy16(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n)
{
	for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) b[q] = a[q];
}
__global__ void __launch_bounds__(256) calib_read4(const float *__restrict__ a, size_t n, float *out)
{
	float s = 0;
	for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) s += a[q];
	if (s == 1.2345e30f) out[0] = s;
}

int main()
{
	const size_t bytes = (size_t)2 << 30;  // 2 GiB per buffer, far beyond the 256 MiB Infinity Cache
	const size_t n = bytes / 8;
	double *a, *b;
	CK(hipMalloc(&a, bytes));
	CK(hipMalloc(&b, bytes));
	CK(hipMemset(a, 0, bytes));
	CK(hipMemset(b, 0, bytes));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const int grid = 256 * 8 * 4;
	for (int rep = 0; rep < 3; rep++) {
		float ms;
		CK(hipEventRecord(e0)); calib_read8<<<grid, 256>>>(a, n, b); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
		std::printf("read8   %zu B  %.3f ms  %.0f GB/s\n", bytes, ms, bytes / ms / 1e6);
		CK(hipEventRecord(e0)); calib_write8<<<grid, 256>>>(b, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
		std::printf("write8  %zu B  %.3f ms  %.0f GB/s\n", bytes, ms, bytes / ms / 1e6);
		CK(hipEventRecord(e0)); calib_copy8<<<grid, 256>>>(a, b, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
		std::printf("copy8   %zu B  %.3f ms  %.0f GB/s (read+write)\n", 2 * bytes, ms, 2 * bytes / ms / 1e6);
		CK(hipEventRecord(e0)); calib_copy8_nt<<<grid, 256>>>(a, b, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
		std::printf("copy8nt %zu B  %.3f ms  %.0f GB/s (read+write, non-temporal stores)\n", 2 * bytes, ms, 2 * bytes / ms / 1e6);
		CK(hipEventRecord(e0)); calib_copy16<<<grid, 256>>>((const double2 *)a, (double2 *)b, n / 2); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
		std::printf("copy16  %zu B  %.3f ms  %.0f GB/s (read+write)\n", 2 * bytes, ms, 2 * bytes / ms / 1e6);
		CK(hipEventRecord(e0)); calib_read4<<<grid, 256>>>((const float *)a, n * 2, (float *)b); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
		std::printf("read4   %zu B  %.3f ms  %.0f GB/s\n", bytes, ms, bytes / ms / 1e6);
	}
	CK(hipFree(a));
	CK(hipFree(b));
	return 0;
}
