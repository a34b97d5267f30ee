// waitvalue_probe.hip -- can a stream be released by a value a kernel writes (hipStreamWaitValue64), and what does that cost
// against a cross-stream event?  Stream A runs a ~60 us kernel whose LAST workgroup writes a sequence number; stream B waits for
// it (wait-value, or event) and runs a tiny kernel that stamps the time.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void busy(double *buf, size_t n, unsigned long long *flag, unsigned long long seq, unsigned int *count, unsigned long long *t_done)
{
	for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) buf[q] = buf[q] * 1.0000001 + 1.0;
	__syncthreads();
	if (threadIdx.x == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		if (atomicAdd(count, 1u) == gridDim.x - 1) {  // last workgroup
			*count = 0;
			*t_done = __builtin_amdgcn_s_memrealtime();
			if (flag) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

__global__ void stamp(unsigned long long *t) { *t = __builtin_amdgcn_s_memrealtime(); }

int main()
{
	const size_t n = 1 << 24;
	double *buf;
	unsigned long long *flag = nullptr, *t_done, *t_start;
	unsigned int *count;
	(void)hipMalloc(&buf, n * 8);
	(void)hipMemset(buf, 0, n * 8);
	hipError_t e = hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory);
	std::printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(e));
	if (e != hipSuccess) return 1;
	(void)hipMemset(flag, 0, 8);
	(void)hipMalloc(&t_done, 8);
	(void)hipMalloc(&t_start, 8);
	(void)hipMalloc(&count, 4);
	(void)hipMemset(count, 0, 4);
	hipStream_t a, b;
	(void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
	(void)hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
	hipEvent_t ev;
	(void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
	for (int mode = 0; mode < 2; mode++) {
		std::vector<double> lat;
		for (int it = 1; it <= 30; it++) {
			const unsigned long long seq = (unsigned long long)(mode * 1000 + it);
			if (mode == 0) {
				e = hipStreamWaitValue64(b, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull);
				if (e != hipSuccess) {
					std::printf("hipStreamWaitValue64: %s\n", hipGetErrorString(e));
					return 1;
				}
				stamp<<<1, 1, 0, b>>>(t_start);
				busy<<<1024, 256, 0, a>>>(buf, n, flag, seq, count, t_done);
			} else {
				busy<<<1024, 256, 0, a>>>(buf, n, nullptr, seq, count, t_done);
				(void)hipEventRecord(ev, a);
				(void)hipStreamWaitEvent(b, ev, 0);
				stamp<<<1, 1, 0, b>>>(t_start);
			}
			(void)hipStreamSynchronize(a);
			(void)hipStreamSynchronize(b);
			unsigned long long td, ts;
			(void)hipMemcpy(&td, t_done, 8, hipMemcpyDeviceToHost);
			(void)hipMemcpy(&ts, t_start, 8, hipMemcpyDeviceToHost);
			lat.push_back(((double)ts - (double)td) / 100.0);  // s_memrealtime ticks at 100 MHz -> us
		}
		std::sort(lat.begin(), lat.end());
		std::printf("%s: dependent kernel on the other stream starts %.2f us (median; min %.2f, max %.2f) after the producer's last workgroup\n",
		            mode == 0 ? "wait-value on a kernel-written flag" : "event record + cross-stream wait  ", lat[lat.size() / 2], lat.front(), lat.back());
	}
	return 0;
}
