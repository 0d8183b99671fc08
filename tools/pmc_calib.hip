// pmc_calib.hip -- known-byte-count kernels in k_polar's access pattern (one raw b32 buffer access per lane,
// 256 contiguous bytes per wave instruction, one independent stream per wave over a buffer far larger than the
// 256 MiB Infinity Cache), to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950:
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o /tmp/pmc_calib
//   rocprofv3 --pmc FETCH_SIZE -d ... -- /tmp/pmc_calib        (and again with WRITE_SIZE)
// calib_read reads BYTES, calib_write writes BYTES, calib_copy does both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr size_t BYTES = 2ull << 30;            // per kernel
constexpr int WAVES = 256 * 16;
constexpr size_t PER_WAVE = BYTES / WAVES;      // 512 KiB stream per wave

__global__ __launch_bounds__(64) void calib_read(const float *src, float *sink)
{
	const float *p = src + (size_t)blockIdx.x * (PER_WAVE / 4) + threadIdx.x;
	float acc = 0.f;
	#pragma unroll 8
	for (size_t i = 0; i < PER_WAVE / 256; ++i)
		acc += p[i * 64];
	if (acc == 123.456f) sink[blockIdx.x] = acc;
}
__global__ __launch_bounds__(64) void calib_write(float *dst)
{
	float *p = dst + (size_t)blockIdx.x * (PER_WAVE / 4) + threadIdx.x;
	#pragma unroll 8
	for (size_t i = 0; i < PER_WAVE / 256; ++i)
		p[i * 64] = (float)i;
}
__global__ __launch_bounds__(64) void calib_copy(const float *src, float *dst)
{
	const float *p = src + (size_t)blockIdx.x * (PER_WAVE / 4) + threadIdx.x;
	float *q = dst + (size_t)blockIdx.x * (PER_WAVE / 4) + threadIdx.x;
	#pragma unroll 8
	for (size_t i = 0; i < PER_WAVE / 256; ++i)
		q[i * 64] = p[i * 64] + 1.f;
}

int main()
{
	float *a, *b, *sink;
	hipMalloc(&a, BYTES); hipMalloc(&b, BYTES); hipMalloc(&sink, WAVES * 4);
	hipMemset(a, 0, BYTES); hipMemset(b, 0, BYTES);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int rep = 0; rep < 2; ++rep) {
		float ms[3];
		hipEventRecord(e0); calib_read<<<WAVES, 64>>>(a, sink); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[0], e0, e1);
		hipEventRecord(e0); calib_write<<<WAVES, 64>>>(b); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[1], e0, e1);
		hipEventRecord(e0); calib_copy<<<WAVES, 64>>>(a, b); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[2], e0, e1);
		printf("rep %d: read %.3f ms (%.0f GB/s)  write %.3f ms (%.0f GB/s)  copy %.3f ms (%.0f GB/s r+w)\n", rep,
			ms[0], BYTES / ms[0] / 1e6, ms[1], BYTES / ms[1] / 1e6, ms[2], 2.0 * BYTES / ms[2] / 1e6);
	}
	printf("bytes per kernel: %zu (KiB %zu)\n", BYTES, BYTES / 1024);
	return 0;
}
