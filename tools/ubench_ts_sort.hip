// ubench_ts_sort.hip -- what one wave pays for the rank kernel's building blocks, alone on its SIMD and with company (round 6):
// ts_sort (512 keys, with and without the inversion count), ts_keys_run, wave_sum_i, ts_sort_lanes - the product's own code
// (this file includes k_theilsen.hip), timed with s_memtime inside the kernel, w waves per SIMD on every CU (one workgroup of
// 4 w waves per CU, LDS padding keeps a second one away).
// hipcc --offload-arch=gfx950 -O3 -mllvm -disable-machine-licm -Imodem_amd/csrc tools/ubench_ts_sort.hip -o /tmp/ubench_ts_sort
#include "../modem_amd/csrc/k_theilsen.hip"
#include <cstdio>
#include <vector>
using namespace rx;
#define REPS 64
template <int WHAT>
__global__ void kb(uint32_t *out, unsigned long long *cyc, float seed)
{
	extern __shared__ uint32_t pad[];
	const int lane = threadIdx.x & 63;
	const TsLane L = ts_lane(lane);
	uint32_t k[8];
	float yv[8];
	#pragma unroll
	for (int t = 0; t < 8; ++t) {
		k[t] = ((uint32_t)(lane * 8 + t) * 2654435761u) >> 1;
		yv[t] = (float)(k[t] & 1023u) * seed;
	}
	int acc = 0;
	__syncthreads();
	const unsigned long long t0 = __builtin_readcyclecounter();
	#pragma unroll 1
	for (int r = 0; r < REPS; ++r) {
		if (WHAT == 0) { acc += ts_sort(k, L, true, 6); k[0] ^= (uint32_t)r * 40503u; k[5] += 977u * (uint32_t)lane; }
		if (WHAT == 1) { acc += ts_sort(k, L, false, 6); k[0] ^= (uint32_t)r * 40503u; k[5] += 977u * (uint32_t)lane; }
		if (WHAT == 2) { TsQuant q = ts_quant(seed * (float)r, -3.f, 3.f, 432); ts_keys_run(k, yv, 8 * lane, 432, seed * (float)r, q); acc += (int)k[3]; }
		if (WHAT == 3) { acc = wave_sum_i(acc + lane + r); }
		if (WHAT == 4) { k[0] = ts_sort_lanes(k[0] + (uint32_t)r * 7919u, L); acc += (int)k[0]; }
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	if (lane == 0)
		cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
	out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)acc + k[1] + k[7] + pad[lane & 15];
}
template <int WHAT> void run(const char *name)
{
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	uint32_t *out; unsigned long long *cyc;
	hipMalloc(&out, (size_t)cus * 1024 * 4);
	hipMalloc(&cyc, (size_t)cus * 16 * 8);
	hipFuncSetAttribute((const void *)kb<WHAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
	for (int w = 1; w <= 4; ++w) {
		const int waves = cus * 4 * w;
		for (int rep = 0; rep < 2; ++rep) {
			hipLaunchKernelGGL(kb<WHAT>, dim3(cus), dim3(256 * w), 96 * 1024, 0, out, cyc, 1.0001f);
			hipDeviceSynchronize();
		}
		std::vector<unsigned long long> h(waves);
		hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
		double s = 0; unsigned long long mx = 0;
		for (auto v : h) { s += (double)v; if (v > mx) mx = v; }
		printf("%-44s waves/SIMD=%d  cycles per call and wave: mean %.0f, slowest %.0f;  per SIMD: one call per %.0f cycles\n", name, w, s / waves / REPS, (double)mx / REPS, s / waves / REPS / w);
	}
	hipFree(out); hipFree(cyc);
}
int main()
{
	run<0>("ts_sort, 512 keys, counting");
	run<1>("ts_sort, 512 keys, plain");
	run<2>("ts_quant + ts_keys_run (8 keys per lane)");
	run<3>("wave_sum_i (dependent)");
	run<4>("ts_sort_lanes (64 keys, dependent)");
	return 0;
}
