// ts_probe.cpp -- timing probe for the Theil-Sen kernel (tools only)
#include <hip/hip_runtime.h>
// usage: ts_probe [cols [noise_sigma]]; -DNO_COUNTERS builds the kernel as the library has it (its time means something then)
__device__ int g_fallbacks, g_iters, g_unc, g_reason[8];
#ifndef NO_COUNTERS
#define TS_PROBE_COUNT (&g_fallbacks)
#define TS_PROBE_ITERS (&g_iters)
#define TS_PROBE_UNC (&g_unc)
#define TS_PROBE_REASON g_reason
#endif
#ifndef TS_SRC
#define TS_SRC "../modem_amd/csrc/k_theilsen.hip"
#endif
#include TS_SRC
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
using namespace rx;
int main(int argc, char **argv)
{
	const int rows = 50 * 1024, cols = argc > 1 ? atoi(argv[1]) : 432;
	std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, argc > 2 ? (float)atof(argv[2]) : 0.1f);
	std::vector<float> y((size_t)rows * cols);
	for (int r = 0; r < rows; ++r) for (int i = 0; i < cols; ++i) y[(size_t)r * cols + i] = 1e-4f * (i - cols / 2) + 0.02f + nd(rng);
	float *dy, *ds, *di; hipMalloc(&dy, y.size() * 4); hipMalloc(&ds, rows * 4); hipMalloc(&di, rows * 4);
	hipMemcpy(dy, y.data(), y.size() * 4, hipMemcpyHostToDevice);
	int zero = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_fallbacks), &zero, 4);
	for (int rep = 0; rep < 2; ++rep) {
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a);
		launch_theil_sen_raw(0, rows, cols, dy, ds, di);
		hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
		int fb; hipMemcpyFromSymbol(&fb, HIP_SYMBOL(g_fallbacks), 4);
		int it = 0; hipMemcpyFromSymbol(&it, HIP_SYMBOL(g_iters), 4);
		int rs[8]; hipMemcpyFromSymbol(rs, HIP_SYMBOL(g_reason), 32); printf("slow-path reasons 1..7: %d %d %d %d %d %d %d\n", rs[1], rs[2], rs[3], rs[4], rs[5], rs[6], rs[7]);
		int un = 0; hipMemcpyFromSymbol(&un, HIP_SYMBOL(g_unc), 4);
		printf("%s: %d rows x %d %.2f ms (fallbacks so far %d, rank counts so far %d, of which with uncertain pairs %d)\n", VARIANT, rows, cols, ms, fb, it, un);
	}
	return 0;
}
