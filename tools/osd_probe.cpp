// osd_probe.cpp -- timing probe for the OSD kernel variants (tools only)
#include "../modem_amd/csrc/k_header.hip"
#include "../modem_amd/csrc/tables.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
using namespace rx;
template <typename T> const T *up(const std::vector<T> &v) { void *p; hipMalloc(&p, v.size() * sizeof(T)); hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); return (const T *)p; }
int main()
{
	HostTables h; build_tables(h, 8000);
	Tables t{}; t.genmat_bits = up(h.genmat_bits); t.osd_pairs = up(h.osd_pairs); t.osd_triples = up(h.osd_triples);
	const int n = getenv("PROBE_N") ? atoi(getenv("PROBE_N")) : 4096;
	std::mt19937 rng(1); std::vector<int8_t> soft(n * 255);
	const char *mode = getenv("PROBE_DATA");
	for (size_t i = 0; i < soft.size(); ++i) {
		int v = (int)(rng() % 200) - 100;
		if (mode && mode[0] == 'c') v = 127;                                  // clean all-zero codeword, saturated
		if (mode && mode[0] == 'n') v = 100 + (int)(rng() % 28);              // all-zero codeword, magnitudes 100..127
		soft[i] = (int8_t)v;
	}
	const int8_t *ds = up(soft); uint8_t *hard; int32_t *uq; hipMalloc(&hard, n * 32); hipMalloc(&uq, n * 4);
	for (int rep = 0; rep < 2; ++rep) {
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a);
		launch_osd_only(0, n, t, ds, hard, uq);
		hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
		printf("%s: %d frames %.2f ms\n", VARIANT, n, ms);
	}
	return 0;
}
