// ubench_issue.hip -- issue rates of VALU / SALU / mixed streams on gfx950 (cycles per wave-instruction per SIMD at
// 1, 2, 4, 5 waves per SIMD, every CU busy): what bounds an instruction-count-bound kernel such as k_theil_sen.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_IT 2048
#define REP8(x) x x x x x x x x
// clk[0..1]: s_memtime ticks and constant-rate (100 MHz) ticks that workgroup 0 spent in the loop: if s_memtime counts shader
// clocks their ratio is the engine clock under this load, if it counts the reference clock too the ratio is 1 (then unknown)
__device__ unsigned long long g_clk[2];
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, float seed)
{
	float a = threadIdx.x * seed, b = seed, c = seed * 3.f;
	uint32_t r = 0;
	const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
	#pragma unroll 1
	for (int i = 0; i < N_IT; ++i) {
		if (OP == 0) asm volatile(REP8("v_add_f32 %0, %1, %0\n v_add_f32 %2, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c));              // 16 independent-ish VALU
		if (OP == 1) asm volatile(REP8("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 3\n") ::: "s20", "s21", "scc");               // 16 SALU
		if (OP == 2) asm volatile(REP8("v_add_f32 %0, %1, %0\n s_add_u32 s20, s20, 1\n") : "+v"(a), "+v"(b) :: "s20", "scc");      // 8 VALU + 8 SALU interleaved
		if (OP == 3) asm volatile(REP8("v_cmp_lt_f32 s[22:23], %0, %1\n v_cmp_gt_f32 s[24:25], %0, %1\n") : "+v"(a), "+v"(b) :: "s22", "s23", "s24", "s25");   // 16 v_cmp to SGPR
		if (OP == 4) asm volatile(REP8("v_cmp_lt_f32 vcc, %0, %1\n s_bcnt1_i32_b64 s20, vcc\n s_add_u32 s21, s21, s20\n") : "+v"(a), "+v"(b) :: "s20", "s21", "vcc", "scc");   // cmp -> bcnt -> add chain
		if (OP == 5) asm volatile(REP8("v_writelane_b32 %0, s20, 5\n s_add_u32 s20, s20, 1\n") : "+v"(r) :: "s20", "scc");         // writelane + salu
		if (OP == 6) asm volatile(REP8("s_ff1_i32_b64 s20, s[22:23]\n s_bitset0_b64 s[22:23], s20\n s_or_b32 s21, s21, s20\n") ::: "s20", "s21", "s22", "s23", "scc");
		if (OP == 7) asm volatile(REP8("v_pk_add_f32 %0, %1, %0\n v_pk_mul_f32 %2, %1, %2\n") : "+v"(*(double *)&a), "+v"(*(double *)&b), "+v"(*(double *)&c));
		if (OP == 8) asm volatile(REP8("v_cmp_lt_f32 vcc, %0, %1\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n") : "+v"(a), "+v"(b), "+v"(r) :: "vcc");
		if (OP == 9) asm volatile(REP8("v_mbcnt_lo_u32_b32 %0, s20, 0\n v_mbcnt_hi_u32_b32 %0, s21, %0\n") : "+v"(r) :: "s20", "s21");
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		g_clk[0] = __builtin_readcyclecounter() - t0;
		g_clk[1] = wall_clock64() - w0;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)a + (uint32_t)b + (uint32_t)c + r;
}
template <int OP> void run(const char *name, int per_iter)
{
	uint32_t *out; hipMalloc(&out, 2048 * 256 * 4);
	for (int wps : {1, 2, 4, 5}) {
		hipLaunchKernelGGL(k<OP>, dim3(256 * wps), dim3(256), 0, 0, out, 1.5f);
		hipDeviceSynchronize();
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0);
		hipLaunchKernelGGL(k<OP>, dim3(256 * wps), dim3(256), 0, 0, out, 1.5f);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		double n = (double)N_IT * per_iter * wps;      // wave-instructions per SIMD
		unsigned long long clk[2] = { 0, 0 };
		(void)hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
		const double ghz = clk[1] ? 0.1 * (double)clk[0] / (double)clk[1] : 0.0;   // wall_clock64 ticks at 100 MHz
		printf("%-34s waves/SIMD=%d  ns/instr/SIMD=%.3f  (cycles at 2.4 GHz: %.2f; s_memtime / 100 MHz ticks = %.3f GHz%s)\n", name, wps,
			ms * 1e6 / n, ms * 1e6 / n * 2.4, ghz, ghz < 0.2 ? ": s_memtime is the reference clock here, engine clock not observable" : "");
	}
	hipFree(out);
}
int main()
{
	run<0>("v_add_f32", 16);
	run<7>("v_pk_add/mul_f32", 16);
	run<1>("s_add_u32", 16);
	run<2>("v_add + s_add interleaved", 16);
	run<3>("v_cmp -> sgpr", 16);
	run<4>("v_cmp vcc, s_bcnt1, s_add", 24);
	run<8>("v_cmp vcc, v_addc", 16);
	run<5>("v_writelane + s_add", 16);
	run<6>("s_ff1, s_bitset0, s_or", 24);
	run<9>("v_mbcnt lo/hi", 16);
	return 0;
}
