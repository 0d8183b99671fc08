// ubench_issue.hip -- issue rate of vector-instruction streams on gfx950 with the occupancy VERIFIED (round 6; the round-5 form
// had two dependent chains per wave and let the dispatcher place its workgroups, so its "waves/SIMD" labels were not what ran).
//   * eight independent accumulators per wave and stream: no instruction waits for its predecessor's result
//   * the instruction kinds of k_theilsen.hip / k_demod.hip: v_fma_f32, v_pk_fma_f32, v_min_u32 / v_max_u32, v_med3_u32, DPP moves,
//     DPP move + v_med3 pairs (a cross-lane stage of the sort), v_fma_f64 (the keys), and round 5's two-chain v_add_f32
//   * placement: every workgroup asks for so much LDS that exactly WG_PER_CU fit a CU, the grid is WG_PER_CU x (number of CUs),
//     every workgroup records (XCC, SE, CU) from HW_ID / XCC_ID and its s_memtime span; the host prints how many distinct CUs ran
//     how many workgroups AT THE SAME TIME, so "w waves per SIMD on every CU" is a checked statement
//   * ns per wave-instruction per SIMD from hipEvents (no clock assumed) and cycles from s_memtime inside the kernel
// Run:  hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o /tmp/ubench_issue && /tmp/ubench_issue
// Under rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... the kernels k<OP> calibrate what those
// counters read for a stream whose rate is known (tools/profile_round.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>
#define N_IT 4096
struct WgRec { unsigned long long t0, t1, w0, w1; uint32_t hw_id, xcc_id; };   // s_memtime span, 100 MHz wall-clock span

#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int OP>
__global__ void k(uint32_t *out, WgRec *rec, float seed)
{
	extern __shared__ uint32_t pad[];
	float a[8], b = seed, c = seed * 0.5f;
	uint32_t u[8], m = (uint32_t)threadIdx.x * 2654435761u;
	double d[8], e = (double)seed;
	#pragma unroll
	for (int i = 0; i < 8; ++i) {
		a[i] = threadIdx.x * seed + i;
		u[i] = (uint32_t)threadIdx.x * 977u + i;
		d[i] = (double)a[i];
	}
	uint32_t hw = 0, xcc = 0;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
	__syncthreads();
	const unsigned long long w0 = wall_clock64(), t0 = __builtin_readcyclecounter();
	#pragma unroll 1
	for (int i = 0; i < N_IT; ++i) {
		// 16 instructions per iteration in every stream
		if (OP == 0) asm volatile(
			"v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
			"v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
			"v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
			"v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));
		if (OP == 1) asm volatile(                                    // 8 packed pairs = 16 fp32 FMAs per 8 instructions; 16 instructions
			"v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
			"v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
			"v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
			"v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
			: "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(d[4]), "v"(d[5]));
		if (OP == 2) asm volatile(
			"v_min_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_max_u32 %3, %3, %8\n"
			"v_min_u32 %4, %4, %8\n v_max_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_max_u32 %7, %7, %8\n"
			"v_max_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_max_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n"
			"v_max_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_max_u32 %6, %6, %8\n v_min_u32 %7, %7, %8\n"
			: "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(m));
		if (OP == 3) asm volatile(
			"v_med3_u32 %0, %0, %8, %9\n v_med3_u32 %1, %1, %8, %9\n v_med3_u32 %2, %2, %8, %9\n v_med3_u32 %3, %3, %8, %9\n"
			"v_med3_u32 %4, %4, %8, %9\n v_med3_u32 %5, %5, %8, %9\n v_med3_u32 %6, %6, %8, %9\n v_med3_u32 %7, %7, %8, %9\n"
			"v_med3_u32 %0, %0, %8, %9\n v_med3_u32 %1, %1, %8, %9\n v_med3_u32 %2, %2, %8, %9\n v_med3_u32 %3, %3, %8, %9\n"
			"v_med3_u32 %4, %4, %8, %9\n v_med3_u32 %5, %5, %8, %9\n v_med3_u32 %6, %6, %8, %9\n v_med3_u32 %7, %7, %8, %9\n"
			: "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(m), "v"(hw));
		if (OP == 4) asm volatile(                                    // DPP moves, eight independent registers
			"v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
			"v_mov_b32_dpp %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n"
			"v_mov_b32_dpp %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
			"v_mov_b32_dpp %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf\n"
			"v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
			"v_mov_b32_dpp %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n"
			"v_mov_b32_dpp %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
			"v_mov_b32_dpp %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf\n"
			: "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]));
		if (OP == 5) asm volatile(                                    // a cross-lane stage of the sort: partner by DPP, then v_med3 (own, partner, bound)
			"v_mov_b32_dpp %8, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
			"v_mov_b32_dpp %10, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %11, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
			"v_med3_u32 %0, %0, %8, %12\n v_med3_u32 %1, %1, %9, %12\n v_med3_u32 %2, %2, %10, %12\n v_med3_u32 %3, %3, %11, %12\n"
			"v_mov_b32_dpp %8, %4 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %5 row_mirror row_mask:0xf bank_mask:0xf\n"
			"v_mov_b32_dpp %10, %6 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %11, %7 row_mirror row_mask:0xf bank_mask:0xf\n"
			"v_med3_u32 %4, %4, %8, %12\n v_med3_u32 %5, %5, %9, %12\n v_med3_u32 %6, %6, %10, %12\n v_med3_u32 %7, %7, %11, %12\n"
			: "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]),
			  "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(m));
		if (OP == 6) asm volatile(
			"v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
			"v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
			"v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
			"v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
			: "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(e), "v"(e));
		if (OP == 7) asm volatile(                                    // round 5's stream: two chains, each instruction waits for the one two back
			"v_add_f32 %0, %2, %0\n v_add_f32 %1, %2, %1\n v_add_f32 %0, %2, %0\n v_add_f32 %1, %2, %1\n"
			"v_add_f32 %0, %2, %0\n v_add_f32 %1, %2, %1\n v_add_f32 %0, %2, %0\n v_add_f32 %1, %2, %1\n"
			"v_add_f32 %0, %2, %0\n v_add_f32 %1, %2, %1\n v_add_f32 %0, %2, %0\n v_add_f32 %1, %2, %1\n"
			"v_add_f32 %0, %2, %0\n v_add_f32 %1, %2, %1\n v_add_f32 %0, %2, %0\n v_add_f32 %1, %2, %1\n"
			: "+v"(a[0]), "+v"(a[1]) : "v"(b));
		if (OP == 8) asm volatile(                                    // the mix of a sort stage with its bookkeeping: 4 DPP, 4 med3, 4 min/max, 2 fma, 2 scalar
			"v_mov_b32_dpp %8, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
			"v_min_u32 %4, %4, %12\n v_max_u32 %5, %5, %12\n s_add_u32 s20, s20, 1\n"
			"v_mov_b32_dpp %10, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %11, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
			"v_med3_u32 %0, %0, %8, %12\n v_med3_u32 %1, %1, %9, %12\n v_min_u32 %6, %6, %12\n v_max_u32 %7, %7, %12\n s_and_b32 s21, s20, 7\n"
			"v_med3_u32 %2, %2, %10, %12\n v_med3_u32 %3, %3, %11, %12\n v_fma_f32 %8, %8, %8, %9\n v_fma_f32 %10, %10, %10, %11\n"
			: "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]),
			  "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(m) : "s20", "s21", "scc");
	}
	const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
	if (threadIdx.x == 0)
		rec[blockIdx.x] = WgRec{ t0, t1, w0, w1, hw, xcc };
	float s = 0.f;
	#pragma unroll
	for (int i = 0; i < 8; ++i)
		s += a[i] + (float)u[i] + (float)d[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s + pad[threadIdx.x & 15];
}

static int g_cus = 256;
template <int OP> void run(const char *name, int valu_per_iter)
{
	uint32_t *out; WgRec *rec;
	hipMalloc(&out, (size_t)2 * g_cus * 1024 * 4);
	hipMalloc(&rec, (size_t)2 * g_cus * sizeof(WgRec));
	hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
	// waves per SIMD w: one workgroup of 4 w waves per CU (w <= 4), two of 2 w (w = 6, 8); LDS such that no further workgroup fits
	struct Cfg { int w, wg_per_cu, threads, lds; };
	const Cfg cfgs[] = { { 1, 1, 256, 96 * 1024 }, { 2, 1, 512, 96 * 1024 }, { 3, 1, 768, 96 * 1024 }, { 4, 1, 1024, 96 * 1024 },
		{ 6, 2, 768, 72 * 1024 }, { 8, 2, 1024, 72 * 1024 } };
	for (const Cfg &c : cfgs) {
		const int grid = c.wg_per_cu * g_cus;
		hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(c.threads), c.lds, 0, out, rec, 1.0001f);
		hipDeviceSynchronize();
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0);
		hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(c.threads), c.lds, 0, out, rec, 1.0001f);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		std::vector<WgRec> h(grid);
		hipMemcpy(h.data(), rec, grid * sizeof(WgRec), hipMemcpyDeviceToHost);
		// placement: (xcc, se, sh, cu) of every workgroup; how many ran on the same CU with overlapping spans
		std::map<uint32_t, std::vector<int>> by_cu;
		double cyc_sum = 0, wall_sum = 0; unsigned long long cyc_max = 0;
		for (int i = 0; i < grid; ++i) {
			wall_sum += (double)(h[i].w1 - h[i].w0) * 10.0;          // ns: wall_clock64 ticks at 100 MHz
			const uint32_t key = ((h[i].xcc_id & 15u) << 16) | (h[i].hw_id & 0xff00u);   // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
			by_cu[key].push_back(i);
			const unsigned long long dt = h[i].t1 - h[i].t0;
			cyc_sum += (double)dt; cyc_max = std::max(cyc_max, dt);
		}
		int max_conc = 0, cus_with_wrong = 0;
		for (auto &kv : by_cu) {
			int conc = 0;
			for (int i : kv.second) {
				int o = 0;
				for (int j : kv.second)
					o += h[j].t0 < h[i].t1 && h[i].t0 < h[j].t1;
				conc = std::max(conc, o);
			}
			max_conc = std::max(max_conc, conc);
			cus_with_wrong += conc != c.wg_per_cu;
		}
		// per SIMD: a workgroup's waves issue N_IT x valu_per_iter x (waves per SIMD OF THAT WORKGROUP) instructions during its own span;
		// with two workgroups on a CU the spans overlap (checked below) and the SIMD issues both streams
		const double n_wg = (double)N_IT * valu_per_iter * c.w / c.wg_per_cu, n = n_wg * c.wg_per_cu;
		const double cyc = cyc_sum / grid, wall_wg = wall_sum / grid;
		printf("%-30s waves/SIMD=%d  ns/valu/SIMD: in-kernel %.3f, launch %.3f  cycles/valu/SIMD=%.2f (slowest workgroup %.2f)  clock=%.2f GHz  workgroup spans cover %.0f%% of the launch"
			"  | placement: %zu distinct CUs, %d workgroups at once on the fullest, %d CUs off the plan\n", name, c.w, wall_wg / n, ms * 1e6 / n, cyc / n, (double)cyc_max / n,
			wall_wg > 0 ? cyc / wall_wg : 0.0, 100.0 * wall_wg / (ms * 1e6), by_cu.size(), max_conc, cus_with_wrong);
	}
	hipFree(out); hipFree(rec);
}
int main()
{
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	g_cus = p.multiProcessorCount;
	printf("# %s, %d CUs; in-kernel = a workgroup's own span on the 100 MHz wall clock (wall_clock64) / the vector instructions its SIMD issued; launch = hipEvent time of the\n"
		"# launch / the same (includes the dispatch of the workgroups); clock = s_memtime ticks / wall-clock span of a workgroup\n", p.gcnArchName, g_cus);
	run<0>("v_fma_f32 x8 independent", 16);
	run<1>("v_pk_fma_f32 x4 independent", 16);
	run<2>("v_min/max_u32 x8", 16);
	run<3>("v_med3_u32 x8", 16);
	run<4>("v_mov_b32 dpp x8", 16);
	run<5>("dpp + v_med3 (sort stage)", 16);
	run<6>("v_fma_f64 x8", 16);
	run<7>("v_add_f32 two chains (r05)", 16);
	run<8>("sort-stage mix + 2 salu", 16);
	return 0;
}
