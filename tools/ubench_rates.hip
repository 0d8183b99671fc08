// ubench_rates.hip -- issue-rate microbenchmark for the integer/cross-lane instructions the
// OSD and polar kernels lean on (gfx950).  Prints cycles per wave-instruction at 1 and 4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define N_IT 4096
template <int OP>
__global__ void k(uint32_t *out, uint32_t seed, long long *cyc)
{
	uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x1234567u, a2 = a0 + 77, a3 = a0 * 3, b = seed | 1, c = seed * 7 + 3;
	__shared__ uint32_t lds[4096];
	for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * seed;
	__syncthreads();
	long long t0 = clock64();
	#pragma unroll 1
	for (int i = 0; i < N_IT; ++i) {
		#pragma unroll
		for (int u = 0; u < 8; ++u) {
			if (OP == 0) { a0 = __popc(a0 & b) + a0; a1 = __popc(a1 & b) + a1; a2 = __popc(a2 & b) + a2; a3 = __popc(a3 & b) + a3; }            // and + bcnt(acc)
			if (OP == 1) { a0 = (a0 ^ b) & c; a1 = (a1 ^ b) & c; a2 = (a2 ^ b) & c; a3 = (a3 ^ b) & c; a0 += i; }                                // bitop3 (xor-and)
			if (OP == 2) { a0 = a0 + b + c; a1 = a1 + b + c; a2 = a2 + b + c; a3 = a3 + b + c; }                                                // add3
			if (OP == 3) { a0 = __builtin_amdgcn_sdot4((int)a0, (int)b, (int)a0, false); a1 = __builtin_amdgcn_sdot4((int)a1, (int)b, (int)a1, false);
			               a2 = __builtin_amdgcn_sdot4((int)a2, (int)c, (int)a2, false); a3 = __builtin_amdgcn_sdot4((int)a3, (int)c, (int)a3, false); }  // dot4 i8
			if (OP == 4) { a0 = __shfl((int)a0, (a0 + i) & 63); a1 = __shfl((int)a1, (a1 + i) & 63); a2 = __shfl((int)a2, (a2 + 1) & 63); a3 = __shfl((int)a3, (a3 + 2) & 63); } // ds_bpermute
			if (OP == 5) { a0 = lds[(a0 + i) & 4095]; a1 = lds[(a1 + i) & 4095]; a2 = lds[(a2 + i) & 4095]; a3 = lds[(a3 + i) & 4095]; }       // random ds_read_b32
			if (OP == 6) { a0 = __shfl_xor((int)a0, 32) + 1; a1 = __shfl_xor((int)a1, 16) + 1; a2 = __shfl_xor((int)a2, 8) + 1; a3 = __shfl_xor((int)a3, 32) + 1; }  // xor shuffles
			if (OP == 7) { float f0 = __uint_as_float(a0), f1 = __uint_as_float(a1); f0 = fminf(fabsf(f0), fabsf(f1)); a0 = __float_as_uint(f0) ^ ((a0 ^ a1) & 0x80000000u);
			               float f2 = __uint_as_float(a2), f3 = __uint_as_float(a3); f2 = fminf(fabsf(f2), fabsf(f3)); a2 = __float_as_uint(f2) ^ ((a2 ^ a3) & 0x80000000u); a1 += i; a3 += i; } // min-sum f
			if (OP == 8) { a0 = __builtin_amdgcn_readlane((int)a0, 3) + a1; a1 = __builtin_amdgcn_readlane((int)a1, 5) + a2; a2 = __builtin_amdgcn_readlane((int)a2, 7) + a3; a3 = __builtin_amdgcn_readlane((int)a3, 9) + a0; } // readlane
		}
	}
	long long t1 = clock64();
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
	if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP> void run(const char *name, int per_iter)
{
	uint32_t *out; long long *cyc, h;
	hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8);
	for (int wps : {1, 4}) {   // waves per SIMD: block of 256 = 1 wave per SIMD; grid 256 CUs x wps
		hipLaunchKernelGGL(k<OP>, dim3(256 * wps), dim3(256), 0, 0, out, 12345u, cyc);
		hipDeviceSynchronize();
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0);
		hipLaunchKernelGGL(k<OP>, dim3(256 * wps), dim3(256), 0, 0, out, 12345u, cyc);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
		double n = (double)N_IT * 8 * per_iter;
		printf("%-28s waves/SIMD=%d  clock64/instr(one wave)=%.2f  ns/instr/SIMD=%.3f\n", name, wps, (double)h / n, ms * 1e6 / (n * wps));
	}
}
int main()
{
	run<0>("and+bcnt(acc) pair", 4);
	run<1>("xor-and (bitop3)", 4);
	run<2>("add3", 4);
	run<3>("sdot4 i8", 4);
	run<4>("ds_bpermute (var idx)", 4);
	run<5>("ds_read_b32 random", 4);
	run<6>("shfl_xor 32/16/8 + add", 4);
	run<7>("min-sum f (pair)", 2);
	run<8>("readlane + add", 4);
	return 0;
}
