// k_channel.hip -- N3: build-owned channel models on the device (AWGN tile; multipath -> CFO -> SFO chain).
#include "dev_common.h"
#include "kernels.h"

namespace rx {

// ---------------------------------------------------------------- channel model utility
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
	x += 0x9e3779b97f4a7c15ull;
	x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
	x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
	return x ^ (x >> 31);
}
// out frame f = base[f % n_base] + complex AWGN(sigma per component); counter-based RNG
__global__ __launch_bounds__(256) void k_awgn_tile(const short2 *__restrict__ base, size_t n_base, short2 *__restrict__ out,
	size_t spf, float sigma, unsigned long long seed, unsigned long long first_frame)
{
	const size_t f = blockIdx.x;
	const unsigned long long key = splitmix64(seed ^ splitmix64(first_frame + f + 0x1234567ull));
	const short2 *src = base + (f % n_base) * spf;
	short2 *dst = out + f * spf;
	for (size_t i = (size_t)blockIdx.y * 256 + threadIdx.x; i < spf; i += (size_t)gridDim.y * 256) {
		unsigned long long r = splitmix64(key + i);
		float u1 = ((float)(unsigned)(r >> 40) + 0.5f) * (1.f / 16777216.f);
		float u2 = ((float)(unsigned)((r >> 8) & 0xffffff) + 0.5f) * (1.f / 16777216.f);
		float mag = sigma * sqrtf(-2.f * logf(u1));
		float sn, cs;
		sincosf(TWO_PI_F * u2, &sn, &cs);
		short2 v = src[i];
		float re = div_32767((float)v.x) + mag * cs, im = div_32767((float)v.y) + mag * sn;
		re = fminf(fmaxf(re, -1.f), 1.f);
		im = fminf(fmaxf(im, -1.f), 1.f);
		dst[i] = make_short2((short)nearbyintf(32767.f * re), (short)nearbyintf(32767.f * im));
	}
}

// grid = number of resident decoders (each needs 2 MiB of `soft`); 0 or >= n: one per codeword

void launch_awgn_tile(hipStream_t s, const int16_t *base, size_t n_base, int16_t *out, size_t n_out,
	size_t spf, float sigma, uint64_t seed, uint64_t first_frame)
{
	hipLaunchKernelGGL(k_awgn_tile, dim3((unsigned)n_out, 64), dim3(256), 0, s, (const short2 *)base, n_base, (short2 *)out,
		spf, sigma, (unsigned long long)seed, (unsigned long long)first_frame);
}

}  // namespace rx

// ---------------------------------------------------------------- build-owned channel chain (N3)
// README.md:49 pipes encode through aicodix/disorders: multipath | cfo | sfo | awgn.  That repository is
// absent; the definitions here are this build's own (same as oracle/channel.c, checked against it):
//   multipath: FIR with integer delays and complex gains;   cfo: x[m] * e^{j 2 pi hz m / rate};
//   sfo: out[i] = resample at t = i (1 + ppm 1e-6), 32-tap Hann-windowed sinc;   awgn: k_awgn_tile.
// 2-channel int16 in and out.  The chain is deterministic, so it is applied to the base frames once
// and k_awgn_tile then adds independent noise per frame.
namespace rx {

struct ChannelParams {
	float cfo_hz, sfo_ppm;
	int ntaps;
	int delays[8];
	float gre[8], gim[8];
};

__global__ __launch_bounds__(256) void k_channel(const short2 *__restrict__ in, short2 *__restrict__ out, size_t spf, ChannelParams cp, int rate)
{
	const size_t f = blockIdx.y;
	const short2 *src = in + f * spf;
	short2 *dst = out + f * spf;
	const double step = 1.0 + (double)cp.sfo_ppm * 1e-6;
	const double w0 = 2.0 * 3.14159265358979323846 * (double)cp.cfo_hz / (double)rate;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < spf; i += (size_t)gridDim.x * 256) {
		auto stage12 = [&](long m) -> cf {   // multipath then cfo at integer sample m
			float re = 0.f, im = 0.f;
			for (int t = 0; t < cp.ntaps; ++t) {
				long idx = m - cp.delays[t];
				if (idx < 0 || (size_t)idx >= spf)
					continue;
				short2 v = src[idx];
				float xr = div_32767((float)v.x), xi = div_32767((float)v.y);
				re += xr * cp.gre[t] - xi * cp.gim[t];
				im += xr * cp.gim[t] + xi * cp.gre[t];
			}
			if (cp.cfo_hz != 0.f) {
				double a = w0 * (double)m;
				float c = (float)cos(a), s = (float)sin(a);
				float r2 = re * c - im * s, i2 = re * s + im * c;
				re = r2; im = i2;
			}
			return mk(re, im);
		};
		float ore, oim;
		if (cp.sfo_ppm == 0.f) {
			cf v = stage12((long)i);
			ore = v.re; oim = v.im;
		} else {
			const int HALF = 16;
			double t = (double)i * step;
			long t0 = (long)floor(t);
			double fr = t - (double)t0, re = 0.0, im = 0.0;
			for (int k = -HALF + 1; k <= HALF; ++k) {
				long idx = t0 + k;
				if (idx < 0 || (size_t)idx >= spf)
					continue;
				double x = (double)k - fr;
				double sinc = fabs(x) < 1e-12 ? 1.0 : sin(3.14159265358979323846 * x) / (3.14159265358979323846 * x);
				double w = 0.5 * (1.0 + cos(3.14159265358979323846 * x / (double)HALF));
				cf v = stage12(idx);
				re += sinc * w * v.re;
				im += sinc * w * v.im;
			}
			ore = (float)re; oim = (float)im;
		}
		ore = fminf(fmaxf(ore, -1.f), 1.f);
		oim = fminf(fmaxf(oim, -1.f), 1.f);
		dst[i] = make_short2((short)nearbyintf(32767.f * ore), (short)nearbyintf(32767.f * oim));
	}
}

void launch_channel(hipStream_t s, int rate, const int16_t *in, int16_t *out, size_t n, size_t spf, const void *params)
{
	ChannelParams cp = *(const ChannelParams *)params;
	hipLaunchKernelGGL(k_channel, dim3(128, (unsigned)n), dim3(256), 0, s, (const short2 *)in, (short2 *)out, spf, cp, rate);
}

}  // namespace rx
