/*
 * oracle/channel.c -- TEST INFRASTRUCTURE (see modem_oracle.h header).
 * Build-owned channel models.  The reference's README.md:44-49 pipes through
 * aicodix/disorders (multipath, cfo, sfo, awgn), a third repository that is
 * absent; the definitions below are this build's own:
 * "awgn LEVEL" adds complex Gaussian noise of power 10^(LEVEL/10) relative to
 * full scale 1.0, split equally between re and im.
 */
#include "modem_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t splitmix64(uint64_t x)
{
	x += 0x9e3779b97f4a7c15ull;
	x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
	x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
	return x ^ (x >> 31);
}

/* counter-based: sample i of frame f under seed s */
void orc_chan_awgn(orc_cf *z, size_t n, float noise_db, uint64_t seed, uint64_t frame)
{
	const float sigma = sqrtf(0.5f * powf(10.f, noise_db / 10.f));
	const uint64_t key = splitmix64(seed ^ splitmix64(frame + 0x1234567ull));
	for (size_t i = 0; i < n; ++i) {
		uint64_t r = splitmix64(key + (uint64_t)i);
		float u1 = ((float)(uint32_t)(r >> 40) + 0.5f) * (1.f / 16777216.f);
		float u2 = ((float)(uint32_t)((r >> 8) & 0xffffff) + 0.5f) * (1.f / 16777216.f);
		float mag = sigma * sqrtf(-2.f * logf(u1));
		z[i].re += mag * cosf(6.28318530717958647692f * u2);
		z[i].im += mag * sinf(6.28318530717958647692f * u2);
	}
}

void orc_chan_cfo(orc_cf *z, size_t n, float hz, int rate)
{
	for (size_t i = 0; i < n; ++i) {
		double a = 2.0 * M_PI * (double)hz * (double)i / (double)rate;
		float c = (float)cos(a), s = (float)sin(a);
		orc_cf v = z[i];
		z[i].re = v.re * c - v.im * s;
		z[i].im = v.re * s + v.im * c;
	}
}

void orc_chan_sfo(orc_cf *out, const orc_cf *in, size_t n, float ppm)
{
	const int HALF = 16;
	const double step = 1.0 + (double)ppm * 1e-6;
	for (size_t i = 0; i < n; ++i) {
		double t = (double)i * step;
		long t0 = (long)floor(t);
		double fr = t - (double)t0;
		double re = 0.0, im = 0.0;
		for (int k = -HALF + 1; k <= HALF; ++k) {
			long idx = t0 + k;
			if (idx < 0 || (size_t)idx >= n)
				continue;
			double x = (double)k - fr;
			double sinc = fabs(x) < 1e-12 ? 1.0 : sin(M_PI * x) / (M_PI * x);
			double w = 0.5 * (1.0 + cos(M_PI * x / (double)HALF));   /* Hann */
			re += sinc * w * in[idx].re;
			im += sinc * w * in[idx].im;
		}
		out[i].re = (float)re;
		out[i].im = (float)im;
	}
}

void orc_chan_multipath(orc_cf *out, const orc_cf *in, size_t n,
	const int *delays, const orc_cf *gains, int ntaps)
{
	for (size_t i = 0; i < n; ++i) {
		float re = 0.f, im = 0.f;
		for (int t = 0; t < ntaps; ++t) {
			long idx = (long)i - delays[t];
			if (idx < 0)
				continue;
			re += in[idx].re * gains[t].re - in[idx].im * gains[t].im;
			im += in[idx].re * gains[t].im + in[idx].im * gains[t].re;
		}
		out[i].re = re;
		out[i].im = im;
	}
}
