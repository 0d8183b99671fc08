/*
 * oracle/dsp.c -- TEST INFRASTRUCTURE (see modem_oracle.h header).
 * PSK mapper/demapper (psk.hh, in-tree, pinned by tests/golden/psk_vectors.json)
 * and the DSP helpers taken from the absent aicodix/dsp headers.
 */
#include "modem_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- psk.hh:90-140  PhaseShiftKeying<8, cmplx, float> -------------------- */
static const float cos_pi_8 = 0.92387953251128675613f;   /* psk.hh:100 */
static const float sin_pi_8 = 0.38268343236508977173f;   /* psk.hh:102 */
static const float rcp_sqrt_2 = 0.70710678118654752440f; /* psk.hh:104 */

/* psk.hh:108-116 quantize, code_type=float: no rounding, no clamp */
static inline float quantize(float dist, float precision, float value)
{
	value *= dist * precision;
	return value;
}

void orc_psk8_hard(float *b, orc_cf c)   /* psk.hh:118-123 */
{
	b[1] = c.re < 0.f ? -1.f : 1.f;
	b[2] = c.im < 0.f ? -1.f : 1.f;
	b[0] = fabsf(c.re) < fabsf(c.im) ? -1.f : 1.f;
}
void orc_psk8_soft(float *b, orc_cf c, float precision)   /* psk.hh:125-130 */
{
	const float DIST = 2 * sin_pi_8;   /* psk.hh:106 */
	b[1] = quantize(DIST, precision, c.re);
	b[2] = quantize(DIST, precision, c.im);
	b[0] = quantize(DIST, precision, rcp_sqrt_2 * (fabsf(c.re) - fabsf(c.im)));
}
orc_cf orc_psk8_map(const float *b)   /* psk.hh:132-139 */
{
	float real = cos_pi_8, imag = sin_pi_8;
	if (b[0] < 0.f) { float t = real; real = imag; imag = t; }
	orc_cf r = { real * b[1], imag * b[2] };
	return r;
}
/* ---- psk.hh:49-88  PhaseShiftKeying<4, cmplx, float> --------------------- */
void orc_psk4_hard(float *b, orc_cf c)   /* psk.hh:70-74 */
{
	b[0] = c.re < 0.f ? -1.f : 1.f;
	b[1] = c.im < 0.f ? -1.f : 1.f;
}
void orc_psk4_soft(float *b, orc_cf c, float precision)   /* psk.hh:76-80 */
{
	const float DIST = 2 * rcp_sqrt_2;   /* psk.hh:59 */
	b[0] = quantize(DIST, precision, c.re);
	b[1] = quantize(DIST, precision, c.im);
}
orc_cf orc_psk4_map(const float *b)   /* psk.hh:82-85 */
{
	orc_cf r = { rcp_sqrt_2 * b[0], rcp_sqrt_2 * b[1] };
	return r;
}

/* ---- DSP::TheilSenEstimator<value,512> (decode.cc:195,488-494) ----------- */
/* value at sorted position k (what std::nth_element leaves at temp[k]) */
static float select_kth(float *a, int n, int k)
{
	int lo = 0, hi = n - 1;
	while (lo < hi) {
		/* median-of-three pivot */
		int mid = lo + (hi - lo) / 2;
		float x = a[lo], y = a[mid], z = a[hi];
		float pivot = (x < y) ? ((y < z) ? y : (x < z ? z : x)) : ((x < z) ? x : (y < z ? z : y));
		int i = lo, j = hi;
		while (i <= j) {
			while (a[i] < pivot) ++i;
			while (pivot < a[j]) --j;
			if (i <= j) { float t = a[i]; a[i] = a[j]; a[j] = t; ++i; --j; }
		}
		if (k <= j) hi = j;
		else if (k >= i) lo = i;
		else return a[k];
	}
	return a[k];
}

void orc_theil_sen(const float *x, const float *y, int n, float *slope, float *yint)
{
	/* all i<j pairs with x[j]!=x[i]; nth_element at count/2 (upper median);
	 * then intercepts y-slope*x with the same rule */
	size_t cap = (size_t)n * (size_t)(n - 1) / 2 + 1;
	float *temp = (float *)malloc(sizeof(float) * (cap > (size_t)n ? cap : (size_t)n + 1));
	int count = 0;
	for (int i = 0; i < n; ++i)
		for (int j = i + 1; j < n; ++j)
			if (x[j] != x[i])
				temp[count++] = (y[j] - y[i]) / (x[j] - x[i]);
	float s = 0.f;
	if (count)
		s = select_kth(temp, count, count / 2);
	count = 0;
	for (int i = 0; i < n; ++i)
		temp[count++] = y[i] - s * x[i];
	float yi = 0.f;
	if (count)
		yi = select_kth(temp, count, count / 2);
	free(temp);
	*slope = s;
	*yint = yi;
}

/* ---- DSP::Hilbert<cmplx,21> (decode.cc:172,193,299) ---------------------- */
/* Kaiser(a=2)-windowed ideal Hilbert transformer; only odd taps non-zero. */
static double bessel_i0(double x)
{
	double sum = 1.0, term = 1.0;
	for (int k = 1; k < 64; ++k) {
		term *= (x / (2.0 * k)) * (x / (2.0 * k));
		sum += term;
		if (term < 1e-20 * sum)
			break;
	}
	return sum;
}
static double kaiser(double a, int n, int N)
{
	double t = 2.0 * n / (double)(N - 1) - 1.0;
	return bessel_i0(M_PI * a * sqrt(1.0 - t * t)) / bessel_i0(M_PI * a);
}
void orc_hilbert_coeffs(float *reco, float *imco)
{
	orc_hilbert_coeffs_n(ORC_FILTER_LEN, reco, imco);
}
void orc_hilbert_coeffs_n(int TAPS, float *reco, float *imco)
{
	*reco = (float)kaiser(2.0, (TAPS - 1) / 2, TAPS);
	for (int i = 0; i < (TAPS - 1) / 4; ++i)
		imco[i] = (float)(kaiser(2.0, (2 * i + 1) + (TAPS - 1) / 2, TAPS) * 2.0 / ((2 * i + 1) * M_PI));
}

/* ---- D0 + D1: ReadWAV scaling, BlockDC, Hilbert (decode.cc:294-301,386) -- */
static inline float sample_at(const void *p, int fmt, size_t idx)
{
	switch (fmt) {
	case ORC_FMT_S16: return (float)((const int16_t *)p)[idx] / 32767.f;
	case ORC_FMT_U8: return (float)((int)((const uint8_t *)p)[idx] - 128) / 127.f;
	default: return ((const float *)p)[idx];
	}
}

void orc_front_end(const void *samples, int fmt, int channels, size_t n, orc_cf *z)
{
	orc_front_end_rate(ORC_RATE, samples, fmt, channels, n, z);
}

void orc_front_end_rate(int rate, const void *samples, int fmt, int channels, size_t n, orc_cf *z)
{
	orc_rate_cfg rc;
	if (!orc_rate_lookup(rate, &rc))
		orc_rate_lookup(ORC_RATE, &rc);
	if (channels == 2) {
		/* decode.cc:297-298: the two channels are taken as (re, im) */
		for (size_t i = 0; i < n; ++i) {
			z[i].re = sample_at(samples, fmt, 2 * i);
			z[i].im = sample_at(samples, fmt, 2 * i + 1);
		}
		return;
	}
	/* decode.cc:299: tmp = hilbert(blockdc(tmp.real())) */
	float reco, imco[32];
	const int NIM = (rc.filter_len - 1) / 4;
	orc_hilbert_coeffs_n(rc.filter_len, &reco, imco);
	/* BlockDC::samples(2*(symbol_len+guard_len)) decode.cc:386 */
	const float s = 2 * (rc.symbol_len + rc.guard_len);
	const float a = (s - 1.f) / s, b = (1.f + a) / 2.f;
	float *dc = (float *)malloc(sizeof(float) * (n + 1));
	if (orc_get_numerics() & ORC_NUM_BLOCKDC_FP32) {
		/* the plain form: the recurrence in fp32, as a scalar build of the reference runs it */
		float x1 = 0.f, y1 = 0.f;
		for (size_t i = 0; i < n; ++i) {
			float x0 = sample_at(samples, fmt, i);
			float y0 = b * (x0 - x1) + a * y1;
			x1 = x0;
			y1 = y0;
			dc[i] = y0;
		}
	} else {
		/* default (round 6): the same recurrence with its state in double, every output rounded to fp32 once.  The fp32 recurrence
		 * feeds its own rounding back through a = 1 - 1 / 2880: its outputs walk about 3e-6 of their magnitude away from the exact ones,
		 * in an order -Ofast does not fix; a blocked scan (the GPU's front end) cannot follow that walk, and two sides 1e-6 apart pick
		 * neighbouring samples where the Schmidl-Cox arg-max sits on a plateau.  Both sides now round the exact value. */
		double x1 = 0.0, y1 = 0.0;
		for (size_t i = 0; i < n; ++i) {
			const double x0 = (double)sample_at(samples, fmt, i);
			const double y0 = (double)b * (x0 - x1) + (double)a * y1;
			x1 = x0;
			y1 = y0;
			dc[i] = (float)y0;
		}
	}
	const int C = (rc.filter_len - 1) / 2;
	for (size_t i = 0; i < n; ++i) {
		/* delay line holds dc[i-20..i]; centre tap = dc[i-10] */
		#define DC(k) (((long)(k) >= 0) ? dc[(k)] : 0.f)
		long c = (long)i - C;
		float re = reco * DC(c);
		float im = imco[0] * (DC(c - 1) - DC(c + 1));
		for (int k = 1; k < NIM; ++k)
			im += imco[k] * (DC(c - (2 * k + 1)) - DC(c + (2 * k + 1)));
		#undef DC
		z[i].re = re;
		z[i].im = im;
	}
	free(dc);
}
