/*
 * oracle/fft.c -- TEST INFRASTRUCTURE (see modem_oracle.h header).
 *
 * Restates the contract of DSP::FastFourierTransform<N,cmplx,SIGN> (fft.hh is
 * ABSENT; contract from decode.cc:43-44,80-82,119-125,191,406,462,473 and
 * encode.cc:42-44,86-97,107-109): out-of-place, UNNORMALISED in both
 * directions (decode.cc:82 divides by N explicitly), natural order in/out,
 * SIGN=-1 forward e^{-j2pi kn/N}, SIGN=+1 backward.
 * Plain fp32 mixed-radix decimation-in-time (radix 4, 2, 5, 3, 7, generic <= 16) with
 * twiddles computed in double and rounded once.
 */
#include "modem_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAX_PLANS 16
typedef struct { int n; orc_cf *tw; } plan_t;
static plan_t plans[MAX_PLANS];
static int n_plans;

static const orc_cf *get_twiddles(int n)
{
	const orc_cf *found = NULL;
	#pragma omp critical(orc_fft_plan)
	{
		for (int i = 0; i < n_plans; ++i)
			if (plans[i].n == n)
				found = plans[i].tw;
		if (!found && n_plans < MAX_PLANS) {
			orc_cf *tw = (orc_cf *)malloc(sizeof(orc_cf) * (size_t)n);
			for (int k = 0; k < n; ++k) {
				double a = -2.0 * M_PI * (double)k / (double)n;
				tw[k].re = (float)cos(a);
				tw[k].im = (float)sin(a);
			}
			plans[n_plans].n = n;
			plans[n_plans].tw = tw;
			++n_plans;
			found = tw;
		}
	}
	return found;
}

static inline orc_cf cmul(orc_cf a, orc_cf b)
{
	orc_cf r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re };
	return r;
}
static inline orc_cf cadd(orc_cf a, orc_cf b) { orc_cf r = { a.re + b.re, a.im + b.im }; return r; }
static inline orc_cf csub(orc_cf a, orc_cf b) { orc_cf r = { a.re - b.re, a.im - b.im }; return r; }
/* multiply by sign*j */
static inline orc_cf cmulj(orc_cf a, int sign)
{
	orc_cf r;
	if (sign < 0) { r.re = a.im; r.im = -a.re; }   /* * (-j) */
	else { r.re = -a.im; r.im = a.re; }            /* * (+j) */
	return r;
}

typedef struct { const orc_cf *tw; int N; int sign; } ctx_t;

static inline orc_cf twid(const ctx_t *c, long idx)
{
	orc_cf w = c->tw[idx % c->N];
	if (c->sign > 0)
		w.im = -w.im;
	return w;
}

static void fft_rec(const ctx_t *c, orc_cf *out, const orc_cf *in, int n, int stride)
{
	if (n == 1) {
		out[0] = in[0];
		return;
	}
	int p = (n % 4 == 0) ? 4 : (n % 2 == 0) ? 2 : (n % 5 == 0) ? 5 : (n % 3 == 0) ? 3 : (n % 7 == 0) ? 7 : n;
	int m = n / p;
	for (int q = 0; q < p; ++q)
		fft_rec(c, out + q * m, in + q * stride, m, stride * p);
	long ts = c->N / n;     /* w_n^k = tw[k*ts] */
	if (p == 2) {
		for (int k = 0; k < m; ++k) {
			orc_cf a = out[k];
			orc_cf b = cmul(out[m + k], twid(c, k * ts));
			out[k] = cadd(a, b);
			out[m + k] = csub(a, b);
		}
	} else if (p == 4) {
		for (int k = 0; k < m; ++k) {
			orc_cf a = out[k];
			orc_cf b = cmul(out[m + k], twid(c, 1L * k * ts));
			orc_cf d = cmul(out[2 * m + k], twid(c, 2L * k * ts));
			orc_cf e = cmul(out[3 * m + k], twid(c, 3L * k * ts));
			orc_cf s0 = cadd(a, d), s1 = csub(a, d);
			orc_cf s2 = cadd(b, e), s3 = cmulj(csub(b, e), c->sign);
			out[k] = cadd(s0, s2);
			out[m + k] = cadd(s1, s3);
			out[2 * m + k] = csub(s0, s2);
			out[3 * m + k] = csub(s1, s3);
		}
	} else {
		orc_cf t[16], y[16];
		for (int k = 0; k < m; ++k) {
			for (int q = 0; q < p; ++q)
				t[q] = q ? cmul(out[q * m + k], twid(c, (long)q * k * ts)) : out[k];
			for (int r = 0; r < p; ++r) {
				orc_cf acc = t[0];
				for (int q = 1; q < p; ++q)
					acc = cadd(acc, cmul(t[q], twid(c, (long)q * r * m * ts)));
				y[r] = acc;
			}
			for (int r = 0; r < p; ++r)
				out[r * m + k] = y[r];
		}
	}
}

void orc_fft(orc_cf *out, const orc_cf *in, int n, int sign)
{
	ctx_t c = { get_twiddles(n), n, sign };
	fft_rec(&c, out, in, n, 1);
}
