// ts_trig_check.cpp -- accuracy of k_theilsen.hip's own sin/cos (and arc tangent) against double precision (tools only)
#include <hip/hip_runtime.h>
#include "../modem_amd/csrc/k_theilsen.hip"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace rx;
__global__ void k_sc(int n, const float *a, float *s, float *c, float *s2, float *c2)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { ts_sincos(a[i], s[i], c[i]); sincosf(a[i], &s2[i], &c2[i]); }
}
#ifdef HAVE_TS_ATAN
__global__ void k_at(int n, const float *im, const float *re, float *o, float *o2)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { o[i] = ts_atan_q(im[i], re[i]); o2[i] = atan2f(im[i], re[i]); }
}
#endif
static double ulp(float got, double want)
{
	float w = (float)want;
	int e; frexpf(w == 0.f ? 1e-30f : w, &e);
	return std::fabs((double)got - want) / std::ldexp(1.0, e - 24);
}
int main()
{
	const int n = 1 << 22;
	std::vector<float> a(n);
	for (int i = 0; i < n; ++i) {
		double t = (double)i / n;
		a[i] = i < n / 2 ? (float)((t * 4 - 1) * 1.7) : (float)((t * 4 - 3) * 900.0);   // [-1.7, 1.7) densely, [-900, 900) coarsely
	}
	float *da, *ds, *dc, *ds2, *dc2;
	hipMalloc(&da, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&ds2, n * 4); hipMalloc(&dc2, n * 4);
	hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k_sc, dim3(n / 256), dim3(256), 0, 0, n, da, ds, dc, ds2, dc2);
	std::vector<float> s(n), c(n), s2(n), c2(n);
	hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
	hipMemcpy(s2.data(), ds2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c2.data(), dc2, n * 4, hipMemcpyDeviceToHost);
	for (int part = 0; part < 2; ++part) {
		double ms = 0, mc = 0, ms2 = 0, mc2 = 0, as = 0, ac = 0;
		for (int i = part * (n / 2); i < (part + 1) * (n / 2); ++i) {
			ms = std::fmax(ms, ulp(s[i], std::sin((double)a[i]))); mc = std::fmax(mc, ulp(c[i], std::cos((double)a[i])));
			ms2 = std::fmax(ms2, ulp(s2[i], std::sin((double)a[i]))); mc2 = std::fmax(mc2, ulp(c2[i], std::cos((double)a[i])));
			as = std::fmax(as, std::fabs(s[i] - std::sin((double)a[i]))); ac = std::fmax(ac, std::fabs(c[i] - std::cos((double)a[i])));
		}
		printf("%s: ts_sincos max error sin %.2f ulp cos %.2f ulp (abs %.3g / %.3g); library sincosf sin %.2f ulp cos %.2f ulp\n",
			part ? "|a| < 900" : "|a| < 1.7", ms, mc, as, ac, ms2, mc2);
	}
#ifdef HAVE_TS_ATAN
	{
		std::vector<float> im(n), re(n), o(n), o2(n);
		for (int i = 0; i < n; ++i) { double ph = ((double)i / n * 2 - 1) * 0.7853981633974483 * 1.0001, r = 0.05 + 1.9 * ((i * 2654435761u) >> 8 & 0xffff) / 65536.0; re[i] = (float)(r * std::cos(ph)); im[i] = (float)(r * std::sin(ph)); }
		float *dim, *dre, *dout, *dout2; hipMalloc(&dim, n * 4); hipMalloc(&dre, n * 4); hipMalloc(&dout, n * 4); hipMalloc(&dout2, n * 4);
		hipMemcpy(dim, im.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dre, re.data(), n * 4, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k_at, dim3(n / 256), dim3(256), 0, 0, n, dim, dre, dout, dout2);
		hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), dout2, n * 4, hipMemcpyDeviceToHost);
		double m = 0, m2 = 0, ab = 0;
		for (int i = 0; i < n; ++i) { double w = std::atan2((double)im[i], (double)re[i]); m = std::fmax(m, ulp(o[i], w)); m2 = std::fmax(m2, ulp(o2[i], w)); ab = std::fmax(ab, std::fabs(o[i] - w)); }
		printf("|phase| <= pi/4: ts_atan_q max error %.2f ulp (abs %.3g); library atan2f %.2f ulp\n", m, ab, m2);
	}
#endif
	return 0;
}
