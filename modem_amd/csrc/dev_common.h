// dev_common.h -- shared device helpers for the gfx950 OFDM receive kernels.
// Wave = 64 lanes everywhere (CDNA4); no warp-size abstraction on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The tuning macros the kernel files define unconditionally are frozen (rounds 2 - 5 measured them: DESIGN.md, profiles/HISTORY.md).
// One of them on the command line (tools/build_variant.sh FLAGS / PERFILE_*) would be silently overridden and an A/B run would
// compare two identical binaries - so it is an error.  (The few that are still open keep an #ifndef next to their definition.)
#if defined(BACK_WALK) || defined(DEMOD_QA_LDS) || defined(DEMOD_RAW_AHEAD) || defined(DEMOD_SWZ) || defined(DEMOD_TWR_BYTES) || defined(POLAR_NT_LEVEL) || defined(POLAR_WPB) || defined(POLAR_XB1) || defined(POLAR_XB2) || defined(POLAR_XB3) || defined(SYNC_FFT_IN_LDS) || defined(SYNC_PER) || defined(SYNC_SPLIT_8K) || defined(SYNC_SPLIT_ROUNDS) || defined(SYNC_WAVES) || defined(SYNC_WAVES_SPLIT_MONO) || defined(TS_CAP) || defined(TS_MARGIN) || defined(TS_MORE_OCC) || defined(TS_MORE_WAVES_N) || defined(TS_OPEN_MARGIN) || defined(TS_OPEN_NEED) || defined(TS_OWN_ATAN) || defined(TS_ROWS_PER_WG) || defined(TS_UNC_STEPS) || defined(TX_NT_LDS) || defined(TX_TW_GLOBAL)
#error "a frozen tuning macro was passed with -D: edit a patched copy of the source instead (tools/experiments/ts_stage_probe.py shows how)"
#endif

namespace rx {

// the lengths that depend on the sample rate: one instantiation per rate like the reference's
// Decoder<value, cmplx, rate> / Encoder<value, cmplx, rate> (decode.cc:590-602, encode.cc:424-436)
template <int RATE> struct RateCfg {
	static constexpr int SL = (int)((1280L * RATE) / 8000);       // symbol_len, decode.cc:171 (1280, 2560, 7056, 7680)
	static constexpr int FILTER_LEN = (((21 * RATE) / 8000) & ~3) | 1;   // decode.cc:172 (21, 41, 113, 125)
	static constexpr int GL = SL / 8;                              // guard_len, decode.cc:173
	static constexpr int STRIDE = SL + GL;
	static constexpr int HS = SL / 2;                              // correlator symbol_len, decode.cc:196
	static constexpr int BUFFER_LEN = 6 * STRIDE;                  // decode.cc:188
	static constexpr int SEARCH_POS = BUFFER_LEN - 4 * STRIDE;     // decode.cc:189
	static constexpr int MATCH_LEN = GL | 1;                       // decode.cc:41
	static constexpr int MATCH_DEL = (MATCH_LEN - 1) / 2;          // decode.cc:42
};
inline bool rate_supported(int rate) { return rate == 8000 || rate == 16000 || rate == 44100 || rate == 48000; }
inline int rate_symbol_len(int rate) { return (int)((1280L * rate) / 8000); }
inline int rate_filter_len(int rate) { return (((21 * rate) / 8000) & ~3) | 1; }
// run `body` with the RateCfg of a run-time rate (launch wrappers)
#define RX_RATE_SWITCH(rate, ...) \
	switch (rate) { \
	case 16000: { constexpr int RATE = 16000; __VA_ARGS__; } break; \
	case 44100: { constexpr int RATE = 44100; __VA_ARGS__; } break; \
	case 48000: { constexpr int RATE = 48000; __VA_ARGS__; } break; \
	default: { constexpr int RATE = 8000; __VA_ARGS__; } break; \
	}
constexpr int CONS_COLS = 432;      // decode.cc:306 (mode 6; other modes via mode_desc)
constexpr int CONS_ROWS = 50;       // decode.cc:453
constexpr int CONS_CNT = 21600;     // decode.cc:372
constexpr int CONS_BITS = 64800;    // decode.cc:310
constexpr int MESG_BITS = 43808;    // decode.cc:311
constexpr int DATA_BITS = 43040;    // decode.cc:174
constexpr int CRC_BITS = 43072;     // decode.cc:175
constexpr int CODE_LEN = 65536;
constexpr int PAYLOAD_BYTES = 5380;
constexpr int MESG_BYTES = 5476;    // 43808 / 8
constexpr int LIST = 8;             // decode.cc:164-169 (AVX2 build)
constexpr int MLS1_LEN = 255;       // decode.cc:185
constexpr int BCH_N = 255, BCH_K = 71;

constexpr float TWO_PI_F = 6.28318530717958647692f;
constexpr float PI_F = 3.14159265358979323846f;

constexpr int CONS_MAX = 32400;     // decode.cc:178 cons_max
constexpr int CARR_MAX = 32400 + 512;   // (rows + 1) x cols carriers per frame (pilot symbol + data rows)
constexpr int ROWS_MAX = 126;       // decode.cc:181 rows_max
constexpr int COLS_MAX = 512;       // decode.cc:180
constexpr int MESG_BITS_MAX = 44096;
constexpr int MESG_BYTES_MAX = 5512;

// decode.cc:302-374 prepare(): the mode table
struct ModeDesc { int cols, rows, mod_bits, cons_bits, mesg_bits, table; };
__host__ __device__ inline ModeDesc mode_desc(int mode)
{
	ModeDesc m;
	m.table = mode >= 10;                       // 0: frozen_64800_43072, 1: frozen_64512_43072
	m.cons_bits = m.table ? 64512 : 64800;
	m.mesg_bits = m.table ? 44096 : 43808;
	switch (mode) {
	case 6: m.cols = 432; m.mod_bits = 3; break;    // decode.cc:305-312
	case 7: m.cols = 400; m.mod_bits = 3; break;    // decode.cc:313-320
	case 8: m.cols = 400; m.mod_bits = 2; break;    // decode.cc:321-328
	case 9: m.cols = 360; m.mod_bits = 2; break;    // decode.cc:329-336
	case 10: m.cols = 512; m.mod_bits = 3; break;   // decode.cc:337-344
	case 11: m.cols = 384; m.mod_bits = 3; break;   // decode.cc:345-352
	case 12: m.cols = 384; m.mod_bits = 2; break;   // decode.cc:353-360
	default: m.cols = 256; m.mod_bits = 2; break;   // 13: decode.cc:361-368
	}
	m.rows = (m.cons_bits / m.mod_bits) / m.cols;   // decode.cc:372,453
	return m;
}

struct cf { float re, im; };

__device__ __forceinline__ cf mk(float re, float im) { cf r; r.re = re; r.im = im; return r; }
__device__ __forceinline__ cf cadd(cf a, cf b) { return mk(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cf csub(cf a, cf b) { return mk(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ cf cmul(cf a, cf b) { return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
__device__ __forceinline__ cf cconj(cf a) { return mk(a.re, -a.im); }
__device__ __forceinline__ float cnorm(cf a) { return a.re * a.re + a.im * a.im; }
__device__ __forceinline__ cf cmul_negj(cf a) { return mk(a.im, -a.re); }   // a * (-j)

// DSP::Complex operator/ = a*conj(b)/norm(b); decode.cc:62-70 / 227-235
__device__ __forceinline__ cf demod_or_erase(cf curr, cf prev)
{
	float d = cnorm(prev);
	if (!(d > 0.f))
		return mk(0.f, 0.f);
	cf n = cmul(curr, cconj(prev));
	cf c = mk(n.re / d, n.im / d);
	if (!(cnorm(c) <= 4.f))
		return mk(0.f, 0.f);
	return c;
}

// ---- psk.hh:90-140 PhaseShiftKeying<8, cmplx, float> ---------------------------------
// psk.hh:49-88 PhaseShiftKeying<4>: map(hard(c)) (psk.hh:70-74,82-85)
__device__ __forceinline__ cf psk4_hard_map(cf c)
{
	const float r = 0.70710678118654752440f;
	return mk(r * (c.re < 0.f ? -1.f : 1.f), r * (c.im < 0.f ? -1.f : 1.f));
}
__device__ __forceinline__ cf psk8_hard_map(cf c)   // map(hard(c)): psk.hh:118-123,132-139
{
	const float cos_pi_8 = 0.92387953251128675613f, sin_pi_8 = 0.38268343236508977173f;
	float b1 = c.re < 0.f ? -1.f : 1.f;
	float b2 = c.im < 0.f ? -1.f : 1.f;
	bool swap = fabsf(c.re) < fabsf(c.im);
	float real = swap ? sin_pi_8 : cos_pi_8, imag = swap ? cos_pi_8 : sin_pi_8;
	return mk(real * b1, imag * b2);
}

// unit phasor e^{j*omega*k}; phase in double like the oracle's closed form of DSP::Phasor
__device__ __forceinline__ cf phasor(float omega, long k)
{
	double a = (double)omega * (double)k;
	const double inv2pi = 0.15915494309189533577, twopi = 6.28318530717958647692;
	a -= twopi * rint(a * inv2pi);
	float s, c;
	sincosf((float)a, &s, &c);
	return mk(c, s);
}

// ---- raw PCM access: what DSP::ReadWAV + next_sample() deliver for 2-channel input
// x / 32767.f and x / 127.f for integer-valued x of the PCM's range, correctly rounded like the division the reference
// runs (pcm.hh), in three instructions instead of the dozen of an IEEE fp32 division: q0 = x c with c = RN(1 / d), the
// exact residual r = x - d q0 (one FMA) and q0 + r c (Markstein).  Checked against the division for every int16 and
// uint8 input (tools/check_pcm_div.c).
__device__ __forceinline__ float div_32767(float x)
{
	const float c = 1.0f / 32767.f;
	const float q0 = x * c;
	return __builtin_fmaf(__builtin_fmaf(-32767.f, q0, x), c, q0);
}
__device__ __forceinline__ float div_127(float x)
{
	const float c = 1.0f / 127.f;
	const float q0 = x * c;
	return __builtin_fmaf(__builtin_fmaf(-127.f, q0, x), c, q0);
}

template <int V> struct IntC { static constexpr int value = V; };
struct SampleSrc {
	const void *base;      // first sample of this frame
	int fmt;               // OFDMRX_FMT_*
	int channels;
	long n;                // samples in the frame
	const cf *analytic;    // mono: output of the front-end kernel (D1), else nullptr
	__device__ __forceinline__ cf at(long i) const
	{
		if (i < 0 || i >= n)
			return mk(0.f, 0.f);
		if (analytic)
			return analytic[i];
		if (fmt == 0) {
			short2 v = ((const short2 *)base)[i];
			return mk(div_32767((float)v.x), div_32767((float)v.y));
		}
		return pair_other(i);
	}
	// 8-bit / float32 pairs: one access per I/Q pair too (check_args keeps frames on sample-frame boundaries)
	__device__ __forceinline__ cf pair_other(long i) const
	{
		if (fmt == 1) {
			const unsigned v = ((const uint16_t *)base)[i];
			return mk(div_127((float)((int)(v & 255u) - 128)), div_127((float)((int)(v >> 8) - 128)));
		}
		const float2 v = ((const float2 *)base)[i];
		return mk(v.x, v.y);
	}
	// the same with the format decided once per kernel instead of once per sample: 0 = analytic (mono), 1 = int16 pairs, 2 = the rest
	__device__ __forceinline__ int mode() const { return analytic ? 0 : (fmt == 0 ? 1 : 2); }
	template <int M> __device__ __forceinline__ cf at_m(long i) const
	{
		if (i < 0 || i >= n)
			return mk(0.f, 0.f);
		if (M == 0)
			return analytic[i];
		if (M == 1) {
			short2 v = ((const short2 *)base)[i];
			return mk(div_32767((float)v.x), div_32767((float)v.y));
		}
		return pair_other(i);
	}
	template <class F> __device__ __forceinline__ void with_mode(F f) const
	{
		const int m = mode();
		if (m == 0) f(IntC<0>{});
		else if (m == 1) f(IntC<1>{});
		else f(IntC<2>{});
	}
};

// ---- Stockham radix stages in LDS, NT threads cooperate on one N-point transform.
// Forward transform e^{-j 2 pi k n / N}; tw = table of 1280 roots e^{-j 2 pi m / 1280}.
// P = product of the radices of the previous stages.  Caller syncs before the first stage.
template <int R> struct Bfly;
template <> struct Bfly<2> {
	static __device__ __forceinline__ void run(cf *v) { cf a = v[0], b = v[1]; v[0] = cadd(a, b); v[1] = csub(a, b); }
};
template <> struct Bfly<4> {
	static __device__ __forceinline__ void run(cf *v)
	{
		cf s0 = cadd(v[0], v[2]), s1 = csub(v[0], v[2]);
		cf s2 = cadd(v[1], v[3]), s3 = cmul_negj(csub(v[1], v[3]));
		v[0] = cadd(s0, s2); v[1] = cadd(s1, s3); v[2] = csub(s0, s2); v[3] = csub(s1, s3);
	}
};
template <> struct Bfly<5> {
	static __device__ __forceinline__ void run(cf *v)
	{
		const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
		const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
		cf a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]);
		cf b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
		cf m1 = mk(v[0].re + c1 * a1.re + c2 * a2.re, v[0].im + c1 * a1.im + c2 * a2.im);
		cf m2 = mk(v[0].re + c2 * a1.re + c1 * a2.re, v[0].im + c2 * a1.im + c1 * a2.im);
		cf n1 = mk(s1 * b1.re + s2 * b2.re, s1 * b1.im + s2 * b2.im);
		cf n2 = mk(s2 * b1.re - s1 * b2.re, s2 * b1.im - s1 * b2.im);
		cf y0 = cadd(v[0], cadd(a1, a2));
		cf jn1 = cmul_negj(n1), jn2 = cmul_negj(n2);   // -j*n
		v[0] = y0;
		v[1] = cadd(m1, jn1); v[4] = csub(m1, jn1);
		v[2] = cadd(m2, jn2); v[3] = csub(m2, jn2);
	}
};

template <> struct Bfly<3> {
	static __device__ __forceinline__ void run(cf *v)
	{
		const float s = 0.86602540378443864676f;
		cf a = cadd(v[1], v[2]), b = csub(v[1], v[2]);
		cf m = mk(v[0].re - 0.5f * a.re, v[0].im - 0.5f * a.im);
		cf jn = cmul_negj(mk(s * b.re, s * b.im));
		v[0] = cadd(v[0], a);
		v[1] = cadd(m, jn);
		v[2] = csub(m, jn);
	}
};
template <> struct Bfly<7> {
	static __device__ __forceinline__ void run(cf *v)
	{
		const float c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f, c3 = -0.90096886790241912624f;
		const float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;
		cf a1 = cadd(v[1], v[6]), a2 = cadd(v[2], v[5]), a3 = cadd(v[3], v[4]);
		cf b1 = csub(v[1], v[6]), b2 = csub(v[2], v[5]), b3 = csub(v[3], v[4]);
		cf m1 = mk(v[0].re + c1 * a1.re + c2 * a2.re + c3 * a3.re, v[0].im + c1 * a1.im + c2 * a2.im + c3 * a3.im);
		cf m2 = mk(v[0].re + c2 * a1.re + c3 * a2.re + c1 * a3.re, v[0].im + c2 * a1.im + c3 * a2.im + c1 * a3.im);
		cf m3 = mk(v[0].re + c3 * a1.re + c1 * a2.re + c2 * a3.re, v[0].im + c3 * a1.im + c1 * a2.im + c2 * a3.im);
		cf n1 = mk(s1 * b1.re + s2 * b2.re + s3 * b3.re, s1 * b1.im + s2 * b2.im + s3 * b3.im);
		cf n2 = mk(s2 * b1.re - s3 * b2.re - s1 * b3.re, s2 * b1.im - s3 * b2.im - s1 * b3.im);
		cf n3 = mk(s3 * b1.re - s1 * b2.re + s2 * b3.re, s3 * b1.im - s1 * b2.im + s2 * b3.im);
		cf y0 = cadd(v[0], cadd(a1, cadd(a2, a3)));
		cf j1 = cmul_negj(n1), j2 = cmul_negj(n2), j3 = cmul_negj(n3);
		v[0] = y0;
		v[1] = cadd(m1, j1); v[6] = csub(m1, j1);
		v[2] = cadd(m2, j2); v[5] = csub(m2, j2);
		v[3] = cadd(m3, j3); v[4] = csub(m3, j3);
	}
};

// NT = 64: the transform belongs to ONE wave (its LDS operations execute in order): no workgroup barrier, the other
// waves of the block run their own transforms unsynchronised
template <int NT> __device__ __forceinline__ void fft_sync()
{
	if (NT == 64)
		__builtin_amdgcn_wave_barrier();
	else
		__syncthreads();
}

// TWC < 0: tw = table of TWN roots e^{-j 2 pi m / TWN}, read at stride TWN / (P R) - fine through L1, but in LDS the strides
// of the early stages are multiples of the 128-byte bank cycle (a 20-way conflict at P = 20).
// TWC >= 0: tw = the plan's COMPACT table (fft_compact_twiddles below): stage (P, R) owns (R - 1) P consecutive entries
// [TWC + (t - 1) P + k] = w^(t k TWN / (P R)), so consecutive butterflies read consecutive words.
template <int N, int R, int P, int NT, int TWN = 1280, int TWC = -1>
__device__ __forceinline__ void fft_stage(cf *buf, const cf *tw, int tid)
{
	constexpr int T = N / R;
	constexpr int NB = (T + NT - 1) / NT;
	constexpr int TWS = TWN / (P * R);   // tw = table of TWN roots e^{-j 2 pi m / TWN}
	cf v[NB][R];
	#pragma unroll
	for (int q = 0; q < NB; ++q) {
		int b = tid + q * NT;
		if (b < T) {
			int k = b % P;
			#pragma unroll
			for (int t = 0; t < R; ++t) {
				cf x = buf[b + t * T];
				if (t && P > 1)
					x = cmul(x, TWC >= 0 ? tw[TWC + (t - 1) * P + k] : tw[(t * k) * TWS]);
				v[q][t] = x;
			}
			Bfly<R>::run(v[q]);
		}
	}
	fft_sync<NT>();
	#pragma unroll
	for (int q = 0; q < NB; ++q) {
		int b = tid + q * NT;
		if (b < T) {
			int k = b % P;
			int j = (b - k) * R + k;
			#pragma unroll
			for (int t = 0; t < R; ++t)
				buf[j + t * P] = v[q][t];
		}
	}
	fft_sync<NT>();
}

// Forward N-point transform in place (LDS or global scratch), natural order in and out, NT threads.
// The radix plan is fixed at compile time: all 5s, then 7s, 3s, 4s and a final 2
//   640 = 5.4.4.4.2   1280 = 5.4.4.4.4   2560 = 5.4.4.4.4.2   3528 = 7.7.3.3.4.2   3840 = 5.3.4.4.4.4
//   5120 = 5.4^5   7056 = 7.7.3.3.4.4   7680 = 5.3.4.4.4.4.2
// tw = table of TWN roots e^{-j 2 pi m / TWN}, N | TWN (TWC < 0), or the compact table of this plan (TWC = 0).
template <int N, int REM, int P, int NT, int TWN, int TWC> struct FftPlan {
	static constexpr int R = REM % 5 == 0 ? 5 : REM % 7 == 0 ? 7 : REM % 3 == 0 ? 3 : REM % 4 == 0 ? 4 : 2;
	static constexpr int OWN = P > 1 ? (R - 1) * P : 0;      // compact entries of this stage
	using Next = FftPlan<N, REM / R, P * R, NT, TWN, (TWC >= 0 ? TWC + OWN : -1)>;
	static constexpr int COMPACT = OWN + Next::COMPACT;
	static __device__ __forceinline__ void run(cf *buf, const cf *tw, int tid)
	{
		fft_stage<N, R, P, NT, TWN, TWC>(buf, tw, tid);
		Next::run(buf, tw, tid);
	}
	// dst[TWC + (t - 1) P + k] = src[t k TWN / (P R)], every stage of the plan, NT threads
	static __device__ __forceinline__ void fill(cf *dst, const cf *src, int tid)
	{
		if (P > 1)
			for (int i = tid; i < OWN; i += NT)
				dst[TWC + i] = src[((i / P + 1) * (i % P)) * (TWN / (P * R))];
		Next::fill(dst, src, tid);
	}
};
template <int N, int P, int NT, int TWN, int TWC> struct FftPlan<N, 1, P, NT, TWN, TWC> {
	static constexpr int COMPACT = 0;
	static __device__ __forceinline__ void run(cf *, const cf *, int) {}
	static __device__ __forceinline__ void fill(cf *, const cf *, int) {}
};
template <int N, int NT, int TWN>
__device__ __forceinline__ void fft_fwd(cf *buf, const cf *tw, int tid)
{
	FftPlan<N, N, 1, NT, TWN, -1>::run(buf, tw, tid);
}
// the same with the plan's compact twiddle table (LDS): fft_compact_size<N, TWN>() entries, built by fft_compact_twiddles
template <int N, int TWN> constexpr int fft_compact_size() { return FftPlan<N, N, 1, 64, TWN, 0>::COMPACT; }
template <int N, int NT, int TWN>
__device__ __forceinline__ void fft_compact_twiddles(cf *dst, const cf *tw, int tid)
{
	FftPlan<N, N, 1, NT, TWN, 0>::fill(dst, tw, tid);
}
template <int N, int NT, int TWN>
__device__ __forceinline__ void fft_fwd_compact(cf *buf, const cf *twc, int tid)
{
	FftPlan<N, N, 1, NT, TWN, 0>::run(buf, twc, tid);
}

// Bank-conflict-free image of a wave-private 256-point buffer: element i lives at i ^ ((i >> 2) & 3) ^ (((i >> 4) & 3) << 2).  The
// radix-4 stages read 64 consecutive elements per instruction (any bijection of the low six bits keeps that conflict-free) and
// write at strides of 4 (P = 1) and 16 / 4 (P = 4) elements, which the plain layout puts four lanes deep on a bank; under the
// swizzle every 16-lane store group touches 32 distinct banks (checked against the bank rules of the guide in a simulation).
// Same operations in the same order as fft_stage<256, 4, P, 64>: only the addresses change.
__device__ __forceinline__ int swz256(int i) { return i ^ ((i >> 2) & 3) ^ (((i >> 4) & 3) << 2); }
template <int P, int TWC> __device__ __forceinline__ void fft256_stage_swz(cf *buf, const cf *tw, int lane, int sl)
{
	cf v[4];
	const int k = lane % P;
	#pragma unroll
	for (int t = 0; t < 4; ++t) {
		cf x = buf[sl + t * 64];                              // swz(lane + 64 t) = swz(lane) + 64 t
		if (t && P > 1)
			x = cmul(x, tw[TWC + (t - 1) * P + k]);
		v[t] = x;
	}
	Bfly<4>::run(v);
	fft_sync<64>();
	const int j = (lane - k) * 4 + k;
	#pragma unroll
	for (int t = 0; t < 4; ++t)
		buf[swz256(j + t * P)] = v[t];
	fft_sync<64>();
}

// ---- a 256-point forward transform of ONE wave in registers (round 6): four radix-4 steps, decimation in frequency; between the
// steps the lane's four values are exchanged with other lanes by data-parallel-primitive moves and lane swaps instead of a trip
// through LDS (k_demod waits for LDS half of its time and uses 29 % of the vector unit: profiles/r06_sq_counters.txt).
//   n = 64 n3 + 16 n2 + 4 n1 + n0.  Lane l = n mod 64 reads buf[sl + 64 t], t = n3 (sl = the caller's address of element l).
//   step A: butterfly over the register index (n3 -> k0), twiddle w256^(l k0);          registers <-> lane bits (5, 4)
//   step B: butterfly (n2 -> k1), twiddle w64^((l & 15) k1);                            registers <-> lane bits (3, 2)
//   step C: butterfly (n1 -> k2), twiddle w16^((l & 3) k2);                             registers <-> lane bits (1, 0)
//   step D: butterfly (n0 -> k3).  Register t of lane l then holds X[k0 + 4 k1 + 16 k2 + 64 t], k0 = l >> 4, k1 = (l >> 2) & 3, k2 = l & 3,
//   and goes to buf[l + 64 t]: the caller reads X[q] at fft256_pos(q).
// An exchange of a register pair (A: register bit 0, B: register bit 1) with lane bit b: A keeps its lanes with bit b clear and takes B's
// value of lane ^ 2^b where bit b is set; B the other way round.  Bits 5 / 4: one v_permlane32_swap / v_permlane16_swap per pair; bits
// 3 / 2: row_ror:8 / row_shr:4 + row_shl:4 with a bank mask, one move per register; bits 1 / 0: quad_perm + a select.
// twl: [9][64] per-lane twiddles, (step, k) major: w256^(l k), w64^((l & 15) k), w16^((l & 3) k), k = 1..3 (fft256_lane_twiddles).
__host__ __device__ constexpr int fft256_pos(int q)
{
	return (q & 192) | ((q & 3) << 4) | (q & 12) | ((q >> 4) & 3);
}
template <int NT> __device__ __forceinline__ void fft256_lane_twiddles(cf *twl, const cf *tw1280, int tid)
{
	for (int i = tid; i < 9 * 64; i += NT) {
		const int l = i & 63, k = (i >> 6) % 3 + 1, step = i / 192;
		const int e = step == 0 ? 5 * (l * k) : step == 1 ? 20 * ((l & 15) * k) : 80 * ((l & 3) * k);   // w256 = w1280^5, w64 = w1280^20, w16 = w1280^80
		twl[i] = tw1280[e % 1280];
	}
}
template <int BIT> __device__ __forceinline__ void fft256_xchg(float &A, float &B, int lane)
{
	const int a = __float_as_int(A), b = __float_as_int(B);
	if constexpr (BIT == 5) {
		auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
		A = __int_as_float((int)r[0]); B = __int_as_float((int)r[1]);
	} else if constexpr (BIT == 4) {
		auto r = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
		A = __int_as_float((int)r[0]); B = __int_as_float((int)r[1]);
	} else if constexpr (BIT == 3) {                              // row_ror:8 = lane ^ 8; banks 2, 3 = lanes with bit 3 set
		A = __int_as_float(__builtin_amdgcn_update_dpp(a, b, 0x128, 0xf, 0xC, false));
		B = __int_as_float(__builtin_amdgcn_update_dpp(b, a, 0x128, 0xf, 0x3, false));
	} else if constexpr (BIT == 2) {                              // row_shr:4 (from lane - 4) into banks 1, 3; row_shl:4 (from lane + 4) into banks 0, 2
		A = __int_as_float(__builtin_amdgcn_update_dpp(a, b, 0x114, 0xf, 0xA, false));
		B = __int_as_float(__builtin_amdgcn_update_dpp(b, a, 0x104, 0xf, 0x5, false));
	} else {
		constexpr int CTRL = BIT == 1 ? 0x4E : 0xB1;              // quad_perm [2,3,0,1] = lane ^ 2, [1,0,3,2] = lane ^ 1
		const int pb = __builtin_amdgcn_update_dpp(0, b, CTRL, 0xf, 0xf, true), pa = __builtin_amdgcn_update_dpp(0, a, CTRL, 0xf, 0xf, true);
		const bool set = (lane >> BIT) & 1;
		A = __int_as_float(set ? pb : a);
		B = __int_as_float(set ? b : pa);
	}
}
template <int HI, int LO> __device__ __forceinline__ void fft256_transpose(cf (&v)[4], int lane)
{
	fft256_xchg<HI>(v[0].re, v[2].re, lane); fft256_xchg<HI>(v[0].im, v[2].im, lane);
	fft256_xchg<HI>(v[1].re, v[3].re, lane); fft256_xchg<HI>(v[1].im, v[3].im, lane);
	fft256_xchg<LO>(v[0].re, v[1].re, lane); fft256_xchg<LO>(v[0].im, v[1].im, lane);
	fft256_xchg<LO>(v[2].re, v[3].re, lane); fft256_xchg<LO>(v[2].im, v[3].im, lane);
}
__device__ __forceinline__ void fft256_regs(cf *buf, const cf *twl, int lane, int sl)
{
	cf v[4];
	#pragma unroll
	for (int t = 0; t < 4; ++t)
		v[t] = buf[sl + t * 64];
	Bfly<4>::run(v);
	#pragma unroll
	for (int k = 1; k < 4; ++k)
		v[k] = cmul(v[k], twl[(k - 1) * 64 + lane]);
	fft256_transpose<5, 4>(v, lane);
	Bfly<4>::run(v);
	#pragma unroll
	for (int k = 1; k < 4; ++k)
		v[k] = cmul(v[k], twl[(3 + k - 1) * 64 + lane]);
	fft256_transpose<3, 2>(v, lane);
	Bfly<4>::run(v);
	#pragma unroll
	for (int k = 1; k < 4; ++k)
		v[k] = cmul(v[k], twl[(6 + k - 1) * 64 + lane]);
	fft256_transpose<1, 0>(v, lane);
	Bfly<4>::run(v);
	fft_sync<64>();                                           // (every lane's reads of buf come before any lane's writes: one wave, LDS in order)
	#pragma unroll
	for (int t = 0; t < 4; ++t)
		buf[lane + t * 64] = v[t];
	fft_sync<64>();
}

// ---- wave helpers ----------------------------------------------------------
// a value every lane holds alike, moved to scalar registers (the compiler cannot know that what came out of a vector load is uniform)
__device__ __forceinline__ long uniform_l(long v)
{
	const int lo = __builtin_amdgcn_readfirstlane((int)v), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
	return ((long)hi << 32) | (long)(unsigned)lo;
}
__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double shfl_d(double v, int src)
{
	int lo = __double2loint(v), hi = __double2hiint(v);
	lo = __shfl(lo, src);
	hi = __shfl(hi, src);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_up_d(double v, int d)
{
	int lo = __double2loint(v), hi = __double2hiint(v);
	lo = __shfl_up(lo, d);
	hi = __shfl_up(hi, d);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_xor_d(double v, int m)
{
	int lo = __double2loint(v), hi = __double2hiint(v);
	lo = __shfl_xor(lo, m);
	hi = __shfl_xor(hi, m);
	return __hiloint2double(hi, lo);
}
// ---- data-parallel-primitive moves (no LDS pipe, no address registers): lanes without a source get 0
//   0x110 + n  row_shr:n (rows of 16 lanes)     0x142 / 0x143  row_bcast:15 / :31 (lane 15 of the row before / lane 31 of the half
//   before, for the rows ROW_MASK names)         0x138  wave_shr:1
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ float dpp_f(float v)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ double dpp_d(double v)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
	return __hiloint2double(hi, lo);
}
// Weighted inclusive scan over the wave, v[l] = sum_{j <= l} A^(l - j) e[j] (the end states of consecutive chunks of a first-order
// recurrence that decays by A per chunk): four shifts inside the rows of 16, then lane 15 / lane 31 broadcast to the row / half
// behind it.  w[k] = A^(2^k), k = 0..3 (uniform); w16 = A^((lane & 15) + 1), w32 = A^((lane & 31) + 1) (per lane).
template <class T> struct WScan;
template <> struct WScan<float> {
	static __device__ __forceinline__ float run(float v, const float (&w)[6], float w16, float w32)
	{
		v = fmaf(w[0], dpp_f<0x111>(v), v);
		v = fmaf(w[1], dpp_f<0x112>(v), v);
		v = fmaf(w[2], dpp_f<0x114>(v), v);
		v = fmaf(w[3], dpp_f<0x118>(v), v);
		v = fmaf(w16, dpp_f<0x142, 0xa>(v), v);
		v = fmaf(w32, dpp_f<0x143, 0xc>(v), v);
		return v;
	}
};
template <> struct WScan<double> {
	static __device__ __forceinline__ double run(double v, const double (&w)[7], double w16, double w32)
	{
		v = fma(w[0], dpp_d<0x111>(v), v);
		v = fma(w[1], dpp_d<0x112>(v), v);
		v = fma(w[2], dpp_d<0x114>(v), v);
		v = fma(w[3], dpp_d<0x118>(v), v);
		v = fma(w16, dpp_d<0x142, 0xa>(v), v);
		v = fma(w32, dpp_d<0x143, 0xc>(v), v);
		return v;
	}
};
// inclusive scan of one double per lane across the wave
__device__ __forceinline__ double wave_scan_incl(double v, int lane)
{
	#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		double o = shfl_up_d(v, d);
		if (lane >= d)
			v += o;
	}
	return v;
}
__device__ __forceinline__ double wave_sum_d(double v)
{
	#pragma unroll
	for (int m = 32; m; m >>= 1)
		v += shfl_xor_d(v, m);
	return v;
}
// sin and cos of the rotation angle (decode.cc:493: a row's phases are yint + slope x, a fraction of a radian).  Cody-Waite
// reduction by pi/2 in two parts (exact enough for |a| < 1e3: n < 640, n * lo's rounding stays under 1e-10), the single-precision
// minimax kernels of fdlibm (k_sinf / k_cosf) on |r| <= pi/4, quadrant by n & 3: <= 1 ulp on either output, 23 vector instructions
// where the library routine (which carries the Payne-Hanek path for huge arguments) took about twice that.  Larger arguments
// (never seen: the estimator's outputs are bounded by the row's phases) go to the library routine.
__device__ __forceinline__ void row_sincos(float a, float &sn, float &cs)
{
	if (!(fabsf(a) < 1000.f)) {
		sincosf(a, &sn, &cs);
		return;
	}
	const float n = rintf(a * 0.636619772f);
	float r = fmaf(-n, 1.57079637f, a);
	r = fmaf(-n, -4.37113883e-8f, r);
	const float z = r * r;
	const float ps = fmaf(fmaf(fmaf(2.71831149e-6f, z, -1.98393348e-4f), z, 8.33332939e-3f), z, -1.66666667e-1f);
	const float pc = fmaf(fmaf(fmaf(2.43904488e-5f, z, -1.38867638e-3f), z, 4.16666233e-2f), z, -0.5f);
	const float sr = fmaf(r * z, ps, r), cr = fmaf(z, pc, 1.f);
	const int q = (int)n;
	const float s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
	sn = __uint_as_float(__float_as_uint(s0) ^ ((uint32_t)(q & 2) << 30));
	cs = __uint_as_float(__float_as_uint(c0) ^ ((uint32_t)((q + 1) & 2) << 30));
}
// decode.cc:493-494: cons[cons_cols * j + i] *= DSP::polar<value>(1, -tse(i + code_off)) for the point in column i of a row
// whose Theil-Sen line is (slope, yint).  Since round 4 the rotated row is never stored: k_theil_sen leaves the line, and the
// kernels behind it (k_back; the CONS_ROT tap) rotate each point where they use it - always through this one function.
__device__ __forceinline__ cf rotate_point(cf c, float slope, float yint, int i, int cols)
{
	const float a = -(yint + slope * (float)(i - cols / 2));
	float sn, cs;
	row_sincos(a, sn, cs);
	return cmul(c, mk(cs, sn));
}

// decode.cc:505-516 for one frame: the running sp / np of the SNR estimate.  Wave w reduces rows w, w + 4, ... by itself
// (per-lane double sums over its 7 carriers, one wave butterfly) - no workgroup barrier per row; after ONE barrier thread 0
// folds the row sums in order into the fp32 running sums and leaves the cumulative precision of every row in prec[].
// A lane owns NPL = ceil(cols / 64) CONSECUTIVE columns lane NPL .. lane NPL + NPL - 1 of every row (round 6; until then the columns
// lane + 64 q: the caller's sign bits of 64 neighbouring points were then ORed into LDS by 64 lanes hitting six words - ten lanes per
// word, serialised - and the kernel spent 70 % of its time in the LDS pipe; with consecutive columns a lane assembles its 21 bits in
// a register and ORs one or two words per row).  raw(j, i) loads the point (row j, column i); begin_row(j, i0) is called once per row
// and lane with the lane's first column, then rotated(j, i, c) for its columns in ascending order: it delivers the point as
// decode.cc:505 sees it (rotated); visit(j, i, c) sees every point once, end_row(j, i0) closes the row (the caller collects what it needs).
// A lane's points of a row - at most eight - are loaded together, and the NEXT row's while this one is worked on: the kernel is
// latency-bound otherwise (one HBM round trip per point and lane: k_back 0.70 ms per 8192 frames with the plain loop).
// Returns false in every thread if a precision is not a positive finite number.
// sp: every hard-decision point has the same norm (psk.hh:82-85,132-139: (cos, sin) of pi/8 in either order, or (r, r)) - also the
// one an erased carrier maps to - so a lane's share of the row's sp is its point count times that constant: exactly the sum of
// its terms (up to 8 x a 49-bit value fits a double), two conversions and two multiplications per point less.
template <typename Raw, typename Begin, typename Rotated, typename Visit, typename End>
__device__ __forceinline__ bool snr_rows(Raw raw, Begin begin_row, Rotated rotated, int rows, int cols, int mod_bits, int tid, double (*rsum)[2],
	float *prec, Visit visit, End end_row)
{
	const int wave = tid >> 6, lane = tid & 63;
	const cf h1 = mod_bits == 3 ? psk8_hard_map(mk(1.f, 0.f)) : psk4_hard_map(mk(1.f, 0.f));
	const double h_norm = (double)h1.re * h1.re + (double)h1.im * h1.im;
	constexpr int NP = COLS_MAX / 64;                             // points per lane and row, at most
	const int npl = (cols + 63) >> 6, i0 = lane * npl;            // this lane's columns: i0 .. i0 + npl - 1 (those below cols)
	const int mine = min(max(cols - i0, 0), npl);
	const double dsp_lane = (double)mine * h_norm;
	cf nxt[NP];
	auto fetch = [&](int j) {
		#pragma unroll
		for (int q = 0; q < NP; ++q)
			if (j < rows && q < mine)
				nxt[q] = raw(j, i0 + q);
	};
	fetch(wave);
	const double dsp = wave_sum_d(dsp_lane);                      // the same for every row of the frame
	for (int j = wave; j < rows; j += 4) {
		cf cur[NP];
		#pragma unroll
		for (int q = 0; q < NP; ++q)
			cur[q] = nxt[q];
		fetch(j + 4);
		double dnp = 0.0;
		begin_row(j, i0);
		#pragma unroll
		for (int q = 0; q < NP; ++q) {
			if (q < mine) {
				const int i = i0 + q;
				const cf c = rotated(j, i, cur[q]);
				const cf h = mod_bits == 3 ? psk8_hard_map(c) : psk4_hard_map(c);   // decode.cc:509-511
				const double er = (double)c.re - h.re, ei = (double)c.im - h.im;
				dnp += er * er + ei * ei;
				visit(j, i, c);
			}
		}
		end_row(j, i0);
		dnp = wave_sum_d(dnp);
		if (lane == 0) { rsum[j][0] = dsp; rsum[j][1] = dnp; }
	}
	__syncthreads();
	__shared__ int snr_ok;
	if (tid == 0) {
		float sp = 0.f, np = 0.f;
		bool ok = true;
		for (int j = 0; j < rows; ++j) {
			sp = (float)((double)sp + rsum[j][0]);
			np = (float)((double)np + rsum[j][1]);
			const float precision = sp / np;                      // decode.cc:516
			ok &= precision > 0.f && precision < 3.0e38f;
			prec[j] = precision;
		}
		snr_ok = ok;
	}
	__syncthreads();
	return snr_ok != 0;
}
// The systematic message (decode.cc:254-261: the codeword at the unfrozen positions, ascending) out of a bit-packed codeword, by a workgroup
// of 256 threads: code word w holds cnt consecutive message bits from bit `off` on - its unfrozen bits pushed together ("compress",
// Hacker's Delight 7-4, the five move masks precomputed per word: tables.cpp info_compress) and ORed into the message at that bit.
// Round 6; until then every message byte gathered its eight bits one by one through the list of unfrozen positions (171 two-byte loads
// and as many LDS reads per thread: 87 us of k_back's 547 per chunk).  code: 2048 words in LDS; mesg32: the message as little-endian
// words in LDS, ZEROED by the caller and synchronised before and after; rec: this table's [2048][8] records.
__device__ __forceinline__ void message_gather(const uint32_t *code, uint32_t *mesg32, const uint32_t *__restrict__ rec, int tid)
{
	#pragma unroll
	for (int e = 0; e < CODE_LEN / 32 / 256; ++e) {
		const int w = tid + 256 * e;
		const uint4 a = ((const uint4 *)rec)[2 * w], b = ((const uint4 *)rec)[2 * w + 1];
		uint32_t x = code[w] & b.y, t;
		t = x & a.x; x = (x ^ t) | (t >> 1);
		t = x & a.y; x = (x ^ t) | (t >> 2);
		t = x & a.z; x = (x ^ t) | (t >> 4);
		t = x & a.w; x = (x ^ t) | (t >> 8);
		t = x & b.x; x = (x ^ t) | (t >> 16);
		const int off = (int)(b.z & 0xffffu), cnt = (int)(b.z >> 16), o = off & 31;
		if (x) {
			atomicOr(&mesg32[off >> 5], x << o);
			if (o + cnt > 32)
				atomicOr(&mesg32[(off >> 5) + 1], x >> (32 - o));
		}
	}
}

// CRC<uint32_t>(0xD419CC15) over the first 43072 bits = 5384 bytes of mesg[] (decode.cc:533-541), zero start, by a workgroup of 256
// threads (round 6; until then 32 threads walked 168 bytes each - two dependent LDS round trips per byte - and one thread folded the
// 32 results: 125 of k_back's 547 us per chunk with seven eighths of the workgroup waiting, profiles/r06_back_by_stage.txt).
// Thread k takes bytes 21 k .. 21 k + 20 (the last one 29) from a zero state; the CRC is linear over GF(2), so that state advanced
// by the bytes that follow the segment - bit by bit through adv[k][32] (tables.cpp: crc32_adv) - is the segment's share of the whole
// CRC, and the shares are XORed over the workgroup.  ctab: the byte table in LDS; red: four words of LDS.  All threads get the CRC.
__device__ __forceinline__ uint32_t crc32_wg256(const uint8_t *mesg, const uint32_t *ctab, const uint32_t *__restrict__ adv, uint32_t *red, int tid)
{
	const uint4 *m = (const uint4 *)(adv + tid * 32);
	uint4 row[8];
	#pragma unroll
	for (int w = 0; w < 8; ++w)
		row[w] = m[w];                                            // (in flight while the bytes are walked)
	const uint8_t *mp = mesg + 21 * tid;
	uint32_t crc = 0;
	#pragma unroll
	for (int i = 0; i < 21; ++i)
		crc = (crc >> 8) ^ ctab[(crc ^ mp[i]) & 255];
	if (tid == 255)
		for (int i = 21; i < 29; ++i)
			crc = (crc >> 8) ^ ctab[(crc ^ mp[i]) & 255];
	uint32_t acc = 0;
	#pragma unroll
	for (int w = 0; w < 8; ++w) {
		acc ^= (0u - ((crc >> (4 * w)) & 1u)) & row[w].x;
		acc ^= (0u - ((crc >> (4 * w + 1)) & 1u)) & row[w].y;
		acc ^= (0u - ((crc >> (4 * w + 2)) & 1u)) & row[w].z;
		acc ^= (0u - ((crc >> (4 * w + 3)) & 1u)) & row[w].w;
	}
	#pragma unroll
	for (int d = 32; d; d >>= 1)
		acc ^= (uint32_t)__shfl_xor((int)acc, d);
	__syncthreads();                                              // (red may still be read from an earlier call)
	if ((tid & 63) == 0)
		red[tid >> 6] = acc;
	__syncthreads();
	return red[0] ^ red[1] ^ red[2] ^ red[3];
}
__device__ __forceinline__ int wave_min_i(int v)
{
	#pragma unroll
	for (int m = 32; m; m >>= 1)
		v = min(v, __shfl_xor(v, m));
	return v;
}

}  // namespace rx
