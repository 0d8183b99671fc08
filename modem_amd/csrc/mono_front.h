// mono_front.h -- D1 for mono input (Decoder::next_sample, decode.cc:294-301: BlockDC, then Hilbert<cmplx, filter_len>) computed
// where the analytic signal is used (round 4), instead of a front pass that writes 8 bytes per sample for the whole stream.
//
// The DC blocker y[n] = b (x[n] - x[n-1]) + a y[n-1] (x[-1] = y[-1] = 0) is y[n] = b x[n] - s[n-1] with the one-pole low pass
// s[n] = a s[n-1] + g x[n], g = b (1 - a), s[-1] = 0.  k_mono_carries (k_sync.hip) leaves s after every 64th sample of a frame
// (8 bytes per 64 samples, in double); from there a consumer runs the recurrence over the samples it needs - a blocked scan over
// its threads - and the filter_len-tap Hilbert FIR out of LDS:
//   MonoCover    the analytic signal of a range of samples into the frame's z buffer by a wave or a workgroup (k_sync: the part of
//                the stream the Schmidl-Cox scan walks, as it walks it; k_sync_accept, k_header: their windows; the demodulator at
//                rates above 8 kHz; k_front_end: the whole frame for the ANALYTIC tap).  The scan of a frame with an early preamble
//                touches a few thousand of its 95 200 samples, and those lines stay in L2.
//   k_demod at 8 kHz keeps the DC-blocked samples of a symbol in LDS and never forms z in memory (k_demod.hip).
// Arithmetic: fp32 inside a span (a few roundings per sample, like the reference's serial fp32 recurrence, whose own rounding walk is
// what the 1e-5 tolerance of the intermediates is for), the state that crosses spans in double.
#pragma once
#include "dev_common.h"
#include "kernels.h"

namespace rx {

constexpr int MONO_CK = 64;                                   // samples per kept state

template <int RATE> struct MonoCfg {
	static constexpr int FL = RateCfg<RATE>::FILTER_LEN, C = (FL - 1) / 2, NIM = (FL - 1) / 4;
	static constexpr int REACH = 2 * C - 1;                   // z[i] reads y[i - REACH .. i - 1] (centre i - C, odd taps up to C - 1 around it)
	static constexpr int HIST = (REACH + 31) / 32 * 32;
};

// one frame's raw samples and kept states
struct MonoFrame {
	const char *base;
	int fmt;
	long n;
	const double *ck;                                         // s after sample 64 (m + 1) - 1, m = 0 ..
	// The PCM's integers as they are, times scale() = the sample.  The recurrences take the integers and fold the scale into their
	// input coefficients g and b (one rounding of a coefficient instead of the three-instruction correctly rounded division per
	// sample: the filters' outputs move by an ulp, like every other rounding of the blocked recurrence against the serial one).
	__device__ __forceinline__ float scale() const { return fmt == 0 ? 1.0f / 32767.f : fmt == 1 ? 1.0f / 127.f : 1.f; }
	__device__ __forceinline__ float raw(long i) const
	{
		if (i < 0 || i >= n) return 0.f;
		if (fmt == 0) return (float)((const int16_t *)base)[i];
		if (fmt == 1) return (float)((int)((const uint8_t *)base)[i] - 128);
		return ((const float *)base)[i];
	}
	// eight consecutive raw samples from pos (pos a multiple of 8): for int16 input inside the frame one 16-byte load, or four 4-byte
	// ones where the frame does not start on 16 bytes (a 44.1 kHz frame is 1 049 580 bytes: three frames in four of a batch; the
	// per-sample path with its range checks took the front pass from 21 to 29 ms there)
	__device__ __forceinline__ void load8(long pos, float (&x)[8]) const
	{
		if (fmt == 0 && pos >= 0 && pos + 8 <= n) {
			int w[4];
			if ((((size_t)base + (size_t)pos * 2) & 15) == 0) {
				const int4 v = *(const int4 *)((const int16_t *)base + pos);
				w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
			} else {
				typedef int __attribute__((aligned(2))) int_at_2;
				const int_at_2 *q = (const int_at_2 *)((const int16_t *)base + pos);
				w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; w[3] = q[3];
			}
			#pragma unroll
			for (int q = 0; q < 4; ++q) {
				x[2 * q] = (float)(short)(w[q] & 0xffff);
				x[2 * q + 1] = (float)(short)(w[q] >> 16);
			}
		} else if (fmt == 1 && pos >= 0 && pos + 8 <= n) {            // 8-bit samples: eight bytes in two accesses
			typedef uint32_t __attribute__((aligned(1))) u32_at_1;
			const u32_at_1 *q = (const u32_at_1 *)((const uint8_t *)base + pos);
			const uint32_t w[2] = { q[0], q[1] };
			#pragma unroll
			for (int i = 0; i < 8; ++i)
				x[i] = (float)((int)((w[i >> 2] >> (8 * (i & 3))) & 255u) - 128);
		} else if (fmt == 2 && pos >= 0 && pos + 8 <= n) {            // float32 samples: two 16-byte accesses
			typedef float4 __attribute__((aligned(4))) float4_at_4;
			const float4_at_4 *q = (const float4_at_4 *)((const float *)base + pos);
			const float4 a = q[0], b = q[1];
			x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
			x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
		} else {
			#pragma unroll
			for (int i = 0; i < 8; ++i)
				x[i] = raw(pos + i);
		}
	}
	__device__ __forceinline__ double state_before(long pos) const   // pos a multiple of MONO_CK: s[pos - 1]
	{
		return pos <= 0 ? 0.0 : ck[pos / MONO_CK - 1];
	}
};
__device__ __forceinline__ MonoFrame mono_frame(const FrameBatch &fb, const double *ck_all, int ck_per_frame, int f)
{
	return MonoFrame{ (const char *)fb.samples + (size_t)f * fb.frame_stride_bytes, fb.fmt, fb.samples_per_frame, ck_all + (size_t)f * ck_per_frame };
}
__device__ __forceinline__ double mono_pow(double a, int e)   // a^e, e >= 0
{
	double pw = 1.0, bs = a;
	while (e) {
		if (e & 1)
			pw *= bs;
		bs *= bs;
		e >>= 1;
	}
	return pw;
}

// z[i] from the DC-blocked samples around it: y(k) = the sample i - REACH + k, k = 0 .. REACH - 1 (decode.cc:299; Hilbert<cmplx, FL>:
// real part = the centre tap, imaginary part = the odd taps, antisymmetric)
template <int RATE, class Y>
__device__ __forceinline__ cf mono_hilbert(const FrontCoef &co, Y y)
{
	typedef MonoCfg<RATE> MC;
	constexpr int c = MC::REACH - MC::C;                      // the centre i - C as an index k
	const float re = co.reco * y(c);
	float im = co.imco[0] * (y(c - 1) - y(c + 1));
	#pragma unroll
	for (int k = 1; k < MC::NIM; ++k)
		im += co.imco[k] * (y(c - (2 * k + 1)) - y(c + (2 * k + 1)));
	return mk(re, im);
}

// the analytic signal of [lo, hi) into z, by NT threads (64: a wave of its own, 256: a workgroup), in spans of 8 NT samples that
// start on a kept state.  All calls are uniform over the NT threads.
template <int RATE, int NT> struct MonoCover {
	typedef MonoCfg<RATE> MC;
	static constexpr int PER = 8, LEN = NT * PER, HIST = MC::HIST, NW = NT / 64;
	static constexpr bool CONTIG = MC::REACH <= 24;           // (8 kHz) see span()
	static constexpr int RS = (HIST + LEN) / 8;               // the long filters' layout: row stride
	static constexpr int ZS = NT + 8;                         // ... and the row stride of the span's outputs on their way to memory
	struct Shared {
		float y[HIST + LEN + (HIST + LEN) / 32 + 1];          // [0, HIST): the samples before the span; one pad word per 32
		double wave_end[NW];
		float zt[CONTIG ? 1 : 2 * 8 * ZS];                    // long filters: a span's z, real and imaginary parts register major, so that
		                                                      // it leaves in the order of the samples (512 contiguous bytes per wave and store)
	};
	// where sample p of the buffer lives.  8 kHz: in place, one pad word per 32 (a thread's eight samples are consecutive words; the pad
	// keeps the lanes on different banks).  The long filters (41 - 125 taps): "register major" - sample 8 c + i at row i, column c -
	// so that the k-th value of EVERY thread's window sits at that thread's column plus a compile-time constant: no address
	// arithmetic per read, consecutive lanes on consecutive words.
	static __device__ __forceinline__ int pad(int p) { return CONTIG ? p + (p >> 5) : (p & 7) * RS + (p >> 3); }

	MonoFrame fr;
	Shared *sh;
	cf *z;                                                    // the frame's analytic signal
	long lo, next;                                            // z is valid on [lo, next) once `any`
	long hi;                                                  // nothing at or beyond it is stored (the frame's end; k_front_end: its stretch's end,
	                                                          // so that two workgroups never store the same sample from different scan groupings)
	bool any;
	double S;                                                 // s[next - 1]
	float Alane, W16, W32;                                    // a^(8 lane); a^(8 ((lane & 15) + 1)), a^(8 ((lane & 31) + 1)): WScan

	__device__ __forceinline__ void init(const MonoFrame &frame, const MonoArgs &ma, Shared *shared, cf *z_frame, int tid)
	{
		fr = frame; sh = shared; z = z_frame;
		lo = next = 0;
		hi = frame.n;
		any = false;
		S = 0.0;
		Alane = (float)mono_pow((double)ma.a, PER * (tid & 63));
		W16 = (float)mono_pow((double)ma.a, PER * ((tid & 15) + 1));
		W32 = (float)mono_pow((double)ma.a, PER * ((tid & 31) + 1));
	}
	// forget what is covered and start over at (a kept state at or before) p - REACH
	__device__ __forceinline__ void start(long p, int tid)
	{
		long q = p - MC::REACH;
		q = q <= 0 ? 0 : q / MONO_CK * MONO_CK;
		next = q;
		lo = p < 0 ? 0 : p;                                   // (the outputs before q + REACH lack their history; nothing before p is written,
		                                                      // so that two workgroups never write the same sample - k_front_end's stretches)
		S = fr.state_before(q);
		sync();
		for (int i = tid; i < HIST; i += NT)
			sh->y[pad(i)] = 0.f;
		sync();
	}
	static __device__ __forceinline__ void sync()
	{
		if (NT == 64)
			fft_sync<64>();
		else
			__syncthreads();
	}
	// one more span: z[next, next + LEN) (inside the frame)
	__device__ __forceinline__ void span(const MonoArgs &ma, int tid)
	{
		const float a = ma.a, g = ma.g * fr.scale(), b = ma.b * fr.scale();
		const int lane = tid & 63, wave = tid >> 6;
		float x[PER], sl[PER];
		fr.load8(next + (long)tid * PER, x);
		float acc = 0.f;
		#pragma unroll
		for (int i = 0; i < PER; ++i) {
			acc = fmaf(a, acc, g * x[i]);
			sl[i] = acc;
		}
		const float v = WScan<float>::run(acc, ma.astep8, W16, W32);   // weighted inclusive scan of the thread ends over the wave
		double carry = S;                                     // the state entering this wave
		if (NW > 1) {
			if (lane == 63)
				sh->wave_end[wave] = (double)v;
			__syncthreads();
			double st = 0.0, aw = 1.0;                        // aw = a^(512 w)
			#pragma unroll
			for (int w = 0; w < NW; ++w) {
				if (w == wave)
					carry = st + aw * S;
				st = sh->wave_end[w] + ma.awave8 * st;
				aw *= ma.awave8;
			}
			S = st + aw * S;
		} else {
			const float e = __shfl(v, 63);
			S = (double)e + ma.awave8 * S;
		}
		const float cin = fmaf(Alane, (float)carry, dpp_f<0x138>(v));   // (the lane before; lane 0: 0)
		float sprev = cin, yv[PER];
		#pragma unroll
		for (int i = 0; i < PER; ++i) {
			const float s = fmaf(ma.apw[i], cin, sl[i]);
			yv[i] = fmaf(b, x[i], -sprev);                        // y[n] = b x[n] - s[n-1]
			sh->y[pad(HIST + tid * PER + i)] = yv[i];
			sprev = s;
		}
		sync();
		if constexpr (CONTIG) {
			// a thread filters its own eight samples: seven of the REACH + 7 values around them are in its registers, REACH come out of LDS
			// (the other mapping reads (REACH + 1) / 2 + 1 values per sample: 88 LDS reads per thread instead of 19 at 8 kHz)
			float w[MC::REACH + PER - 1];
			#pragma unroll
			for (int k = 0; k < MC::REACH; ++k)
				w[k] = sh->y[pad(HIST + tid * PER - MC::REACH + k)];
			#pragma unroll
			for (int i = 0; i < PER - 1; ++i)
				w[MC::REACH + i] = yv[i];
			const long i0 = next + (long)tid * PER;
			cf r[PER];
			#pragma unroll
			for (int q = 0; q < PER; ++q)
				r[q] = mono_hilbert<RATE>(ma.co, [&](int k) { return w[q + k]; });
			if (next >= lo && next + LEN <= hi) {                 // (uniform) the whole span is wanted: 64 contiguous bytes per thread
				#pragma unroll
				for (int q = 0; q < PER; ++q)
					z[i0 + q] = r[q];
			} else {
				#pragma unroll
				for (int q = 0; q < PER; ++q)
					if (i0 + q >= lo && i0 + q < hi)
						z[i0 + q] = r[q];
			}
		} else {
			// The long filters: a thread filters its own eight samples too, out of a window of REACH + 7 values around them.  The odd
			// taps of a sample reach values of the other parity only, so the window is taken one parity class at a time: the class
			// holds the taps of four of the samples (q0, q0 + 2, q0 + 4, q0 + 6) and the centre taps of the other four.  Tap pair k of
			// those four samples reads four neighbouring values of the class below the centre and four above it, and the next pair the
			// same two windows moved by one: two new LDS reads per pair, eight live values - the sums run in mono_hilbert's order.
			// (Round 5, first form: the whole class in registers - 66 at 48 kHz, 125 VGPRs, four workgroups per CU: 16.1 ms per 8192
			// frames of 48 kHz, 15.4 this way - the pass writes 8 bytes per sample, and 37 GB of writes are 13 - 15 ms on this memory
			// system.  Before round 5 a lane took samples tid + 256 q and read its 63 values per sample through pad(): three address
			// instructions per read, 34 - 36 ms.)
			static_assert(PER == 8 && HIST % 8 == 0 && HIST >= MC::REACH && MC::NIM >= 2, "window layout");
			constexpr int CEN = MC::REACH - MC::C, K0 = HIST - MC::REACH;   // w(k) = buffer sample K0 + 8 tid + k
			const float *col = sh->y + tid;
			auto wv = [&](int k) { return col[((K0 + k) & 7) * RS + ((K0 + k) >> 3)]; };   // (k: a compile-time constant wherever it is called)
			float rre[PER], rim[PER];
			#pragma unroll
			for (int ph = 0; ph < 2; ++ph) {
				const int q0 = ((CEN + 1 + ph) & 1);                  // the samples whose odd taps lie in this class: q0 + 2 j
				float lo[4], hi[4], im[4];
				#pragma unroll
				for (int j = 0; j < 4; ++j) {
					lo[j] = wv(q0 + 2 * j + CEN - 1);
					hi[j] = wv(q0 + 2 * j + CEN + 1);
					im[j] = ma.co.imco[0] * (lo[j] - hi[j]);
				}
				#pragma unroll
				for (int k = 1; k < MC::NIM; ++k) {
					#pragma unroll
					for (int j = 3; j > 0; --j)
						lo[j] = lo[j - 1];                            // the window below the centres moves down by one value of the class
					lo[0] = wv(q0 + CEN - (2 * k + 1));
					#pragma unroll
					for (int j = 0; j < 3; ++j)
						hi[j] = hi[j + 1];                            // ... the one above them up
					hi[3] = wv(q0 + 6 + CEN + (2 * k + 1));
					#pragma unroll
					for (int j = 0; j < 4; ++j)
						im[j] += ma.co.imco[k] * (lo[j] - hi[j]);
				}
				#pragma unroll
				for (int j = 0; j < 4; ++j) {
					rim[q0 + 2 * j] = im[j];
					rre[(q0 ^ 1) + 2 * j] = ma.co.reco * wv((q0 ^ 1) + 2 * j + CEN);   // the centre taps of the other four samples
				}
			}
			#pragma unroll
			for (int q = 0; q < PER; ++q) {
				sh->zt[q * ZS + tid] = rre[q];
				sh->zt[(8 + q) * ZS + tid] = rim[q];
			}
			sync();
			#pragma unroll
			for (int q = 0; q < PER; ++q) {
				const int j = tid + NT * q;                       // consecutive samples across the lanes
				const long i = next + j;
				const int a = (j & 7) * ZS + (j >> 3);
				if (i >= lo && i < hi)
					z[i] = mk(sh->zt[a], sh->zt[8 * ZS + a]);
			}
		}
		sync();
		for (int i = tid; i < HIST; i += NT)                  // the span's tail is the next span's history
			sh->y[pad(i)] = sh->y[pad(LEN + i)];
		next += LEN;
		sync();
	}
	// make z valid on [p_lo, p_hi) (clipped to the frame); what was valid stays valid unless p_lo lies before it
	__device__ __forceinline__ void cover(const MonoArgs &ma, long p_lo, long p_hi, int tid)
	{
		if (p_lo < 0) p_lo = 0;
		if (p_hi > fr.n) p_hi = fr.n;
		if (p_hi <= p_lo)
			return;
		if (!any || p_lo < lo || p_lo > next) {               // nothing yet, or a window that begins before / beyond what is covered
			if (any && p_lo < lo && next > p_hi)
				p_hi = next;                                  // (the scan goes on from where it was)
			start(p_lo, tid);
			any = true;
		}
		while (next < p_hi)
			span(ma, tid);
	}
};

}  // namespace rx
