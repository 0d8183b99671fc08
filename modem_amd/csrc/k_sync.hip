// k_sync.hip -- D1 (front end) and D2/D3 (Schmidl-Cox search) for gfx950.
//
// D1  Decoder::next_sample (decode.cc:294-301): BlockDC + Hilbert<cmplx,21> for
//     mono input.  The 1st-order DC blocker is a linear recurrence: every thread
//     runs it over a contiguous chunk from a zero state, the 256 chunk carries
//     are composed serially, then the chunks are corrected by a^k * carry.
// D2  SchmidlCox::operator() per-sample part (decode.cc:84-108): the three
//     SMA4 sliding sums are running sums of (in - out) differences, prefix-summed
//     in double precision per 1024-sample tile (one wave per frame, 16 samples
//     per lane).  Trigger logic (Schmitt + falling edge + arg-max with the
//     saturating age counter) is evaluated on the tile with wave reductions.
// D3  trigger part (decode.cc:110-151): derotate 640, FFT640, differential
//     demod across bins, FFT640, x conj(FFT(mls0))/640, IFFT640, peak search.
#include "dev_common.h"
#include "kernels.h"

// Decision-critical fp32 expressions (cfo_rad = shift*2pi/640 - frac_cfo sits at magnitude ~6 before the
// wrap, so ONE ulp there is 5e-7 rad/sample = 7e-4 rad per symbol): no FMA contraction in this file, like
// the strict-IEEE CPU restatement.
#pragma clang fp contract(off)

namespace rx {

// ---------------------------------------------------------------- D1 front end
// Decoder::next_sample (decode.cc:294-301) for mono input: y = BlockDC(x), z = Hilbert<cmplx,21>(y).
// The DC blocker y[n] = b(x[n]-x[n-1]) + a y[n-1] is a linear recurrence over the whole stream.  Round 3: one workgroup per
// (frame, TILE of 4096 samples) instead of one per frame walking its 24 tiles in turn (6.9 ms per 8192 frames, more than
// Theil-Sen, all of it a serial chain of barriers at low occupancy).  Two passes:
//   k_front_dc      every tile from a ZERO state: 16 consecutive samples per thread, the 256 thread-end states combined by a
//                   weighted inclusive scan (v[t] += A^d v[t-d], A = a^16).  Kept per tile: its end state and the zero-start
//                   values of its last FE_HIST samples (the history the next tile's Hilbert taps reach into).
//   k_front_end     the true state at a tile's start is the fold of the earlier tiles' end states, C_t = e_{t-1} + a^4096 C_{t-1}
//                   (<= 24 terms; 140 at 48 kHz); the tile's recurrence is run again and every sample corrected by
//                   a^(k+1) C_t, the history samples by the same rule from the kept values and C_{t-1}; then the 21-tap
//                   Hilbert FIR out of LDS and 128 contiguous bytes of z per lane.
// The recurrence is evaluated in double and rounded once per sample: a parallel scan cannot reproduce the rounding sequence of
// the serial fp32 recurrence anyway, so the GPU side is made (nearly) exact and the difference to the CPU's serial fp32 filter
// is that filter's own rounding.  Other rates: Hilbert<cmplx, filter_len> with filter_len = 41 / 113 / 125 (decode.cc:172).
constexpr int FE_PER = 16, FE_TILE = 256 * FE_PER;
template <int RATE> struct FeCfg {
	static constexpr int FL = RateCfg<RATE>::FILTER_LEN, HIST = (FL - 1 + 31) / 32 * 32, C = (FL - 1) / 2, NIM = (FL - 1) / 4;
	static constexpr int REC = 1 + HIST;                      // doubles kept per tile: end state + the history values
};

// the tile's samples (thread tid: FE_PER consecutive ones from s0), the recurrence from a zero state per thread (y[i]), and the
// state entering this thread when the TILE starts from zero (cin0): the pieces both passes need
template <int RATE>
__device__ __forceinline__ void fe_tile_local(const SampleSrc &src, const char *base, int fmt, long n, long s0, double a, double b,
	const double (&Apow)[7], int lane, int wave, double *wave_end, double (&y)[FE_PER], double &cin0, double &tile_end)
{
	float x[FE_PER + 1];
	x[0] = (s0 - 1 >= 0 && s0 - 1 < n) ? src.scalar(s0 - 1) : 0.f;
	if (fmt == 0 && s0 + FE_PER <= n && (((size_t)base + (size_t)s0 * 2) & 15) == 0) {
		const int4 *p = (const int4 *)((const int16_t *)base + s0);
		int4 v0 = p[0], v1 = p[1];
		const int w[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
		#pragma unroll
		for (int q = 0; q < 8; ++q) {
			x[1 + 2 * q] = div_32767((float)(short)(w[q] & 0xffff));
			x[2 + 2 * q] = div_32767((float)(short)(w[q] >> 16));
		}
	} else {
		#pragma unroll
		for (int i = 0; i < FE_PER; ++i)
			x[1 + i] = s0 + i < n ? src.scalar(s0 + i) : 0.f;
	}
	double yl = 0.0;
	#pragma unroll
	for (int i = 0; i < FE_PER; ++i) {
		yl = b * (double)(x[i + 1] - x[i]) + a * yl;
		y[i] = yl;
	}
	// weighted inclusive scan of the thread end states over the 64 lanes of the wave
	double v = yl;
	#pragma unroll
	for (int sft = 0; sft < 6; ++sft) {
		double o = shfl_up_d(v, 1 << sft);
		if (lane >= (1 << sft))
			v += Apow[sft] * o;
	}
	if (lane == 63)
		wave_end[wave] = v;
	__syncthreads();
	double st = 0.0;                                              // state at the end of the previous waves of this tile
	for (int w2 = 0; w2 < wave; ++w2)
		st = wave_end[w2] + Apow[6] * st;                         // A^64 decays a whole wave (1024 samples)
	const double prev = shfl_up_d(v, 1);
	double decay = 1.0;                                           // A^lane
	#pragma unroll
	for (int sft = 0; sft < 6; ++sft)
		if (lane & (1 << sft))
			decay *= Apow[sft];
	cin0 = (lane ? prev : 0.0) + decay * st;
	tile_end = ((wave_end[3] + Apow[6] * wave_end[2]) + Apow[6] * Apow[6] * wave_end[1]) + Apow[6] * Apow[6] * Apow[6] * wave_end[0];
}
template <int RATE>
__device__ __forceinline__ void fe_powers(double a, double (&apow)[FE_PER], double (&Apow)[7])
{
	apow[0] = a;                                                  // a^1..a^16; A^(2^s) = a^(16*2^s) for the scan steps
	#pragma unroll
	for (int i = 1; i < FE_PER; ++i)
		apow[i] = apow[i - 1] * a;
	Apow[0] = apow[FE_PER - 1];
	#pragma unroll
	for (int sft = 1; sft < 7; ++sft)
		Apow[sft] = Apow[sft - 1] * Apow[sft - 1];
}

template <int RATE>
__global__ __launch_bounds__(256) void k_front_dc(FrameBatch fb, FrontCoef co, double *__restrict__ rec_all, int tiles)
{
	using FC = FeCfg<RATE>;
	const int f = blockIdx.y, t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const long n = fb.samples_per_frame;
	const char *base = (const char *)fb.samples + (size_t)f * fb.frame_stride_bytes;
	SampleSrc src{ base, fb.fmt, 1, n, nullptr };
	__shared__ double wave_end[4];
	const double a = (double)co.dc_a, b = (double)co.dc_b;
	double apow[FE_PER], Apow[7], y[FE_PER], cin0, tile_end;
	fe_powers<RATE>(a, apow, Apow);
	const long s0 = (long)t * FE_TILE + (long)tid * FE_PER;
	fe_tile_local<RATE>(src, base, fb.fmt, n, s0, a, b, Apow, lane, wave, wave_end, y, cin0, tile_end);
	double *rec = rec_all + ((size_t)f * tiles + t) * FC::REC;
	if (tid == 0)
		rec[0] = tile_end;
	#pragma unroll
	for (int i = 0; i < FE_PER; ++i) {
		const int k = tid * FE_PER + i - (FE_TILE - FC::HIST);    // index into the kept history
		if (k >= 0)
			rec[1 + k] = y[i] + apow[i] * cin0;
	}
}

template <int RATE>
__global__ __launch_bounds__(256) void k_front_end(FrameBatch fb, FrontCoef co, const double *__restrict__ rec_all, int tiles, cf *__restrict__ z_all)
{
	using FC = FeCfg<RATE>;
	constexpr int FE_HIST = FC::HIST, FE_C = FC::C, FE_NIM = FC::NIM;
	const int f = blockIdx.y, t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const long n = fb.samples_per_frame;
	const char *base = (const char *)fb.samples + (size_t)f * fb.frame_stride_bytes;
	SampleSrc src{ base, fb.fmt, 1, n, nullptr };
	cf *z = z_all + (size_t)f * fb.samples_per_frame;
	// y of the tile, [0, FE_HIST) = tail of the previous tile.  One pad word per 32: the recurrence writes 16 consecutive samples
	// per thread (stride 16 words across the lanes: two banks without the pad), the FIR reads consecutive samples across the lanes
	constexpr int YN = FE_HIST + FE_TILE;
	__shared__ float ydc[YN + YN / 32 + 1];
	auto pad = [](int p) { return p + (p >> 5); };
	__shared__ double wave_end[4];
	const double a = (double)co.dc_a, b = (double)co.dc_b;
	double apow[FE_PER], Apow[7], y[FE_PER], cin0, tile_end;
	fe_powers<RATE>(a, apow, Apow);
	// state at the start of the previous tile (Cp) and of this one (Ct): fold of the earlier tiles' end states
	const double Atile = Apow[6] * Apow[6] * Apow[6] * Apow[6];   // a^4096
	const double *rec_f = rec_all + (size_t)f * tiles * FC::REC;
	__shared__ double ends_sh[256];
	double Cp = 0.0, Ct = 0.0;
	for (int q0 = 0; q0 < t; q0 += 256) {                         // (the end states through LDS: one round trip per 256 tiles, not per tile)
		__syncthreads();
		if (q0 + tid < t)
			ends_sh[tid] = rec_f[(size_t)(q0 + tid) * FC::REC];
		__syncthreads();
		const int m = t - q0 < 256 ? t - q0 : 256;
		for (int q = 0; q < m; ++q) {
			Cp = Ct;
			Ct = ends_sh[q] + Atile * Ct;
		}
	}
	if (tid < FE_HIST) {                                          // the previous tile's last samples, from their zero-start values
		float h = 0.f;
		if (t > 0) {
			int e = FE_TILE - FE_HIST + tid + 1;                  // a^(position in that tile + 1)
			double pw = 1.0, bs = a;
			while (e) {
				if (e & 1)
					pw *= bs;
				bs *= bs;
				e >>= 1;
			}
			h = (float)(rec_f[(size_t)(t - 1) * FC::REC + 1 + tid] + pw * Cp);
		}
		ydc[pad(tid)] = h;
	}
	const long s0 = (long)t * FE_TILE + (long)tid * FE_PER;
	fe_tile_local<RATE>(src, base, fb.fmt, n, s0, a, b, Apow, lane, wave, wave_end, y, cin0, tile_end);
	{
		double decay = 1.0;                                       // a^(16 * tid): what the tile's own entry state has decayed to at this thread
		#pragma unroll
		for (int sft = 0; sft < 6; ++sft)
			if (lane & (1 << sft))
				decay *= Apow[sft];
		for (int w2 = 0; w2 < wave; ++w2)
			decay *= Apow[6];
		const double cin = cin0 + decay * Ct;
		#pragma unroll
		for (int i = 0; i < FE_PER; ++i)
			ydc[pad(FE_HIST + tid * FE_PER + i)] = (float)(y[i] + apow[i] * cin);
	}
	__syncthreads();
	// Hilbert<cmplx,21>: centre tap 10 back, odd taps +-1,3,5,7,9 around it.  Thread tid takes the samples tid + 256 i: consecutive
	// words of LDS across the lanes, 512 contiguous bytes of z per wave instruction
	const long t0 = (long)t * FE_TILE;
	#pragma unroll 4
	for (int i = 0; i < FE_PER; ++i) {
		const int li = FE_HIST + tid + 256 * i;                   // index of sample t0 + tid + 256 i ; centre at li-10
		const int c = li - FE_C;
		float re = co.reco * ydc[pad(c)];
		float im = co.imco[0] * (ydc[pad(c - 1)] - ydc[pad(c + 1)]);
		#pragma unroll
		for (int k = 1; k < FE_NIM; ++k)
			im += co.imco[k] * (ydc[pad(c - (2 * k + 1))] - ydc[pad(c + (2 * k + 1))]);
		if (t0 + tid + 256 * i < n)
			z[t0 + tid + 256 * i] = mk(re, im);
	}
}

// ---------------------------------------------------------------- D2 + D3 sync
#ifndef SYNC_PER
#define SYNC_PER 8     // sample times per lane and tile: 8 -> 224 VGPRs and 22 KB of LDS, so that a sync wave fits on a SIMD beside
                       // three resident polar decoders (the whole front runs inside the polar phase); 16 -> 256 VGPRs, 30 KB
#endif
constexpr int PER = SYNC_PER, TILE = 64 * PER;
// ring of the last TILE + match_len metric values (a power of two): 1024 at 8 / 16 kHz, 2048 at 44.1 / 48 kHz
template <int RATE> struct SyncRing { static constexpr int N = TILE + RateCfg<RATE>::MATCH_LEN <= 1024 ? 1024 : 2048; };

// 8 kHz: the two 640-point work arrays of the trigger part live in LDS.  Other rates (1280 / 3528 / 3840
// points) keep them in a per-frame global scratch so the scanning loop's occupancy does not pay for them.
#ifndef SYNC_FFT_IN_LDS
#define SYNC_FFT_IN_LDS 1    // 8 kHz: the two 640-point buffers of the S&C trigger part in LDS (0: global scratch like the other rates)
#endif
#define SYNC_FFT_LDS(R) ((R) == 8000 && SYNC_FFT_IN_LDS)
#ifndef SYNC_WAVES
#define SYNC_WAVES 2
#endif
#ifndef SYNC_WAVES_SPLIT
#define SYNC_WAVES_SPLIT 4      // register budget of the split scan (k_sync<RATE, true>): 124 VGPRs, 10 KB of LDS = 16 waves per CU
#endif
// Both arrays are read and written at lane * PER + e (+ a ring offset): one pad word per eight keeps the lanes of a group on
// distinct banks (stride 9 instead of 8).  The ring holds fp32 metric values (they are summed in double after the read).
__device__ __forceinline__ int sync_pad(int i) { return i + (i >> 3); }
template <int RATE, bool SPLIT = false> struct SyncShared {   // (the split scan never runs the accept path's transforms)
	float m[SyncRing<RATE>::N + SyncRing<RATE>::N / 8];
	float timing[TILE + TILE / 8];
	cf buf[SYNC_FFT_LDS(RATE) && !SPLIT ? RateCfg<RATE>::HS : 1];
	cf xr[SYNC_FFT_LDS(RATE) && !SPLIT ? RateCfg<RATE>::HS : 1];
};

__device__ __forceinline__ int first_index(const float *timing, int T0, int lane, int lo_t, int hi_t, bool greater, float thr)
{
	int best = 0x7fffffff;
	#pragma unroll
	for (int e = 0; e < PER; ++e) {
		int t = T0 + lane * PER + e;
		float v = timing[sync_pad(lane * PER + e)];
		bool c = greater ? (v > thr) : (v < thr);
		if (t >= lo_t && t < hi_t && c && best == 0x7fffffff)
			best = t;
	}
	return wave_min_i(best);
}

// P at time t by direct summation (decode.cc:86), double accumulate
template <int RATE>
__device__ __forceinline__ void direct_P(const SampleSrc &src, long t, int lane, double &re, double &im)
{
	constexpr int BUFFER_LEN = RateCfg<RATE>::BUFFER_LEN, SEARCH_POS = RateCfg<RATE>::SEARCH_POS, HALF_LEN = RateCfg<RATE>::HS;
	double sr = 0.0, si = 0.0;
	long a0 = t - (BUFFER_LEN - 1 - (SEARCH_POS + HALF_LEN));   // newest u
	for (int q = 0; q < (HALF_LEN + 63) / 64; ++q) {
		if (q * 64 + lane >= HALF_LEN)
			break;
		long u = a0 - (q * 64 + lane);
		cf x = src.at(u), y = src.at(u + HALF_LEN);
		sr += (double)x.re * y.re + (double)x.im * y.im;
		si += (double)x.im * y.re - (double)x.re * y.im;
	}
	re = wave_sum_d(sr);
	im = wave_sum_d(si);
}
template <int RATE>
__device__ __forceinline__ double direct_R(const SampleSrc &src, long t, int lane)
{
	constexpr int BUFFER_LEN = RateCfg<RATE>::BUFFER_LEN, SEARCH_POS = RateCfg<RATE>::SEARCH_POS, HALF_LEN = RateCfg<RATE>::HS;
	double s = 0.0;
	long a0 = t - (BUFFER_LEN - 1 - (SEARCH_POS + 2 * HALF_LEN));
	for (int q = 0; q < (2 * HALF_LEN + 63) / 64; ++q) {
		if (q * 64 + lane >= 2 * HALF_LEN)
			break;
		cf x = src.at(a0 - (q * 64 + lane));
		s += (double)x.re * x.re + (double)x.im * x.im;
	}
	return wave_sum_d(s);
}

// decode.cc:110-151 ; returns accept, fills symbol_pos (window coord) and cfo_rad
template <int RATE>
__device__ bool sc_process(cf *buf, cf *xr, const SampleSrc &src, const cf *tw, const cf *kern,
	long t, int index_max, float phase_max, int lane, int &symbol_pos_out, float &cfo_out)
{
	typedef RateCfg<RATE> RC;
	constexpr int BUFFER_LEN = RC::BUFFER_LEN, SEARCH_POS = RC::SEARCH_POS, HALF_LEN = RC::HS, GUARD_LEN = RC::GL;
	const float frac_cfo = phase_max / (float)HALF_LEN;       // decode.cc:110
	int symbol_pos = SEARCH_POS - index_max;                   // decode.cc:114
#ifdef SYNC_PROBE_NO_SC
	symbol_pos_out = symbol_pos; cfo_out = 0.f; return true;   // timing probe: the trigger scan without the accept path
#endif
	const long base = t - (BUFFER_LEN - 1);
	__syncthreads();
	{   // decode.cc:117-118.  e^{j frac_cfo i}, i = 64 q + lane: one closed-form phasor per lane times one per q (lane q holds the
		// q-th: HALF_LEN / 64 <= 64), instead of a double-precision range reduction and a sincos per sample
		static_assert(HALF_LEN <= 64 * 64, "one lane per block of 64 samples");
		const cf p_lane = phasor(frac_cfo, lane), p_blk = phasor(frac_cfo, 64L * lane);
		for (int q = 0; q * 64 < HALF_LEN; ++q) {
			const int i = q * 64 + lane;
			const cf r = mk(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(p_blk.re), q)),
				__int_as_float(__builtin_amdgcn_readlane(__float_as_int(p_blk.im), q)));
			if (i < HALF_LEN)
				buf[i] = cmul(src.at(base + i + symbol_pos + HALF_LEN), cmul(p_lane, r));
		}
	}
	__syncthreads();
	fft_fwd<HALF_LEN, 64, RC::SL>(buf, tw, lane);
	for (int i = lane; i < HALF_LEN; i += 64)                  // decode.cc:120-121
		xr[i] = demod_or_erase(buf[i], buf[(i + HALF_LEN - 1) % HALF_LEN]);
	__syncthreads();
	for (int i = lane; i < HALF_LEN; i += 64)
		buf[i] = xr[i];
	__syncthreads();
	fft_fwd<HALF_LEN, 64, RC::SL>(buf, tw, lane);
	// x kern, then backward transform as conj(FFT(conj(.)))
	for (int i = lane; i < HALF_LEN; i += 64)
		buf[i] = cconj(cmul(buf[i], kern[i]));
	__syncthreads();
	fft_fwd<HALF_LEN, 64, RC::SL>(buf, tw, lane);
	// decode.cc:127-139: peak = max, shift = first index of it, next = runner-up
	float pk = -1.f;
	int sh_i = 0x7fffffff;
	for (int i = lane; i < HALF_LEN; i += 64) {
		float p = cnorm(buf[i]);
		if (p > pk) { pk = p; sh_i = i; }
	}
	#pragma unroll
	for (int m = 32; m; m >>= 1) {
		float op = __shfl_xor(pk, m);
		int oi = __shfl_xor(sh_i, m);
		if (op > pk || (op == pk && oi < sh_i)) { pk = op; sh_i = oi; }
	}
	float nx = 0.f;
	for (int i = lane; i < HALF_LEN; i += 64) {
		float p = cnorm(buf[i]);
		if (i != sh_i && p > nx) nx = p;
	}
	#pragma unroll
	for (int m = 32; m; m >>= 1)
		nx = fmaxf(nx, __shfl_xor(nx, m));
	const float peak = fmaxf(pk, 0.f);
	const int shift = peak > 0.f ? sh_i : 0;
	if (peak <= nx * 4.f)                                      // decode.cc:140-141
		return false;
	cf v = cconj(buf[shift]);
	int pos_err = (int)nearbyintf(atan2f(v.im, v.re) * (float)HALF_LEN / TWO_PI_F);
	if (abs(pos_err) > GUARD_LEN / 2)                          // decode.cc:144-145
		return false;
	symbol_pos -= pos_err;
	float cfo_rad = (float)shift * (TWO_PI_F / (float)HALF_LEN) - frac_cfo;   // decode.cc:148
	if (cfo_rad >= PI_F)
		cfo_rad -= TWO_PI_F;
	symbol_pos_out = symbol_pos;
	cfo_out = cfo_rad;
	return true;
}

// SPLIT = false: the whole of decode.cc:86-151 for one frame by one wave (the catch-all after SYNC_SPLIT_ROUNDS rejected triggers;
// the only sync kernel at 8 kHz until round 3).
// SPLIT = true: the scan stops at the first trigger and leaves (g, index_max, phase_max) in the frame's state; the accept
// path runs as k_sync_accept with a whole workgroup per frame.  At 44.1 / 48 kHz the three 3528 / 3840-point transforms of
// the accept path by ONE wave through global scratch (256 VGPRs + spills) were 60 % of this kernel's time.
template <int RATE, bool SPLIT>
__global__ __launch_bounds__(64, SPLIT ? SYNC_WAVES_SPLIT : SYNC_WAVES) void k_sync(FrameBatch fb, const cf *__restrict__ z_all, const cf *__restrict__ tw,
	const cf *__restrict__ kern, SyncState *__restrict__ st_all, cf *__restrict__ scratch)
{
	typedef RateCfg<RATE> RC;
	constexpr int BUFFER_LEN = RC::BUFFER_LEN, SEARCH_POS = RC::SEARCH_POS, HALF_LEN = RC::HS, GUARD_LEN = RC::GL;
	constexpr int MATCH_LEN = RC::MATCH_LEN, MATCH_DEL = RC::MATCH_DEL;
	constexpr int MRING = SyncRing<RATE>::N;
	static_assert(TILE + MATCH_LEN <= MRING, "m ring too small");
	const int f = blockIdx.x, lane = threadIdx.x;
	SyncState st = st_all[f];
	const long n = fb.samples_per_frame;
	if (!st.active || st.found || st.t_next >= n)              // found: accepted earlier in this round (split schedule)
		return;
	SampleSrc src{ (const char *)fb.samples + (size_t)f * fb.frame_stride_bytes, fb.fmt, fb.channels, n,
		fb.channels == 1 ? z_all + (size_t)f * fb.samples_per_frame : nullptr };
	__shared__ SyncShared<RATE, SPLIT> sh;
	cf *fbuf = SYNC_FFT_LDS(RATE) && !SPLIT ? sh.buf : scratch + (size_t)f * 2 * HALF_LEN;
	cf *fxr = SYNC_FFT_LDS(RATE) && !SPLIT ? sh.xr : scratch + (size_t)f * 2 * HALF_LEN + HALF_LEN;
	for (int i = lane; i < MRING + MRING / 8; i += 64)
		sh.m[i] = 0.f;
	const float thr_lo = (float)(0.17 * MATCH_LEN), thr_hi = (float)(0.19 * MATCH_LEN);   // decode.cc:76
	const float min_R = 0.0001f * HALF_LEN;                                                // decode.cc:88
	const long t_start = st.t_next;
	long T0 = t_start - (MATCH_LEN - 1);
	if (T0 < 0) T0 = 0;
	// running window sums at time T0-1
	double Wr = 0.0, Wi = 0.0, Wp = 0.0, Wm = 0.0;
	if (T0 > 0) {
		direct_P<RATE>(src, T0 - 1, lane, Wr, Wi);
		Wp = direct_R<RATE>(src, T0 - 1, lane);
	}
	bool collecting = false, found = false, pending = false;
	float tmax = 0.f;
	long nmax = 0;
	int rejects = st.rejects;
	__syncthreads();
	for (; T0 < n && !found && !pending; T0 += TILE) {
		// ---- phase 1: P, R, m for the 16 times of this lane
		double dr[PER], di[PER], dp[PER];
		{
			const long tb = T0 + lane * PER;
			auto phase1 = [&](auto at) {
				#pragma unroll
				for (int e = 0; e < PER; ++e) {
					long a = tb + e - (BUFFER_LEN - 1 - (SEARCH_POS + HALF_LEN));
					cf zA = at(a - HALF_LEN), zB = at(a), zC = at(a + HALF_LEN);
					double inr = (double)zB.re * zC.re + (double)zB.im * zC.im;
					double ini = (double)zB.im * zC.re - (double)zB.re * zC.im;
					double outr = (double)zA.re * zB.re + (double)zA.im * zB.im;
					double outi = (double)zA.im * zB.re - (double)zA.re * zB.im;
					double pin = (double)zC.re * zC.re + (double)zC.im * zC.im;
					double pout = (double)zA.re * zA.re + (double)zA.im * zA.im;
					double pr = inr - outr, pi = ini - outi, pp = pin - pout;
					dr[e] = (e ? dr[e - 1] : 0.0) + pr;
					di[e] = (e ? di[e - 1] : 0.0) + pi;
					dp[e] = (e ? dp[e - 1] : 0.0) + pp;
				}
			};
			// the whole tile's sample window inside the frame and int16 pairs (the rule): no format switch, no bounds checks
			const long w_lo = T0 - (BUFFER_LEN - 1 - (SEARCH_POS + HALF_LEN)) - HALF_LEN, w_hi = w_lo + TILE + 2 * HALF_LEN;
			if (src.mode() == 1 && w_lo >= 0 && w_hi <= n) {
				const short2 *p = (const short2 *)src.base;
				phase1([&](long i) { const short2 v = p[i]; return mk(div_32767((float)v.x), div_32767((float)v.y)); });
			} else {
				phase1([&](long i) { return src.at(i); });
			}
		}
		double or_ = wave_scan_incl(dr[PER - 1], lane) - dr[PER - 1] + Wr;
		double oi_ = wave_scan_incl(di[PER - 1], lane) - di[PER - 1] + Wi;
		double op_ = wave_scan_incl(dp[PER - 1], lane) - dp[PER - 1] + Wp;
		#pragma unroll
		for (int e = 0; e < PER; ++e) {
			float Pre = (float)(or_ + dr[e]), Pim = (float)(oi_ + di[e]);
			float R = 0.5f * (float)(op_ + dp[e]);
			R = fmaxf(R, min_R);
			// decode.cc:90: the fp32 expression of the reference, term by term (no contraction); only its moving sum runs in double
			const float mf = __fdiv_rn(__fadd_rn(__fmul_rn(Pre, Pre), __fmul_rn(Pim, Pim)), __fmul_rn(R, R));
			sh.m[sync_pad((int)((T0 + lane * PER + e) & (MRING - 1)))] = mf;
		}
		Wr = shfl_d(or_ + dr[PER - 1], 63);
		Wi = shfl_d(oi_ + di[PER - 1], 63);
		Wp = shfl_d(op_ + dp[PER - 1], 63);
		__syncthreads();
		// ---- phase 2: timing = sliding sum of m over 161 (decode.cc:90)
		double dm[PER];
		#pragma unroll
		for (int e = 0; e < PER; ++e) {
			long t = T0 + lane * PER + e;
			double in = (double)sh.m[sync_pad((int)(t & (MRING - 1)))];
			double out = (t - MATCH_LEN >= 0) ? (double)sh.m[sync_pad((int)((t - MATCH_LEN) & (MRING - 1)))] : 0.0;
			dm[e] = (e ? dm[e - 1] : 0.0) + (in - out);
		}
		double om = wave_scan_incl(dm[PER - 1], lane) - dm[PER - 1] + Wm;
		#pragma unroll
		for (int e = 0; e < PER; ++e)
			sh.timing[sync_pad(lane * PER + e)] = (float)(om + dm[e]);
		Wm = shfl_d(om + dm[PER - 1], 63);
		__syncthreads();
		// ---- trigger logic on the tile (decode.cc:93-108)
		long cur = T0 > t_start ? T0 : t_start;
		long tile_end = T0 + TILE < n ? T0 + TILE : n;
		while (cur < tile_end && !found && !pending) {
			if (!collecting) {
				int fr = first_index(sh.timing, (int)T0, lane, (int)cur, (int)tile_end, true, thr_hi);
				if (fr == 0x7fffffff)
					break;
				collecting = true;
				cur = fr;
			}
			int g = first_index(sh.timing, (int)T0, lane, (int)cur, (int)tile_end, false, thr_lo);
			long stop = g == 0x7fffffff ? tile_end : (long)g + 1;
			// first arg-max over [cur, stop)
			float bv = -1.f;
			int bi = 0x7fffffff;
			#pragma unroll
			for (int e = 0; e < PER; ++e) {
				int t = (int)T0 + lane * PER + e;
				float v = sh.timing[sync_pad(lane * PER + e)];
				if (t >= cur && t < stop && v > bv) { bv = v; bi = t; }
			}
			#pragma unroll
			for (int mm = 32; mm; mm >>= 1) {
				float ov = __shfl_xor(bv, mm);
				int oi = __shfl_xor(bi, mm);
				if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
			}
			if (tmax < bv) { tmax = bv; nmax = bi; }           // decode.cc:99-102
			if (g == 0x7fffffff) {
				cur = tile_end;
				break;
			}
			collecting = false;
			long age = MATCH_DEL + ((long)g - nmax);           // decode.cc:103-105
			int index_max = (int)(age < HALF_LEN + GUARD_LEN + MATCH_DEL ? age : HALF_LEN + GUARD_LEN + MATCH_DEL);
			float phase_max = 0.f;
			{
				long tp = nmax - MATCH_DEL;                    // decode.cc:91 delay(arg(P))
				if (tp >= 0) {
					double pr, pi;
					direct_P<RATE>(src, tp, lane, pr, pi);
					phase_max = atan2f((float)pi, (float)pr);
				}
			}
			tmax = 0.f;                                        // decode.cc:115-116
			if constexpr (SPLIT) {
				pending = true;                                // k_sync_accept decides; a rejected frame resumes at g + 1
				st.pend_g = g;
				st.pend_index_max = index_max;
				st.pend_phase = phase_max;
				st.t_next = (long)g + 1;
			} else {
				int sp;
				float cfo;
				if (sc_process<RATE>(fbuf, fxr, src, tw, kern, g, index_max, phase_max, lane, sp, cfo)) {
					found = true;
					st.symbol_pos = sp;
					st.cfo_rad = cfo;
					st.sc_start = (long)g - (BUFFER_LEN - 1) + sp;
					st.t_next = (long)g + 1;
				} else {
					++rejects;
				}
			}
			cur = (long)g + 1;
		}
		__syncthreads();
	}
	st.rejects = rejects;
	st.found = found ? 1 : 0;
	st.pending = pending ? 1 : 0;
	if (!found && !pending)
		st.t_next = n;
	if (lane == 0)
		st_all[f] = st;
}

// decode.cc:110-151 for a pending trigger, one workgroup per frame: the same arithmetic as sc_process (the transforms
// are the same Stockham stages, butterfly by butterfly - only shared among 256 threads, in LDS), bit-identical results.
template <int RATE>
__global__ __launch_bounds__(256) void k_sync_accept(FrameBatch fb, const cf *__restrict__ z_all, const cf *__restrict__ tw,
	const cf *__restrict__ kern, SyncState *__restrict__ st_all)
{
	typedef RateCfg<RATE> RC;
	constexpr int BUFFER_LEN = RC::BUFFER_LEN, SEARCH_POS = RC::SEARCH_POS, HALF_LEN = RC::HS, GUARD_LEN = RC::GL, NT = 256;
	const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	SyncState st = st_all[f];
	if (!st.active || st.found || !st.pending)
		return;
	const long n = fb.samples_per_frame;
	SampleSrc src{ (const char *)fb.samples + (size_t)f * fb.frame_stride_bytes, fb.fmt, fb.channels, n,
		fb.channels == 1 ? z_all + (size_t)f * fb.samples_per_frame : nullptr };
	__shared__ cf buf[HALF_LEN], xr[HALF_LEN];
	__shared__ cf rot[(HALF_LEN + NT - 1) / NT];
	__shared__ float red_p[4];
	__shared__ int red_i[4];
	const long g = st.pend_g;
	const float phase_max = st.pend_phase;
	const float frac_cfo = phase_max / (float)HALF_LEN;       // decode.cc:110
	int symbol_pos = SEARCH_POS - st.pend_index_max;           // decode.cc:114
	const long base = g - (BUFFER_LEN - 1);
	if (tid < (HALF_LEN + NT - 1) / NT)
		rot[tid] = phasor(frac_cfo, (long)NT * tid);
	const cf p_thread = phasor(frac_cfo, tid);
	__syncthreads();
	for (int i = tid; i < HALF_LEN; i += NT)                   // decode.cc:117-118
		buf[i] = cmul(src.at(base + i + symbol_pos + HALF_LEN), cmul(p_thread, rot[i / NT]));
	__syncthreads();
	fft_fwd<HALF_LEN, NT, RC::SL>(buf, tw, tid);
	for (int i = tid; i < HALF_LEN; i += NT)                   // decode.cc:120-121
		xr[i] = demod_or_erase(buf[i], buf[(i + HALF_LEN - 1) % HALF_LEN]);
	__syncthreads();
	fft_fwd<HALF_LEN, NT, RC::SL>(xr, tw, tid);
	// x kern, then backward transform as conj(FFT(conj(.)))
	for (int i = tid; i < HALF_LEN; i += NT)
		xr[i] = cconj(cmul(xr[i], kern[i]));
	__syncthreads();
	fft_fwd<HALF_LEN, NT, RC::SL>(xr, tw, tid);
	// decode.cc:127-139: peak = max, shift = first index of it, next = runner-up
	float pk = -1.f;
	int sh_i = 0x7fffffff;
	for (int i = tid; i < HALF_LEN; i += NT) {
		float p = cnorm(xr[i]);
		if (p > pk) { pk = p; sh_i = i; }
	}
	#pragma unroll
	for (int m = 32; m; m >>= 1) {
		float op = __shfl_xor(pk, m);
		int oi = __shfl_xor(sh_i, m);
		if (op > pk || (op == pk && oi < sh_i)) { pk = op; sh_i = oi; }
	}
	if (lane == 0) { red_p[wave] = pk; red_i[wave] = sh_i; }
	__syncthreads();
	pk = red_p[0]; sh_i = red_i[0];
	#pragma unroll
	for (int w = 1; w < 4; ++w)
		if (red_p[w] > pk || (red_p[w] == pk && red_i[w] < sh_i)) { pk = red_p[w]; sh_i = red_i[w]; }
	__syncthreads();
	float nx = 0.f;
	for (int i = tid; i < HALF_LEN; i += NT) {
		float p = cnorm(xr[i]);
		if (i != sh_i && p > nx) nx = p;
	}
	#pragma unroll
	for (int m = 32; m; m >>= 1)
		nx = fmaxf(nx, __shfl_xor(nx, m));
	if (lane == 0)
		red_p[wave] = nx;
	__syncthreads();
	nx = fmaxf(fmaxf(red_p[0], red_p[1]), fmaxf(red_p[2], red_p[3]));
	const float peak = fmaxf(pk, 0.f);
	const int shift = peak > 0.f ? sh_i : 0;
	bool accept = peak > nx * 4.f;                             // decode.cc:140-141
	int pos_err = 0;
	if (accept) {
		cf v = cconj(xr[shift]);
		pos_err = (int)nearbyintf(atan2f(v.im, v.re) * (float)HALF_LEN / TWO_PI_F);
		if (abs(pos_err) > GUARD_LEN / 2)                      // decode.cc:144-145
			accept = false;
	}
	if (tid == 0) {
		st.pending = 0;
		if (accept) {
			symbol_pos -= pos_err;
			float cfo_rad = (float)shift * (TWO_PI_F / (float)HALF_LEN) - frac_cfo;   // decode.cc:148
			if (cfo_rad >= PI_F)
				cfo_rad -= TWO_PI_F;
			st.found = 1;
			st.symbol_pos = symbol_pos;
			st.cfo_rad = cfo_rad;
			st.sc_start = g - (BUFFER_LEN - 1) + symbol_pos;
		} else {
			st.rejects += 1;
		}
		st_all[f] = st;
	}
}

}  // namespace rx

namespace rx {

__global__ void k_init_sync(int n, SyncState *st, const int32_t *skip, int *chunk_flags, int32_t *attempt_counts)
{
	int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f == 0 && chunk_flags)
		chunk_flags[0] = 0;                                       // "some frame has more rows than one Theil-Sen launch covers" (k_theilsen.hip)
	if (f >= n)
		return;
	SyncState s;
	s.t_next = 0;
	s.sc_start = -1;
	s.active = 1;
	s.found = 0;
	s.symbol_pos = 0;
	s.cfo_rad = 0.f;
	s.rejects = 0;
	s.skip_left = skip ? skip[f] : 0;
	s.status = 1;   // OFDMRX_NO_SYNC until a preamble is accepted
	s.oper_mode = 0;
	s.call_sign = 0;
	s.hdr_rounds = 0;
	s.okay = 0;
	s.pend_g = 0;
	s.pend_index_max = 0;
	s.pend_phase = 0.f;
	s.pending = 0;
	st[f] = s;
	if (attempt_counts)
		attempt_counts[f] = 0;
}

void launch_init_sync(hipStream_t s, int n, SyncState *st, const int32_t *skip_counts, int *chunk_flags, int32_t *attempt_counts)
{
	hipLaunchKernelGGL(k_init_sync, dim3((n + 255) / 256), dim3(256), 0, s, n, st, skip_counts, chunk_flags, attempt_counts);
}
size_t front_end_scratch_bytes(int rate, int n, long samples_per_frame)
{
	const size_t tiles = (size_t)((samples_per_frame + FE_TILE - 1) / FE_TILE);
	size_t rec = 0;
	RX_RATE_SWITCH(rate, rec = FeCfg<RATE>::REC);
	return (size_t)n * tiles * rec * sizeof(double);
}
void launch_front_end(hipStream_t s, int rate, int n, FrameBatch fb, FrontCoef co, double *scratch, cf *z)
{
	const int tiles = (int)((fb.samples_per_frame + FE_TILE - 1) / FE_TILE);
	for (int f0 = 0; f0 < n; f0 += 65535) {                       // gridDim.y
		const int nf = n - f0 < 65535 ? n - f0 : 65535;
		FrameBatch fbq = fb;
		fbq.samples = (const char *)fb.samples + (size_t)f0 * fb.frame_stride_bytes;
		double *sc = scratch + (size_t)f0 * (front_end_scratch_bytes(rate, 1, fb.samples_per_frame) / sizeof(double));
		cf *zq = z + (size_t)f0 * fb.samples_per_frame;
		RX_RATE_SWITCH(rate,
			hipLaunchKernelGGL(k_front_dc<RATE>, dim3(tiles, nf), dim3(256), 0, s, fbq, co, sc, tiles);
			hipLaunchKernelGGL(k_front_end<RATE>, dim3(tiles, nf), dim3(256), 0, s, fbq, co, sc, tiles, zq));
	}
}
#ifndef SYNC_SPLIT_ROUNDS
#define SYNC_SPLIT_ROUNDS 2   // rates above 8 kHz: scan + accept pairs before the one-wave catch-all (a frame needs the catch-all only
                              // after that many rejected triggers; finished frames leave every later launch at once)
#endif
void launch_sync(hipStream_t s, int rate, int n, FrameBatch fb, const cf *z, Tables tb, SyncState *st, cf *scratch)
{
#ifndef SYNC_SPLIT_8K
#define SYNC_SPLIT_8K 1       // 8 kHz too since round 3: the fused one-wave kernel needs 240 VGPRs and 20 KB of LDS (8 waves per CU,
                              // four rounds of 2048 frames per chunk); the scan alone runs at 16 per CU: 0.83 -> 0.73 ms per 8192 frames
#endif
	if (rate != 8000 || SYNC_SPLIT_8K) {
		for (int r = 0; r < SYNC_SPLIT_ROUNDS; ++r) {
			RX_RATE_SWITCH(rate,
				hipLaunchKernelGGL((k_sync<RATE, true>), dim3(n), dim3(64), 0, s, fb, z, tb.tw_sym, tb.sc_kern, st, scratch);
				hipLaunchKernelGGL(k_sync_accept<RATE>, dim3(n), dim3(256), 0, s, fb, z, tb.tw_sym, tb.sc_kern, st));
		}
	}
	RX_RATE_SWITCH(rate, hipLaunchKernelGGL((k_sync<RATE, false>), dim3(n), dim3(64), 0, s, fb, z, tb.tw_sym, tb.sc_kern, st, scratch));
}

}  // namespace rx
