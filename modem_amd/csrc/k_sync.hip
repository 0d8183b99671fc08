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
#include <type_traits>
#include "dev_common.h"
#include "kernels.h"
#include "mono_front.h"

// Decision-critical fp32 expressions (cfo_rad = shift*2pi/640 - frac_cfo sits at magnitude ~6 before the
// wrap, so ONE ulp there is 5e-7 rad/sample = 7e-4 rad per symbol): no FMA contraction in this file, like
// the strict-IEEE CPU restatement.
#pragma clang fp contract(off)

namespace rx {

// ---------------------------------------------------------------- D1 front end
// Decoder::next_sample (decode.cc:294-301) for mono input: y = BlockDC(x), z = Hilbert<cmplx, filter_len>(y).  Round 4: the analytic
// signal is formed by its consumers (mono_front.h); what runs over the whole stream is only this pass, which leaves the DC blocker's
// low-pass state s[n] = a s[n-1] + g x[n] after every 64th sample - 2 bytes read and 1/8 byte written per sample where the front end
// of rounds 1-3 wrote 8 (and k_sync / k_header / k_demod read them back).
//   one workgroup per frame, TILE of 4096 samples = 16 consecutive samples per thread, CAR_TB tiles in flight: every tile from a ZERO
//   state (thread recurrences, a weighted inclusive scan of the 256 thread-end states: v[t] += A^d v[t-d], A = a^16), then the
//   tiles' true entry states C_t = e_{t-1} + a^4096 C_{t-1} in turn and s = (zero-start value) + a^(position + 1) C_t.
// In double: a parallel scan cannot reproduce the rounding sequence of the serial fp32 recurrence anyway, so the kept states are
// made (nearly) exact and the difference to the CPU's serial fp32 filter is that filter's own rounding.
constexpr int FE_PER = 16, FE_TILE = 256 * FE_PER, CAR_TB = 6;

template <int RATE>
__global__ __launch_bounds__(256) void k_mono_carries(FrameBatch fb, FrontCoef co, double *__restrict__ ck_all, int ck_per_frame)
{
	const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const long n = fb.samples_per_frame;
	const char *base = (const char *)fb.samples + (size_t)f * fb.frame_stride_bytes;
	MonoFrame fr{ base, fb.fmt, n, nullptr };
	double *ck = ck_all + (size_t)f * ck_per_frame;
	__shared__ double wave_end[CAR_TB][4];
	const double a = (double)co.dc_a, g = (double)co.dc_b * (1.0 - a) * (double)fr.scale();   // (the samples come as the PCM's integers)
	double Apow[7];                                               // a^(16 2^k): Apow[6] = a^1024 decays a whole wave
	Apow[0] = mono_pow(a, FE_PER);
	#pragma unroll
	for (int k = 1; k < 7; ++k)
		Apow[k] = Apow[k - 1] * Apow[k - 1];
	const double Atile = Apow[6] * Apow[6] * Apow[6] * Apow[6];   // a^4096
	double Dlane = 1.0;                                           // a^(16 (lane + 1))
	#pragma unroll
	for (int k = 0; k < 6; ++k)
		if ((lane + 1) & (1 << k))
			Dlane *= Apow[k];
	if (lane == 63)
		Dlane = Apow[6];
	const double W16 = mono_pow(a, FE_PER * ((lane & 15) + 1)), W32 = mono_pow(a, FE_PER * ((lane & 31) + 1));
	double Dthread = Dlane;                                       // a^(16 (tid + 1))
	for (int w = 0; w < wave; ++w)
		Dthread *= Apow[6];
	const int tiles = (int)((n + FE_TILE - 1) / FE_TILE);
	double C = 0.0;                                               // the state entering tile t0
	for (int t0 = 0; t0 < tiles; t0 += CAR_TB) {
		double v[CAR_TB];
		#pragma unroll
		for (int q = 0; q < CAR_TB; ++q) {
			const long s0 = (long)(t0 + q) * FE_TILE + (long)tid * FE_PER;
			float x[FE_PER];
			{
				float x8[8];
				fr.load8(s0, x8);
				#pragma unroll
				for (int i = 0; i < 8; ++i) x[i] = x8[i];
				fr.load8(s0 + 8, x8);
				#pragma unroll
				for (int i = 0; i < 8; ++i) x[8 + i] = x8[i];
			}
			double sl = 0.0;
			#pragma unroll
			for (int i = 0; i < FE_PER; ++i)
				sl = a * sl + g * (double)x[i];
			sl = WScan<double>::run(sl, Apow, W16, W32);          // weighted inclusive scan of the thread ends over the wave
			if (lane == 63)
				wave_end[q][wave] = sl;
			v[q] = sl;
		}
		__syncthreads();
		#pragma unroll
		for (int q = 0; q < CAR_TB; ++q) {
			double st = 0.0;                                      // zero-start state at the end of the earlier waves of this tile
			for (int w = 0; w < wave; ++w)
				st = wave_end[q][w] + Apow[6] * st;
			const double tile_end = ((wave_end[q][3] + Apow[6] * wave_end[q][2]) + Apow[6] * Apow[6] * wave_end[q][1])
				+ Apow[6] * Apow[6] * Apow[6] * wave_end[q][0];
			const double s_true = (v[q] + Dlane * st) + Dthread * C;
			const long m = (long)(t0 + q) * (FE_TILE / MONO_CK) + (tid >> 2);   // this thread ends sample 16 tid + 15 of its tile
			if ((tid & 3) == 3 && m < ck_per_frame)
				ck[m] = s_true;
			C = tile_end + Atile * C;
		}
		__syncthreads();
	}
}

// the whole analytic signal of the frames (rates above 8 kHz; the ANALYTIC tap): a workgroup per stretch of FE_STRETCH samples, which
// starts on the kept state before it - four spans of 2048 samples cover a stretch and the filter's reach + the 63 samples back to
// that state
constexpr int FE_STRETCH = 4 * 2048 - 256;
template <int RATE>
__global__ __launch_bounds__(256, RATE == 8000 ? 4 : 5) void k_front_end(FrameBatch fb, MonoArgs ma, cf *__restrict__ z_all)
{
	static_assert(MonoCfg<RATE>::REACH + MONO_CK <= 256, "a stretch and its lead-in fit four spans");
	const int f = blockIdx.y, tid = threadIdx.x;
	const long lo = (long)blockIdx.x * FE_STRETCH;
	__shared__ typename MonoCover<RATE, 256>::Shared msh;
	MonoCover<RATE, 256> mc;
	mc.init(mono_frame(fb, ma.ck, ma.ck_per_frame, f), ma, &msh, z_all + (size_t)f * fb.samples_per_frame, tid);
	mc.hi = lo + FE_STRETCH < mc.fr.n ? lo + FE_STRETCH : mc.fr.n;   // (cover() runs whole spans: they reach into the next workgroup's stretch)
	mc.cover(ma, lo, lo + FE_STRETCH, tid);
}

// ---------------------------------------------------------------- D2 + D3 sync
#define SYNC_PER 8     // sample times per lane and tile: 8 -> 224 VGPRs and 22 KB of LDS, so that a sync wave fits on a SIMD beside
                       // three resident polar decoders (the whole front runs inside the polar phase); 16 -> 256 VGPRs, 30 KB
constexpr int PER = SYNC_PER, TILE = 64 * PER;
// ring of the last TILE + match_len metric values (a power of two): 1024 at 8 / 16 kHz, 2048 at 44.1 / 48 kHz
template <int RATE> struct SyncRing { static constexpr int N = TILE + RateCfg<RATE>::MATCH_LEN <= 1024 ? 1024 : 2048; };

// 8 kHz: the two 640-point work arrays of the trigger part live in LDS.  Other rates (1280 / 3528 / 3840
// points) keep them in a per-frame global scratch so the scanning loop's occupancy does not pay for them.
#define SYNC_FFT_IN_LDS 1    // 8 kHz: the two 640-point buffers of the S&C trigger part in LDS (0: global scratch like the other rates)
#define SYNC_FFT_LDS(R) ((R) == 8000 && SYNC_FFT_IN_LDS)
#define SYNC_WAVES 2
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
// MONO: z_all is written here - the analytic signal of the samples the scan walks, span by span ahead of it (mono_front.h)
struct NoShared {};
#define SYNC_WAVES_SPLIT_MONO 3 // the same for mono input (the scan forms the analytic signal too): 168 VGPRs
template <int RATE, bool SPLIT, bool MONO>
__global__ __launch_bounds__(64, SPLIT ? (MONO ? SYNC_WAVES_SPLIT_MONO : SYNC_WAVES_SPLIT) : SYNC_WAVES) void k_sync(FrameBatch fb, cf *__restrict__ z_all, const cf *__restrict__ tw,
	const cf *__restrict__ kern, SyncState *__restrict__ st_all, cf *__restrict__ scratch, MonoArgs ma)
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
	__shared__ std::conditional_t<MONO, typename MonoCover<RATE, 64>::Shared, NoShared> msh;
	MonoCover<RATE, 64> mc;
	constexpr int D = BUFFER_LEN - 1 - (SEARCH_POS + HALF_LEN);   // P at time t: its newest pair is (t - D, t - D + HALF_LEN)
	if constexpr (MONO)
		mc.init(mono_frame(fb, ma.ck, ma.ck_per_frame, f), ma, &msh, z_all + (size_t)f * fb.samples_per_frame, lane);
	auto need = [&](long p_lo, long p_hi) {                       // MONO: the analytic signal of [p_lo, p_hi) before it is read
		if constexpr (MONO) {
			mc.cover(ma, p_lo, p_hi, lane);
			__syncthreads();
		}
	};
	cf *fbuf = SYNC_FFT_LDS(RATE) && !SPLIT ? sh.buf : scratch + (size_t)f * 2 * HALF_LEN;
	cf *fxr = SYNC_FFT_LDS(RATE) && !SPLIT ? sh.xr : scratch + (size_t)f * 2 * HALF_LEN + HALF_LEN;
	for (int i = lane; i < MRING + MRING / 8; i += 64)
		sh.m[i] = 0.f;
	const float thr_lo = (float)(0.17 * MATCH_LEN), thr_hi = (float)(0.19 * MATCH_LEN);   // decode.cc:76
	const float min_R = 0.0001f * HALF_LEN;                                                // decode.cc:88
	const long t_start = st.t_next;
	long T0 = t_start - (MATCH_LEN - 1);
	// The correlator looks at the samples up to t - FIRST at time t (decode.cc:86: the delay lines in front of it are BUFFER_LEN long):
	// before t = FIRST it sees none of the stream, P = R = 0, the metric and its moving sum are exactly 0 and nothing can trigger.
	// A scan from the start of a stream - every frame of a batch, first round - therefore starts there instead of walking eight
	// tiles of zeros (round 4: the scan was 0.48 ms per 8192 frames, 60 % of it those tiles).
	constexpr long FIRST = D - HALF_LEN;
	if (T0 < FIRST - (MATCH_LEN - 1)) T0 = FIRST - (MATCH_LEN - 1);
	// running window sums at time T0-1
	double Wr = 0.0, Wi = 0.0, Wp = 0.0, Wm = 0.0;
	if (T0 > FIRST) {
		need(T0 - 1 - D - (HALF_LEN - 1), T0 - D + HALF_LEN);
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
			if constexpr (MONO) {
				// this tile's window was formed a tile ago (the first tile's: here); the NEXT tile's span is formed now, so that its
				// stores are long done when they are read - the fence at the top of the next round then costs nothing
				if (!mc.any || mc.next < w_hi)
					need(w_lo, w_hi);
				else
					__syncthreads();
				mc.cover(ma, w_lo, w_hi + TILE, lane);
			}
			if (src.mode() == 1 && w_lo >= 0 && w_hi <= n) {
				const short2 *p = (const short2 *)src.base;
				phase1([&](long i) { const short2 v = p[i]; return mk(div_32767((float)v.x), div_32767((float)v.y)); });
			} else if (MONO && w_lo >= 0 && w_hi <= n) {
				const cf *p = src.analytic;
				phase1([&](long i) { return p[i]; });
			} else if (src.mode() == 1 && w_hi <= n) {
				// the head of the stream (the three tiles behind FIRST): zeros in front of sample 0, no branches per sample
				const short2 *p = (const short2 *)src.base;
				phase1([&](long i) {
					const short2 v = p[i < 0 ? 0 : i];
					return i < 0 ? mk(0.f, 0.f) : mk(div_32767((float)v.x), div_32767((float)v.y));
				});
			} else if (MONO && w_hi <= n) {
				const cf *p = src.analytic;
				phase1([&](long i) { const cf v = p[i < 0 ? 0 : i]; return i < 0 ? mk(0.f, 0.f) : v; });
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
					need(tp - D - (HALF_LEN - 1), tp - D + HALF_LEN + 1);
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
				need((long)g - (BUFFER_LEN - 1) + (SEARCH_POS - index_max) + HALF_LEN, (long)g - (BUFFER_LEN - 1) + (SEARCH_POS - index_max) + 2 * HALF_LEN);
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
template <int RATE, bool MONO>
__global__ __launch_bounds__(256) void k_sync_accept(FrameBatch fb, cf *__restrict__ z_all, const cf *__restrict__ tw,
	const cf *__restrict__ kern, SyncState *__restrict__ st_all, MonoArgs ma)
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
	if constexpr (MONO) {                                      // the window's analytic signal (mono_front.h)
		__shared__ typename MonoCover<RATE, 256>::Shared msh;
		MonoCover<RATE, 256> mc;
		mc.init(mono_frame(fb, ma.ck, ma.ck_per_frame, f), ma, &msh, z_all + (size_t)f * fb.samples_per_frame, tid);
		mc.cover(ma, base + symbol_pos + HALF_LEN, base + symbol_pos + 2 * HALF_LEN, tid);
	}
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
void launch_mono_carries(hipStream_t s, int rate, int n, FrameBatch fb, FrontCoef co, double *ck)
{
	RX_RATE_SWITCH(rate, hipLaunchKernelGGL(k_mono_carries<RATE>, dim3(n), dim3(256), 0, s, fb, co, ck, mono_ck_per_frame(fb.samples_per_frame)));
}
void launch_front_end(hipStream_t s, int rate, int n, FrameBatch fb, MonoArgs ma, cf *z)
{
	const int stretches = (int)((fb.samples_per_frame + FE_STRETCH - 1) / FE_STRETCH);
	for (int f0 = 0; f0 < n; f0 += 65535) {                       // gridDim.y
		const int nf = n - f0 < 65535 ? n - f0 : 65535;
		FrameBatch fbq = fb;
		fbq.samples = (const char *)fb.samples + (size_t)f0 * fb.frame_stride_bytes;
		MonoArgs maq = ma;
		maq.ck = ma.ck + (size_t)f0 * ma.ck_per_frame;
		RX_RATE_SWITCH(rate, hipLaunchKernelGGL(k_front_end<RATE>, dim3(stretches, nf), dim3(256), 0, s, fbq, maq, z + (size_t)f0 * fb.samples_per_frame));
	}
}
#define SYNC_SPLIT_ROUNDS 2   // rates above 8 kHz: scan + accept pairs before the one-wave catch-all (a frame needs the catch-all only
                              // after that many rejected triggers; finished frames leave every later launch at once)
template <int RATE, bool MONO>
static void sync_rounds(hipStream_t s, int n, FrameBatch fb, cf *z, Tables tb, SyncState *st, cf *scratch, const MonoArgs &ma)
{
#define SYNC_SPLIT_8K 1       // 8 kHz too since round 3: the fused one-wave kernel needs 240 VGPRs and 20 KB of LDS (8 waves per CU,
                              // four rounds of 2048 frames per chunk); the scan alone runs at 16 per CU: 0.83 -> 0.73 ms per 8192 frames
	if (RATE != 8000 || SYNC_SPLIT_8K) {
		for (int r = 0; r < SYNC_SPLIT_ROUNDS; ++r) {
			hipLaunchKernelGGL((k_sync<RATE, true, MONO>), dim3(n), dim3(64), 0, s, fb, z, tb.tw_sym, tb.sc_kern, st, scratch, ma);
			hipLaunchKernelGGL((k_sync_accept<RATE, MONO>), dim3(n), dim3(256), 0, s, fb, z, tb.tw_sym, tb.sc_kern, st, ma);
		}
	}
	hipLaunchKernelGGL((k_sync<RATE, false, MONO>), dim3(n), dim3(64), 0, s, fb, z, tb.tw_sym, tb.sc_kern, st, scratch, ma);
}
void launch_sync(hipStream_t s, int rate, int n, FrameBatch fb, cf *z, Tables tb, SyncState *st, cf *scratch, const MonoArgs &ma)
{
	if (fb.channels == 1 && mono_fused(rate)) {
		sync_rounds<8000, true>(s, n, fb, z, tb, st, scratch, ma);
	} else {
		RX_RATE_SWITCH(rate, (sync_rounds<RATE, false>(s, n, fb, z, tb, st, scratch, ma)));
	}
}

}  // namespace rx
