// k_demod.hip -- D4 (51 x FFT1280 + time-differential demod) for gfx950; D5 (Theil-Sen) lives in k_theilsen.hip, D6-D8 in
// k_finish.hip (k_back).  Compiled with -fno-slp-vectorize (Makefile): the
// SLP vectoriser packs the butterflies into v_pk_*_f32 (6.4 cycles per issue against 3.3, measured) and pays a v_mov per
// pack - the 8 kHz demodulator is issue-bound and runs 1.75 -> 1.4 ms per 8192 frames without it.
#include "dev_common.h"
#include "kernels.h"
#include "mono_front.h"

namespace rx {

// ---------------------------------------------------------------- D4
// decode.cc:453-477.  One workgroup per frame walks the pilot and the data symbols in order; samples are read once, straight from
// the raw PCM.  symbol_len = R1 x NS (1280 = 5 x 256, 2560 = 5 x 512, 7056 = 7 x 1008, 7680 = 5 x 1536): radix-R1 decimation in
// frequency in registers, then R1 row transforms of NS points - see the kernel.
#define DEMOD_CONS_OUT(R) ((R) != 44100)   // the demodulator forms cons = X_j / X_{j-1} itself, from the carriers of the previous symbol parked in
                            // LDS (5 KB at 8 kHz): k_theil_sen is VALU-bound and 0.42 ms per chunk shorter without the row formation, k_demod is
                            // LDS-bound and 0.16 ms longer with it.  44.1 kHz measured 3 % slower with it and keeps the other form: the
                            // carriers go to HBM and k_theil_sen forms the rows where it reads them
template <int RATE> struct DifCfg {
	static constexpr int SL = RateCfg<RATE>::SL;
#define DEMOD_DIF_W(R) ((R) == 48000 ? 3 : (R) == 44100 ? 2 : 1)
	static constexpr int W = DEMOD_DIF_W(RATE);              // waves that share one row's transform (1: wave-private, wave barriers only)
	static constexpr int R1 = SL % 5 == 0 ? 5 : 7, NS = SL / R1, NT = 64 * W * R1;
	static constexpr int NQ = (NS + NT - 1) / NT;             // points n' per loader thread
#define DEMOD_TWR_BYTES 8192
#ifndef DEMOD_WAVES_8K
#define DEMOD_WAVES_8K 7
#endif
#define DEMOD_DIF_WAVES(R) ((R) >= 44100 ? 3 : (R) == 8000 ? DEMOD_WAVES_8K : 5)   // 8 kHz: 72 VGPRs = five workgroups per CU (26.6 KB of LDS each); 64 spills with the swizzle
#ifndef DEMOD_TWR_8K
#define DEMOD_TWR_8K 1    // 8 kHz: 1 = the w^(n' r) of the radix-5 step in LDS (8 KB: five workgroups per CU); 0 = read from the global root table through L1:
                          // 1.21 ms per 8192 frames against 1.16, and 1.36 with the register budget of eight waves per SIMD that lets a sixth workgroup in (round 6)
#endif
	static constexpr bool TWR_LDS = ((R1 - 1) * NS * 8 <= DEMOD_TWR_BYTES && (RATE != 8000 || DEMOD_TWR_8K)) || SL == 7056;   // w^(n' r) table in LDS (8 / 16 kHz; 44.1 kHz: 47 KB - its one
	                                                          // workgroup per CU has the room) or read from the global root table (48 kHz: registers)
	static constexpr int WAVES = DEMOD_DIF_WAVES(RATE);      // waves per SIMD the register budget is set for (two workgroups per CU at 44.1 / 48 kHz)
};

#define DEMOD_PREFETCH_NQ(R) ((R) >= 44100 ? 2 : 1)   // loader points per thread up to which the NEXT symbol's samples are fetched during the
                        // transform.  44.1 / 48 kHz run ONE workgroup of 14 / 15 waves per CU (registers and LDS), so nothing else hides the load
                        // at the top of a symbol: with the next symbol's two points per loader in flight meanwhile 13.5 -> 11.3 and 12.0 -> 11.2 ms
                        // per 8192 frames; 16 kHz (two points too, four workgroups per CU) 3.44 -> 3.51: stays without
#define DEMOD_SWZ 1     // 8 kHz (the only rate with wave-private 256-point transforms): demod 1.48 -> 1.35 ms per chunk with the seven-waves budget below
// (swz256 / fft256_stage_swz: dev_common.h, shared with the transmitter's transforms)
#ifndef DEMOD_REGS
#define DEMOD_REGS 1    // 8 kHz: the wave's 256-point transform in registers (fft256_regs, dev_common.h): the rows are read from LDS once and the
                        // spectrum written once, the three exchanges between the radix-4 steps are lane moves (32 LDS accesses of 75 per wave and symbol gone)
#endif

// Mono input (mono_front.h).  MONO = 2 (8 kHz): the workgroup walks the frame's symbols in order, so it carries the DC blocker's
// state along: a symbol's span is its guard interval and body, [t0 - guard_len, t0 + symbol_len) - 1440 samples, five per thread
// on 288 threads - and the spans of successive symbols join without a gap.  The recurrence runs a symbol ahead in two parts that
// hang on the transform's own barriers (thread chunks + wave scan before the barrier that follows the transform, the waves' entry
// states + y[n] = b x[n] - s[n-1] into LDS after it), the 21-tap Hilbert filter forms a loader thread's five samples n' + 256 a
// out of LDS where the analytic path converts five int16 pairs.  Nothing of the analytic signal is ever in memory.
// Mono input at the other rates arrives as the analytic signal z a front pass wrote (k_front_end), MONO = 0 like 2-channel input.
#ifndef DEMOD_MONO_WAVES
#define DEMOD_MONO_WAVES 5    // register budget of the MONO = 2 instantiation (waves per SIMD): 5 = 96 VGPRs, four workgroups per CU (LDS + registers).
                              // Measured per 8192 frames: 7 (72 VGPRs, five workgroups, spills) 2.46 - 2.67 ms, 6 (80) 2.04 - 2.10, 5 (96) 2.02, 4 (128, three
                              // workgroups) 3.05
#endif
template <int RATE, int MONO>
__global__ __launch_bounds__(DifCfg<RATE>::NT, MONO == 2 ? DEMOD_MONO_WAVES : DifCfg<RATE>::WAVES) void k_demod(FrameBatch fb, cf *__restrict__ z_all, MonoArgs ma, Tables tb,
	const SyncState *__restrict__ st_all, cf *__restrict__ cons_all, cf *__restrict__ carr_all)
{
	constexpr int SYMBOL_LEN = RateCfg<RATE>::SL, SYM_STRIDE = RateCfg<RATE>::STRIDE, GUARD_LEN = RateCfg<RATE>::GL;
	const int f = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const SyncState st = st_all[f];
	if (!st.okay)
		return;
	SampleSrc src{ (const char *)fb.samples + (size_t)f * fb.frame_stride_bytes, fb.fmt, fb.channels, fb.samples_per_frame,
		fb.channels == 1 ? z_all + (size_t)f * fb.samples_per_frame : nullptr };
	const ModeDesc md = mode_desc(st.oper_mode);
	cf *cons = cons_all + (size_t)f * CONS_MAX;
	const long body0 = st.sc_start + 2 * SYM_STRIDE;          // pilot body, decode.cc:456-459
	const float omega = -st.cfo_rad;                          // decode.cc:403
	const int code_off = -md.cols / 2;                        // decode.cc:454
	{
		// symbol_len = R1 x NS.  X[R1 q + r] = sum_n' wNS^(n' q) [ w^(n' r) sum_a x[n' + NS a] wR1^(a r) ], w = e^{-j 2 pi / symbol_len}:
		// a loader thread takes the R1 samples n' + NS a (stride NS: the raw PCM is read coalesced), runs the radix-R1 butterfly in
		// registers, applies the R1 - 1 twiddles and parks output r in row r of an LDS buffer; wave r (R1 waves) then transforms
		// row r - NS points, NS / (64 R) butterflies per lane and stage, wave barriers only, twiddles from a compact LDS table.
		// Two workgroup barriers per symbol (after the rows are written, before the carriers are read) against two per radix
		// stage of a cooperative transform; a third barrier protects the single row buffer.
		typedef DifCfg<RATE> DC;
		constexpr int NT = DC::NT, R1 = DC::R1, NS = DC::NS, NQ = DC::NQ;
		static_assert(R1 * NS == SYMBOL_LEN, "plan");
		constexpr int TWC = fft_compact_size<NS, SYMBOL_LEN>();
		__shared__ cf row[R1 * NS];
		constexpr bool REGS = DEMOD_REGS && DEMOD_SWZ && NS == 256 && DC::W == 1;
		__shared__ cf tw_sub[REGS ? 1 : TWC];                 // compact twiddles of the NS-point plan
		__shared__ cf twl[REGS ? 9 * 64 : 1];                 // ... or the per-lane twiddles of the in-register transform
		__shared__ cf tw_r[DC::TWR_LDS ? (R1 - 1) * NS : 1];  // w^(n' r), r = 1..R1-1
		__shared__ cf rotA[R1], rotQ[NQ], symrot[ROWS_MAX + 1];
		// a thread's (at most two) carriers of the previous symbol.  A frame has at most COLS_MAX = 512 carriers: with 512 threads or
		// more only e = 0 and tid < 512 ever come here (48 kHz: 4 KB instead of 15 - what keeps a second workgroup off the CU)
		constexpr int PC_E = NT >= COLS_MAX ? 1 : 2, PC_T = NT >= COLS_MAX ? COLS_MAX : NT;
		__shared__ cf prevc[PC_E][DEMOD_CONS_OUT(RATE) ? PC_T : 1];
		// MONO = 2: y of the span from (about) 32 samples before the body on (the filter reaches 19 back), and the waves' end states
		constexpr int MPER = 5, MTH = SYM_STRIDE / MPER, YOFF = (GUARD_LEN - 32) / MPER * MPER, YB = GUARD_LEN - YOFF;   // body sample m at ybuf[YB + m]
		static_assert(MONO != 2 || (MTH * MPER == SYM_STRIDE && MTH <= NT && NS == 256 && NQ == 1 && MonoCfg<RATE>::REACH <= 32 && GUARD_LEN % MPER == 0
			&& (MTH + 63) / 64 == 5 && MTH % 64 == 32), "span layout");
		__shared__ float ybuf[MONO == 2 ? SYM_STRIDE - YOFF : 1];
		__shared__ float mwe[5];
		if constexpr (REGS)
			fft256_lane_twiddles<NT>(twl, tb.tw_sym, tid);
		else
			fft_compact_twiddles<NS, NT, SYMBOL_LEN>(tw_sub, tb.tw_sym, tid);
		if (DC::TWR_LDS)
			for (int i = tid; i < (R1 - 1) * NS; i += NT)
				tw_r[i] = tb.tw_sym[(i / NS + 1) * (i % NS)];
		// NCO e^{j omega (symbol_len + s stride + n' + NS a)}, n' = tid + NT q: (per thread, once per frame) x (per q) x (per a) and, per
		// symbol, symrot[s]: the transform is linear, so that factor multiplies the <= 512 carriers on their way out instead of
		// the symbol_len samples
		if (tid < R1)
			rotA[tid] = phasor(omega, (long)NS * tid);
		if (tid >= 64 && tid < 64 + NQ)
			rotQ[tid - 64] = phasor(omega, (long)NT * (tid - 64));
		if (tid <= md.rows)
			symrot[tid] = phasor(omega, (long)tid * SYM_STRIDE);
		const cf p0 = phasor(omega, (long)SYMBOL_LEN + tid);
		__syncthreads();
#define DEMOD_QA_LDS 0        // 1: every instantiation forms the R1 - 1 NCO phasors per symbol from LDS (the mono one, short of registers, always does)
		constexpr bool QA_LDS = MONO == 2 || DEMOD_QA_LDS;
		cf qa[NQ == 1 ? R1 : 1];                             // one point per loader: the R1 phasors stay in registers
		if (NQ == 1) {
			#pragma unroll
			for (int a = 0; a < R1; ++a)
				qa[a] = (a && !QA_LDS) ? cmul(p0, rotA[a]) : p0;
		}
		// the (at most two) carriers of this thread sit at the same place of the rows in every symbol
		int coff[2];
		#pragma unroll
		for (int e = 0; e < 2; ++e) {
			const int i = tid + NT * e, k = (i + code_off + SYMBOL_LEN) % SYMBOL_LEN;
			coff[e] = i < md.cols ? (k % R1) * NS + (REGS ? fft256_pos(k / R1) : (DEMOD_SWZ && NS == 256 && DC::W == 1) ? swz256(k / R1) : k / R1) : -1;
		}
		// ---- MONO = 2: the span recurrence (see the head of the kernel)
		const MonoFrame mfr = mono_frame(fb, ma.ck, ma.ck_per_frame, f);
		const long span0 = MONO == 2 ? uniform_l(body0 - GUARD_LEN) : 0;   // first sample of symbol 0's span
		float mS = 0.f, mAspan = 1.f, mAw = 1.f, mAwave = 1.f;        // s before the span that is next to run; a^1440; a^(320 wave); a^320
		const float mg = ma.g * mfr.scale(), mb = ma.b * mfr.scale();   // (the samples are taken as the PCM's integers, mono_front.h)
		float mAlane = 1.f, mW16 = 1.f, mW32 = 1.f, my0[MPER] = {};
		// the span's samples, fetched a symbol ahead of their use and left as they come (the PCM's integers; f32 input: the bits) - a
		// conversion here would wait for the loads where they are issued
		int mxi[MPER] = {};
		bool mpacked = false;                                         // int16 inside the frame: the five samples as 10 bytes in mxi[0..2]
		const bool mfloat = mfr.fmt == 2;
		auto mono_raw = [&](int sym) {
			const long e0 = span0 + (long)sym * SYM_STRIDE;           // (uniform; a frame has fewer than 2^31 samples)
			const int n32 = (int)mfr.n;
			if (tid < MTH) {
				mpacked = mfr.fmt == 0 && e0 >= 0 && e0 + SYM_STRIDE <= mfr.n;   // the span inside the frame, int16: the rule
				if (mpacked) {
					typedef uint32_t __attribute__((aligned(2))) word_at_2;   // (10 tid bytes into the span: two-byte aligned)
					const int16_t *q = (const int16_t *)mfr.base + e0 + tid * MPER;
					mxi[0] = (int)((const word_at_2 *)q)[0];
					mxi[1] = (int)((const word_at_2 *)q)[1];
					mxi[2] = (int)((const uint16_t *)q)[4];
				} else {                                              // any format, zeros outside the frame; straight-line code
					#pragma unroll
					for (int i = 0; i < MPER; ++i) {
						const int j = (int)e0 + tid * MPER + i, jc = min(max(j, 0), n32 - 1);
						int v;
						if (mfr.fmt == 0) v = ((const int16_t *)mfr.base)[jc];
						else if (mfr.fmt == 1) v = (int)((const uint8_t *)mfr.base)[jc] - 128;
						else v = ((const int *)mfr.base)[jc];
						mxi[i] = j == jc ? v : 0;
					}
				}
			}
		};
		// y[n] = b x[n] - s[n-1] and s[n-1] = (the thread's chunk from a zero state) + a^i (the state entering the thread), which is the
		// scan's value of the lane before + a^(5 lane) (the state entering the wave): everything but the last term is known before
		// the barrier - my0 - and the last term is a per-thread power of a times a per-wave number that needs the other waves' ends
		auto mono_part1 = [&]() {
			float acc = 0.f, sl[MPER], mx[MPER];
			if (mpacked) {
				mx[0] = (float)(short)(mxi[0] & 0xffff); mx[1] = (float)(mxi[0] >> 16);
				mx[2] = (float)(short)(mxi[1] & 0xffff); mx[3] = (float)(mxi[1] >> 16);
				mx[4] = (float)(short)mxi[2];
			} else {
				#pragma unroll
				for (int i = 0; i < MPER; ++i)
					mx[i] = mfloat ? __int_as_float(mxi[i]) : (float)mxi[i];
			}
			#pragma unroll
			for (int i = 0; i < MPER; ++i) {
				sl[i] = acc;                                          // the chunk's state BEFORE sample i
				acc = fmaf(ma.a, acc, mg * mx[i]);
			}
			const float v = WScan<float>::run(acc, ma.astep5, mW16, mW32);   // weighted inclusive scan of the chunk ends over the wave
			if (lane == (wave < 4 ? 63 : 31))                         // the span ends with thread 287 = lane 31 of wave 4
				mwe[wave] = v;
			const float cw0 = dpp_f<0x138>(v);                        // the lane before (lane 0: 0)
			my0[0] = fmaf(mb, mx[0], -cw0);
			#pragma unroll
			for (int i = 1; i < MPER; ++i)
				my0[i] = fmaf(mb, mx[i], -fmaf(ma.apw[i - 1], cw0, sl[i]));
		};
		auto mono_part2 = [&]() {                                     // (behind a barrier) the waves' entry states, y into LDS
			float st = 0.f, stw = 0.f;                                // the span from a zero state: at the start of wave w / of this wave
			#pragma unroll
			for (int w = 0; w < 4; ++w) {
				st = fmaf(mAwave, st, mwe[w]);
				if (w < wave)
					stw = st;
			}
			const float cl = mAlane * uniform_f(fmaf(mAw, mS, stw));  // a^(5 lane) x the state entering the wave
			mS = uniform_f(fmaf(mAspan, mS, fmaf(ma.astep5[5], st, mwe[4])));   // a^160 st_4: wave 4 ends after 32 chunks
			if (tid >= YOFF / MPER && tid < MTH) {
				float *yo = ybuf + (tid * MPER - YOFF);
				yo[0] = my0[0] - cl;
				#pragma unroll
				for (int i = 1; i < MPER; ++i)
					yo[i] = fmaf(-ma.apw[i - 1], cl, my0[i]);
			}
		};
		if constexpr (MONO == 2) {
			// s before the first span: the kept state at or before it, then the (at most 63) samples between, one per lane
			mAlane = (float)mono_pow((double)ma.a, MPER * lane);
			mW16 = (float)mono_pow((double)ma.a, MPER * ((lane & 15) + 1));
			mW32 = (float)mono_pow((double)ma.a, MPER * ((lane & 31) + 1));
			mAspan = uniform_f((float)mono_pow((double)ma.a, SYM_STRIDE));
			mAw = uniform_f((float)mono_pow((double)ma.a, 64 * MPER * wave));
			mAwave = (float)ma.awave5;
			if (span0 > 0) {
				const long c0 = span0 / MONO_CK * MONO_CK;
				const int gap = (int)(span0 - c0);
				const double term = lane < gap ? (double)mg * (double)mfr.raw(c0 + lane) * mono_pow((double)ma.a, gap - 1 - lane) : 0.0;
				mS = uniform_f((float)(mono_pow((double)ma.a, gap) * mfr.state_before(c0) + wave_sum_d(term)));
			}
			mono_raw(0);
			mono_part1();
			__syncthreads();
			mono_part2();
			mono_raw(1);
			__syncthreads();
		}
		// 48 kHz: the w^(n' r) of a loader's two points, the same in every symbol, stay in registers (they came from the global root
		// table in every symbol: eight loads per thread at the top of the symbol, in front of the row writes)
		constexpr bool TWR_REG = !DC::TWR_LDS && NQ * (R1 - 1) <= 8 && RATE != 8000;   // (8 kHz has no registers to spare: 64 = eight waves per SIMD)
		cf twq[TWR_REG ? NQ : 1][TWR_REG ? R1 - 1 : 1];
		if constexpr (TWR_REG) {
			#pragma unroll
			for (int q = 0; q < NQ; ++q)
				#pragma unroll
				for (int r = 1; r < R1; ++r)
					twq[q][r - 1] = tid + NT * q < NS ? tb.tw_sym[r * (tid + NT * q)] : mk(0.f, 0.f);
		}
		// ... and so do the NCO phasors of its two points (two complex multiplications per sample and symbol otherwise).  48 kHz only: its
		// one workgroup per CU has the registers (102 -> 117 of 128; 11.2 -> 10.7 ms per 8192 frames with both); at 16 kHz the same
		// twenty registers cost the fourth workgroup per CU (3.45 -> 4.08 ms)
		constexpr bool QA_REG2 = NQ == 2 && R1 == 5 && RATE == 48000;
		cf qa2[QA_REG2 ? NQ : 1][QA_REG2 ? R1 : 1];
		if constexpr (QA_REG2) {
			#pragma unroll
			for (int q = 0; q < NQ; ++q) {
				const cf pq = q ? cmul(p0, rotQ[q]) : p0;
				#pragma unroll
				for (int a = 0; a < R1; ++a)
					qa2[q][a] = a ? cmul(pq, rotA[a]) : pq;
			}
		}
		auto symbols = [&](auto M) {
		constexpr int MODE = decltype(M)::value;
		cf pre[NQ][R1];
#define DEMOD_RAW_AHEAD 1     // int16 pairs fetched a symbol ahead stay as they come - one register per point, no conversion (and no wait for
                              // the load) where the load is issued - and are converted where the symbol is taken up
		constexpr bool RAW = DEMOD_RAW_AHEAD && MODE == 1 && MONO == 0 && NQ <= DEMOD_PREFETCH_NQ(RATE);
		int praw[RAW ? NQ : 1][RAW ? R1 : 1];
		auto fetch = [&](int sym) {
			const long t0 = body0 + (long)sym * SYM_STRIDE;       // wave-uniform
			if constexpr (MONO == 2) {
				// Hilbert<cmplx, 21> of the loader's five samples out of LDS; sample t0 + m sits at ybuf[YB + m] (outside the frame: 0,
				// like SampleSrc::at)
				const bool inside = t0 >= 0 && t0 + SYMBOL_LEN <= src.n;
				if (tid < NS) {
					#pragma unroll
					for (int a = 0; a < R1; ++a) {
						const int m = tid + NS * a;
						cf zv = mono_hilbert<RATE>(ma.co, [&](int k) { return ybuf[YB + m - MonoCfg<RATE>::REACH + k]; });
						if (!inside && (t0 + m < 0 || t0 + m >= src.n))
							zv = mk(0.f, 0.f);
						pre[0][a] = zv;
						__builtin_amdgcn_sched_barrier(0);            // one sample's eleven LDS reads in flight, not five samples' (registers)
					}
				}
				return;
			}
			if (MODE == 1 && sym <= md.rows && t0 >= 0 && t0 + SYMBOL_LEN <= src.n) {
				// the whole symbol lies inside the frame (the rule): int16 pairs from a uniform base, no per-sample checks
				const short2 *p = (const short2 *)src.base + t0;
				#pragma unroll
				for (int q = 0; q < NQ; ++q) {
					const int np = tid + NT * q;
					if (np < NS) {
						#pragma unroll
						for (int a = 0; a < R1; ++a) {
							if constexpr (RAW)
								praw[q][a] = ((const int *)p)[np + NS * a];
							else {
								const short2 x = p[np + NS * a];
								pre[q][a] = mk(div_32767((float)x.x), div_32767((float)x.y));
							}
						}
					}
				}
			} else if constexpr (RAW) {                           // a symbol that leaves the frame: zeros outside (SampleSrc::at)
				#pragma unroll
				for (int q = 0; q < NQ; ++q) {
					const int np = tid + NT * q;
					#pragma unroll
					for (int a = 0; a < R1; ++a) {
						const long i = t0 + np + NS * a;
						praw[q][a] = (np < NS && sym <= md.rows && i >= 0 && i < src.n) ? ((const int *)src.base)[i] : 0;
					}
				}
			} else {
				#pragma unroll
				for (int q = 0; q < NQ; ++q) {
					const int np = tid + NT * q;
					#pragma unroll
					for (int a = 0; a < R1; ++a)
						pre[q][a] = (np < NS && sym <= md.rows) ? src.template at_m<MODE>(t0 + np + NS * a) : mk(0.f, 0.f);
				}
			}
		};
		constexpr bool AHEAD = NQ <= DEMOD_PREFETCH_NQ(RATE) && MONO == 0;   // (mono: the points are formed at the top of the symbol)
		if (AHEAD)
			fetch(0);
		cf *carr = carr_all + (size_t)f * CARR_MAX;
		for (int s = 0; s <= md.rows; ++s) {
			if (!AHEAD)
				fetch(s);                                         // several points per thread: their loads overlap each other
			#pragma unroll
			for (int q = 0; q < NQ; ++q) {
				const int np = tid + NT * q;
				if (np < NS) {
					cf v[R1];
					if constexpr (RAW) {
						#pragma unroll
						for (int a = 0; a < R1; ++a)
							pre[q][a] = mk(div_32767((float)(short)(praw[q][a] & 0xffff)), div_32767((float)(praw[q][a] >> 16)));
					}
					if (NQ == 1) {
						#pragma unroll
						for (int a = 0; a < R1; ++a)
							v[a] = cmul(pre[q][a], (QA_LDS && a) ? cmul(p0, rotA[a]) : qa[a]);
					} else if constexpr (QA_REG2) {
						#pragma unroll
						for (int a = 0; a < R1; ++a)
							v[a] = cmul(pre[q][a], qa2[q][a]);
					} else {
						const cf pq = q ? cmul(p0, rotQ[q]) : p0;
						#pragma unroll
						for (int a = 0; a < R1; ++a)
							v[a] = cmul(pre[q][a], a ? cmul(pq, rotA[a]) : pq);
					}
					Bfly<R1>::run(v);
					const int npw = (DEMOD_SWZ && NS == 256 && DC::W == 1) ? swz256(np) : np;
					row[npw] = v[0];
					#pragma unroll
					for (int r = 1; r < R1; ++r)
						row[r * NS + npw] = cmul(v[r], DC::TWR_LDS ? tw_r[(r - 1) * NS + np] : TWR_REG ? twq[TWR_REG ? q : 0][TWR_REG ? r - 1 : 0] : tb.tw_sym[r * np]);
				}
			}
			if (AHEAD)
				fetch(s + 1);
			__syncthreads();
			if constexpr (REGS) {
				fft256_regs(row + wave * NS, twl, lane, swz256(lane));
			} else if constexpr (DEMOD_SWZ && NS == 256 && DC::W == 1) {
				cf *sub = row + wave * NS;
				const int sl = swz256(lane);
				fft256_stage_swz<1, 0>(sub, tw_sub, lane, sl);
				fft256_stage_swz<4, 0>(sub, tw_sub, lane, sl);
				fft256_stage_swz<16, 12>(sub, tw_sub, lane, sl);
				fft256_stage_swz<64, 60>(sub, tw_sub, lane, sl);
			} else
				fft_fwd_compact<NS, 64 * DC::W, SYMBOL_LEN>(row + (tid / (64 * DC::W)) * NS, tw_sub, tid % (64 * DC::W));
			if constexpr (MONO == 2)
				if (s < md.rows)
					mono_part1();                                 // the next symbol's span (its samples arrived during the transform)
			__syncthreads();
			// the payload carriers of symbol s: the time-differential step cons = X_j / X_{j-1} (decode.cc:474-475) against the previous
			// symbol's carriers (DEMOD_CONS_OUT 1; 0: the carriers go to HBM and k_theil_sen does it where it reads them).
			// osc() call count: symbol_len (header) + s*stride + i, decode.cc:459-470
			const cf w = symrot[s];
			#pragma unroll
			for (int e = 0; e < 2; ++e)
				if (coff[e] >= 0) {
					const cf cur = cmul(row[coff[e]], w);
					if constexpr (DEMOD_CONS_OUT(RATE)) {
						// decode.cc:474-475 here: the previous symbol's carrier is the one this thread parked a symbol ago (its own LDS slot)
						if (s > 0)
							cons[(size_t)(s - 1) * md.cols + tid + NT * e] = demod_or_erase(cur, prevc[e][tid]);
						prevc[e][tid] = cur;
					} else
						carr[tid + NT * e] = cur;
				}
			carr += md.cols;
			if constexpr (MONO == 2)
				if (s < md.rows) {
					mono_part2();
					if (s + 2 <= md.rows)
						mono_raw(s + 2);
				}
			__syncthreads();
		}
		};
		if constexpr (MONO != 0)
			symbols(IntC<0>{});                                   // (mono: the points come out of LDS / the z scratch)
		else
			src.with_mode(symbols);
	}
}

// ---------------------------------------------------------------- debug FFT entry
// len = symbol_len or symbol_len/2 of the handle's rate, both directions (backward = conj . forward . conj)
template <int RATE>
__global__ __launch_bounds__(256) void k_fft_debug(int len, int sign, const cf *__restrict__ in, cf *__restrict__ out, const cf *__restrict__ tw)
{
	constexpr int SYMBOL_LEN = RateCfg<RATE>::SL;
	const int f = blockIdx.x, tid = threadIdx.x;
	__shared__ cf buf[SYMBOL_LEN];
	for (int i = tid; i < len; i += 256) {
		cf v = in[(size_t)f * len + i];
		buf[i] = sign > 0 ? cconj(v) : v;
	}
	__syncthreads();
	if (len == SYMBOL_LEN) fft_fwd<SYMBOL_LEN, 256, SYMBOL_LEN>(buf, tw, tid);
	else fft_fwd<SYMBOL_LEN / 2, 256, SYMBOL_LEN>(buf, tw, tid);
	for (int i = tid; i < len; i += 256)
		out[(size_t)f * len + i] = sign > 0 ? cconj(buf[i]) : buf[i];
}

bool demod_forms_cons(int rate) { return DEMOD_CONS_OUT(rate); }
void launch_demod(hipStream_t s, int rate, int n, FrameBatch fb, cf *z, const MonoArgs &ma, Tables tb, const SyncState *st, cf *cons, cf *carr)
{
	if (fb.channels == 1 && mono_fused(rate)) {
		hipLaunchKernelGGL((k_demod<8000, 2>), dim3(n), dim3(DifCfg<8000>::NT), 0, s, fb, z, ma, tb, st, cons, carr);
	} else {
		RX_RATE_SWITCH(rate, hipLaunchKernelGGL((k_demod<RATE, 0>), dim3(n), dim3(DifCfg<RATE>::NT), 0, s, fb, z, ma, tb, st, cons, carr));
	}
}
void launch_fft_debug(hipStream_t s, int rate, int n, int len, int sign, const cf *in, cf *out, Tables tb)
{
	RX_RATE_SWITCH(rate, hipLaunchKernelGGL(k_fft_debug<RATE>, dim3(n), dim3(256), 0, s, len, sign, in, out, tb.tw_sym));
}

}  // namespace rx
