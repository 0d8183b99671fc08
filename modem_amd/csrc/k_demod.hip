// k_demod.hip -- D4 (51 x FFT1280 + time-differential demod), D6/D7 (cumulative SNR estimate + 8PSK soft demap),
// D8 (lengthen) for gfx950; D5 (Theil-Sen) lives in k_theilsen.hip.  Compiled with -fno-slp-vectorize (Makefile): the
// SLP vectoriser packs the butterflies into v_pk_*_f32 (6.4 cycles per issue against 3.3, measured) and pays a v_mov per
// pack - the 8 kHz demodulator is issue-bound and runs 1.75 -> 1.4 ms per 8192 frames without it.
#include "dev_common.h"
#include "kernels.h"

namespace rx {

// ---------------------------------------------------------------- D4
#ifndef DEMOD_TPS
#define DEMOD_TPS 256      // 8 kHz: threads per symbol transform.  256 (four waves share one 1280-point buffer, 126 VGPRs, 10 KB of LDS:
                           // 4.0 ms per 8192 frames alone) beats one wave per symbol (64: 252 VGPRs, 40 KB, 4.3 ms) and 128 (166 VGPRs, 4.7 ms)
#endif
#ifndef DEMOD_NT8
#define DEMOD_NT8 256      // 8 kHz: threads per workgroup (DEMOD_NT8 / DEMOD_TPS symbols in flight per workgroup)
#endif
#ifndef DEMOD_TW_LDS
#define DEMOD_TW_LDS 1     // 8 kHz: the 1280 twiddles are copied to LDS once per frame (4.1 -> 3.4 ms per 8192 frames; 0 = read through L1)
#endif
template <int RATE> struct DemodShared {                     // 8 kHz: one 1280-point buffer per symbol slot
	cf fft[DEMOD_NT8 / DEMOD_TPS][RateCfg<RATE>::SL];
};
template <int RATE> struct DemodSharedBlock {                // other rates: one buffer, the whole block per symbol
	cf fft[RateCfg<RATE>::SL];
	cf carr[2][COLS_MAX];
};

// decode.cc:453-477.  One workgroup (4 waves) per frame.
// 8 kHz: wave w transforms symbol 4g+w of group g in its own LDS buffer (radix 5,4,4,4,4 Stockham
// stages); the payload carriers of each symbol are parked in an 8-slot LDS ring so that
// cons = X_j / X_{j-1} needs no second pass over HBM.  Samples are read once, straight from the raw
// PCM (int16 pairs).  16 / 44.1 / 48 kHz (2560 / 7056 / 7680 points, 20-61 KB): 1024 threads share one buffer
// (two to eight points per thread and radix stage) and walk the symbols in order, two carrier slots.
#ifndef DEMOD_WAVE_PER_SYMBOL
#define DEMOD_WAVE_PER_SYMBOL(R) ((R) == 8000)
#endif
#ifndef DEMOD_CONS_OUT
#define DEMOD_CONS_OUT(R) ((R) != 44100)   // 1 (since the end of round 3; 44.1 kHz measured 3 % slower with it and keeps the old form): the register-decimated demodulator forms cons = X_j / X_{j-1} itself, from the carriers of the
                            // previous symbol parked in LDS (5 KB at 8 kHz) - k_theil_sen is VALU-bound and 0.42 ms per chunk shorter without the row
                            // formation, k_demod is LDS-bound and 0.16 ms longer with it; 0: the carriers go to HBM and k_theil_sen forms the rows
#endif
#ifndef DEMOD_DIF
#define DEMOD_DIF 1        // radix-5 / radix-7 decimation in frequency in registers, then 5 / 7 wave-private transforms (all rates)
#endif
// symbol_len = R1 x NS: 1280 = 5 x 256, 2560 = 5 x 512, 7056 = 7 x 1008, 7680 = 5 x 1536
template <int RATE> struct DifCfg {
	static constexpr int SL = RateCfg<RATE>::SL;
#ifndef DEMOD_DIF_W
#define DEMOD_DIF_W(R) ((R) == 48000 ? 3 : (R) == 44100 ? 2 : 1)
#endif
	static constexpr int W = DEMOD_DIF_W(RATE);              // waves that share one row's transform (1: wave-private, wave barriers only)
	static constexpr int R1 = SL % 5 == 0 ? 5 : 7, NS = SL / R1, NT = 64 * W * R1;
	static constexpr int NQ = (NS + NT - 1) / NT;             // points n' per loader thread
#ifndef DEMOD_TWR_BYTES
#define DEMOD_TWR_BYTES 8192
#endif
#ifndef DEMOD_DBL_BYTES
#define DEMOD_DBL_BYTES 0
#endif
#ifndef DEMOD_DIF_WAVES
#define DEMOD_DIF_WAVES(R) ((R) >= 44100 ? 3 : (R) == 8000 ? 7 : 4)   // 8 kHz: 72 VGPRs = five workgroups per CU (26.6 KB of LDS each); 64 spills with the swizzle
#endif
	static constexpr bool TWR_LDS = (R1 - 1) * NS * 8 <= DEMOD_TWR_BYTES;   // w^(n' r) table in LDS (8 / 16 kHz) or read from the global root table
	static constexpr bool DOUBLE = 2 * SL * 8 <= DEMOD_DBL_BYTES;       // two row buffers, no third barrier per symbol: OFF by default (DEMOD_DBL_BYTES 0:
	                                                                     // the second buffer cost a workgroup per CU and measured slower); -DDEMOD_DBL_BYTES=20480 turns it on at 8 kHz
	static constexpr int WAVES = DEMOD_DIF_WAVES(RATE);      // waves per SIMD the register budget is set for (two workgroups per CU at 44.1 / 48 kHz)
};
template <int RATE> struct DemodCfg {
	static constexpr bool DIF = DEMOD_DIF != 0;
	static constexpr int NT = DIF ? DifCfg<RATE>::NT : DEMOD_WAVE_PER_SYMBOL(RATE) ? DEMOD_NT8 : 1024;     // threads per frame
#ifndef DEMOD_MINB
#define DEMOD_MINB (DEMOD_DIF ? 2 : DEMOD_TPS == 256 ? 4 : 2)
#endif
	static constexpr int MINB = DIF ? DifCfg<RATE>::WAVES : DEMOD_WAVE_PER_SYMBOL(RATE) ? DEMOD_MINB : 1;        // waves per SIMD the register budget is set for
};

// Bank-conflict-free image of a wave-private 256-point buffer: element i lives at i ^ ((i >> 2) & 3) ^ (((i >> 4) & 3) << 2).  The
// radix-4 stages read 64 consecutive elements per instruction (any bijection of the low six bits keeps that conflict-free) and
// write at strides of 4 (P = 1) and 16 / 4 (P = 4) elements, which the plain layout puts four lanes deep on a bank; under the
// swizzle every 16-lane store group touches 32 distinct banks (checked against the bank rules of the guide in a simulation).
// Same operations in the same order as fft_stage<256, 4, P, 64>: only the addresses change.
#ifndef DEMOD_SWZ
#define DEMOD_SWZ 1     // 8 kHz (the only rate with wave-private 256-point transforms): demod 1.48 -> 1.35 ms per chunk with the seven-waves budget below
#endif
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

template <int RATE>
__global__ __launch_bounds__(DemodCfg<RATE>::NT, DemodCfg<RATE>::MINB) void k_demod(FrameBatch fb, const cf *__restrict__ z_all, Tables tb,
	const SyncState *__restrict__ st_all, cf *__restrict__ cons_all, cf *__restrict__ carr_all)
{
	constexpr int SYMBOL_LEN = RateCfg<RATE>::SL, SYM_STRIDE = RateCfg<RATE>::STRIDE;
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
	if constexpr (DemodCfg<RATE>::DIF) {
		// symbol_len = R1 x NS.  X[R1 q + r] = sum_n' wNS^(n' q) [ w^(n' r) sum_a x[n' + NS a] wR1^(a r) ], w = e^{-j 2 pi / symbol_len}:
		// a loader thread takes the R1 samples n' + NS a (stride NS: the raw PCM is read coalesced), runs the radix-R1 butterfly in
		// registers, applies the R1 - 1 twiddles and parks output r in row r of an LDS buffer; wave r (R1 waves) then transforms
		// row r - NS points, NS / (64 R) butterflies per lane and stage, wave barriers only, twiddles from a compact LDS table.
		// Two workgroup barriers per symbol (after the rows are written, before the carriers are read) against two per radix
		// stage of a cooperative transform; at 8 / 16 kHz the rows are double-buffered, at 44.1 / 48 kHz (56 / 61 KB) a third
		// barrier protects the single buffer.
		typedef DifCfg<RATE> DC;
		constexpr int NT = DC::NT, R1 = DC::R1, NS = DC::NS, NQ = DC::NQ;
		static_assert(R1 * NS == SYMBOL_LEN && NT == DemodCfg<RATE>::NT, "plan");
		constexpr int TWC = fft_compact_size<NS, SYMBOL_LEN>();
		__shared__ cf rows[DC::DOUBLE ? 2 : 1][R1 * NS];
		__shared__ cf tw_sub[TWC];                            // compact twiddles of the NS-point plan
		__shared__ cf tw_r[DC::TWR_LDS ? (R1 - 1) * NS : 1];  // w^(n' r), r = 1..R1-1
		__shared__ cf rotA[R1], rotQ[NQ], symrot[ROWS_MAX + 1];
		__shared__ cf prevc[2][DEMOD_CONS_OUT(RATE) ? NT : 1];   // a thread's (at most two) carriers of the previous symbol
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
		cf qa[NQ == 1 ? R1 : 1];                             // one point per loader: the R1 phasors stay in registers
		if (NQ == 1) {
			#pragma unroll
			for (int a = 0; a < R1; ++a)
				qa[a] = a ? cmul(p0, rotA[a]) : p0;
		}
		// the (at most two) carriers of this thread sit at the same place of the rows in every symbol
		int coff[2];
		#pragma unroll
		for (int e = 0; e < 2; ++e) {
			const int i = tid + NT * e, k = (i + code_off + SYMBOL_LEN) % SYMBOL_LEN;
			coff[e] = i < md.cols ? (k % R1) * NS + ((DEMOD_SWZ && NS == 256 && DC::W == 1) ? swz256(k / R1) : k / R1) : -1;
		}
		src.with_mode([&](auto M) {
		constexpr int MODE = decltype(M)::value;
		cf pre[NQ][R1];
		auto fetch = [&](int sym) {
			const long t0 = body0 + (long)sym * SYM_STRIDE;       // wave-uniform
			if (MODE == 1 && sym <= md.rows && t0 >= 0 && t0 + SYMBOL_LEN <= src.n) {
				// the whole symbol lies inside the frame (the rule): int16 pairs from a uniform base, no per-sample checks
				const short2 *p = (const short2 *)src.base + t0;
				#pragma unroll
				for (int q = 0; q < NQ; ++q) {
					const int np = tid + NT * q;
					if (np < NS) {
						#pragma unroll
						for (int a = 0; a < R1; ++a) {
							const short2 x = p[np + NS * a];
							pre[q][a] = mk(div_32767((float)x.x), div_32767((float)x.y));
						}
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
#ifndef DEMOD_PREFETCH_NQ
#define DEMOD_PREFETCH_NQ 1   // loader points per thread up to which the NEXT symbol's samples are fetched during the transform
#endif
		constexpr bool AHEAD = NQ <= DEMOD_PREFETCH_NQ;
		if (AHEAD)
			fetch(0);
		cf *carr = carr_all + (size_t)f * CARR_MAX;
		for (int s = 0; s <= md.rows; ++s) {
			cf *row = rows[DC::DOUBLE ? (s & 1) : 0];
			if (!AHEAD)
				fetch(s);                                         // several points per thread: their loads overlap each other
			#pragma unroll
			for (int q = 0; q < NQ; ++q) {
				const int np = tid + NT * q;
				if (np < NS) {
					cf v[R1];
					if (NQ == 1) {
						#pragma unroll
						for (int a = 0; a < R1; ++a)
							v[a] = cmul(pre[q][a], qa[a]);
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
						row[r * NS + npw] = cmul(v[r], DC::TWR_LDS ? tw_r[(r - 1) * NS + np] : tb.tw_sym[r * np]);
				}
			}
			if (AHEAD)
				fetch(s + 1);
			__syncthreads();
			if constexpr (DEMOD_SWZ && NS == 256 && DC::W == 1) {
				cf *sub = row + wave * NS;
				const int sl = swz256(lane);
				fft256_stage_swz<1, 0>(sub, tw_sub, lane, sl);
				fft256_stage_swz<4, 0>(sub, tw_sub, lane, sl);
				fft256_stage_swz<16, 12>(sub, tw_sub, lane, sl);
				fft256_stage_swz<64, 60>(sub, tw_sub, lane, sl);
			} else
				fft_fwd_compact<NS, 64 * DC::W, SYMBOL_LEN>(row + (tid / (64 * DC::W)) * NS, tw_sub, tid % (64 * DC::W));
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
			if (!DC::DOUBLE)
				__syncthreads();
		}
		});
	} else if constexpr (DEMOD_WAVE_PER_SYMBOL(RATE)) {
		__shared__ DemodShared<RATE> sh;
		constexpr int TPS = DEMOD_TPS, SLOTS = DEMOD_NT8 / TPS;   // TPS threads share one transform; SLOTS symbols in flight per workgroup
		const int slot = tid / TPS, lt = tid % TPS;
		(void)wave; (void)lane;
		// NCO as at the other rates: e^{j omega (k0 + lt + TPS q)} = one closed-form phasor per thread and symbol times a table
		// of e^{j omega TPS q} built once per frame - a complex multiply per sample instead of a double-precision range
		// reduction and a sincos (DEMOD_NCO_TABLE 0: the closed form for every sample)
#ifndef DEMOD_NCO_TABLE
#define DEMOD_NCO_TABLE 1
#endif
		__shared__ cf rot8[SYMBOL_LEN / TPS];
		// twiddles in LDS, in the transform plan's compact per-stage layout (the plain 1280-entry table read at the early
		// stages' strides put every lane on one bank: 44 % of the kernel's LDS time was conflict replays)
		__shared__ cf tw_l[DEMOD_TW_LDS ? fft_compact_size<SYMBOL_LEN, SYMBOL_LEN>() : 1];
		if (DEMOD_TW_LDS) {
			fft_compact_twiddles<SYMBOL_LEN, DEMOD_NT8, SYMBOL_LEN>(tw_l, tb.tw_sym, tid);
			__syncthreads();
		}
		if (DEMOD_NCO_TABLE) {
			if (tid < SYMBOL_LEN / TPS)
				rot8[tid] = phasor(omega, (long)TPS * tid);
			__syncthreads();
		}
		const int groups = (md.rows + 1 + SLOTS - 1) / SLOTS;
		// the samples of the NEXT symbol are fetched while the current one is transformed (a symbol is 5 KB of raw PCM: its
		// HBM round trip is as long as the whole transform)
		constexpr int NQ = SYMBOL_LEN / TPS;
		src.with_mode([&](auto M) {
		cf pre[NQ];
		auto fetch = [&](int sym) {
			#pragma unroll
			for (int q = 0; q < NQ; ++q)
				pre[q] = sym <= md.rows ? src.template at_m<decltype(M)::value>(body0 + (long)sym * SYM_STRIDE + lt + TPS * q) : mk(0.f, 0.f);
		};
		fetch(slot);
		for (int g = 0; g < groups; ++g) {
			const int s = SLOTS * g + slot;                       // 0 = pilot, 1..rows = data rows
			const bool valid = s <= md.rows;
			cf *buf = sh.fft[slot];
			const cf base = phasor(omega, (long)SYMBOL_LEN + (long)s * SYM_STRIDE + lt);
			#pragma unroll
			for (int q = 0; q < NQ; ++q) {
				int i = lt + TPS * q;
				cf v = mk(0.f, 0.f);
				if (valid)   // osc() call count: symbol_len (header) + s*stride + i, decode.cc:459-470
					v = cmul(pre[q], DEMOD_NCO_TABLE ? cmul(base, rot8[q]) : phasor(omega, (long)SYMBOL_LEN + (long)s * SYM_STRIDE + i));
				buf[i] = v;
			}
			fetch(s + SLOTS);
			fft_sync<TPS>();
			// TPS = 64: one wave, its own buffer, no workgroup barriers; otherwise the slots run in lock-step
			if (DEMOD_TW_LDS)
				fft_fwd_compact<SYMBOL_LEN, TPS, SYMBOL_LEN>(buf, tw_l, lt);
			else
				fft_fwd<SYMBOL_LEN, TPS, SYMBOL_LEN>(buf, tb.tw_sym, lt);
			// the payload carriers of symbol s go to HBM (cols x 8 B); the time-differential step
			// cons = X_j / X_{j-1} (decode.cc:474-475) happens where they are read (k_theil_sen): no carrier ring,
			// no dependence between the waves, 40 KB of LDS per workgroup
			if (valid) {
				cf *carr = carr_all + (size_t)f * CARR_MAX + (size_t)s * md.cols;
				for (int i = lt; i < md.cols; i += TPS)
					carr[i] = buf[(i + code_off + SYMBOL_LEN) % SYMBOL_LEN];
			}
			fft_sync<TPS>();                                      // the next group refills the buffer
		}
		});
	} else {
		__shared__ DemodSharedBlock<RATE> sh;
		// NCO: e^{j omega (k0 + tid + 256 q)} = (one closed-form phasor per thread and symbol) x (a table of
		// e^{j omega 256 q}, q < symbol_len / 256, built once per frame): one complex multiply per sample instead
		// of a double-precision range reduction and a sincos (2560..7680 samples per symbol at these rates)
		constexpr int NT = DemodCfg<RATE>::NT, NQ = (SYMBOL_LEN + NT - 1) / NT;
		__shared__ cf rot[NQ];
		if (tid < NQ)
			rot[tid] = phasor(omega, (long)NT * tid);
		__syncthreads();
		for (int s = 0; s <= md.rows; ++s) {
			const cf base = phasor(omega, (long)SYMBOL_LEN + (long)s * SYM_STRIDE + tid);
			#pragma unroll 2
			for (int q = 0; q < NQ; ++q) {
				const int i = tid + NT * q;
				if (i < SYMBOL_LEN)
					sh.fft[i] = cmul(src.at(body0 + (long)s * SYM_STRIDE + i), cmul(base, rot[q]));
			}
			__syncthreads();
			fft_fwd<SYMBOL_LEN, NT, SYMBOL_LEN>(sh.fft, tb.tw_sym, tid);
			for (int i = tid; i < md.cols; i += NT) {
				cf x = sh.fft[(i + code_off + SYMBOL_LEN) % SYMBOL_LEN];
				sh.carr[s & 1][i] = x;
				if (s >= 1)                                       // decode.cc:474-475 (same thread wrote slot (s-1)&1)
					cons[(s - 1) * md.cols + i] = demod_or_erase(x, sh.carr[(s - 1) & 1][i]);
			}
			__syncthreads();
		}
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

bool demod_writes_carriers(int rate) { return DEMOD_DIF || DEMOD_WAVE_PER_SYMBOL(rate); }
bool demod_forms_cons(int rate) { return !demod_writes_carriers(rate) || (DEMOD_DIF && DEMOD_CONS_OUT(rate)); }
void launch_demod(hipStream_t s, int rate, int n, FrameBatch fb, const cf *z, Tables tb, const SyncState *st, cf *cons, cf *carr)
{
	RX_RATE_SWITCH(rate, hipLaunchKernelGGL(k_demod<RATE>, dim3(n), dim3(DemodCfg<RATE>::NT), 0, s, fb, z, tb, st, cons, carr));
}
void launch_fft_debug(hipStream_t s, int rate, int n, int len, int sign, const cf *in, cf *out, Tables tb)
{
	RX_RATE_SWITCH(rate, hipLaunchKernelGGL(k_fft_debug<RATE>, dim3(n), dim3(256), 0, s, len, sign, in, out, tb.tw_sym));
}

}  // namespace rx
