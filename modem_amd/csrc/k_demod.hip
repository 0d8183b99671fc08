// k_demod.hip -- D4 (51 x FFT1280 + time-differential demod) for gfx950; D5 (Theil-Sen) lives in k_theilsen.hip, D6-D8 in
// k_finish.hip (k_back).  Compiled with -fno-slp-vectorize (Makefile): the
// SLP vectoriser packs the butterflies into v_pk_*_f32 (6.4 cycles per issue against 3.3, measured) and pays a v_mov per
// pack - the 8 kHz demodulator is issue-bound and runs 1.75 -> 1.4 ms per 8192 frames without it.
#include "dev_common.h"
#include "kernels.h"

namespace rx {

// ---------------------------------------------------------------- D4
// decode.cc:453-477.  One workgroup per frame walks the pilot and the data symbols in order; samples are read once, straight from
// the raw PCM.  symbol_len = R1 x NS (1280 = 5 x 256, 2560 = 5 x 512, 7056 = 7 x 1008, 7680 = 5 x 1536): radix-R1 decimation in
// frequency in registers, then R1 row transforms of NS points - see the kernel.
#ifndef DEMOD_CONS_OUT
#define DEMOD_CONS_OUT(R) ((R) != 44100)   // the demodulator forms cons = X_j / X_{j-1} itself, from the carriers of the previous symbol parked in
                            // LDS (5 KB at 8 kHz): k_theil_sen is VALU-bound and 0.42 ms per chunk shorter without the row formation, k_demod is
                            // LDS-bound and 0.16 ms longer with it.  44.1 kHz measured 3 % slower with it and keeps the other form: the
                            // carriers go to HBM and k_theil_sen forms the rows where it reads them
#endif
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
#ifndef DEMOD_DIF_WAVES
#define DEMOD_DIF_WAVES(R) ((R) >= 44100 ? 3 : (R) == 8000 ? 7 : 4)   // 8 kHz: 72 VGPRs = five workgroups per CU (26.6 KB of LDS each); 64 spills with the swizzle
#endif
	static constexpr bool TWR_LDS = (R1 - 1) * NS * 8 <= DEMOD_TWR_BYTES;   // w^(n' r) table in LDS (8 / 16 kHz) or read from the global root table
	static constexpr int WAVES = DEMOD_DIF_WAVES(RATE);      // waves per SIMD the register budget is set for (two workgroups per CU at 44.1 / 48 kHz)
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
__global__ __launch_bounds__(DifCfg<RATE>::NT, DifCfg<RATE>::WAVES) void k_demod(FrameBatch fb, const cf *__restrict__ z_all, Tables tb,
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
			__syncthreads();
		}
		});
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
void launch_demod(hipStream_t s, int rate, int n, FrameBatch fb, const cf *z, Tables tb, const SyncState *st, cf *cons, cf *carr)
{
	RX_RATE_SWITCH(rate, hipLaunchKernelGGL(k_demod<RATE>, dim3(n), dim3(DifCfg<RATE>::NT), 0, s, fb, z, tb, st, cons, carr));
}
void launch_fft_debug(hipStream_t s, int rate, int n, int len, int sign, const cf *in, cf *out, Tables tb)
{
	RX_RATE_SWITCH(rate, hipLaunchKernelGGL(k_fft_debug<RATE>, dim3(n), dim3(256), 0, s, len, sign, in, out, tb.tw_sym));
}

}  // namespace rx
