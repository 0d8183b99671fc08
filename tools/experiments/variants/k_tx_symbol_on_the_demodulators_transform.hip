// VARIANT (round 5, measured, NOT adopted: 17.2 ms per 8192 frames against 13.9 for the shipped form; profiles/r05_v5_transmitter_on_the_demodulators_transform.txt).
// k_tx.hip with a second 8 kHz symbol kernel, k_tx_symbol_dif, that runs the nine transforms of a data symbol one after the other on
// k_demod's five-wave 5 x 256 transform (OFDMRX_TX_OLD=1 selects the shipped kernel for the A/B).  Build: SRC_k_tx=tools/experiments/variants/<this file>
// tools/build_variant.sh NAME "".
// k_tx.hip -- N2: the transmitter (Encoder<value,cmplx,8000>, encode.cc:27-318) on the device, so that
// synthetic batches with DISTINCT payloads never cross PCIe.  Not on the receive hot path; built to the same
// parity bar (waveform within +-1 LSB of the CPU restatement after the int16 quantiser, identical bits).
//   k_tx_code     payload -> scramble -> CRC-32 -> systematic polar codeword (encode.cc:293-303,415-419)
//   k_tx_symbol   one OFDM symbol: carriers (pilot / Schmidl-Cox / meta / differential 8PSK|QPSK rows) ->
//                 PAPR clip via 4x oversampling (encode.cc:80-100) -> IFFT1280 -> scale (encode.cc:101-109)
//   k_tx_assemble raised-cosine guard cross-fade with the previous symbol (encode.cc:110-114), quantise to
//                 int16 like DSP::WriteWAV, silence before and after (encode.cc:423,441)
#include "dev_common.h"
#include "kernels.h"
#include <cstdlib>

namespace rx {

struct TxParams {
	int oper_mode, offset, channels, nsym;     // offset = freq_off*symbol_len/rate bins (encode.cc:283)
	unsigned long long md;                     // (call_sign << 8) | mode  (encode.cc:291)
	long frame_samples;
	int count, bits;                           // payloads per stream (encode.cc:289 loop), 8 or 16 bit samples
	int symbol_len;
};

// ---------------------------------------------------------------- polar systematic encoder
__global__ __launch_bounds__(256) void k_tx_code(const uint8_t *__restrict__ payload_all, Tables tb, TxParams tp,
	uint32_t *__restrict__ code_all)
{
	const int f = blockIdx.x, tid = threadIdx.x;
	const ModeDesc md = mode_desc(tp.oper_mode);
	const uint32_t *frozen = tb.frozen + (md.table ? 2048 : 0);
	__shared__ uint8_t msg[PAYLOAD_BYTES + 4];
	__shared__ uint32_t cw[2048];
	__shared__ uint32_t crctab[256];
	__shared__ int rank[2048];                 // unfrozen positions before word w
	__shared__ uint32_t csh[1024], cpart[32];
	crctab[tid] = tb.crc32_tab[tid];
	#pragma unroll
	for (int q = 0; q < 4; ++q)
		csh[tid + 256 * q] = tb.crc32_shift168[tid + 256 * q];
	for (int i = tid; i < PAYLOAD_BYTES; i += 256)
		msg[i] = payload_all[(size_t)f * PAYLOAD_BYTES + i] ^ tb.scramble[i];   // encode.cc:417-419
	__syncthreads();
	// CRC<uint32_t>(0xD419CC15) over the scrambled bytes, encode.cc:295-297: 32 threads run the byte table over 168-byte
	// segments from a zero state, the partial states are folded in order with the "advance by 168 zero bytes" operator
	// (k_finish.hip has the same scheme; 5380 = 32 x 168 + 4).  Unfrozen positions before each word: a wave scan.
	constexpr int SEG = 168, NSEG = 32, TAIL = PAYLOAD_BYTES - SEG * NSEG;
	if (tid < NSEG) {
		const uint8_t *mp = msg + tid * SEG;
		uint32_t crc = 0;
		for (int i = 0; i < SEG; ++i)
			crc = (crc >> 8) ^ crctab[(crc ^ mp[i]) & 255];
		cpart[tid] = crc;
	} else if (tid >= 64 && tid < 128) {       // rank[w] = unfrozen positions before word w: 32 words per lane, then a prefix over the lanes
		const int l = tid - 64;
		int own = 0;
		for (int w = 32 * l; w < 32 * l + 32; ++w)
			own += 32 - __popc(frozen[w]);
		int incl = own;
		#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const int o = __shfl_up(incl, d, 64);
			if (l >= d)
				incl += o;
		}
		int acc = incl - own;
		for (int w = 32 * l; w < 32 * l + 32; ++w) {
			rank[w] = acc;
			acc += 32 - __popc(frozen[w]);
		}
	}
	__syncthreads();
	if (tid == 0) {
		uint32_t crc = 0;
		for (int q = 0; q < NSEG; ++q) {
			crc = csh[crc & 255] ^ csh[256 + ((crc >> 8) & 255)] ^ csh[512 + ((crc >> 16) & 255)] ^ csh[768 + (crc >> 24)];
			crc ^= cpart[q];
		}
		for (int i = SEG * NSEG; i < SEG * NSEG + TAIL; ++i)
			crc = (crc >> 8) ^ crctab[(crc ^ msg[i]) & 255];
		for (int b = 0; b < 4; ++b)
			msg[PAYLOAD_BYTES + b] = (uint8_t)(crc >> (8 * b));   // appended LSB first, encode.cc:298-299
	}
	__syncthreads();
	// u: message bit k at the k-th unfrozen position (bits >= 43072 are +1 = 0), frozen positions 0
	for (int w = tid; w < 2048; w += 256) {
		uint32_t fz = frozen[w], v = 0;
		int k = rank[w];
		for (int b = 0; b < 32; ++b)
			if (!((fz >> b) & 1)) {
				if (k < CRC_BITS)
					v |= (uint32_t)((msg[k >> 3] >> (k & 7)) & 1) << b;
				++k;
			}
		cw[w] = v;
	}
	__syncthreads();
	// x = u F^{(x)16} twice with the frozen positions cleared in between (CODE::PolarSysEnc, encode.cc:302)
	for (int pass = 0; pass < 2; ++pass) {
		for (int w = tid; w < 2048; w += 256) {
			uint32_t v = cw[w];
			if (pass)
				v &= ~frozen[w];
			v ^= (v >> 1) & 0x55555555u;
			v ^= (v >> 2) & 0x33333333u;
			v ^= (v >> 4) & 0x0f0f0f0fu;
			v ^= (v >> 8) & 0x00ff00ffu;
			v ^= (v >> 16) & 0x0000ffffu;
			cw[w] = v;
		}
		__syncthreads();
		for (int h = 1; h < 2048; h <<= 1) {
			for (int q = tid; q < 1024; q += 256) {
				int a = ((q & ~(h - 1)) << 1) | (q & (h - 1));   // word index with bit h clear
				cw[a] ^= cw[a + h];
			}
			__syncthreads();
		}
	}
	for (int w = tid; w < 2048; w += 256)
		code_all[(size_t)f * 2048 + w] = cw[w];
}

// ---------------------------------------------------------------- one OFDM symbol
// The 4x oversampled PAPR buffer (encode.cc:50-51 fdom4/tdom4) is 5120 / 10240 points at 8 / 16 kHz and sits in
// LDS (41 / 82 KB); at 44.1 / 48 kHz it is 28224 / 30720 points (226 / 246 KB > LDS) and lives in a per-workgroup
// global scratch, worked on by 1024 threads so the in-place radix stages still fit the register file.
#define TX_TW_GLOBAL 2     // 0: compact stage twiddles copied to LDS per workgroup; 1: the root table read at a stride through L1;
                           // 2: the compact table read from global memory (consecutive words, L1-resident): 10 KB of LDS less per
                           // workgroup = three workgroups per CU, and a quarter of the transforms' LDS reads gone
template <int RATE> struct TxCfg {
	static constexpr bool BIG_IN_LDS = RATE <= 16000;
#define TX_NT_LDS 256
	static constexpr int NT = BIG_IN_LDS ? TX_NT_LDS : 1024;
};
// The payload carriers of data row j are pilot x the product of the PSK symbols of rows 0..j (the transmitter's
// differential step, encode.cc:304-309: fdom[] keeps multiplying).  One pass per payload forms all rows in that order -
// the very sequence of fp32 complex products the reference runs - and parks them (rows x cols cf, 173 KB in mode 6), so a
// symbol's block reads its row instead of redoing j products per carrier (that recomputation was 80 % of the
// transmitter's instructions).
__global__ __launch_bounds__(256) void k_tx_rows(const uint32_t *__restrict__ code_all, Tables tb, TxParams tp, cf *__restrict__ rowsym_all)
{
	const int fp = blockIdx.x, tid = threadIdx.x;             // one block per payload (stream x count)
	const ModeDesc md = mode_desc(tp.oper_mode);
	const int SL = tp.symbol_len;
	const float code_fac = sqrtf((float)SL / (float)md.cols);    // encode.cc:135
	const uint32_t *code = code_all + (size_t)fp * 2048;
	cf *out = rowsym_all + (size_t)fp * CONS_MAX;
	const float cos_pi_8 = 0.92387953251128675613f, sin_pi_8 = 0.38268343236508977173f, r2 = 0.70710678118654752440f;
	for (int i = tid; i < md.cols; i += 256) {
		cf acc = mk(code_fac * tb.mls2_nrz[i], 0.f);
		for (int r = 0; r < md.rows; ++r) {
			const int p = md.mod_bits * (md.cols * r + i);
			float b[3];
			#pragma unroll
			for (int t = 0; t < 3; ++t) {
				int q = p + t;
				b[t] = t < md.mod_bits ? (float)(1 - 2 * (int)((code[q >> 5] >> (q & 31)) & 1)) : 1.f;
			}
			cf m;
			if (md.mod_bits == 3) {                           // psk.hh:132-139
				float re = cos_pi_8, im = sin_pi_8;
				if (b[0] < 0.f) { re = sin_pi_8; im = cos_pi_8; }
				m = mk(re * b[1], im * b[2]);
			} else {
				m = mk(r2 * b[0], r2 * b[1]);                 // psk.hh:82-85
			}
			acc = cmul(acc, m);
			out[(size_t)r * md.cols + i] = acc;
		}
	}
}

// (8-byte alignment said out loud: `cf` alone promises 4, and the transforms then run on pairs of 32-bit LDS accesses - two-way
// bank conflicts on every one of them, 54 % of the kernel's LDS cycles - instead of ds_read_b64 / ds_write_b64)
template <int RATE> struct alignas(16) TxShared {
	alignas(16) cf big[TxCfg<RATE>::BIG_IN_LDS ? 4 * RateCfg<RATE>::SL : 2];   // LDS path: four decimated sequences [r][symbol_len]
	alignas(16) cf fdom[RateCfg<RATE>::SL];
	alignas(16) cf twc[(TxCfg<RATE>::BIG_IN_LDS && !TX_TW_GLOBAL) ? fft_compact_size<RateCfg<RATE>::SL, RateCfg<RATE>::SL>() : 1];   // compact twiddles of the symbol_len plan
};

// symbol kinds in transmission order (encode.cc:288-313): pilot | S&C | meta | pilot | rows x data | zero
// The pilot, Schmidl-Cox, meta-data and zero symbols do not depend on the payload (one mode, offset and call sign per call):
// they are formed once, for frame 0 / payload 0, and every other frame's cross-fade reads them from there (5 of 55 symbol
// transforms per mode-6 frame less).
__device__ __forceinline__ size_t tx_symbol_slot(int f, int sidx, int nsym, int rows)
{
	const int per = 3 + rows, last = nsym - 1;
	if (sidx == 0 || sidx == last)
		return (size_t)sidx;
	const int w = (sidx - 1) % per;
	return w < 3 ? (size_t)(1 + w) : (size_t)f * nsym + sidx;
}
template <int RATE>
#ifndef TX_WAVES
#define TX_WAVES 2        // waves per SIMD the register budget of k_tx_symbol is set for
#endif
__global__ __launch_bounds__(TxCfg<RATE>::NT, (TxCfg<RATE>::BIG_IN_LDS ? TX_WAVES : 1)) void k_tx_symbol(const cf *__restrict__ rowsym_all, Tables tb, TxParams tp,
	const cf *__restrict__ tw5120, cf *__restrict__ tdom_all, cf *__restrict__ big_scratch)
{
	constexpr int SYMBOL_LEN = RateCfg<RATE>::SL, NT = TxCfg<RATE>::NT;
	auto bin1280 = [](int c) { return (c + SYMBOL_LEN) % SYMBOL_LEN; };               // encode.cc:68-71
	auto bin5120 = [](int c) { return (c + 4 * SYMBOL_LEN) % (4 * SYMBOL_LEN); };     // encode.cc:72-75
	const int f = blockIdx.x / tp.nsym, sidx = blockIdx.x % tp.nsym, tid = threadIdx.x;
	const ModeDesc md = mode_desc(tp.oper_mode);
	if (tx_symbol_slot(f, sidx, tp.nsym, md.rows) != (size_t)f * tp.nsym + sidx)
		return;                                               // a payload-independent symbol: frame 0 makes it
	__shared__ TxShared<RATE> sh;
	cf *big = (cf *)__builtin_assume_aligned(TxCfg<RATE>::BIG_IN_LDS ? sh.big : big_scratch + (size_t)blockIdx.x * (4 * SYMBOL_LEN), 8);
	const int code_off = tp.offset - md.cols / 2;             // encode.cc:284
	const int mls0_off = tp.offset - 127 + 1;                 // encode.cc:285
	const int mls1_off = tp.offset - 255 / 2;                 // encode.cc:286
	for (int i = tid; i < SYMBOL_LEN; i += NT)
		sh.fdom[i] = mk(0.f, 0.f);
	__syncthreads();
	bool papr = true;
	const int last = tp.nsym - 1;
	// stream layout (encode.cc:288-313): pilot | count x (S&C, meta, pilot, rows x data) | zero symbol
	const int per = 3 + md.rows, q = sidx - 1, pay = (sidx > 0 && sidx < last) ? q / per : 0, w = (sidx > 0 && sidx < last) ? q % per : 2;
	if (sidx != 0 && sidx != last && w == 0) {                // schmidl_cox(): encode.cc:142-154
		papr = false;
		if (tid == 0) {
			float c = sqrtf((float)(2 * SYMBOL_LEN) / 127.f);
			sh.fdom[bin1280(mls0_off - 2)] = mk(c, 0.f);
			for (int i = 0; i < 127; ++i) {
				c *= tb.mls0_nrz[i];
				sh.fdom[bin1280(2 * i + mls0_off)] = mk(c, 0.f);
			}
		}
	} else if (sidx != 0 && sidx != last && w == 1) {         // meta_data(): encode.cc:155-179
		if (tid == 0) {
			uint32_t cwd[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
			unsigned long long m = tp.md;
			// 55 bits of md LSB first, then CRC-16(md << 9) LSB first; codeword = XOR of generator rows
			unsigned crc = 0;
			unsigned long long d9 = m << 9;
			for (int i = 0; i < 64; ++i) {
				unsigned t = crc ^ (unsigned)((d9 >> i) & 1);
				crc = (crc >> 1) ^ ((t & 1) * 0xA8F4u);
			}
			crc &= 0xffffu;
			for (int i = 0; i < 71; ++i) {
				int bit = i < 55 ? (int)((m >> i) & 1) : (int)((crc >> (i - 55)) & 1);
				if (bit)
					for (int w = 0; w < 8; ++w)
						cwd[w] ^= tb.genmat_bits[i * 8 + w];
			}
			float c = sqrtf((float)SYMBOL_LEN / 255.f);
			sh.fdom[bin1280(mls1_off - 1)] = mk(c, 0.f);
			for (int i = 0; i < 255; ++i) {
				int bit = (cwd[i >> 5] >> (i & 31)) & 1;
				c *= (float)(1 - 2 * bit);
				sh.fdom[bin1280(i + mls1_off)] = mk(c * tb.mls1_nrz[i], 0.f);   // scrambled after the differential step
			}
		}
	} else if (sidx != last) {
		// pilot (w == 2, and the leading one) or data row j = w - 3: fdom = pilot * product of the rows' PSK symbols
		// (encode.cc:304-309), formed once per payload by k_tx_rows
		const int j = w - 3;
		const float code_fac = sqrtf((float)SYMBOL_LEN / (float)md.cols);   // encode.cc:135
		const cf *rowsym = rowsym_all + ((size_t)f * tp.count + pay) * CONS_MAX + (size_t)(j < 0 ? 0 : j) * md.cols;
		for (int i = tid; i < md.cols; i += NT)
			sh.fdom[bin1280(i + code_off)] = j < 0 ? mk(code_fac * tb.mls2_nrz[i], 0.f) : rowsym[i];
	}
	__syncthreads();
	// symbol(): encode.cc:101-109
	const float s8 = sqrtf((float)(8 * SYMBOL_LEN));
	cf *out = tdom_all + ((size_t)f * tp.nsym + sidx) * SYMBOL_LEN;
	if constexpr (TxCfg<RATE>::BIG_IN_LDS) {
		// improve_papr() (encode.cc:80-100) runs a 4x oversampled transform pair: backward 4 N points of the N-bin spectrum,
		// clip, forward 4 N points of which only the N original bins are kept.  Decimated by four both are FOUR independent
		// N-point transforms: x[4m + r] = IFFT_N(F[c] w^(c r))[m] and X[c] = sum_r w^(c r) FFT_N(x[4m + r])[c], w = e^{-j 2 pi /
		// 4N}.  Wave r owns residue r: its two transforms run in its own quarter of LDS with wave barriers only (no
		// workgroup barrier between the spectrum and the combine), N-point stages instead of 4N-point ones, twiddles from
		// the compact LDS table.
		constexpr int RT = NT / 4;                            // threads per residue (64: wave-private transforms, wave barriers only)
		static_assert(RT % 64 == 0 && RT >= 64, "whole waves per residue");
		const int wave = tid / RT, lane = tid % RT;           // residue, thread within it
		if (!TX_TW_GLOBAL)
			fft_compact_twiddles<SYMBOL_LEN, NT, SYMBOL_LEN>(sh.twc, tb.tw_sym, tid);
		const float s4 = sqrtf((float)(4 * SYMBOL_LEN)), r4 = 1.f / s4;
		auto div_s4 = [&](float x) { const float q0 = x * r4; return __builtin_fmaf(__builtin_fmaf(-s4, q0, x), r4, q0); };   // x / s4
		__syncthreads();
		if (papr && sidx != last) {
			cf *sub = (cf *)__builtin_assume_aligned(big + wave * SYMBOL_LEN, 8);
			auto w4 = [&](int c) {                            // w^(c * wave)
				int t = (c * wave) % (4 * SYMBOL_LEN);
				return tw5120[t < 0 ? t + 4 * SYMBOL_LEN : t];
			};
			for (int i = lane; i < SYMBOL_LEN; i += RT) {
				const int c = i - SYMBOL_LEN / 2, b = bin1280(c);
				const cf o = sh.fdom[b];
				cf g = mk(0.f, 0.f);
				if (o.re != 0.f || o.im != 0.f)
					g = wave ? cmul(cconj(o), w4(c)) : cconj(o);   // conj in, conj out = backward transform
				sub[b] = g;
			}
			fft_sync<RT>();
			if (TX_TW_GLOBAL == 1) fft_fwd<SYMBOL_LEN, RT, SYMBOL_LEN>(sub, tb.tw_sym, lane); else fft_fwd_compact<SYMBOL_LEN, RT, SYMBOL_LEN>(sub, TX_TW_GLOBAL ? tb.tw_symc : sh.twc, lane);
			for (int i = lane; i < SYMBOL_LEN; i += RT) {
				cf v = cconj(sub[i]);
				v = mk(div_s4(v.re), div_s4(v.im));
				const float amp = fmaxf(fabsf(v.re), fabsf(v.im));
				if (amp > 1.f)
					v = mk(v.re / amp, v.im / amp);
				sub[i] = v;
			}
			fft_sync<RT>();
			if (TX_TW_GLOBAL == 1) fft_fwd<SYMBOL_LEN, RT, SYMBOL_LEN>(sub, tb.tw_sym, lane); else fft_fwd_compact<SYMBOL_LEN, RT, SYMBOL_LEN>(sub, TX_TW_GLOBAL ? tb.tw_symc : sh.twc, lane);
			__syncthreads();
			for (int i = tid; i < SYMBOL_LEN; i += NT) {
				const int c = i - SYMBOL_LEN / 2, b = bin1280(c);
				const cf o = sh.fdom[b];
				cf keep = mk(0.f, 0.f);
				if (cnorm(o) != 0.f) {
					cf acc = big[b];
					#pragma unroll
					for (int r = 1; r < 4; ++r) {
						int t = (c * r) % (4 * SYMBOL_LEN);
						acc = cadd(acc, cmul(big[r * SYMBOL_LEN + b], tw5120[t < 0 ? t + 4 * SYMBOL_LEN : t]));
					}
					keep = mk(div_s4(acc.re), div_s4(acc.im));
				}
				sh.fdom[b] = cconj(keep);
			}
		} else {
			for (int i = tid; i < SYMBOL_LEN; i += NT)
				sh.fdom[i] = cconj(sh.fdom[i]);
		}
		__syncthreads();
		if (TX_TW_GLOBAL == 1) fft_fwd<SYMBOL_LEN, NT, SYMBOL_LEN>(sh.fdom, tb.tw_sym, tid); else fft_fwd_compact<SYMBOL_LEN, NT, SYMBOL_LEN>(sh.fdom, TX_TW_GLOBAL ? tb.tw_symc : sh.twc, tid);
		const float r8 = 1.f / s8;
		for (int i = tid; i < SYMBOL_LEN; i += NT) {
			cf v = cconj(sh.fdom[i]);
			// v / s8 as q0 = v r8, q0 + fma(-s8, q0, v) r8: the correctly rounded quotient in three instructions (Markstein; within the
			// +-1 LSB contract of this path in any case)
			const float qr = v.re * r8, qi = v.im * r8;
			out[i] = mk(__builtin_fmaf(__builtin_fmaf(-s8, qr, v.re), r8, qr), __builtin_fmaf(__builtin_fmaf(-s8, qi, v.im), r8, qi));
		}
	} else {
		cf *temp = sh.fdom;                                   // global scratch for the 4N-point buffer, fdom itself for the symbol
		if (papr && sidx != last) {
			// improve_papr(): encode.cc:80-100
			for (int i = tid; i < 4 * SYMBOL_LEN; i += NT)
				big[i] = mk(0.f, 0.f);
			__syncthreads();
			for (int i = tid; i < SYMBOL_LEN; i += NT) {
				int c = i - SYMBOL_LEN / 2;
				big[bin5120(c)] = cconj(sh.fdom[bin1280(c)]);  // conj in, conj out = backward transform
			}
			__syncthreads();
			fft_fwd<4 * SYMBOL_LEN, NT, 4 * SYMBOL_LEN>(big, tw5120, tid);
			const float s4 = sqrtf((float)(4 * SYMBOL_LEN));
			for (int i = tid; i < 4 * SYMBOL_LEN; i += NT) {
				cf v = cconj(big[i]);
				v = mk(v.re / s4, v.im / s4);
				float amp = fmaxf(fabsf(v.re), fabsf(v.im));
				if (amp > 1.f)
					v = mk(v.re / amp, v.im / amp);
				big[i] = v;
			}
			__syncthreads();
			fft_fwd<4 * SYMBOL_LEN, NT, 4 * SYMBOL_LEN>(big, tw5120, tid);
			constexpr int NK = (SYMBOL_LEN + NT - 1) / NT;
			cf keep[NK];
			#pragma unroll
			for (int q = 0; q < NK; ++q) {
				int i = tid + NT * q, c = i - SYMBOL_LEN / 2;
				keep[q] = mk(0.f, 0.f);
				if (i < SYMBOL_LEN) {
					cf o = sh.fdom[bin1280(c)], v = big[bin5120(c)];
					if (cnorm(o) != 0.f)
						keep[q] = mk(v.re / s4, v.im / s4);
				}
			}
			__syncthreads();
			#pragma unroll
			for (int q = 0; q < NK; ++q) {
				int i = tid + NT * q, c = i - SYMBOL_LEN / 2;
				if (i < SYMBOL_LEN)
					temp[bin1280(c)] = cconj(keep[q]);
			}
		} else {
			for (int i = tid; i < SYMBOL_LEN; i += NT)
				temp[i] = cconj(sh.fdom[i]);
		}
		__syncthreads();
		fft_fwd<SYMBOL_LEN, NT, SYMBOL_LEN>(temp, tb.tw_sym, tid);
		for (int i = tid; i < SYMBOL_LEN; i += NT) {
			cf v = cconj(temp[i]);
			out[i] = mk(v.re / s8, v.im / s8);
		}
	}
}

// ---------------------------------------------------------------- one OFDM symbol, 8 kHz (round 5)
// The same arithmetic (encode.cc:80-109) on the demodulator's transform (k_demod.hip): 1280 = 5 x 256 - a loader thread takes the five
// inputs n' + 256 a, runs the radix-5 butterfly in registers, applies w^(n' r) and parks output r in row r; wave r then transforms
// its row (256 points, ONE radix-4 butterfly per lane and stage, swizzled so that every access is bank-conflict-free, wave barriers
// only); X[5 q + r] sits at row r, place swz256(q).  Two workgroup barriers per transform, five waves on EVERY transform (the first
// form gave each residue of the PAPR step to one wave: twenty points per lane and stage, dependent chains five times as long, twelve
// waves per CU).  The nine transforms of a data symbol run one after the other; the four residues' spectra are summed in registers
// (a thread owns four bins), so the 40 KB buffer of the four decimated sequences is gone: 40 KB of LDS per workgroup, four per CU.
template <int RATE>
__global__ __launch_bounds__(320, 5) void k_tx_symbol_dif(const cf *__restrict__ rowsym_all, Tables tb, TxParams tp, const cf *__restrict__ tw5120,
	cf *__restrict__ tdom_all)
{
	constexpr int SL = RateCfg<RATE>::SL, NS = 256, R1 = 5, NT = 320;
	static_assert(SL == R1 * NS, "8 kHz: 1280 = 5 x 256");
	constexpr int TWC = fft_compact_size<NS, SL>();
	auto bin = [](int c) { return (c + SL) % SL; };
	const int f = blockIdx.x / tp.nsym, sidx = blockIdx.x % tp.nsym, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const ModeDesc md = mode_desc(tp.oper_mode);
	if (tx_symbol_slot(f, sidx, tp.nsym, md.rows) != (size_t)f * tp.nsym + sidx)
		return;                                               // a payload-independent symbol: frame 0 makes it
	__shared__ cf rowA[SL], rowB[SL], fdom[SL];
	__shared__ cf tw_sub[TWC], tw_r[(R1 - 1) * NS];
	fft_compact_twiddles<NS, NT, SL>(tw_sub, tb.tw_sym, tid);
	for (int i = tid; i < (R1 - 1) * NS; i += NT)
		tw_r[i] = tb.tw_sym[(i / NS + 1) * (i % NS)];
	const int code_off = tp.offset - md.cols / 2;             // encode.cc:284
	const int mls0_off = tp.offset - 127 + 1;                 // encode.cc:285
	const int mls1_off = tp.offset - 255 / 2;                 // encode.cc:286
	for (int i = tid; i < SL; i += NT)
		fdom[i] = mk(0.f, 0.f);
	__syncthreads();
	bool papr = true;
	const int last = tp.nsym - 1;
	const int per = 3 + md.rows, q = sidx - 1, pay = (sidx > 0 && sidx < last) ? q / per : 0, w = (sidx > 0 && sidx < last) ? q % per : 2;
	if (sidx != 0 && sidx != last && w == 0) {                // schmidl_cox(): encode.cc:142-154
		papr = false;
		if (tid == 0) {
			float c = sqrtf((float)(2 * SL) / 127.f);
			fdom[bin(mls0_off - 2)] = mk(c, 0.f);
			for (int i = 0; i < 127; ++i) {
				c *= tb.mls0_nrz[i];
				fdom[bin(2 * i + mls0_off)] = mk(c, 0.f);
			}
		}
	} else if (sidx != 0 && sidx != last && w == 1) {         // meta_data(): encode.cc:155-179
		if (tid == 0) {
			uint32_t cwd[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
			unsigned long long m = tp.md;
			unsigned crc = 0;
			unsigned long long d9 = m << 9;
			for (int i = 0; i < 64; ++i) {
				unsigned t = crc ^ (unsigned)((d9 >> i) & 1);
				crc = (crc >> 1) ^ ((t & 1) * 0xA8F4u);
			}
			crc &= 0xffffu;
			for (int i = 0; i < 71; ++i) {
				int bit = i < 55 ? (int)((m >> i) & 1) : (int)((crc >> (i - 55)) & 1);
				if (bit)
					for (int ww = 0; ww < 8; ++ww)
						cwd[ww] ^= tb.genmat_bits[i * 8 + ww];
			}
			float c = sqrtf((float)SL / 255.f);
			fdom[bin(mls1_off - 1)] = mk(c, 0.f);
			for (int i = 0; i < 255; ++i) {
				int bit = (cwd[i >> 5] >> (i & 31)) & 1;
				c *= (float)(1 - 2 * bit);
				fdom[bin(i + mls1_off)] = mk(c * tb.mls1_nrz[i], 0.f);
			}
		}
	} else if (sidx != last) {
		const int j = w - 3;
		const float code_fac = sqrtf((float)SL / (float)md.cols);   // encode.cc:135
		const cf *rowsym = rowsym_all + ((size_t)f * tp.count + pay) * CONS_MAX + (size_t)(j < 0 ? 0 : j) * md.cols;
		for (int i = tid; i < md.cols; i += NT)
			fdom[bin(i + code_off)] = j < 0 ? mk(code_fac * tb.mls2_nrz[i], 0.f) : rowsym[i];
	}
	__syncthreads();
	// one transform: in(n) for n = tid + 256 a (loader threads tid < 256) -> rows of `dst`; X[k] = dst[(k % 5) * 256 + swz256(k / 5)]
	auto transform = [&](cf *dst, auto in) {
		if (tid < NS) {
			cf v[R1];
			#pragma unroll
			for (int a = 0; a < R1; ++a)
				v[a] = in(tid + NS * a);
			Bfly<R1>::run(v);
			const int sw = swz256(tid);
			dst[sw] = v[0];
			#pragma unroll
			for (int r = 1; r < R1; ++r)
				dst[r * NS + sw] = cmul(v[r], tw_r[(r - 1) * NS + tid]);
		}
		__syncthreads();
		cf *sub = dst + wave * NS;
		const int sl = swz256(lane);
		fft256_stage_swz<1, 0>(sub, tw_sub, lane, sl);
		fft256_stage_swz<4, 0>(sub, tw_sub, lane, sl);
		fft256_stage_swz<16, 12>(sub, tw_sub, lane, sl);
		fft256_stage_swz<64, 60>(sub, tw_sub, lane, sl);
		__syncthreads();
	};
	auto at = [&](const cf *rows, int k) { return rows[(k % R1) * NS + swz256(k / R1)]; };
	if (papr && sidx != last) {
		// improve_papr(), encode.cc:80-100, decimated by four: x[4 m + r] = IFFT_N(F[c] w^(c r))[m], X[c] = sum_r w^(c r) FFT_N(x[4 m + r])[c],
		// w = e^{-j 2 pi / 4 N}
		const float s4 = sqrtf((float)(4 * SL)), r4 = 1.f / s4;
		auto div_s4 = [&](float x) { const float q0 = x * r4; return __builtin_fmaf(__builtin_fmaf(-s4, q0, x), r4, q0); };   // x / s4
		auto w4 = [&](int c, int r) {                         // w^(c r)
			int t = (c * r) % (4 * SL);
			return tw5120[t < 0 ? t + 4 * SL : t];
		};
		cf acc[4];
		bool occ[4];
		#pragma unroll
		for (int e = 0; e < 4; ++e) {
			acc[e] = mk(0.f, 0.f);
			const cf o = fdom[tid + NT * e];
			occ[e] = o.re != 0.f || o.im != 0.f;
		}
		#pragma unroll 1
		for (int r = 0; r < 4; ++r) {
			transform(rowA, [&](int n) {                      // conj in, conj out = backward transform
				const cf o = fdom[n];
				if (o.re == 0.f && o.im == 0.f)
					return mk(0.f, 0.f);
				const int c = n < SL / 2 ? n : n - SL;
				return r ? cmul(cconj(o), w4(c, r)) : cconj(o);
			});
			transform(rowB, [&](int m) {
				cf v = cconj(at(rowA, m));
				v = mk(div_s4(v.re), div_s4(v.im));
				const float amp = fmaxf(fabsf(v.re), fabsf(v.im));
				if (amp > 1.f)
					v = mk(v.re / amp, v.im / amp);
				return v;
			});
			#pragma unroll
			for (int e = 0; e < 4; ++e)
				if (occ[e]) {
					const int b = tid + NT * e, c = b < SL / 2 ? b : b - SL;
					const cf y = at(rowB, b);
					acc[e] = r ? cadd(acc[e], cmul(y, w4(c, r))) : y;
				}
		}
		#pragma unroll
		for (int e = 0; e < 4; ++e)                           // (every transform above is behind a barrier: nobody reads fdom any more)
			fdom[tid + NT * e] = occ[e] ? cconj(mk(div_s4(acc[e].re), div_s4(acc[e].im))) : mk(0.f, 0.f);
	} else {
		#pragma unroll
		for (int e = 0; e < 4; ++e)
			fdom[tid + NT * e] = cconj(fdom[tid + NT * e]);
	}
	__syncthreads();
	transform(rowA, [&](int n) { return fdom[n]; });
	// symbol(): encode.cc:101-109
	const float s8 = sqrtf((float)(8 * SL)), r8 = 1.f / s8;
	cf *out = tdom_all + ((size_t)f * tp.nsym + sidx) * SL;
	#pragma unroll
	for (int e = 0; e < 4; ++e) {
		const int i = tid + NT * e;
		const cf v = cconj(at(rowA, i));
		const float qr = v.re * r8, qi = v.im * r8;           // v / s8, correctly rounded (Markstein)
		out[i] = mk(__builtin_fmaf(__builtin_fmaf(-s8, qr, v.re), r8, qr), __builtin_fmaf(__builtin_fmaf(-s8, qi, v.im), r8, qi));
	}
}

// ---------------------------------------------------------------- guard cross-fade + int16
template <int RATE>
__global__ __launch_bounds__(256) void k_tx_assemble(const cf *__restrict__ tdom_all, TxParams tp, void *__restrict__ pcm_all)
{
	constexpr int SYMBOL_LEN = RateCfg<RATE>::SL, GUARD_LEN = RateCfg<RATE>::GL, SYM_STRIDE = RateCfg<RATE>::STRIDE;
	const int f = blockIdx.x / (tp.nsym + 2), part = blockIdx.x % (tp.nsym + 2), tid = threadIdx.x;
	const int ch = tp.channels;
	int16_t *pcm = (int16_t *)pcm_all + (size_t)f * tp.frame_samples * ch;
	uint8_t *pcm8 = (uint8_t *)pcm_all + (size_t)f * tp.frame_samples * ch;
	const bool b8 = tp.bits == 8;
	auto put = [&](long n, cf v) {   // DSP::WriteWAV: clamp, scale by 2^(bits-1)-1, round; 8 bit = unsigned, offset 128
		float re = fminf(fmaxf(v.re, -1.f), 1.f), im = fminf(fmaxf(v.im, -1.f), 1.f);
		if (b8) {
			pcm8[n * ch] = (uint8_t)((int)nearbyintf(127.f * re) + 128);
			if (ch == 2)
				pcm8[n * ch + 1] = (uint8_t)((int)nearbyintf(127.f * im) + 128);
		} else {
			pcm[n * ch] = (int16_t)nearbyintf(32767.f * re);
			if (ch == 2)
				pcm[n * ch + 1] = (int16_t)nearbyintf(32767.f * im);
		}
	};
	if (part >= tp.nsym) {                                    // silence(rate) before and after, encode.cc:423,441
		long base = part == tp.nsym ? 0 : RATE + (long)tp.nsym * SYM_STRIDE;
		for (int i = tid; i < RATE; i += 256)
			put(base + i, mk(0.f, 0.f));
		return;
	}
	const int rows = mode_desc(tp.oper_mode).rows;
	const cf *cur = tdom_all + tx_symbol_slot(f, part, tp.nsym, rows) * SYMBOL_LEN;
	const cf *prv = part ? tdom_all + tx_symbol_slot(f, part - 1, tp.nsym, rows) * SYMBOL_LEN : nullptr;
	const long base = RATE + (long)part * SYM_STRIDE;
	for (int i = tid; i < GUARD_LEN; i += 256) {              // encode.cc:110-114
		float x = (float)i / (float)(GUARD_LEN - 1);
		x = 0.5f * (1.f - cosf(PI_F * x));
		cf a = prv ? prv[i] : mk(0.f, 0.f), b = cur[i + SYMBOL_LEN - GUARD_LEN];
		put(base + i, mk((1.f - x) * a.re + x * b.re, (1.f - x) * a.im + x * b.im));
	}
	for (int i = tid; i < SYMBOL_LEN; i += 256)
		put(base + GUARD_LEN + i, cur[i]);
}

size_t tx_big_scratch_bytes(int rate, int n, int nsym)
{
	return rate <= 16000 ? 0 : (size_t)n * nsym * 4 * (size_t)rate_symbol_len(rate) * sizeof(cf);
}

void launch_tx(hipStream_t s, int rate, int n, const uint8_t *payload, Tables tb, const void *tp_, const cf *tw5120,
	uint32_t *code, cf *rowsym, cf *tdom, cf *big_scratch, void *pcm)
{
	TxParams tp = *(const TxParams *)tp_;
	tp.symbol_len = rate_symbol_len(rate);
	hipLaunchKernelGGL(k_tx_code, dim3(n * tp.count), dim3(256), 0, s, payload, tb, tp, code);
	hipLaunchKernelGGL(k_tx_rows, dim3(n * tp.count), dim3(256), 0, s, code, tb, tp, rowsym);
	static const bool old_form = std::getenv("OFDMRX_TX_OLD") != nullptr;   // A/B: the first form of the 8 kHz symbol kernel
	if (rate == 8000 && !old_form) {
		hipLaunchKernelGGL(k_tx_symbol_dif<8000>, dim3(n * tp.nsym), dim3(320), 0, s, rowsym, tb, tp, tw5120, tdom);
		hipLaunchKernelGGL(k_tx_assemble<8000>, dim3(n * (tp.nsym + 2)), dim3(256), 0, s, tdom, tp, pcm);
		return;
	}
	RX_RATE_SWITCH(rate,
		hipLaunchKernelGGL(k_tx_symbol<RATE>, dim3(n * tp.nsym), dim3(TxCfg<RATE>::NT), 0, s, rowsym, tb, tp, tw5120, tdom, big_scratch);
		hipLaunchKernelGGL(k_tx_assemble<RATE>, dim3(n * (tp.nsym + 2)), dim3(256), 0, s, tdom, tp, pcm));
}

}  // namespace rx
