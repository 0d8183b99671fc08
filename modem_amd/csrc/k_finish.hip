// k_finish.hip -- D10 (systematic message, CRC-32 lane selection, bit packing, descramble) for gfx950.
#include "dev_common.h"
#include "kernels.h"

namespace rx {

// ---------------------------------------------------------------- D6-D8 + syndrome certificate + D10 for the frames it decides
// k_back replaces k_llr (k_demod.hip) when the certificate is on.  Per frame:
//   1. the row loop of k_llr (decode.cc:505-523: running sp / np, per-row precision) - and, beside it, the SIGN of every soft bit
//      (psk.hh:76-80,125-130: sign(re), sign(im), sign(|re| - |im|); the precision is a positive factor) as one bit per code
//      position in LDS.  No LLR is written yet.
//   2. Syndrome certificate.  If those hard decisions x already ARE a codeword - u = x F^(x16) is zero on every frozen position -
//      and no LLR is zero, the list decoder's answer is known without running it (any list size):
//        * along the path that follows the hard decisions every node's LLR vector carries the signs of that node's sub-codeword
//          (f: sign = product of the signs = a xor b; g with the matching left partial sum: b and +-a have the same sign, so the
//          sum keeps the sign of b), no magnitude is ever zero, so every frozen leaf sees a positive LLR (no penalty) and every
//          information leaf's matching candidate costs nothing: the path keeps metric 0 from start to end;
//        * metrics are sums of non-negative penalties (the seven placeholder paths start at 1000), every other candidate costs
//          more than 0, candidates are ranked by (metric, index): the metric-0 path is lane 0 after every fork and at the end.
//      So lane 0's re-encoded codeword is x itself, its flip count (decode.cc:546-555) 0, and decode.cc:532-541 takes lane 0 if
//      its CRC-32 is 0.
//   3. Certified and CRC-32 of x at the unfrozen positions = 0: the frame is FINISHED here - payload bytes (descrambled) and the
//      result record, exactly what k_polar + k_finish would have produced (best_lane 0, bit_flips 0); cert = 1, and neither of
//      those kernels looks at the frame again.  No LLR, no partial-sum array is ever written for it.
//   4. Otherwise (not a codeword, a zero LLR, or a codeword with the wrong CRC - then the reference goes on to lanes 1..7):
//      cert = 0, the 65536 LLRs are written now (second pass over the row, same arithmetic as k_llr) and the list decoder and
//      k_finish run as always.
// At the benchmark's noise level (-30 dB) every frame is decided here; from -24 dB on almost none (tools/flips_probe.py).
// cert_all: [n] verdicts, then [n] unused, [n + 1] = frames left to the list decoder (cleared before the launch).
__global__ __launch_bounds__(256) void k_back(int sym_stride, const SyncState *__restrict__ st_all, const cf *__restrict__ cons_all,
	const float *__restrict__ slope_all, const float *__restrict__ yint_all, float *__restrict__ precision_all,
	float *__restrict__ llr_all, Result *__restrict__ res_all, float *__restrict__ esn0_rows, Tables tb, int descramble,
	uint8_t *__restrict__ payload_all, int *__restrict__ cert_all)
{
	const int f = blockIdx.x, tid = threadIdx.x;
	const SyncState st = st_all[f];
	uint8_t *payload = payload_all + (size_t)f * PAYLOAD_BYTES;
	if (esn0_rows && tid < ROWS_MAX)
		esn0_rows[(size_t)f * ROWS_MAX + tid] = 0.f;             // rows this frame does not have (all of them without a header)
	Result r = res_all[f];
	r.status = st.status;
	r.symbol_pos = st.symbol_pos;
	r.sc_start = st.sc_start;
	r.cfo_rad = st.cfo_rad;
	r.oper_mode = st.oper_mode;
	r.call_sign = st.call_sign;
	r.n_sync_rejects = st.rejects;
	r.best_lane = -1;
	r.bit_flips = 0;
	if (!st.okay) {                                               // what k_finish writes for a frame without a header
		r.cfo_fine = st.cfo_rad;
		r.sfo_slope = 0.f;
		r.esn0_db_last = 0.f;
		for (int i = tid; i < PAYLOAD_BYTES; i += 256)
			payload[i] = 0;
		if (tid == 0) {
			res_all[f] = r;
			cert_all[f] = 1;
		}
		return;
	}
	__shared__ double rsum[ROWS_MAX][2];
	__shared__ uint32_t bits[CODE_LEN / 32];
	__shared__ float prec[ROWS_MAX];
	__shared__ uint8_t mesg[MESG_BYTES_MAX];
	__shared__ uint32_t ctab[256], csh[1024], cpart[32];
	__shared__ uint32_t crc_sh;
	const ModeDesc md = mode_desc(st.oper_mode);
	const cf *cons = cons_all + (size_t)f * CONS_MAX;
	float *llr = llr_all + (size_t)f * CODE_LEN;
	const float rcp_sqrt_2 = 0.70710678118654752440f;         // psk.hh:57,104
	const float DIST = md.mod_bits == 3 ? 2.f * 0.38268343236508977173f : 2.f * rcp_sqrt_2;   // psk.hh:106 / psk.hh:59
	for (int q = tid; q < CODE_LEN / 32; q += 256)
		bits[q] = 0;
	ctab[tid] = tb.crc32_tab[tid];
	#pragma unroll
	for (int q = 0; q < 4; ++q)
		csh[tid + 256 * q] = tb.crc32_shift168[tid + 256 * q];
	__syncthreads();
	// ---- 1. decode.cc:505-523 (snr_rows, shared with k_llr) + the signs of the soft bits
	bool odd = false;                                             // a zero / NaN LLR somewhere: no certificate
	const int mod_bits = md.mod_bits, cols = md.cols;
	const bool snr_ok = snr_rows(cons, md.rows, cols, mod_bits, tid, rsum, prec, [&](int j, int i, cf c) {
		const float are = fabsf(c.re), aim = fabsf(c.im);
		uint32_t v;
		if (mod_bits == 3) {
			v = (are < aim ? 1u : 0u) | (c.re < 0.f ? 2u : 0u) | (c.im < 0.f ? 4u : 0u);
			odd |= !(are > 0.f) | !(aim > 0.f) | (are == aim);
		} else {
			v = (c.re < 0.f ? 1u : 0u) | (c.im < 0.f ? 2u : 0u);
			odd |= !(are > 0.f) | !(aim > 0.f);
		}
		const int p0 = mod_bits * (j * cols + i), o = p0 & 31;
		if (v) {
			atomicOr(&bits[p0 >> 5], v << o);
			if (o + mod_bits > 32)
				atomicOr(&bits[(p0 >> 5) + 1], v >> (32 - o));
		}
	});
	odd |= !snr_ok;                                               // (LLR = value * DIST * precision)
	if (tid < md.rows) {
		precision_all[(size_t)f * ROWS_MAX + tid] = prec[tid];
		if (esn0_rows)
			esn0_rows[(size_t)f * ROWS_MAX + tid] = 10.f * log10f(prec[tid]);   // decode.cc:518
	}
	if (tid == 0) {
		float sum_slope = 0.f, sum_yint = 0.f;
		for (int j = 0; j < md.rows; ++j) {                   // decode.cc:491-492
			sum_slope += slope_all[(size_t)f * ROWS_MAX + j];
			sum_yint += yint_all[(size_t)f * ROWS_MAX + j];
		}
		r.sfo_slope = sum_slope / (float)md.rows;
		r.cfo_fine = st.cfo_rad + (sum_yint / (float)md.rows) / (float)sym_stride;   // decode.cc:501
		r.esn0_db_last = 10.f * log10f(prec[md.rows - 1]);    // decode.cc:518
	}
	// ---- 3a. (before the transform overwrites the bit array) the systematic message = x at the unfrozen positions, decode.cc:254-261
	const uint16_t *info_pos = tb.info_pos + (md.table ? MESG_BITS_MAX : 0);
	const int mesg_bytes = md.mesg_bits / 8;
	for (int bi = tid; bi < mesg_bytes; bi += 256) {
		uint32_t o = 0;
		#pragma unroll
		for (int b = 0; b < 8; ++b) {
			const int p = info_pos[8 * bi + b];
			o |= ((bits[p >> 5] >> (p & 31)) & 1u) << b;
		}
		mesg[bi] = (uint8_t)o;
	}
	// ---- 2. u = x F: at every level the left half of a block takes the XOR with the right half (the involution the partial-sum
	// combines of the decoder apply the other way round).  Word a = tid + 256 q: distances >= 256 words are inside the thread.
	uint32_t w[8];
	#pragma unroll
	for (int q = 0; q < 8; ++q) {
		uint32_t v = bits[tid + 256 * q];
		v ^= (v >> 1) & 0x55555555u;
		v ^= (v >> 2) & 0x33333333u;
		v ^= (v >> 4) & 0x0f0f0f0fu;
		v ^= (v >> 8) & 0x00ff00ffu;
		v ^= (v >> 16) & 0x0000ffffu;
		w[q] = v;
	}
	#pragma unroll
	for (int q = 0; q < 4; ++q) w[q] ^= w[q + 4];               // 1024 words
	w[0] ^= w[2]; w[1] ^= w[3]; w[4] ^= w[6]; w[5] ^= w[7];       // 512
	w[0] ^= w[1]; w[2] ^= w[3]; w[4] ^= w[5]; w[6] ^= w[7];       // 256
	for (int dw = 128; dw >= 1; dw >>= 1) {
		__syncthreads();
		#pragma unroll
		for (int q = 0; q < 8; ++q)
			bits[tid + 256 * q] = w[q];
		__syncthreads();
		if ((tid & dw) == 0) {
			#pragma unroll
			for (int q = 0; q < 8; ++q)
				w[q] ^= bits[tid + dw + 256 * q];
		}
	}
	const uint32_t *frozen = tb.frozen + (md.table ? CODE_LEN / 32 : 0);
	uint32_t syn = 0;
	#pragma unroll
	for (int q = 0; q < 8; ++q)
		syn |= w[q] & frozen[tid + 256 * q];
	int bad = __syncthreads_or((syn != 0) | (odd ? 1 : 0));
	// ---- 3b. CRC<uint32_t>(0xD419CC15) over the first 43072 bits (decode.cc:533-541), as k_finish does it for a lane: 32 segments
	// of 168 bytes from a zero state, folded in order with the "advance by 168 zero bytes" operator
	if (!bad) {
		constexpr int SEG = 168, NSEG = 32, TAIL = CRC_BITS / 8 - SEG * NSEG;   // 5384 = 32 * 168 + 8
		if (tid < NSEG) {
			const uint8_t *mp = mesg + tid * SEG;
			uint32_t crc = 0;
			for (int i = 0; i < SEG; ++i)
				crc = (crc >> 8) ^ ctab[(crc ^ mp[i]) & 255];
			cpart[tid] = crc;
		}
		__syncthreads();
		if (tid == 0) {
			uint32_t crc = 0;
			for (int q = 0; q < NSEG; ++q) {
				crc = csh[crc & 255] ^ csh[256 + ((crc >> 8) & 255)] ^ csh[512 + ((crc >> 16) & 255)] ^ csh[768 + (crc >> 24)];
				crc ^= cpart[q];
			}
			for (int i = SEG * NSEG; i < SEG * NSEG + TAIL; ++i)
				crc = (crc >> 8) ^ ctab[(crc ^ mesg[i]) & 255];
			crc_sh = crc;
		}
		__syncthreads();
		bad = crc_sh != 0;
	}
	if (!bad) {
		for (int i = tid; i < PAYLOAD_BYTES; i += 256)
			payload[i] = mesg[i] ^ (descramble ? tb.scramble[i] : (uint8_t)0);
		if (tid == 0) {
			r.best_lane = 0;
			res_all[f] = r;
			cert_all[f] = 1;
		}
		return;
	}
	// ---- 4. the list decoder has to look: its LLRs (decode.cc:520-529), same arithmetic as k_llr
	if (tid == 0) {
		res_all[f] = r;                                       // k_finish completes the record
		cert_all[f] = 0;
		atomicAdd(cert_all + gridDim.x + 1, 1);
	}
	for (int j = 0; j < md.rows; ++j) {
		const float sc = DIST * prec[j];
		#pragma unroll
		for (int q = 0; q < 2; ++q) {
			const int i = tid + 256 * q;
			if (i < md.cols) {                                // psk.hh:76-80,125-130
				const cf c = cons[j * md.cols + i];
				float *b = llr + md.mod_bits * (j * md.cols + i);
				if (md.mod_bits == 3) {
					b[1] = c.re * sc;
					b[2] = c.im * sc;
					b[0] = (rcp_sqrt_2 * (fabsf(c.re) - fabsf(c.im))) * sc;
				} else {
					b[0] = c.re * sc;
					b[1] = c.im * sc;
				}
			}
		}
	}
	for (int i = md.cons_bits + tid; i < CODE_LEN; i += 256)   // lengthen(), decode.cc:252
		llr[i] = 9000.f;
}

// ---------------------------------------------------------------- D10
// decode.cc:254-261 (systematic message = codeword at the unfrozen positions),
// decode.cc:532-541 (first lane whose CRC-32 over 43072 bits is 0), decode.cc:546-555
// (LE bit packing + flip count), decode.cc:613-615 (descramble).
__global__ __launch_bounds__(256) void k_finish(const SyncState *__restrict__ st_all, const float *__restrict__ llr_all,
	const uint8_t *__restrict__ hard_all, Tables tb, int descramble, int list, int n_frames, uint8_t *__restrict__ lane_mesg_all,
	uint8_t *__restrict__ payload_all, Result *__restrict__ res_all, const int *__restrict__ cert_all)
{
	// cert_all (nullable): frames with 1 were finished by k_back (syndrome certificate)
	const int f = blockIdx.x, tid = threadIdx.x;
	if (cert_all && cert_all[f] == 1)
		return;
	const SyncState st = st_all[f];
	uint8_t *payload = payload_all + (size_t)f * PAYLOAD_BYTES;
	__shared__ uint8_t mesg[LIST][MESG_BYTES_MAX];
	__shared__ uint32_t crcs[LIST];
	__shared__ int flips_red[4];
	__shared__ uint32_t ctab[256], csh[1024], cpart[LIST][32];
	ctab[tid] = tb.crc32_tab[tid];
	#pragma unroll
	for (int q = 0; q < 4; ++q)
		csh[tid + 256 * q] = tb.crc32_shift168[tid + 256 * q];
	Result r = res_all[f];
	r.status = st.status;
	r.symbol_pos = st.symbol_pos;
	r.sc_start = st.sc_start;
	r.cfo_rad = st.cfo_rad;
	r.oper_mode = st.oper_mode;
	r.call_sign = st.call_sign;
	r.n_sync_rejects = st.rejects;
	r.best_lane = -1;
	r.bit_flips = 0;
	if (!st.okay) {
		r.cfo_fine = st.cfo_rad;
		r.sfo_slope = 0.f;
		r.esn0_db_last = 0.f;
		for (int i = tid; i < PAYLOAD_BYTES; i += 256)
			payload[i] = 0;
		if (tid == 0)
			res_all[f] = r;
		return;
	}
	// list 4: k_polar decodes the frames (2u, 2u+1) as a pair when both have a header and share the frozen table; the pair's
	// partial sums are ONE byte array in the first frame's slot, bits 0..3 = first frame, bits 4..7 = second
	int hshift = 0;
	size_t hframe = (size_t)f;
	if (list == 4) {
		const int mate = f ^ 1;
		if (mate < n_frames) {
			const SyncState sm = st_all[mate];
			if (sm.okay && !(cert_all && cert_all[mate] == 1) && (sm.oper_mode >= 10) == (st.oper_mode >= 10)) {   // (k_polar's pairing rule)
				hframe = (size_t)(f & ~1);
				hshift = (f & 1) * 4;
			}
		}
	}
	const uint8_t *hard = hard_all + hframe * CODE_LEN;
	const float *llr = llr_all + (size_t)f * CODE_LEN;
	const ModeDesc md = mode_desc(st.oper_mode);
	const uint16_t *info_pos = tb.info_pos + (md.table ? MESG_BITS_MAX : 0);
	const int mesg_bytes = md.mesg_bits / 8;
	// transpose: 8 code positions (one byte each, bit k = path k) -> one message byte per path
	for (int bi = tid; bi < mesg_bytes; bi += 256) {
		uint32_t o[LIST] = { 0, 0, 0, 0, 0, 0, 0, 0 };
		#pragma unroll
		for (int b = 0; b < 8; ++b) {
			uint32_t x = (uint32_t)hard[info_pos[8 * bi + b]] >> hshift;
			#pragma unroll
			for (int k = 0; k < LIST; ++k)
				o[k] |= ((x >> k) & 1u) << b;
		}
		#pragma unroll
		for (int k = 0; k < LIST; ++k)
			mesg[k][bi] = (uint8_t)o[k];
	}
	__syncthreads();
	if (lane_mesg_all)
		for (int i = tid; i < LIST * MESG_BYTES; i += 256)
			lane_mesg_all[(size_t)f * LIST * MESG_BYTES + i] = mesg[i / MESG_BYTES][i % MESG_BYTES];
	// CRC<uint32_t>(0xD419CC15) over the first 43072 bits of each lane (decode.cc:533-541), 32 threads per lane:
	// every thread runs the byte-table CRC over its own 168-byte segment from a zero state, then the 32 partial
	// states are folded in order with the "advance by 168 zero bytes" operator (CRC is linear: state(A|B) =
	// advance(state(A), |B|) ^ state(B)); that operator is four 256-entry tables built on the host.
	{
		constexpr int SEG = 168, NSEG = 32, TAIL = CRC_BITS / 8 - SEG * NSEG;   // 5384 = 32 * 168 + 8
		const int lk = tid >> 5, seg = tid & 31;
		const uint8_t *mp = mesg[lk] + seg * SEG;
		uint32_t crc = 0;
		for (int i = 0; i < SEG; ++i)
			crc = (crc >> 8) ^ ctab[(crc ^ mp[i]) & 255];
		cpart[lk][seg] = crc;
		__syncthreads();
		if (tid < LIST) {
			crc = 0;
			for (int q = 0; q < NSEG; ++q) {
				crc = csh[crc & 255] ^ csh[256 + ((crc >> 8) & 255)] ^ csh[512 + ((crc >> 16) & 255)] ^ csh[768 + (crc >> 24)];
				crc ^= cpart[tid][q];
			}
			for (int i = SEG * NSEG; i < SEG * NSEG + TAIL; ++i)
				crc = (crc >> 8) ^ ctab[(crc ^ mesg[tid][i]) & 255];
			crcs[tid] = crc;
		}
	}
	__syncthreads();
	int best = -1;
	for (int k = list - 1; k >= 0; --k)                       // decode.cc:532-541 over the list's lanes
		if (crcs[k] == 0)
			best = k;
	r.best_lane = best;
	if (best < 0) {
		r.status = 6;                                         // decode.cc:542-545
		for (int i = tid; i < PAYLOAD_BYTES; i += 256)
			payload[i] = 0;
		if (tid == 0)
			res_all[f] = r;
		return;
	}
	// decode.cc:546-554: received hard decision against decoded bit over the data bits.  Message bit i is x at the i-th
	// unfrozen position (ascending), so the walk runs over CODE positions - coalesced reads of the LLRs, the partial-sum
	// bytes and the frozen bitmap instead of two dependent gathers per bit: positions below the one of data bit DATA_BITS
	// that are not frozen.
	int flips = 0;
	{
		const uint32_t *frozen = tb.frozen + (md.table ? CODE_LEN / 32 : 0);
		const int p_end = info_pos[DATA_BITS];
		for (int p = 4 * tid; p < p_end; p += 4 * 256) {         // four positions per thread and step: 16 B of LLRs, 4 partial-sum bytes
			const float4 l = *(const float4 *)(llr + p);
			const uint32_t h = (*(const uint32_t *)(hard + p) >> hshift) >> best;
			const uint32_t fz = frozen[p >> 5] >> (p & 31);
			const float lv[4] = { l.x, l.y, l.z, l.w };
			#pragma unroll
			for (int e = 0; e < 4; ++e) {
				const int received = lv[e] < 0.f, decoded = (h >> (8 * e)) & 1;
				const int counts = (p + e < p_end) & !((fz >> e) & 1);
				flips += counts & (received != decoded);
			}
		}
	}
	#pragma unroll
	for (int m = 32; m; m >>= 1)
		flips += __shfl_xor(flips, m);
	if ((tid & 63) == 0)
		flips_red[tid >> 6] = flips;
	for (int i = tid; i < PAYLOAD_BYTES; i += 256)
		payload[i] = mesg[best][i] ^ (descramble ? tb.scramble[i] : (uint8_t)0);
	__syncthreads();
	if (tid == 0) {
		r.bit_flips = flips_red[0] + flips_red[1] + flips_red[2] + flips_red[3];
		res_all[f] = r;
	}
}

void launch_finish(hipStream_t s, int list, int n, const SyncState *st, const float *llr, const uint8_t *hard, Tables tb,
	int descramble, uint8_t *lane_mesg, uint8_t *payload, Result *res, const int *cert)
{
	hipLaunchKernelGGL(k_finish, dim3(n), dim3(256), 0, s, st, llr, hard, tb, descramble, list == 4 ? 4 : 8, n, lane_mesg, payload, res, cert);
}
__global__ void k_cert_clear(int *__restrict__ counters) { counters[threadIdx.x] = 0; }
__global__ void k_cert_log(const int *__restrict__ counters, int *__restrict__ log) { *log = counters[1]; }
void launch_back(hipStream_t s, int rate, int n, const SyncState *st, const cf *cons, const float *slope, const float *yint,
	float *precision, float *llr, Result *res, float *esn0_rows, Tables tb, int descramble, uint8_t *payload, int *cert, int *log)
{
	const int sym_stride = rate_symbol_len(rate) + rate_symbol_len(rate) / 8;
	// (one-thread kernels instead of hipMemsetAsync / a 4-byte hipMemcpyAsync: the runtime's blit copy cost 0.26 ms of stream time each)
	hipLaunchKernelGGL(k_cert_clear, dim3(1), dim3(2), 0, s, cert + n);
	hipLaunchKernelGGL(k_back, dim3(n), dim3(256), 0, s, sym_stride, st, cons, slope, yint, precision, llr, res, esn0_rows, tb, descramble,
		payload, cert);
	if (log)
		hipLaunchKernelGGL(k_cert_log, dim3(1), dim3(1), 0, s, cert + n, log);
}

}  // namespace rx
