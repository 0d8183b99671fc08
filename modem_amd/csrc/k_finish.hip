// k_finish.hip -- the back end behind the Theil-Sen stage for gfx950: k_back = D5's rotation + D6-D8 (SNR estimate, soft demapper,
// lengthen) + the syndrome certificate (+ D10 for the frames it decides); k_finish = D10 for the frames the list decoder
// decoded; and the small kernels that run the list decoder's work queue.
#include "dev_common.h"
#include "kernels.h"

namespace rx {

// ---------------------------------------------------------------- k_back
// One workgroup per frame:
//   1. decode.cc:493-494 on the fly: every point of the frame is rotated by its row's Theil-Sen line where it is used
//      (rotate_point; k_theil_sen leaves slope / yint and no longer writes a rotated copy); decode.cc:505-523: running sp / np,
//      per-row precision - and, beside it, the SIGN of every soft bit (psk.hh:76-80,125-130: sign(re), sign(im),
//      sign(|re| - |im|); the precision is a positive factor) as one bit per code position in LDS.  No LLR is written yet.
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
//      result record, exactly what k_polar + k_finish would have produced (best_lane 0, bit_flips 0).  No LLR, no partial-sum
//      array is ever written for it.
//   4. Otherwise (not a codeword, a zero LLR, or a codeword with the wrong CRC - then the reference goes on to lanes 1..7, or
//      the certificate was not tried): the frame takes the next slot of the list decoder's queue (kernels.h: ListQueue) and its
//      65536 LLRs are written there in a second pass over the rows (decode.cc:520-529, 252).
// At the benchmark's noise level (-30 dB) every frame is decided here; from -24 dB on almost none (tools/flips_probe.py).  So
// the certificate is ADAPTIVE (cert_mode 1): k_queue_snap, which runs behind every launch, switches it to a probe sample (one
// frame in sixteen) when fewer than 5 % of the frames it was tried for were finished by it, and back when a fifth of the
// sample is.  Either way every decision is exact: the list decoder is the general path.  cert_mode 0: never tried (the
// reference's behaviour, OFDMRX_FLAG_SCL_ALWAYS; handles with debug taps).
// slot_of[f] (per chunk): the queue slot of the frame, -1 if it needs none (no header; finished here).
#ifndef BACK_WAVES
#define BACK_WAVES 6      // register budget (waves per SIMD): 6 = 80 VGPRs, two of them spilled; the 19.5 KB of LDS allow eight workgroups per CU
#endif
__global__ __launch_bounds__(256, BACK_WAVES) void k_back(int sym_stride, int cert_mode, const SyncState *__restrict__ st_all, const cf *__restrict__ cons_all,
	const float *__restrict__ slope_all, const float *__restrict__ yint_all, float *__restrict__ precision_all,
	Result *__restrict__ res_all, float *__restrict__ esn0_rows, Tables tb, int descramble, uint8_t *__restrict__ payload_all,
	ListQueue *__restrict__ q, ListSlot *__restrict__ slots, float *__restrict__ llr_q, int *__restrict__ slot_of,
	uint8_t *__restrict__ payload_later, Result *__restrict__ res_later, ScRing sc, int chunk_seq)
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
	if (!st.okay) {                                               // no header -> nothing to decode (decode.cc:450-451)
		r.cfo_fine = st.cfo_rad;
		r.sfo_slope = 0.f;
		r.esn0_db_last = 0.f;
		for (int i = tid; i < PAYLOAD_BYTES; i += 256)
			payload[i] = 0;
		if (tid == 0) {
			res_all[f] = r;
			slot_of[f] = -1;
		}
		return;
	}
	__shared__ double rsum[ROWS_MAX][2];
	__shared__ uint32_t bits[CODE_LEN / 32];
	__shared__ float prec[ROWS_MAX], row_slope[ROWS_MAX], row_yint[ROWS_MAX];
	__shared__ cf row_step[ROWS_MAX];
	__shared__ uint32_t mesg32[MESG_BYTES_MAX / 4];               // the systematic message, little-endian words
	uint8_t *const mesg = (uint8_t *)mesg32;
	__shared__ uint32_t ctab[256], cpart[4];
	__shared__ int slot_sh;
	ListQueue *const q_cert = q;                                      // the syndrome certificate's switch and counters (the list queue's block)
	const bool try_cert = cert_mode && (q->cert_on || (f & 15) == 0);   // (uniform in the workgroup)
	const ModeDesc md = mode_desc(st.oper_mode);
	const cf *cons = cons_all + (size_t)f * CONS_MAX;
	const float rcp_sqrt_2 = 0.70710678118654752440f;         // psk.hh:57,104
	const float DIST = md.mod_bits == 3 ? 2.f * 0.38268343236508977173f : 2.f * rcp_sqrt_2;   // psk.hh:106 / psk.hh:59
	if (tid < md.rows) {
		const float sl = slope_all[(size_t)f * ROWS_MAX + tid];
		row_slope[tid] = sl;
		row_yint[tid] = yint_all[(size_t)f * ROWS_MAX + tid];
		float sn, cs;
		row_sincos(-sl, sn, cs);                                  // the rotation advances by this factor from a lane's column i to i + 1
		row_step[tid] = mk(cs, sn);
	}
	if (try_cert) {
		for (int w = tid; w < CODE_LEN / 32; w += 256)
			bits[w] = 0;
		for (int w = tid; w < MESG_BYTES_MAX / 4; w += 256)
			mesg32[w] = 0;
		ctab[tid] = tb.crc32_tab[tid];
	}
	__syncthreads();
	// ---- 1. decode.cc:493-494 + 505-523 (snr_rows) + the signs of the soft bits
	const int mod_bits = md.mod_bits, cols = md.cols;
	auto raw = [&](int j, int i) { return cons[j * cols + i]; };
#define BACK_WALK 1
	// the first pass walks a lane's (at most eight) consecutive columns of a row: one sin / cos for the first, then one complex
	// multiplication per column (the phasor's error grows by an ulp per step: seven steps) - a third of the rotation's instructions
	cf rot_cur = mk(1.f, 0.f), rot_step = mk(1.f, 0.f);
	uint32_t acc_lo = 0;                                          // the sign bits of the lane's points of the running row, from bit 0
	int acc_n = 0;
	auto begin_row = [&](int j, int i0) {
		if (BACK_WALK) {
			float sn, cs;
			row_sincos(-(row_yint[j] + row_slope[j] * (float)(i0 - cols / 2)), sn, cs);
			rot_cur = mk(cs, sn);
			rot_step = row_step[j];
		}
		acc_lo = 0;
		acc_n = 0;
	};
	auto rotated = [&](int j, int i, cf c0) {
		if (!BACK_WALK)
			return rotate_point(c0, row_slope[j], row_yint[j], i, cols);
		const cf c = cmul(c0, rot_cur);
		rot_cur = cmul(rot_cur, rot_step);
		return c;
	};
	bool odd = false;                                             // a zero / NaN LLR somewhere: no certificate
	const bool snr_ok = snr_rows(raw, begin_row, rotated, md.rows, cols, mod_bits, tid, rsum, prec, [&](int j, int i, cf c) {
		if (!try_cert)
			return;
		// This pass walks the row's phasor (a few ulps of drift), the LLRs the decoders see come from rotate_point: a soft bit within
		// that distance of zero could carry the other sign there.  Such a point is "odd" like a zero: the frame is not certified
		// and takes the general route, where its LLRs are the second pass's (round-4 advisor).  2e-6 |c| is about 16 ulps.
		const float are = fabsf(c.re), aim = fabsf(c.im), guard = 2e-6f * (are + aim);
		uint32_t v;
		if (mod_bits == 3) {
			v = (are < aim ? 1u : 0u) | (c.re < 0.f ? 2u : 0u) | (c.im < 0.f ? 4u : 0u);
			odd |= !(are > guard) | !(aim > guard) | !(fabsf(are - aim) > guard);
		} else {
			v = (c.re < 0.f ? 1u : 0u) | (c.im < 0.f ? 2u : 0u);
			odd |= !(are > guard) | !(aim > guard);
		}
		acc_lo |= v << acc_n;                                     // (at most 8 x 3 = 24 bits)
		acc_n += mod_bits;
	}, [&](int j, int i0) {
		// the lane's acc_n bits go to code positions mod_bits (j cols + i0) ...: one word, or two
		if (try_cert && acc_lo) {
			const int p0 = mod_bits * (j * cols + i0), o = p0 & 31;
			atomicOr(&bits[p0 >> 5], acc_lo << o);
			if (o + acc_n > 32)
				atomicOr(&bits[(p0 >> 5) + 1], acc_lo >> (32 - o));
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
			sum_slope += row_slope[j];
			sum_yint += row_yint[j];
		}
		r.sfo_slope = sum_slope / (float)md.rows;
		r.cfo_fine = st.cfo_rad + (sum_yint / (float)md.rows) / (float)sym_stride;   // decode.cc:501
		r.esn0_db_last = 10.f * log10f(prec[md.rows - 1]);    // decode.cc:518
	}
	int bad = 1;
	if (try_cert) {
		// ---- 3a. (before the transform overwrites the bit array) the systematic message = x at the unfrozen positions, decode.cc:254-261
		message_gather(bits, mesg32, tb.info_compress + (md.table ? 2048 * 8 : 0), tid);   // (mesg32 was zeroed with the bit array)
		// ---- 2. u = x F: at every level the left half of a block takes the XOR with the right half (the involution the partial-sum
		// combines of the decoder apply the other way round).  Word a = tid + 256 w: distances >= 256 words are inside the thread.
		uint32_t w[8];
		#pragma unroll
		for (int e = 0; e < 8; ++e) {
			uint32_t v = bits[tid + 256 * e];
			v ^= (v >> 1) & 0x55555555u;
			v ^= (v >> 2) & 0x33333333u;
			v ^= (v >> 4) & 0x0f0f0f0fu;
			v ^= (v >> 8) & 0x00ff00ffu;
			v ^= (v >> 16) & 0x0000ffffu;
			w[e] = v;
		}
		#pragma unroll
		for (int e = 0; e < 4; ++e) w[e] ^= w[e + 4];               // 1024 words
		w[0] ^= w[2]; w[1] ^= w[3]; w[4] ^= w[6]; w[5] ^= w[7];       // 512
		w[0] ^= w[1]; w[2] ^= w[3]; w[4] ^= w[5]; w[6] ^= w[7];       // 256
		for (int dw = 128; dw >= 1; dw >>= 1) {
			__syncthreads();
			#pragma unroll
			for (int e = 0; e < 8; ++e)
				bits[tid + 256 * e] = w[e];
			__syncthreads();
			if ((tid & dw) == 0) {
				#pragma unroll
				for (int e = 0; e < 8; ++e)
					w[e] ^= bits[tid + dw + 256 * e];
			}
		}
		const uint32_t *frozen = tb.frozen + (md.table ? CODE_LEN / 32 : 0);
		uint32_t syn = 0;
		#pragma unroll
		for (int e = 0; e < 8; ++e)
			syn |= w[e] & frozen[tid + 256 * e];
		bad = __syncthreads_or((syn != 0) | (odd ? 1 : 0));
			// ---- 3b. CRC<uint32_t>(0xD419CC15) over the first 43072 bits (decode.cc:533-541): crc32_wg256, dev_common.h
		if (!bad)                                                 // (uniform in the workgroup)
			bad = crc32_wg256(mesg, ctab, tb.crc32_adv, cpart, tid) != 0;
		if (!bad) {
			for (int i = tid; i < PAYLOAD_BYTES; i += 256)
				payload[i] = mesg[i] ^ (descramble ? tb.scramble[i] : (uint8_t)0);
			if (tid == 0) {
				r.best_lane = 0;
				res_all[f] = r;
				slot_of[f] = -1;
				atomicAdd(&q->tried, 1u);
				atomicAdd(&q->certified, 1u);
			}
			return;
		}
	}
	// ---- 4. a decoder has to look: a slot and the LLRs (decode.cc:520-529) - in the SC ring when that pass is on (all frames, or its
	// probe sample: k_sc.hip), else in the list decoder's queue
	// The list-1 pass decides a frame only while its path metric stays under min_fork: on AWGN that ends where the cumulative Es/N0
	// estimate of decode.cc:516 falls to 11 dB in the 8PSK modes (mode 6, -18.3 dB noise level: all of 4096 frames decided at 11.25 dB,
	// none at 10.7 dB; mode 10 decides at 10.9 dB and not at 10.5; the README's multipath chain already fails at 11.8 dB) and to 6.3 -
	// 6.5 dB in the QPSK modes (8, 9, 12: decided at -13.5 ... -14 dB noise level, not half a dB below; mode 13 further down): below
	// 10.4 / 5.8 dB a frame goes straight to the list decoder.  Above it the adaptive switch decides (k_sc_adapt): everything, or a
	// probe sample - one frame in sixteen of every fourth chunk.
	const bool sc_hope = prec[md.rows - 1] >= (md.mod_bits == 3 ? 10.96f : 3.8f);   // 10^(10.4 / 10), 10^(5.8 / 10)
	const bool to_sc = sc.q && sc_hope && (sc.q->cert_on || ((f & 15) == 8 && (sc.q->epoch & 3u) == 0u));   // (uniform in the workgroup)
	if (to_sc) {
		q = sc.q;
		slots = sc.slots;
		llr_q = sc.llr;
	}
	if (tid == 0) {
		const unsigned e = atomicAdd(&q->tail, 1u);
		const int slot = (int)(e % q->cap);
		ListSlot ls;
		ls.payload = payload_later ? payload_later + (size_t)f * PAYLOAD_BYTES : payload;   // where k_finish delivers (see launch_back)
		ls.res = (res_later ? res_later : res_all) + f;
		ls.payload_now = payload;
		ls.res_now = res_all + f;
		ls.oper_mode = st.oper_mode;
		ls.frame = f;
		ls.chunk = chunk_seq;
		ls.pad = 0;
		slots[slot] = ls;
		slot_of[f] = to_sc ? -2 - slot : slot;
		res_all[f] = r;                                       // k_sc_finish / k_finish completes the record (best_lane, bit_flips, status)
		if (try_cert)
			atomicAdd(&q_cert->tried, 1u);
		slot_sh = slot;
	}
	__syncthreads();
	float *llr = llr_q + (size_t)slot_sh * CODE_LEN;
	typedef float vf4 __attribute__((ext_vector_type(4)));
	const int lane = tid & 63;
	float *const stage = (float *)bits + (tid >> 6) * 192;        // (the bit array has done its work)
	cf nxt[2];
	auto fetch = [&](int j) {
		#pragma unroll
		for (int e = 0; e < 2; ++e)
			if (j < md.rows && tid + 256 * e < cols)
				nxt[e] = cons[j * cols + tid + 256 * e];
	};
	fetch(0);
	for (int j = 0; j < md.rows; ++j) {
		const float sc = DIST * prec[j];
		const cf cur[2] = { nxt[0], nxt[1] };
		fetch(j + 1);                                             // (the next row travels while this one is written)
		#pragma unroll
		for (int e = 0; e < 2; ++e) {
			const int i = tid + 256 * e;
			if (i < cols) {                                   // psk.hh:76-80,125-130
				const cf c = rotate_point(cur[e], row_slope[j], row_yint[j], i, cols);
				float *b = llr + mod_bits * (j * cols + i);
				if (mod_bits == 3) {
					stage[3 * lane + 1] = c.re * sc;
					stage[3 * lane + 2] = c.im * sc;
					stage[3 * lane + 0] = (rcp_sqrt_2 * (fabsf(c.re) - fabsf(c.im))) * sc;
				} else {
					b[0] = c.re * sc;
					b[1] = c.im * sc;
				}
			}
			if (mod_bits == 3) {
				// three soft bits per point: a lane's own stores would be 4 bytes every 12 (every store instruction a third of each line it touches:
				// 2.6 TB/s).  The wave's 64 points are 768 consecutive bytes: through LDS, written as 16-byte pieces
				const int i0 = i - lane, nfl = 3 * min(64, cols - i0);              // (wave-uniform; <= 0: this wave has no points here)
				__builtin_amdgcn_wave_barrier();
				if (nfl > 0) {
					float *dst = llr + 3 * (j * cols + i0);
					if (lane < (nfl >> 2))
						__builtin_nontemporal_store(((const vf4 *)stage)[lane], (vf4 *)dst + lane);
					else if (lane < (nfl >> 2) + (nfl & 3))
						dst[(nfl & ~3) + lane - (nfl >> 2)] = stage[(nfl & ~3) + lane - (nfl >> 2)];
				}
				__builtin_amdgcn_wave_barrier();
			}
		}
	}
	for (int i = md.cons_bits + tid; i < CODE_LEN; i += 256)   // lengthen(), decode.cc:252: the shortened positions are the index tail
		llr[i] = 9000.f;
}

// the rotated constellation of one frame (OFDMRX_TAP_CONS_ROT): what decode.cc:494 leaves in cons[]
__global__ __launch_bounds__(256) void k_rotate_tap(const SyncState *__restrict__ st, const cf *__restrict__ cons, const float *__restrict__ slope,
	const float *__restrict__ yint, cf *__restrict__ out)
{
	const ModeDesc md = mode_desc(st->oper_mode);
	for (int p = blockIdx.x * 256 + threadIdx.x; p < CONS_MAX; p += gridDim.x * 256) {
		const int j = p / md.cols, i = p % md.cols;
		out[p] = (st->okay && j < md.rows) ? rotate_point(cons[p], slope[j], yint[j], i, md.cols) : mk(0.f, 0.f);
	}
}
void launch_rotate_tap(hipStream_t s, const SyncState *st, const cf *cons, const float *slope, const float *yint, cf *out)
{
	hipLaunchKernelGGL(k_rotate_tap, dim3(32), dim3(256), 0, s, st, cons, slope, yint, out);
}

// ---------------------------------------------------------------- the list decoder's queue (kernels.h: ListQueue)
__global__ void k_queue_reset(ListQueue *__restrict__ q, unsigned cap)
{
	q->tail = q->head = 0;
	q->snap[0] = q->snap[1] = 0;
	q->run_head[0] = q->run_head[1] = q->run_n[0] = q->run_n[1] = 0;
	q->next_unit[0] = q->next_unit[1] = 0;
	q->cert_on = 1;
	q->tried = q->certified = 0;
	q->cap = cap;
	q->done_total = 0;
	q->epoch = 0;
	q->probe_tried = q->probe_done = 0;
}
// behind k_back of a chunk: what the flush of that chunk may take, and the adaptive certificate's next state
__global__ void k_queue_snap(ListQueue *__restrict__ q, int par)
{
	q->snap[par] = q->tail;
	const unsigned tried = q->tried, cert = q->certified;
	if (q->cert_on) {
		if (tried >= 64 && cert * 20 < tried) {
			q->cert_on = 0;
			q->probe_tried = q->probe_done = 0;
		}
	} else {
		// the probe sample (one frame in sixteen) of a small chunk is a handful of frames: summed over chunks until eight have been tried
		const unsigned pt = q->probe_tried + tried, pd = q->probe_done + cert;
		if (pt >= 8) {
			if (pd * 5 >= pt)
				q->cert_on = 1;
			q->probe_tried = q->probe_done = 0;
		} else {
			q->probe_tried = pt;
			q->probe_done = pd;
		}
	}
	q->tried = q->certified = 0;
}
// in front of k_polar: the entries of this flush.  Nothing until `unit` entries wait (one full residency of the list decoder),
// then whole multiples of it - the rest stays queued for the next flush; force: everything (the last flush of a call, and
// every flush of a call whose outputs leave chunk by chunk).
__global__ void k_queue_plan(ListQueue *__restrict__ q, int par, unsigned unit, int force)
{
	const unsigned head = q->head, n = q->snap[par] - head;
	const unsigned run = force ? n : (n / unit) * unit;
	q->run_head[par] = head;
	q->run_n[par] = run;
	q->head = head + run;
	q->next_unit[par] = 0;
}
// debug entries: n frames in slots 0 .. n-1 in order, all to be decoded
__global__ void k_queue_fill(ListQueue *__restrict__ q, ListSlot *__restrict__ slots, int n, uint8_t *payload, Result *res, int oper_mode)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		ListSlot ls;
		ls.payload = payload + (size_t)i * PAYLOAD_BYTES;
		ls.res = res + i;
		ls.payload_now = ls.payload;
		ls.res_now = ls.res;
		ls.oper_mode = oper_mode;
		ls.frame = i;
		ls.chunk = 0;
		ls.pad = 0;
		slots[i] = ls;
	}
	if (i == 0) {
		q->tail = (unsigned)n;
		q->snap[0] = (unsigned)n;
	}
}
void launch_queue_reset(hipStream_t s, ListQueue *q, unsigned cap) { hipLaunchKernelGGL(k_queue_reset, dim3(1), dim3(1), 0, s, q, cap); }
void launch_queue_snap(hipStream_t s, ListQueue *q, int par) { hipLaunchKernelGGL(k_queue_snap, dim3(1), dim3(1), 0, s, q, par); }
void launch_queue_plan(hipStream_t s, ListQueue *q, int par, unsigned unit, int force) { hipLaunchKernelGGL(k_queue_plan, dim3(1), dim3(1), 0, s, q, par, unit ? unit : 1u, force); }
void launch_queue_fill(hipStream_t s, ListQueue *q, ListSlot *slots, int n, uint8_t *payload, Result *res, int oper_mode)
{
	hipLaunchKernelGGL(k_queue_fill, dim3((n + 255) / 256), dim3(256), 0, s, q, slots, n, payload, res, oper_mode);
}

// ---------------------------------------------------------------- D10
// decode.cc:254-261 (systematic message = codeword at the unfrozen positions),
// decode.cc:532-541 (first lane whose CRC-32 over 43072 bits is 0), decode.cc:546-555
// (LE bit packing + flip count), decode.cc:613-615 (descramble).
// One workgroup per entry of the flush `par` of the list decoder's queue; workgroups beyond the run leave at once.
__global__ __launch_bounds__(256) void k_finish(const ListQueue *__restrict__ q, int par, const ListSlot *__restrict__ slots,
	const float *__restrict__ llr_q, const uint8_t *__restrict__ hard_q, Tables tb, int descramble, int list, uint8_t *__restrict__ lane_mesg_q)
{
	const int rel = blockIdx.x, tid = threadIdx.x;
	const unsigned run_n = q->run_n[par], run_head = q->run_head[par], cap = q->cap;
	if ((unsigned)rel >= run_n)
		return;
	const int slot = (int)((run_head + (unsigned)rel) % cap);
	const ListSlot ls = slots[slot];
	uint8_t *payload = ls.payload;
	__shared__ uint8_t mesg[LIST][MESG_BYTES_MAX];
	__shared__ uint32_t crcs[LIST];
	__shared__ int flips_red[4];
	__shared__ uint32_t ctab[256], csh[1024], cpart[LIST][32];
	ctab[tid] = tb.crc32_tab[tid];
	#pragma unroll
	for (int e = 0; e < 4; ++e)
		csh[tid + 256 * e] = tb.crc32_shift168[tid + 256 * e];
	// list 4: k_polar decodes the entries (2u, 2u+1) of a run as a pair when they share the frozen table; the pair's
	// partial sums are ONE byte array in the first entry's slot, bits 0..3 = first frame, bits 4..7 = second
	int hshift = 0, hslot = slot;
	if (list == 4) {
		const unsigned mate = (unsigned)rel ^ 1u;
		if (mate < run_n) {
			const int mslot = (int)((run_head + mate) % cap);
			const bool adjacent = (rel & 1) ? mslot + 1 == slot : slot + 1 == mslot;
			if (adjacent && (slots[mslot].oper_mode >= 10) == (ls.oper_mode >= 10)) {   // (k_polar's pairing rule)
				hslot = (rel & 1) ? mslot : slot;
				hshift = (rel & 1) * 4;
			}
		}
	}
	const uint8_t *hard = hard_q + (size_t)hslot * CODE_LEN;
	const float *llr = llr_q + (size_t)slot * CODE_LEN;
	const ModeDesc md = mode_desc(ls.oper_mode);
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
	if (lane_mesg_q)
		for (int i = tid; i < LIST * MESG_BYTES; i += 256)
			lane_mesg_q[(size_t)slot * LIST * MESG_BYTES + i] = mesg[i / MESG_BYTES][i % MESG_BYTES];
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
			for (int e = 0; e < NSEG; ++e) {
				crc = csh[crc & 255] ^ csh[256 + ((crc >> 8) & 255)] ^ csh[512 + ((crc >> 16) & 255)] ^ csh[768 + (crc >> 24)];
				crc ^= cpart[tid][e];
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
	if (best < 0) {
		for (int i = tid; i < PAYLOAD_BYTES; i += 256)
			payload[i] = 0;
		if (tid == 0) {
			ls.res->status = 6;                               // decode.cc:542-545
			ls.res->best_lane = -1;
		}
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
		ls.res->best_lane = best;
		ls.res->bit_flips = flips_red[0] + flips_red[1] + flips_red[2] + flips_red[3];
	}
}

void launch_finish(hipStream_t s, int list, int max_entries, const ListQueue *q, int par, const ListSlot *slots, const float *llr_q,
	const uint8_t *hard_q, Tables tb, int descramble, uint8_t *lane_mesg_q)
{
	hipLaunchKernelGGL(k_finish, dim3(max_entries), dim3(256), 0, s, q, par, slots, llr_q, hard_q, tb, descramble, list == 4 ? 4 : 8, lane_mesg_q);
}
void launch_back(hipStream_t s, int rate, int n, int cert_mode, const SyncState *st, const cf *cons, const float *slope, const float *yint,
	float *precision, Result *res, float *esn0_rows, Tables tb, int descramble, uint8_t *payload, ListQueue *q, ListSlot *slots,
	float *llr_q, int *slot_of, uint8_t *payload_later, Result *res_later, ScRing sc, int chunk_seq)
{
	const int sym_stride = rate_symbol_len(rate) + rate_symbol_len(rate) / 8;
	hipLaunchKernelGGL(k_back, dim3(n), dim3(256), 0, s, sym_stride, cert_mode, st, cons, slope, yint, precision, res, esn0_rows, tb, descramble,
		payload, q, slots, llr_q, slot_of, payload_later, res_later, sc, chunk_seq);
}

}  // namespace rx
