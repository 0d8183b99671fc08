// k_header.hip -- D3h: header symbol (decode.cc:403-447) for gfx950.
//   derotate 1280 samples at sc_start+1440, FFT1280, MLS1 descramble, differential
//   BPSK across 255 bins -> int8 soft values -> OrderedStatisticsDecoder<255,71,4>
//   -> 55-bit metadata + CRC-16 -> oper_mode / call sign checks.
// One 256-thread workgroup per frame.  The OSD keeps the permuted generator matrix
// bit-packed in LDS (71 rows x 8 words), reduces it with Gauss-Jordan (pivot search and
// column swaps exactly as the row-echelon + back-substitution of the restated osd.hh:
// the reduced form is unique for a given column order), then the 1 031 347 flip
// patterns of weight <= 4 are spread over the threads as (a,b) work items pulled from an
// LDS counter; candidate metrics come from a byte-sliced LDS lookup table.
#include "dev_common.h"
#include "kernels.h"
#include "mono_front.h"

namespace rx {

constexpr int NPAIRS = BCH_K * (BCH_K - 1) / 2;   // 2485
constexpr int NTRIPLES = BCH_K * (BCH_K - 1) * (BCH_K - 2) / 6;   // 57155

struct OsdShared {
	uint32_t G[BCH_K][9];      // 8 words + 1 pad: row stride 9 words spreads rows over the LDS banks
	uint32_t plane[8][8];      // plane[b][w]: bit i = bit b of the two's-complement byte of x[32w+i]
	short ax[BCH_K];           // |x| of the 71 most reliable (systematic) positions
	int S0;                    // sum of x over the set bits of the order-0 codeword's systematic part
	short x[256];
	short perm[256];
	unsigned char rel[256];
	signed char soft[256];
	uint32_t cw[8];
	uint32_t cw2[8];           // best codeword so far (permuted order)
	int pivot_row, pivot_col;
	int next_item;
	int red_best[256], red_next[256], red_id[256];
	int X;
	uint32_t gen[BCH_K * 8];   // the generator's bit rows, copied from global memory once (the permutation below reads 255 bits of every row)
};

// After Gauss-Jordan the permuted generator is [I | P]: a candidate differs from the order-0
// codeword in exactly its flipped systematic positions (each adds |x| to S) plus the XOR of
// the rows' parity parts.  S over the 184 parity positions (words 2..7, word 2 bits >= 7) is a
// weighted popcount over the 8 bit planes of x: 48 AND + 48 BCNT, registers only.
constexpr int PW = 6;   // parity words 2..7
struct Planes {
	uint32_t v[8][PW];
	__device__ __forceinline__ int eval(const uint32_t *e) const
	{
		int acc = 0;
		#pragma unroll
		for (int b = 0; b < 8; ++b) {
			int c = 0;
			#pragma unroll
			for (int w = 0; w < PW; ++w)
				c += __popc(e[w] & v[b][w]);
			acc += b == 7 ? -(c << 7) : (c << b);
		}
		return acc;
	}
};

struct Track {
	int best, next, id;
	__device__ __forceinline__ void update(int met, int cand)
	{
		if (met > best) { next = best; best = met; id = cand; }
		else if (met > next) { next = met; }
	}
};

// Exact optimality certificate.  For codewords c, c' : metric(c) - metric(c') = 2 * sum over the
// positions where they differ of w_i(c), w_i(c) = (1 - 2 c_i) x_i.  The generator's roots include
// alpha^1..alpha^58 (the 24 minimal polynomials of decode.cc:379-384), so by the BCH bound two distinct
// codewords differ in >= 59 positions.  If -(sum of |w_i| over the n_neg positions with w_i < 0)
// + (sum of the 59 - n_neg smallest non-negative w_i) > 0, every other codeword - hence every other
// OSD candidate - has a strictly smaller metric: c is the result and it is unique (best != next), exactly
// what the full enumeration of decode.cc:417 would return.  cwbits: 8 words in LDS (permuted order).
__device__ bool osd_certify(OsdShared &s, const uint32_t *cwbits, int tid)
{
	const int DMIN = 59;
	int w = 0;
	if (tid < BCH_N) {
		int c = (cwbits[tid >> 5] >> (tid & 31)) & 1;
		w = c ? -(int)s.x[tid] : (int)s.x[tid];
	}
	s.red_id[tid] = w;
	// sum of |w| over the negative positions and their count: packed (sum << 9 | count), one wave reduction + 4 LDS words
	int pk = w < 0 ? ((-w) << 9) | 1 : 0;
	#pragma unroll
	for (int m = 32; m; m >>= 1)
		pk += __shfl_xor(pk, m);
	if ((tid & 63) == 0)
		s.red_next[tid >> 6] = pk;
	__syncthreads();
	pk = s.red_next[0] + s.red_next[1] + s.red_next[2] + s.red_next[3];
	const int neg = pk >> 9, nneg = pk & 511;
	const int need = DMIN - nneg;
	bool ok = false;
	if (need > 0) {
		int take = 0;
		if (tid < BCH_N && w >= 0) {
			int r = 0;
			for (int i = 0; i < BCH_N; ++i) {
				int v = s.red_id[i];
				r += v >= 0 && (v < w || (v == w && i < tid));
			}
			if (r < need)
				take = w;
		}
		#pragma unroll
		for (int m = 32; m; m >>= 1)
			take += __shfl_xor(take, m);
		__syncthreads();
		if ((tid & 63) == 0)
			s.red_best[tid >> 6] = take;
		__syncthreads();
		ok = s.red_best[0] + s.red_best[1] + s.red_best[2] + s.red_best[3] > neg;
	}
	__syncthreads();
	return ok;
}

// soft[255] in s.soft must be valid; returns unique flag, writes hard bits (BE) to hard_out[32] (thread 0)
__device__ __forceinline__ bool osd_decode(OsdShared &s, const uint32_t *__restrict__ genmat_bits, const uint8_t *__restrict__ pairs,
	const uint8_t *__restrict__ triples,
	uint8_t *hard_out /* LDS or global, 32 B */, int tid)
{
	// ---- syndrome certificate (round 3).  The search below maximises the correlation sum_i x_i (1 - 2 c_i) over the candidate
	// codewords c (oracle/bch_osd.c; osd.hh).  Its upper bound sum |x_i| is reached exactly by the words that equal the hard
	// decisions h_i = [x_i < 0] wherever x_i != 0.  If h itself is a codeword it is the order-0 candidate of ANY information set
	// (a codeword is determined by its bits on one) and therefore in the list; any other codeword with the same correlation
	// differs from h only where x_i = 0, i.e. in at least d_min >= 59 such positions (the generator has the roots alpha^1 ..
	// alpha^58).  So with fewer zeros than that h is the unique best: "hard = h, unique = true" is what the search returns,
	// with no sort, no elimination and no candidate walked.  h is a codeword iff re-encoding its 71 systematic bits with the
	// systematic generator gives back all 255 bits.
	{
		uint32_t *hw = (uint32_t *)s.red_best;
		const bool neg = tid < BCH_N && s.soft[tid] < 0, zero = tid < BCH_N && s.soft[tid] == 0;
		const unsigned long long bn = __ballot(neg), bz = __ballot(zero);
		if ((tid & 63) == 0) {
			hw[2 * (tid >> 6)] = (uint32_t)bn;
			hw[2 * (tid >> 6) + 1] = (uint32_t)(bn >> 32);
			s.red_next[tid >> 6] = __popcll(bz);
		}
		__syncthreads();
		if (tid < 8) {
			uint32_t acc = 0;
			for (int j = 0; j < BCH_K; ++j)
				if ((hw[j >> 5] >> (j & 31)) & 1)
					acc ^= genmat_bits[j * 8 + tid];
			s.red_id[tid] = acc != hw[tid];
		}
		__syncthreads();
		const int zeros = s.red_next[0] + s.red_next[1] + s.red_next[2] + s.red_next[3];
		int bad = 0;
		#pragma unroll
		for (int w = 0; w < 8; ++w)
			bad |= s.red_id[w];
		if (!bad && zeros <= 16) {                            // (16 << 59: the margin costs nothing)
			if (tid < 32)                                         // big-endian bits: bit 7 - b of byte p = h[8 p + b]
				hard_out[tid] = (uint8_t)(__brev((hw[tid >> 2] >> (8 * (tid & 3))) & 255u) >> 24);
			__syncthreads();
			return true;
		}
		__syncthreads();
	}
	// reliabilities, stable descending sort by rank counting
	if (tid < 256)
		s.rel[tid] = tid < BCH_N ? (unsigned char)abs(max((int)s.soft[tid], -127)) : 0;
	__syncthreads();
	if (tid < BCH_N) {
		int r = 0, me = s.rel[tid];
		for (int j = 0; j < BCH_N; ++j) {
			int o = s.rel[j];
			r += (o > me) || (o == me && j < tid);
		}
		s.perm[r] = (short)tid;
	}
	if (tid == 255)
		s.perm[255] = 255;
	__syncthreads();
	// permuted generator: G[j] bit i = genmat[j][perm[i]]
	for (int it = tid; it < BCH_K * 8; it += 256)
		s.gen[it] = genmat_bits[it];
	__syncthreads();
	for (int it = tid; it < BCH_K * 8; it += 256) {
		int j = it >> 3, w = it & 7;
		uint32_t v = 0;
		for (int b = 0; b < 32; ++b) {
			int i = 32 * w + b;
			if (i < BCH_N) {
				int c = s.perm[i];
				v |= ((s.gen[j * 8 + (c >> 5)] >> (c & 31)) & 1u) << b;
			}
		}
		s.G[j][w] = v;
	}
	__syncthreads();
	// Gauss-Jordan with the pivoting rule of row_echelon(): pivot = the first row >= k with a one in column k (found
	// by all rows at once: LDS atomicMin), else the first later column that has a one in some row >= k (rare: serial)
	if (tid == 0)
		s.pivot_row = 255;
	__syncthreads();
	for (int k = 0; k < BCH_K; ++k) {
		if (tid >= k && tid < BCH_K && ((s.G[tid][k >> 5] >> (k & 31)) & 1))
			atomicMin(&s.pivot_row, tid);
		__syncthreads();
		int pr = s.pivot_row, pc = k;
		if (pr == 255) {                                      // same value in every thread
			__syncthreads();
			if (tid == 0) {
				int r2 = -1, c2 = k;
				for (int c = k + 1; r2 < 0 && c < BCH_N; ++c)
					for (int h = k; h < BCH_K; ++h)
						if ((s.G[h][c >> 5] >> (c & 31)) & 1) { r2 = h; c2 = c; break; }
				s.pivot_row = r2;
				s.pivot_col = c2;
			}
			__syncthreads();
			pr = s.pivot_row;
			pc = s.pivot_col;
		}
		if (pc != k) {   // column swap k <-> pc in every row, and in perm
			if (tid < BCH_K) {
				uint32_t bk = (s.G[tid][k >> 5] >> (k & 31)) & 1, bc = (s.G[tid][pc >> 5] >> (pc & 31)) & 1;
				if (bk != bc) {
					s.G[tid][k >> 5] ^= 1u << (k & 31);
					s.G[tid][pc >> 5] ^= 1u << (pc & 31);
				}
			}
			if (tid == 255) { short t = s.perm[k]; s.perm[k] = s.perm[pc]; s.perm[pc] = t; }
			__syncthreads();
		}
		if (pr != k && tid < 8) {   // row swap
			uint32_t t = s.G[k][tid]; s.G[k][tid] = s.G[pr][tid]; s.G[pr][tid] = t;
		}
		__syncthreads();
		if (tid == 0)
			s.pivot_row = 255;                                // for the next pivot search (read again only after two barriers)
		{   // clear column k in every other row: read phase, barrier, write phase
			uint32_t pv[3];
			bool hit[3];
			#pragma unroll
			for (int q = 0; q < 3; ++q) {
				int it = tid + 256 * q;
				hit[q] = false;
				pv[q] = 0;
				if (it < BCH_K * 8) {
					int j = it >> 3, w = it & 7;
					hit[q] = j != k && ((s.G[j][k >> 5] >> (k & 31)) & 1);
					pv[q] = s.G[k][w];
				}
			}
			__syncthreads();
			#pragma unroll
			for (int q = 0; q < 3; ++q) {
				int it = tid + 256 * q;
				if (it < BCH_K * 8 && hit[q])
					s.G[it >> 3][it & 7] ^= pv[q];
			}
		}
		__syncthreads();
	}
	// permuted soft values and the byte-sliced table
	if (tid < 256)
		s.x[tid] = tid < BCH_N ? (short)max((int)s.soft[s.perm[tid]], -127) : 0;
	__syncthreads();
	if (tid < 64) {   // bit planes of x (two's complement bytes)
		int b = tid >> 3, w = tid & 7;
		uint32_t v = 0;
		for (int i = 0; i < 32; ++i)
			v |= (uint32_t)(((int)s.x[32 * w + i] >> b) & 1) << i;
		if (w == 2)
			v &= ~((1u << (BCH_K - 64)) - 1);   // positions 64..70 are systematic
		s.plane[b][w] = v;
	}
	if (tid < BCH_K)
		s.ax[tid] = (short)abs((int)s.x[tid]);
	if (tid < 8) {   // order-0 codeword: hard decisions on the 71 most reliable positions
		uint32_t v = 0;
		for (int i = 0; i < BCH_K; ++i)
			if (s.x[i] < 0)
				v ^= s.G[i][tid];
		s.cw[tid] = v;
	}
	{   // X = sum of x over all positions, S0 = sum of x over the negative systematic positions
		int vx = tid < BCH_N ? (int)s.x[tid] : 0, v0 = (tid < BCH_K && vx < 0) ? vx : 0;
		#pragma unroll
		for (int m = 32; m; m >>= 1) {
			vx += __shfl_xor(vx, m);
			v0 += __shfl_xor(v0, m);
		}
		if ((tid & 63) == 0) {
			s.red_best[tid >> 6] = vx;
			s.red_next[tid >> 6] = v0;
		}
		if (tid == 0)
			s.next_item = 0;
	}
	__syncthreads();
	if (tid == 0) {
		s.X = s.red_best[0] + s.red_best[1] + s.red_best[2] + s.red_best[3];
		s.S0 = s.red_next[0] + s.red_next[1] + s.red_next[2] + s.red_next[3];
	}
	__syncthreads();
	const int X = s.X, S0 = s.S0;
	Planes pl;
	#pragma unroll
	for (int b = 0; b < 8; ++b)
		#pragma unroll
		for (int w = 0; w < PW; ++w)
			pl.v[b][w] = s.plane[b][w + 2];
	uint32_t base[PW];
	#pragma unroll
	for (int w = 0; w < PW; ++w)
		base[w] = s.cw[w + 2];
	Track tr;
	tr.best = X - 2 * (S0 + pl.eval(base));   // candidate id 0 = no flips
	tr.next = -1;
	tr.id = 0;
	if (tid != 0) { tr.best = -0x7fffffff; tr.next = -0x7fffffff; }
	// ids: singles 1+a ; pairs/triples/quads packed as (a+1) | (b+1)<<7 | (c+1)<<14 | (d+1)<<21
	if (tid < BCH_K) {
		uint32_t e[PW];
		#pragma unroll
		for (int w = 0; w < PW; ++w)
			e[w] = base[w] ^ s.G[tid][w + 2];
		tr.update(X - 2 * (S0 + s.ax[tid] + pl.eval(e)), tid + 1);
	}
	// pairs: one candidate each, strided over the threads
	for (int item = tid; item < NPAIRS; item += 256) {
		const int a = pairs[2 * item], b = pairs[2 * item + 1];
		uint32_t e[PW];
		#pragma unroll
		for (int w = 0; w < PW; ++w)
			e[w] = base[w] ^ s.G[a][w + 2] ^ s.G[b][w + 2];
		tr.update(X - 2 * (S0 + s.ax[a] + s.ax[b] + pl.eval(e)), (a + 1) | ((b + 1) << 7));
	}
	// reduce the per-thread (best, runner-up, id) triples; thread 0 rebuilds the best codeword into s.cw2
	// Reduce the per-thread (best, runner-up, id) triples: gb = the largest best (its id from the LOWEST thread that holds
	// it, as a serial scan would find it), gn = the largest value among all other bests and all runner-ups (gn = gb when
	// two threads hold gb); then rebuild the best codeword into s.cw2.  Wave shuffles + four LDS slots, no serial scan.
	auto reduce_tracks = [&]() {
		const int lane = tid & 63, wv = tid >> 6;
		int b = tr.best, id = tr.id, t = tid;
		#pragma unroll
		for (int m = 32; m; m >>= 1) {
			const int ob = __shfl_xor(b, m), oi = __shfl_xor(id, m), ot = __shfl_xor(t, m);
			if (ob > b || (ob == b && ot < t)) { b = ob; id = oi; t = ot; }
		}
		if (lane == 0) { s.red_best[wv] = b; s.red_id[wv] = id; s.red_next[wv] = t; }
		__syncthreads();
		int gb = s.red_best[0], gi = s.red_id[0], gt = s.red_next[0];
		#pragma unroll
		for (int q = 1; q < 4; ++q)
			if (s.red_best[q] > gb) { gb = s.red_best[q]; gi = s.red_id[q]; gt = s.red_next[q]; }
		__syncthreads();
		// runner-up: every other thread's best, every thread's runner-up; a second holder of gb makes the optimum ambiguous
		int n2 = tr.next;
		if (tid != gt && tr.best > n2)
			n2 = tr.best;
		#pragma unroll
		for (int m = 32; m; m >>= 1)
			n2 = max(n2, __shfl_xor(n2, m));
		if (lane == 0)
			s.red_next[wv] = n2;
		__syncthreads();
		const int gn = max(max(s.red_next[0], s.red_next[1]), max(s.red_next[2], s.red_next[3]));
		if (tid < 8) {
			uint32_t v = s.cw[tid];
			#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int r = (gi >> (7 * q)) & 127;
				if (r)
					v ^= s.G[r - 1][tid];
			}
			s.cw2[tid] = v;
		}
		if (tid == 0)
			s.pivot_row = (gb != gn);
		__syncthreads();
	};
	auto emit = [&](bool unique) {
		// un-permute the best codeword: bit i of cw2 is code position perm[i]; big-endian bits in 32 bytes, assembled with
		// LDS atomics on eight words (byte p >> 3 of the output = byte (p >> 3) & 3 of word p >> 5)
		uint32_t *hw = (uint32_t *)s.red_best;
		if (tid < 8)
			hw[tid] = 0;
		__syncthreads();
		if (tid < BCH_N && ((s.cw2[tid >> 5] >> (tid & 31)) & 1)) {
			const int p = s.perm[tid];
			atomicOr(&hw[p >> 5], (0x80u >> (p & 7)) << (8 * ((p >> 3) & 3)));
		}
		__syncthreads();
		if (tid < 32)
			hard_out[tid] = (uint8_t)(hw[tid >> 2] >> (8 * (tid & 3)));
		if (tid == 0)
			s.pivot_row = unique;
		__syncthreads();
		return s.pivot_row != 0;
	};
	// orders 0..2 are done (2557 candidates).  If their best is provably the unique optimum, stop here.
	reduce_tracks();
	if (osd_certify(s, s.cw2, tid))
		return emit(true);
	// triples (a<b<c) sorted by c: item = the triple itself plus its d-loop (d > c).  Consecutive
	// items have equal trip counts, so the 64 lanes of a wave stay converged.
	for (int item = tid; item < NTRIPLES; item += 256) {
		const int a = triples[3 * item], b = triples[3 * item + 1], c = triples[3 * item + 2];
		uint32_t eabc[PW], e[PW];
		#pragma unroll
		for (int w = 0; w < PW; ++w)
			eabc[w] = base[w] ^ s.G[a][w + 2] ^ s.G[b][w + 2] ^ s.G[c][w + 2];
		const int idabc = (a + 1) | ((b + 1) << 7) | ((c + 1) << 14);
		const int sabc = S0 + s.ax[a] + s.ax[b] + s.ax[c];
		tr.update(X - 2 * (sabc + pl.eval(eabc)), idabc);
		#pragma unroll 2
		for (int d = c + 1; d < BCH_K; ++d) {
			#pragma unroll
			for (int w = 0; w < PW; ++w)
				e[w] = eabc[w] ^ s.G[d][w + 2];
			tr.update(X - 2 * (sabc + s.ax[d] + pl.eval(e)), idabc | ((d + 1) << 21));
		}
	}
	reduce_tracks();
	return emit(s.pivot_row != 0);
}

__device__ __forceinline__ int be_bit(const uint8_t *b, int i) { return (b[i >> 3] >> (7 - (i & 7))) & 1; }

// CRC<uint16_t>(0xA8F4)(uint64_t): reflected, low byte first (decode.cc:428-429)
__device__ __forceinline__ unsigned crc16_u64(unsigned long long data)
{
	unsigned crc = 0;
	for (int i = 0; i < 64; ++i) {
		unsigned tmp = crc ^ (unsigned)((data >> i) & 1);
		crc = (crc >> 1) ^ ((tmp & 1) * 0xA8F4u);
	}
	return crc & 0xffffu;
}

#ifndef HDR_WAVES
#define HDR_WAVES 5        // waves per SIMD the register budget is set for: 96 VGPRs (a few spills) instead of 216 - the kernel is a chain of short
                           // barrier-separated steps, latency-bound: 1.48 ms per 8192 frames at 2, 0.88 at 4, 0.79 at 5; the uncertified order-3
                           // search (OSD_NO_CERTIFICATE: 30 ms) does not care
#endif
template <int RATE, bool MONO>
__global__ __launch_bounds__(256, HDR_WAVES) void k_header(FrameBatch fb, cf *__restrict__ z_all, MonoArgs ma, Tables tb,
	SyncState *__restrict__ st_all, int8_t *__restrict__ hdr_soft, Attempt *__restrict__ attempts, int32_t *__restrict__ attempt_counts)
{
	constexpr int SYMBOL_LEN = RateCfg<RATE>::SL, SYM_STRIDE = RateCfg<RATE>::STRIDE;
	const int f = blockIdx.x, tid = threadIdx.x;
	__shared__ OsdShared s;
	__shared__ cf buf[SYMBOL_LEN];
	__shared__ uint8_t hard[32];
	SyncState st = st_all[f];
	if (!st.active)
		return;
	if (!st.found) {
		if (tid == 0) {
			st.status = st.status ? st.status : 1;   // NO_SYNC (decode.cc:393-394)
			st.okay = 0;
			st.active = 0;
			st_all[f] = st;
		}
		return;
	}
	SampleSrc src{ (const char *)fb.samples + (size_t)f * fb.frame_stride_bytes, fb.fmt, fb.channels, fb.samples_per_frame,
		fb.channels == 1 ? z_all + (size_t)f * fb.samples_per_frame : nullptr };
	const long body = st.sc_start + SYM_STRIDE;               // decode.cc:405
	if constexpr (MONO) {                                     // the symbol's analytic signal first (mono_front.h; its LDS is buf's)
		typedef MonoCover<RATE, 256> MCov;
		static_assert(sizeof(typename MCov::Shared) <= sizeof(buf), "the cover's staging area lives in the symbol buffer");
		MCov mc;
		mc.init(mono_frame(fb, ma.ck, ma.ck_per_frame, f), ma, reinterpret_cast<typename MCov::Shared *>(buf), z_all + (size_t)f * fb.samples_per_frame, tid);
		mc.cover(ma, body, body + SYMBOL_LEN, tid);
	}
	for (int i = tid; i < SYMBOL_LEN; i += 256)
		buf[i] = cmul(src.at(body + i), phasor(-st.cfo_rad, i));
	__syncthreads();
	fft_fwd<SYMBOL_LEN, 256, SYMBOL_LEN>(buf, tb.tw_sym, tid);
	if (tid < MLS1_LEN) {                                     // decode.cc:407-416
		const int mls1_off = -MLS1_LEN / 2;
		int b1 = (tid + mls1_off + SYMBOL_LEN) % SYMBOL_LEN, b0 = (tid - 1 + mls1_off + SYMBOL_LEN) % SYMBOL_LEN;
		cf cur = buf[b1], prv = buf[b0];
		float s1 = tb.mls1_nrz[tid], s0 = tid ? tb.mls1_nrz[tid - 1] : 1.f;
		cur = mk(cur.re * s1, cur.im * s1);
		prv = mk(prv.re * s0, prv.im * s0);
		float v = nearbyintf(127.f * demod_or_erase(cur, prv).re);
		v = fminf(fmaxf(v, -128.f), 127.f);
		s.soft[tid] = (signed char)v;
		hdr_soft[(size_t)f * 256 + tid] = (int8_t)v;
	}
	if (tid == 255)
		s.soft[255] = 0;
	__syncthreads();
	bool unique = osd_decode(s, tb.genmat_bits, tb.osd_pairs, tb.osd_triples, hard, tid);
	if (tid == 0) {
		int status = 0;
		unsigned long long md = 0;
		if (!unique) {
			status = 2;                                       // decode.cc:418-421
		} else {
			for (int i = 0; i < 55; ++i)
				md |= (unsigned long long)be_bit(hard, i) << i;
			unsigned cs = 0;
			for (int i = 0; i < 16; ++i)
				cs |= (unsigned)be_bit(hard, i + 55) << i;
			if (crc16_u64(md << 9) != cs) {
				status = 3;                                   // decode.cc:429-432
			} else {
				st.oper_mode = (int)(md & 255);
				st.call_sign = md >> 8;
				if (st.oper_mode < 6 || st.oper_mode > 13)
					status = 4;                               // decode.cc:434-437
				else if ((md >> 8) == 0 || (md >> 8) >= 129961739795077ULL)
					status = 5;                               // decode.cc:439-442
			}
		}
		st.status = status;
		st.okay = status == 0;
		if (attempts && st.hdr_rounds < ATTEMPTS_MAX) {           // decode.cc:400-401,417-446: what the reference prints for this preamble
			Attempt a;
			a.status = status;
			a.symbol_pos = st.symbol_pos;
			a.cfo_rad = st.cfo_rad;
			a.oper_mode = st.oper_mode;
			a.call_sign = st.call_sign;
			attempts[(size_t)f * ATTEMPTS_MAX + st.hdr_rounds] = a;
			attempt_counts[f] = st.hdr_rounds + 1;
		}
		st.hdr_rounds += 1;
		if (st.skip_left > 0) { st.skip_left -= 1; st.active = 1; st.found = 0; }   // decode.cc:448: the search goes on behind this preamble
		else st.active = 0;
		st_all[f] = st;
	}
}

// parity-test entry: OSD alone on caller-provided soft values
__global__ __launch_bounds__(256, 2) void k_osd_only(Tables tb, const int8_t *__restrict__ soft, uint8_t *__restrict__ hard_out,
	int32_t *__restrict__ unique_out)
{
	const int f = blockIdx.x, tid = threadIdx.x;
	__shared__ OsdShared s;
	__shared__ uint8_t hard[32];
	if (tid < 256)
		s.soft[tid] = tid < BCH_N ? soft[(size_t)f * BCH_N + tid] : 0;
	__syncthreads();
	bool u = osd_decode(s, tb.genmat_bits, tb.osd_pairs, tb.osd_triples, hard, tid);
	if (tid < 32)
		hard_out[(size_t)f * 32 + tid] = hard[tid];
	if (tid == 0)
		unique_out[f] = u ? 1 : 0;
}

void launch_header(hipStream_t s, int rate, int n, FrameBatch fb, cf *z, const MonoArgs &ma, Tables tb, SyncState *st, int8_t *hdr_soft,
	Attempt *attempts, int32_t *attempt_counts)
{
	Attempt *att = attempt_counts ? attempts : nullptr;       // (both or none)
	if (fb.channels == 1 && mono_fused(rate)) {
		hipLaunchKernelGGL((k_header<8000, true>), dim3(n), dim3(256), 0, s, fb, z, ma, tb, st, hdr_soft, att, attempt_counts);
	} else {
		RX_RATE_SWITCH(rate, hipLaunchKernelGGL((k_header<RATE, false>), dim3(n), dim3(256), 0, s, fb, z, ma, tb, st, hdr_soft, att, attempt_counts));
	}
}
void launch_osd_only(hipStream_t s, int n, Tables tb, const int8_t *soft, uint8_t *hard, int32_t *unique)
{
	hipLaunchKernelGGL(k_osd_only, dim3(n), dim3(256), 0, s, tb, soft, hard, unique);
}

}  // namespace rx
