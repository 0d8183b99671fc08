// k_theilsen.hip -- D5 (Theil-Sen phase-slope correction, decode.cc:479-504) for gfx950.
// Split from k_demod.hip so that the two can be compiled with their own flags (the demodulator's transforms are faster
// without SLP-packed fp32, this kernel is tuned instruction by instruction with the default settings).
#include "dev_common.h"
#include "kernels.h"

namespace rx {

// ---------------------------------------------------------------- D5 Theil-Sen
// DSP::TheilSenEstimator<value,512>::compute (decode.cc:488): median (rank count/2) of all
// pairwise slopes, then median of the intercepts.  Exact selection by a 3-pass radix select
// (11+11+10 bits of the order-preserving key) with the slopes recomputed on the fly from the
// 432 phases held in LDS; fp32 division is correctly rounded so slopes are bit-identical to
// the CPU's.
#ifndef TS_LIST_CAP_V
#define TS_LIST_CAP_V 4096   // 16 KB: with hist/part the block needs ~29 KB of LDS -> 5 workgroups per CU
#endif
// 1 / (k + 1), k = 0..639, correctly rounded at compile time (index d - 1 for distance d)
struct TsRcpTab {
	float v[640];
	constexpr TsRcpTab() : v() { for (int k = 0; k < 640; ++k) v[k] = 1.0f / (float)(k + 1); }
};
__constant__ TsRcpTab TS_RCP;

struct TsShared {
	float y[640];              // [n, n + 128) = +3e38 (pairs that do not exist sort above everything)
	float buf[TS_LIST_CAP_V];    // list of bracketed pairs -> their exact slopes; then the intercepts
	int hist[2048];
	int part[256];
	int red[4];
	unsigned prefix;
	int rank;
	int list_n;
	float pick;
	int bin[2];                // linear-histogram bracket: bins holding the two wanted ranks
};

__device__ __forceinline__ unsigned fkey(float v)
{
	unsigned b = __float_as_uint(v);
	return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k)
{
	unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
	return __uint_as_float(b);
}

template <typename F>
__device__ __forceinline__ void for_each_pair(int n, int tid, F fn)
{
	// fold distance q with n-q so every fold has n items (i, d): all i<j pairs exactly once.
	// Every thread runs the same trip counts (fn gets a `valid` flag) so wave-level ballots inside
	// fn see all 64 lanes.
	const int folds = (n - 1) / 2, n_up = (n + 255) & ~255;
	for (int q = 1; q <= folds; ++q)
		for (int e = tid; e < n_up; e += 256) {
			int i, d;
			if (e < n - q) { i = e; d = q; }
			else { i = e - (n - q); d = n - q; }
			const bool valid = e < n;
			fn(valid ? i : 0, valid ? d : 1, valid);
		}
	if ((n & 1) == 0) {
		const int d = n / 2;
		for (int e = tid; e < n_up; e += 256) {
			const bool valid = e < n - d;
			fn(valid ? e : 0, d, valid);
		}
	}
}

// one radix digit of a rank selection: histogram `bits` bits at `shift` of the keys produced by
// `each` that match the prefix found so far, then narrow (prefix, rank) to the digit's bin
template <typename Each>
__device__ void radix_digit(TsShared &s, int tid, int shift, int bits, unsigned mask_hi, Each each)
{
	for (int i = tid; i < 2048; i += 256)
		s.hist[i] = 0;
	__syncthreads();
	const unsigned prefix = s.prefix, bmask = (1u << bits) - 1;
	each([&](float v) {
		unsigned k = fkey(v);
		if ((k & mask_hi) == prefix)
			atomicAdd(&s.hist[(k >> shift) & bmask], 1);
	});
	__syncthreads();
	const int nb = 1 << bits, per = nb / 256;
	int acc = 0;
	for (int q = 0; q < per; ++q)
		acc += s.hist[tid * per + q];
	// block-wide exclusive scan of the 256 partial counts; the thread whose span holds the rank
	// walks its own <= 8 bins (no serial scan over the histogram)
	const int lane = tid & 63, wave = tid >> 6;
	int incl = acc;
	#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		int o = __shfl_up(incl, d);
		if (lane >= d)
			incl += o;
	}
	if (lane == 63)
		s.red[wave] = incl;
	const int r = s.rank;
	__syncthreads();
	int off = 0;
	for (int w = 0; w < wave; ++w)
		off += s.red[w];
	const int excl = incl - acc + off;
	if (r >= excl && r < excl + acc) {
		int rr = r - excl, b = tid * per;
		while (rr >= s.hist[b]) { rr -= s.hist[b]; ++b; }
		s.rank = rr;
		s.prefix = prefix | ((unsigned)b << shift);
	}
	__syncthreads();
}
// bins of s.hist[0..2048) that hold sorted positions r0 and r1 (s.bin[0], s.bin[1]; -1 if beyond the total)
__device__ void hist_locate2(TsShared &s, int tid, int r0, int r1)
{
	int acc = 0;
	#pragma unroll
	for (int q = 0; q < 8; ++q)
		acc += s.hist[tid * 8 + q];
	const int lane = tid & 63, wave = tid >> 6;
	int incl = acc;
	#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		int o = __shfl_up(incl, d);
		if (lane >= d)
			incl += o;
	}
	if (lane == 63)
		s.red[wave] = incl;
	if (tid < 2)
		s.bin[tid] = -1;
	__syncthreads();
	int off = 0;
	for (int w = 0; w < wave; ++w)
		off += s.red[w];
	const int excl = incl - acc + off;
	#pragma unroll
	for (int which = 0; which < 2; ++which) {
		const int r = which ? r1 : r0;
		if (r >= excl && r < excl + acc) {
			int rr = r - excl, b = tid * 8;
			while (rr >= s.hist[b]) { rr -= s.hist[b]; ++b; }
			s.bin[which] = b;
		}
	}
	__syncthreads();
}

// value at sorted position `rank` of the multiset enumerated by `each` (exact, 3 digits 11+11+10).
// EDGE = -1 / +1 stops after two digits and returns the lower / upper edge of the 22-bit key cell that
// holds the rank (a value <= / >= the order statistic, within 2^-13 relative): enough for a bracket.
template <int EDGE = 0, typename Each>
__device__ float select_rank(TsShared &s, int tid, int rank, Each each)
{
	if (tid == 0) { s.prefix = 0; s.rank = rank; }
	__syncthreads();
	radix_digit(s, tid, 21, 11, 0u, each);
	radix_digit(s, tid, 10, 11, 0xffe00000u, each);
	if (EDGE == 0)
		radix_digit(s, tid, 0, 10, 0xfffffc00u, each);
	unsigned key = s.prefix;
	if (EDGE > 0)
		key |= 0x3ffu;
	float v = fkey_inv(key);
	__syncthreads();
	return v;
}

// Exact order statistic by ONE linear histogram over [lo, hi] (values outside land in the two end bins) and a resolve among
// the members of the bin that holds the rank.  The binning is monotone in v - fp32 subtraction, multiplication by a
// positive constant and truncation all are - so every member of a lower bin is <= every member of a higher one and
// the wanted value is the (rank - count of the lower bins)-th smallest of its bin.  One histogram pass instead of the
// three digit passes of select_rank when the values are known to sit in a narrow range (the bracketed slopes).
// Returns false - nothing decided, the caller uses select_rank - if that bin holds more than 256 values (ties).
template <typename Each>
__device__ bool select_rank_linear(TsShared &s, int tid, int rank, float lo, float hi, Each each, float &out)
{
	const float inv = 2046.f / (hi - lo);
	if (!(hi > lo) || !(inv < 3.0e38f))
		return false;                                         // block-uniform
	auto bin = [&](float v) {
		const float q = (v - lo) * inv;
		return q < 0.f ? 0 : (q >= 2046.f ? 2047 : 1 + (int)q);
	};
	for (int i = tid; i < 2048; i += 256)
		s.hist[i] = 0;
	if (tid == 0)
		s.list_n = 0;
	__syncthreads();
	each([&](float v) { atomicAdd(&s.hist[bin(v)], 1); });
	__syncthreads();
	{   // the bin that holds sorted position `rank`, and the position inside it
		int acc = 0;
		#pragma unroll
		for (int q = 0; q < 8; ++q)
			acc += s.hist[tid * 8 + q];
		const int lane = tid & 63, wave = tid >> 6;
		int incl = acc;
		#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			int o = __shfl_up(incl, d);
			if (lane >= d)
				incl += o;
		}
		if (lane == 63)
			s.red[wave] = incl;
		if (tid == 0)
			s.bin[0] = -1;
		__syncthreads();
		int off = 0;
		for (int w = 0; w < wave; ++w)
			off += s.red[w];
		const int excl = incl - acc + off;
		if (rank >= excl && rank < excl + acc) {
			int rr = rank - excl, b = tid * 8;
			while (rr >= s.hist[b]) { rr -= s.hist[b]; ++b; }
			s.bin[0] = b;
			s.rank = rr;
		}
		__syncthreads();
	}
	const int B = s.bin[0], rr = s.rank;
	if (B < 0)
		return false;
	float *mem = (float *)s.part;
	each([&](float v) {
		if (bin(v) == B) {
			const int idx = atomicAdd(&s.list_n, 1);
			if (idx < 256)
				mem[idx] = v;
		}
	});
	__syncthreads();
	const int cnt = s.list_n;
	if (cnt > 256) {
		__syncthreads();
		return false;
	}
	if (tid < cnt) {
		const float mine = mem[tid];
		int pos = 0;
		for (int m = 0; m < cnt; ++m) {
			const float o = mem[m];
			pos += (o < mine) | ((o == mine) & (m < tid));
		}
		if (pos == rr)
			s.pick = mine;
	}
	__syncthreads();
	out = s.pick;
	__syncthreads();
	return true;
}

// y - s*x with the product rounded on its own (HIP's __fmul_rn is a plain '*' and would be
// contracted into an FMA): keeps the intercepts bit-identical to the CPU's
__device__ __forceinline__ float sub_mul_nofma(float y, float s, float x)
{
	#pragma clang fp contract(off)
	float p = s * x;
	return y - p;
}


constexpr int TS_LIST_CAP = TS_LIST_CAP_V;
constexpr int TS_GRID_ROWS = 50;

// y[0..n) in s.y ; x[i] = i - n/2.  Returns slope and yint in all threads.
// Exact median of the n(n-1)/2 pairwise slopes (decode.cc:488, rank count/2):
//  1. a 1/8 sample of the pairs (every distance that is a multiple of 8, cheap a*rcp(d) slopes, never
//     stored: recomputed by each radix pass) brackets the median: [T_lo, T_hi] = the sample order
//     statistics 3.2 sigma of Binomial(m, 1/2) either side of the sample median;
//  2. ONE pass over all pairs with the cheap slope a*rcp(d) (relative error < 3*2^-24 of the
//     correctly rounded quotient): pairs certainly below T_lo are counted, pairs certainly
//     above T_hi are dropped, the rest (~2-3%) are appended to an LDS list as (i, d);
//  3. the listed pairs get the exact fp32 division and the wanted order statistic is selected exactly
//     inside the list.  The result is accepted only if it lies in [T_lo, T_hi] (then its rank is
//     provably exact); otherwise, or when the list overflows (ties), the exact 3-digit radix select
//     over all pairs with exact divisions runs instead.
__device__ void theil_sen_block(TsShared &s, int n, int tid, float &slope, float &yint)
{
	const int count = n * (n - 1) / 2, target = count / 2;
	if (tid < 128)
		s.y[n + tid] = 3.0e38f;                               // padding: see the classification pass
	__syncthreads();
	auto all_pairs_exact = [&](auto emit) {
		for_each_pair(n, tid, [&](int i, int d, bool valid) { if (valid) emit((s.y[i + d] - s.y[i]) / (float)d); });
	};
	bool done = false;
	// ---- 1. sample
	const int TS_SAMPLE_STEP = 8;         // sample = all pairs whose distance is a multiple of this
	int m = 0;
	for (int d = TS_SAMPLE_STEP; d < n; d += TS_SAMPLE_STEP)
		m += n - d;
	if (m >= 512) {
		// the sample is never stored: its slopes are recomputed (cheap reciprocal form - a bracket needs
		// no exactness) in each of the four histogram passes
		auto sample = [&](auto emit) {
			for (int d = TS_SAMPLE_STEP; d < n; d += TS_SAMPLE_STEP) {
				const float rd = __builtin_amdgcn_rcpf((float)d);
				for (int i = tid; i < n - d; i += 256)
					emit((s.y[i + d] - s.y[i]) * rd);
			}
		};
		int K = (int)(1.6f * sqrtf((float)m)) + 2;            // 3.2 sigma of Binomial(m, 1/2)
		int rlo = m / 2 - K, rhi = m / 2 + K;
		if (rlo < 0) rlo = 0;
		if (rhi > m - 1) rhi = m - 1;
		// The bracket comes from ONE pass over the sample: a 2048-bin histogram that is linear in the slope,
		// centred on the mean of a small pre-sample (distances that are multiples of 64) and six mean absolute
		// deviations wide, so the few per cent of the sample around its median spread over tens of bins and
		// LDS atomics rarely collide.  T_lo / T_hi = outer edges of the bins holding the two ranks.  A bracket
		// is only a hint - the result is validated below - so rounding in the binning is harmless; if a rank
		// falls off the histogram the bracket comes from the radix select instead.
		float T_lo, T_hi;
		bool have = false;
#ifndef TS_NO_LINEAR_HIST
		{
			const int lane = tid & 63, wave = tid >> 6;
			// block sums through alternating LDS slots (red / bin-pair scratch in part[]): the slot of sum k is not written
			// again before the barrier of sum k+1, so one barrier per sum is enough
			int flip = 0;
			auto block_sum = [&](float v) {
				#pragma unroll
				for (int mm = 32; mm; mm >>= 1)
					v += __shfl_xor(v, mm);
				int *slot = flip ? s.part : s.red;
				flip ^= 1;
				if (lane == 0)
					slot[wave] = __float_as_int(v);
				__syncthreads();
				return (__int_as_float(slot[0]) + __int_as_float(slot[1])) + (__int_as_float(slot[2]) + __int_as_float(slot[3]));
			};
			auto pre = [&](auto emit) {
				for (int d = 64; d < n; d += 64) {
					const float rd = __builtin_amdgcn_rcpf((float)d);
					for (int i = tid; i < n - d; i += 256)
						emit((s.y[i + d] - s.y[i]) * rd);
				}
			};
			int cnti = 0;
			for (int d = 64; d < n; d += 64)
				cnti += n - d;
			const float cnt = (float)cnti;
			float a0 = 0.f;
			pre([&](float q) { a0 += q; });
			const float c = block_sum(a0) / cnt;
			float a1 = 0.f;
			pre([&](float q) { a1 += fabsf(q - c); });
			const float dev = block_sum(a1) / cnt;
			const float W = 6.f * dev, lo = c - W, inv = 1024.f / W, wbin = W * (1.f / 1024.f);
			if (W > 0.f && inv < 3.0e38f) {
				for (int i = tid; i < 2048; i += 256)
					s.hist[i] = 0;
				__syncthreads();
				sample([&](float q) {
					int b = (int)((q - lo) * inv);
					b = b < 0 ? 0 : (b > 2047 ? 2047 : b);
					atomicAdd(&s.hist[b], 1);
				});
				__syncthreads();
				hist_locate2(s, tid, rlo, rhi);
				const int b0 = s.bin[0], b1 = s.bin[1];
				if (b0 > 0 && b1 >= b0 && b1 < 2047) {
					T_lo = lo + (float)b0 * wbin;
					T_hi = lo + (float)(b1 + 1) * wbin;
					have = true;
				}
				__syncthreads();
			}
		}
#endif
		if (!have) {
			T_lo = select_rank<-1>(s, tid, rlo, sample);
			T_hi = select_rank<+1>(s, tid, rhi, sample);
		}
		// ---- 2. classify every pair: lane = point i, loop = distance d (wave-uniform), sixteen distances per step.
		// x is the integer grid (decode.cc:485), so "slope < T" is "y_j - T x_j < y_i - T x_i": the row is transformed ONCE
		// per bound, zl = y - T_lo' x and zh = y - T_hi' x, and a pair costs two compares of the neighbour's (zl, zh)
		// against the lane's own two values - no subtraction, no reciprocal table, no multiply per pair.  The bounds are
		// moved outwards by 1e-6 relative (T_lo', T_hi') and the lane's values by four times the worst rounding error of
		// the transform, so "below" implies that the correctly rounded slope is < T_lo and "above" that it is > T_hi;
		// everything else is listed and gets the exact division.  The (zl, zh) pairs sit in the histogram's LDS (idle
		// during this pass) and are padded with +3e38 beyond n, points i >= n compare against -3e38: pairs that do not
		// exist drop out as "above" - no index clamps, no validity masks.  "below" is a scalar popcount of the compare
		// mask; kept pairs go to a wave-private quarter of the LDS list (fill count in a scalar register, slot = fill +
		// mbcnt) only when the keep mask of the wave instruction is not empty.  Every wave walks all blocks of 64 points
		// and takes every fourth group of sixteen distances: the trip counts balance, and so do the kept pairs.
		const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: uniform loops
		constexpr int WCAP = TS_LIST_CAP / 4;
		int below = 0, cfill = 0;
		unsigned *list = (unsigned *)s.buf + wave * WCAP;
		const float T_lo_m = T_lo - (1e-6f * fabsf(T_lo) + 1e-36f);
		const float T_hi_m = T_hi + (1e-6f * fabsf(T_hi) + 1e-36f);
		// |computed z - (y - T x)| <= 2^-24 (|T x| + |y - T x|) <= 2^-24 (pi + 2 * 216 |T|) for |y| <= pi, |x| <= 216 (cols <= 512:
		// 256); two values are compared, and the margin doubles that again
		const float tmax = fmaxf(fabsf(T_lo_m), fabsf(T_hi_m));
		const float margin = 4.f * 5.97e-8f * (3.1416f + 1024.f * tmax) + 1e-37f;
		float2 *z2 = (float2 *)s.hist;
		{
			const int xoff = n / 2;
			for (int i = tid; i < n + 128; i += 256) {
				float2 v = make_float2(3.0e38f, 3.0e38f);
				if (i < n) {
					const float x = (float)(i - xoff), yv = s.y[i];
					v = make_float2(yv - T_lo_m * x, yv - T_hi_m * x);
				}
				z2[i] = v;
			}
		}
		__syncthreads();
#ifndef TS_PROBE_SKIP_MAIN
		{
			// Per pair: two subtractions and two v_alignbit - the SIGN of (zl_j - thr_lo) is "below", the sign of
			// (thr_hi - zh_j) is "above", and acc = (acc << 1) | sign shifts each into a per-lane 32-bit history, so a
			// window of 32 distances costs no compare, no scalar instruction and no branch (on this chip a VALU
			// instruction costs a SIMD 3.3 cycles, a compare into an SGPR 5.3 and every scalar instruction 4.8 -
			// tools/ubench_issue.hip).  Per window: below += popcount(history) per lane (one v_bcnt), the pairs that are
			// neither below nor above (2-3 %) are peeled off bit by bit into the wave's list.
			const int nblk = (n + 63) >> 6;
			int cbl = 0;                                          // this lane's "below" count
			constexpr int W = 32;                                 // distances per window = bits of the history registers
			for (int b = 0; b < nblk; ++b) {
				const int i = b * 64 + lane;
				const float2 zi = z2[i];
				const float thr_lo = i < n ? zi.x - margin : -3.0e38f;
				const float thr_hi = i < n ? zi.y + margin : -3.0e38f;
				const int dmax = n - 1 - b * 64;              // largest distance with any existing pair
				const float2 *zp = z2 + i + 1;
				for (int d0 = wave * W; d0 < dmax; d0 += 4 * W) {   // distances d0+1 .. d0+32; wave w takes every 4th window
					uint32_t accB = 0, accA = 0;
#ifndef TS_S8_UNROLL
#define TS_S8_UNROLL 1
#endif
					#pragma unroll TS_S8_UNROLL
					for (int s8 = 0; s8 < W; s8 += 8) {
						float2 zv[8];
						#pragma unroll
						for (int u = 0; u < 8; ++u)
							zv[u] = zp[d0 + s8 + u];
						__builtin_amdgcn_sched_barrier(0);
						#pragma unroll
						for (int u = 0; u < 8; ++u) {
							// (plain v_sub_f32 by hand: left to itself the compiler packs the two subtractions into one v_pk_add_f32
							// plus two register moves, 13 SIMD cycles instead of 6.6)
							float tb, ta;
#ifdef TS_PLAIN_SUB
							tb = zv[u].x - thr_lo;                       // needs -fno-slp-vectorize, or the two become one v_pk_add_f32 + moves
							ta = thr_hi - zv[u].y;
#else
							asm("v_sub_f32_e32 %0, %1, %2" : "=v"(tb) : "v"(zv[u].x), "v"(thr_lo));
							asm("v_sub_f32_e32 %0, %1, %2" : "=v"(ta) : "v"(thr_hi), "v"(zv[u].y));
#endif
							accB = __builtin_amdgcn_alignbit(accB, __float_as_uint(tb), 31);   // (accB << 1) | sign(tb)
							accA = __builtin_amdgcn_alignbit(accA, __float_as_uint(ta), 31);
						}
					}
					cbl += __popc(accB);
					uint32_t in = ~(accB | accA);                 // bit 31-u: the pair at distance d0+u+1 needs the exact division
#ifdef TS_PEEL_SCAN
					// (variant, measured neutral: -12 % scalar, +1 % vector instructions) Slots of the listed pairs: ONE wave scan of the per-lane counts per window (six DPP adds), then every lane
					// peels its own bits into its own run of slots - the loop body is clz / clear / pack / store, no ballot, no
					// mbcnt, no scalar population count per round (the order of the list does not matter: it is selected by rank)
					{
						const int c = __popc(in);
						int inc = c;
						inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, false);   // row_shr:1
						inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, false);   // row_shr:2
						inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, false);   // row_shr:4
						inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, false);   // row_shr:8
						inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
						inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
						int slot = cfill + inc - c;
						cfill += __builtin_amdgcn_readlane(inc, 63);
						while (__builtin_amdgcn_ballot_w64(in != 0)) {
							if (in != 0) {
								const int u = __clz(in);
								list[slot < WCAP ? slot : WCAP - 1] = (unsigned)i | ((unsigned)(d0 + u + 1) << 16);   // an overflowing wave is detected below
								in &= ~(0x80000000u >> u);
								++slot;
							}
						}
					}
#else
					for (;;) {
						const unsigned long long m = __builtin_amdgcn_ballot_w64(in != 0);
						if (!m)
							break;
						if (in != 0) {
							const int u = __clz(in);
							int slot = cfill + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
							slot = slot < WCAP ? slot : WCAP - 1;     // an overflowing wave is detected below
							list[slot] = (unsigned)i | ((unsigned)(d0 + u + 1) << 16);
							in &= ~(0x80000000u >> u);
						}
						cfill += __popcll(m);
					}
#endif
				}
			}
			#pragma unroll
			for (int mm = 32; mm; mm >>= 1)
				cbl += __shfl_xor(cbl, mm);
			below = cbl;
		}
#endif
		if (lane == 0) {
			s.red[wave] = below;
			s.part[wave] = cfill;
		}
		__syncthreads();
		below = s.red[0] + s.red[1] + s.red[2] + s.red[3];
		const int f0 = s.part[0], f1 = s.part[1], f2 = s.part[2], f3 = s.part[3];
		const int real = f0 + f1 + f2 + f3;
		const int r = target - below;
		__syncthreads();
#ifdef TS_PROBE_DEBUG
		if (tid == 0 && blockIdx.x < 3) printf("row %d: T_lo %g T_hi %g below %d fills %d %d %d %d target %d r %d\n", (int)blockIdx.x, T_lo, T_hi, below, f0, f1, f2, f3, target, r);
#endif
#ifdef TS_PROBE_SKIP_LIST
		if (true) { slope = 0.f; done = true; } else
#endif
		if (f0 <= WCAP && f1 <= WCAP && f2 <= WCAP && f3 <= WCAP && r >= 0 && r < real) {
			auto segs = [&](auto fn) {                        // every listed slot, segment by segment
				for (int i = tid; i < f0; i += 256) fn(i);
				for (int i = tid; i < f1; i += 256) fn(WCAP + i);
				for (int i = tid; i < f2; i += 256) fn(2 * WCAP + i);
				for (int i = tid; i < f3; i += 256) fn(3 * WCAP + i);
			};
			segs([&](int i) {                                 // pair index -> exact slope, in place
				const unsigned pk = ((const unsigned *)s.buf)[i];
				const int pi = pk & 0xffff, pd = pk >> 16;
				s.buf[i] = (s.y[pi + pd] - s.y[pi]) / (float)pd;
			});
			__syncthreads();
			auto lst = [&](auto emit) { segs([&](int i) { emit(s.buf[i]); }); };
			float v;
#ifndef TS_NO_LINEAR_SELECT
			if (!select_rank_linear(s, tid, r, T_lo, T_hi, lst, v))   // the listed slopes sit in or next to [T_lo, T_hi]
#endif
				v = select_rank(s, tid, r, lst);
			if (v >= T_lo && v <= T_hi) {
				slope = v;
				done = true;
			}
		}
	}
#ifdef TS_PROBE_COUNT
	if (!done && tid == 0) atomicAdd(TS_PROBE_COUNT, 1);
#endif
#ifdef TS_PROBE_NO_FALLBACK
	if (!done) { slope = 0.f; done = true; }
#endif
	if (!done)   // wave-uniform: every thread computed the same decision
		slope = select_rank(s, tid, target, all_pairs_exact);
	// ---- intercepts y - slope*x, median (rank n/2)
	const int xoff = n / 2;
	for (int i = tid; i < n; i += 256)
		s.buf[i] = sub_mul_nofma(s.y[i], slope, (float)(i - xoff));
	__syncthreads();
	auto icpt = [&](auto emit) { for (int i = tid; i < n; i += 256) emit(s.buf[i]); };
#ifdef TS_PROBE_SKIP_YINT
	yint = 0.f;
#else
	yint = select_rank(s, tid, n / 2, icpt);
#endif
	// the key order treats -0 < +0; nth_element would return whichever sits there: same value
}

// decode.cc:479-504: one workgroup per (frame, row)
// carr_all != nullptr (8 kHz): the row is formed here from the carriers of two consecutive symbols; cons_raw_all
// (nullable) receives the unrotated row for the CONS_RAW tap
#ifndef TS_WAVES
#define TS_WAVES 5     // waves per SIMD the register budget is set for (5 -> <= 96 VGPRs, five workgroups per CU by LDS)
#endif
__global__ __launch_bounds__(256, TS_WAVES) void k_theil_sen(const SyncState *__restrict__ st_all, cf *__restrict__ cons_all,
	const cf *__restrict__ carr_all, cf *__restrict__ cons_raw_all, float *__restrict__ slope_all, float *__restrict__ yint_all)
{
#ifdef TS_PRIO
	__builtin_amdgcn_s_setprio(TS_PRIO);                      // experiments: issue priority against the co-resident polar decoders
#endif
	// grid = frames x 50 (mode 6 has exactly 50 rows: one row per block); modes with more rows loop
	const int f = blockIdx.x / TS_GRID_ROWS, tid = threadIdx.x;
	if (!st_all[f].okay)
		return;
	const ModeDesc md = mode_desc(st_all[f].oper_mode);
	__shared__ TsShared s;
	for (int j = blockIdx.x % TS_GRID_ROWS; j < md.rows; j += TS_GRID_ROWS) {
		cf *row = cons_all + (size_t)f * CONS_MAX + (size_t)j * md.cols;
		cf cv[2];
		#pragma unroll
		for (int q = 0; q < 2; ++q) {                         // decode.cc:482-487
			const int i = tid + 256 * q;
			cv[q] = mk(0.f, 0.f);
			if (i < md.cols) {
				cf c;
				if (carr_all) {                               // decode.cc:474-475
					const cf *cr = carr_all + (size_t)f * CARR_MAX + (size_t)j * md.cols;
					c = demod_or_erase(cr[md.cols + i], cr[i]);
					if (cons_raw_all)
						cons_raw_all[(size_t)f * CONS_MAX + (size_t)j * md.cols + i] = c;
				} else {
					c = row[i];
				}
				cv[q] = c;
				cf d = cmul(c, cconj(md.mod_bits == 3 ? psk8_hard_map(c) : psk4_hard_map(c)));
				s.y[i] = atan2f(d.im, d.re);
			}
		}
		__syncthreads();
		float slope, yint;
#ifdef TS_SPECIALIZE_432
		if (md.cols == CONS_COLS)                             // mode 6: compile-time trip counts (a second copy of the whole block)
			theil_sen_block(s, CONS_COLS, tid, slope, yint);
		else
#endif
			theil_sen_block(s, md.cols, tid, slope, yint);
		#pragma unroll
		for (int q = 0; q < 2; ++q) {                         // decode.cc:493-494
			const int i = tid + 256 * q;
			if (i < md.cols) {
				float a = -(yint + slope * (float)(i - md.cols / 2));
				float sn, cs;
				sincosf(a, &sn, &cs);
				row[i] = cmul(cv[q], mk(cs, sn));
			}
		}
		if (tid == 0) {
			slope_all[(size_t)f * ROWS_MAX + j] = slope;
			yint_all[(size_t)f * ROWS_MAX + j] = yint;
		}
		__syncthreads();
	}
}

__global__ __launch_bounds__(256, TS_WAVES) void k_theil_sen_raw(int cols, const float *__restrict__ y, float *__restrict__ slope_all,
	float *__restrict__ yint_all)
{
	const int r = blockIdx.x, tid = threadIdx.x;
	__shared__ TsShared s;
	for (int i = tid; i < cols; i += 256)
		s.y[i] = y[(size_t)r * cols + i];
	__syncthreads();
	float slope, yint;
	theil_sen_block(s, cols, tid, slope, yint);
	if (tid == 0) { slope_all[r] = slope; yint_all[r] = yint; }
}

void launch_theil_sen(hipStream_t s, int n, const SyncState *st, cf *cons, const cf *carr, cf *cons_raw, float *slope, float *yint)
{
	hipLaunchKernelGGL(k_theil_sen, dim3(n * TS_GRID_ROWS), dim3(256), 0, s, st, cons, carr, cons_raw, slope, yint);
}
void launch_theil_sen_raw(hipStream_t s, int rows, int cols, const float *y, float *slope, float *yint)
{
	hipLaunchKernelGGL(k_theil_sen_raw, dim3(rows), dim3(256), 0, s, cols, y, slope, yint);
}

}  // namespace rx
