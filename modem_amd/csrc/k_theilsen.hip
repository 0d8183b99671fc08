// k_theilsen.hip -- D5 (Theil-Sen phase-slope correction, decode.cc:479-504) for gfx950: ONE WAVE PER ROW, rank counting.
//
// DSP::TheilSenEstimator<value,512>::compute (decode.cc:488) = the median (sorted position count/2) of the n(n-1)/2
// pairwise slopes s_ij = fl(fl(y_j - y_i) / d), then the median of the intercepts.  Round 2 classified every pair
// against a sampled bracket (round 2's kernel, in the history: O(n^2), 33 k wave-instructions per row).  This kernel never walks
// the pairs.  x is the integer grid (decode.cc:485), so for a threshold T
//
//     s_ij < T   <=>   y_j - T x_j  <  y_i - T x_i          (up to the two fp32 roundings of s_ij)
//
// and #{s_ij < T} is the number of INVERSIONS of z = y - T x taken in index order: one sort of n keys counts them.
//  * keys: z in fp64 (exact to 2^-53), quantised to 22 bits over the row's range, index in the low 9 bits - a 31-bit
//    integer, so the order of the keys is the order of z with ties broken by index (a tie is no inversion).
//  * the sort is a bitonic merge sort of 512 keys held 8 per lane: lane-local stages are v_min / v_max on registers,
//    cross-lane stages one DPP / swizzle move and one v_med3_u32 (median(own, partner, 0 or ~0) = min or max).  Each
//    merge level adds the inversions between the two runs it merges: an element of the earlier run that lands on
//    merged position p had p - (its position in its own run) elements of the later run in front of it, so the level's
//    count is a sum of positions over the elements whose index bit says "earlier run" - two instructions per element.
//  * pairs whose keys differ by at most TS_MQ quanta are "uncertain" (quantisation + the fp32 roundings of s_ij); they
//    are neighbours in sorted order, found by a short scan, and get the exact fp32 division.  The count is then exact.
//  * search: least-squares slope as the first threshold, the density of slopes from the interquartile range of z, then
//    secant steps on EXACT counts.  As soon as a count lands within TS_OPEN_NEED ranks of the wanted one the bracket is
//    closed by an end that is NOT counted (the missing ranks + TS_OPEN_MARGIN, over the density): the pairs inside a
//    bracket are the pairs whose order differs between the keys at its two ends - neighbours in sorted order, a windowed
//    scan finds ALL of them whatever the far end is - they get the exact division, and with the exact count on the
//    counted side the wanted slope is the (target - count)-th of them from below or the (count - target)-th from above
//    (a 64-key sort over the lanes picks it).  A list that is too short or too long has its far end counted after all
//    and the search goes on with a closed bracket of at most TS_CAP pairs.  2.05 counts per row; every decision rests
//    on an exact count and a complete list: no probabilistic bracket to validate, no list of 2-3 % of the pairs.
//  * the intercepts (decode.cc:488 second median): the keys left by the last count are the row sorted by y - T x with T
//    next to the slope, so the candidate on position n/2 is checked by an exact rank count and moved by neighbours;
//    a full 512-key sort only if that does not settle.
//  * rows the search cannot finish (hundreds of tied slopes, NaNs) take an exact 8-bit radix select over all pairs.
// Bit-identical to nth_element on the CPU (tests/test_gpu_parity.py::test_theil_sen_bit_exact, ..._rank_search_rows,
// ..._soak_rows; tests/ts_soak.py: 60 000 random rows).
#include "dev_common.h"
#include "kernels.h"

namespace rx {

#define TS_CAP 56          // pairs the final bracket may hold (the list holds 64: a few candidates fall outside after the exact division)
#define TS_OPEN_NEED 30    // an open bracket (one counted end) is tried when the wanted rank is at most this far from the count
#define TS_OPEN_MARGIN 6   // ranks the uncounted end is placed beyond the wanted one (times 1.25 for the density estimate)
#define TS_MARGIN 12       // a secant step aims this many ranks past the target, on the side still open
#define TS_UNC_STEPS 24    // neighbour distance up to which uncertain pairs are resolved one by one
constexpr int TS_QBITS = 22;
constexpr uint32_t TS_QREAL = (1u << TS_QBITS) - 8192u;   // quanta [0, TS_QREAL) for the row's points
constexpr uint32_t TS_QPAD0 = (1u << TS_QBITS) - 4096u;   // slots beyond n: above every real key, 8 quanta apart
// |q_i - q_j| <= TS_MQ: the pair gets the exact division.  Why 2 is enough: |q_i - q_j| >= 3 means the fp64 values differ by
// more than 2 quanta, the exact ones by more than 2 - 2^-27; the two fp32 roundings of s_ij move it by at most
// 2^-22 |y_j - y_i| relative to T d, i.e. by less than one quantum (scale <= 2^22 / (ymax - ymin)).
constexpr uint32_t TS_MQ = 2;
constexpr int TS_MQ1S = (int)((TS_MQ + 1) << 9);
constexpr int TS_LIST = 64;                               // candidate pairs of the final bracket: one per lane
constexpr int TS_MAX_IT = 48;
constexpr uint32_t TS_SENTINEL = 0x7fffffffu;             // above every key, and differences to it stay positive as int32

// LDS images are "register major": element e = 8 * lane + t lives at row t, column lane, so that "my elements" and "the
// elements kk places after mine" are consecutive words across the lanes (index 8 * lane + t would put 8 lanes on a bank).
// Row strides 68 / 76 words keep column accesses AND the row-crossing writes of the loader (i = lane + 64 q) conflict-free.
constexpr int TS_YS = 68, TS_PS = 76;                     // sk: columns 64..75 hold sentinels (positions past the end)
// Rows (waves) per workgroup.  A row's wave never meets another wave: LDS operations of one wave execute in program order, so a
// value written by one lane is visible to a later read by any lane of the same wave; TS_SYNC only has to keep the compiler from
// moving LDS accesses across the point.  With one row per workgroup that is __syncthreads() (no s_barrier is emitted for a
// one-wave workgroup); with several rows it must not be a barrier (rows take different paths).  The hardware holds at most
// 16 workgroups per CU, so one-wave workgroups cap the kernel at 16 waves per CU where its registers would allow 20 - and
// still 1 row per workgroup is the fastest: 3.3 ms per 8192 frames against 3.5 (2 rows) and 4.1 (4 rows; a workgroup's LDS
// and slots are released only when its slowest row is through), profiles/r03_v3_theil_sen_rows_per_workgroup.txt.
#define TS_ROWS_PER_WG 1
#define TS_SYNC() __syncthreads()
struct TsLds {
	float y[8 * TS_YS];
	uint2 sk[8 * TS_PS];       // .x sorted keys of the last count; .y the same elements keyed at the other end of the bracket
	uint32_t lst[TS_LIST];
};
__device__ __forceinline__ int ts_yaddr(int i) { return (i & 7) * TS_YS + (i >> 3); }
__device__ __forceinline__ int ts_paddr(int p) { return (p & 7) * TS_PS + (p >> 3); }

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
// y - s*x with the product rounded on its own (a plain '*' would be contracted into an FMA): keeps the intercepts
// bit-identical to the CPU's
__device__ __forceinline__ float sub_mul_nofma(float y, float s, float x)
{
	#pragma clang fp contract(off)
	float p = s * x;
	return y - p;
}

// ---------------------------------------------------------------- cross-lane plumbing (wave64)
__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t r;
	asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);   // every lane has a valid source in the patterns used
}
template <int PAT> __device__ __forceinline__ uint32_t swz_u(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, PAT);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_XOR3 = 0x1B, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROR8 = 0x128;
constexpr int SWZ_XOR4 = 0x101F, SWZ_XOR16 = 0x401F, SWZ_XOR31 = 0x7C1F;   // bit-mask mode: and 0x1f, or 0, xor in [14:10]

struct TsLane {
	uint32_t c1, c2, c4, c8, c16, c32;   // ~0 where that bit of the lane id is set: the third operand of v_med3_u32
	int a63;                             // byte address of lane ^ 63 for ds_bpermute
	int lane;
};
__device__ __forceinline__ TsLane ts_lane(int lane)
{
	TsLane L;
	L.c1 = (lane & 1) ? ~0u : 0u; L.c2 = (lane & 2) ? ~0u : 0u; L.c4 = (lane & 4) ? ~0u : 0u;
	L.c8 = (lane & 8) ? ~0u : 0u; L.c16 = (lane & 16) ? ~0u : 0u; L.c32 = (lane & 32) ? ~0u : 0u;
	L.a63 = (lane ^ 63) << 2;
	L.lane = lane;
	return L;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
	v += (int)dpp_u<DPP_XOR1>((uint32_t)v);
	v += (int)dpp_u<DPP_XOR2>((uint32_t)v);
	v += (int)dpp_u<DPP_HALF_MIRROR>((uint32_t)v);
	v += (int)dpp_u<DPP_MIRROR>((uint32_t)v);
	return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
// a value every lane holds alike, moved to a scalar register (fp32 arithmetic on wave-uniform values still runs on the vector
// unit and would otherwise keep its result in a vector register for as long as it lives)
__device__ __forceinline__ float ts_uni(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
template <typename Op> __device__ __forceinline__ float wave_reduce_f(float v, Op op)
{
	v = op(v, __uint_as_float(dpp_u<DPP_XOR1>(__float_as_uint(v))));
	v = op(v, __uint_as_float(dpp_u<DPP_XOR2>(__float_as_uint(v))));
	v = op(v, __uint_as_float(dpp_u<DPP_HALF_MIRROR>(__float_as_uint(v))));
	v = op(v, __uint_as_float(dpp_u<DPP_MIRROR>(__float_as_uint(v))));
	const float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 0)), b = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 16));
	const float c = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 32)), d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 48));
	return ts_uni(op(op(a, b), op(c, d)));
}

// ---------------------------------------------------------------- the sort
// 512 keys, element e = 8 * lane + t in register k[t].  Bitonic merge sort in the "flip" form: a merge of two ascending
// runs first compares e with its mirror image inside the merged run, then halves the distance down to 1; everything is
// ascending, the lower element of a pair keeps the minimum.
__device__ __forceinline__ void ts_ce(uint32_t &a, uint32_t &b)
{
	const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
	a = lo; b = hi;
}
template <int D> __device__ __forceinline__ void ts_stage(uint32_t (&k)[8], const TsLane &L)   // partner = lane ^ D, same register
{
	// all eight moves first, then the eight medians: a DPP move must not read a register a VALU instruction wrote less than
	// two issue slots earlier (left to itself the scheduler pairs move and median and pays an s_nop per element)
	uint32_t p[8];
	#pragma unroll
	for (int t = 0; t < 8; ++t) {
		if (D == 1) p[t] = dpp_u<DPP_XOR1>(k[t]);
		else if (D == 2) p[t] = dpp_u<DPP_XOR2>(k[t]);
		else if (D == 4) p[t] = swz_u<SWZ_XOR4>(k[t]);
		else if (D == 8) p[t] = dpp_u<DPP_ROR8>(k[t]);
		else p[t] = swz_u<SWZ_XOR16>(k[t]);
	}
	__builtin_amdgcn_sched_barrier(0);
	const uint32_t c = D == 1 ? L.c1 : D == 2 ? L.c2 : D == 4 ? L.c4 : D == 8 ? L.c8 : L.c16;
	#pragma unroll
	for (int t = 0; t < 8; ++t)
		k[t] = med3u(k[t], p[t], c);
	__builtin_amdgcn_sched_barrier(0);
}
template <int M> __device__ __forceinline__ void ts_level(uint32_t (&k)[8], const TsLane &L)   // merges runs of 4 * 2^M keys
{
	uint32_t p[8];
	#pragma unroll
	for (int t = 0; t < 8; ++t) {                                 // flip: e <-> e ^ (8 * 2^M - 1): lane ^ (2^M - 1), register 7 - t
		if (M == 1) p[t] = dpp_u<DPP_XOR1>(k[7 - t]);
		else if (M == 2) p[t] = dpp_u<DPP_XOR3>(k[7 - t]);
		else if (M == 3) p[t] = dpp_u<DPP_HALF_MIRROR>(k[7 - t]);
		else if (M == 4) p[t] = dpp_u<DPP_MIRROR>(k[7 - t]);
		else if (M == 5) p[t] = swz_u<SWZ_XOR31>(k[7 - t]);
		else p[t] = (uint32_t)__builtin_amdgcn_ds_bpermute(L.a63, (int)k[7 - t]);
	}
	__builtin_amdgcn_sched_barrier(0);
	{
		const uint32_t c = M == 1 ? L.c1 : M == 2 ? L.c2 : M == 3 ? L.c4 : M == 4 ? L.c8 : M == 5 ? L.c16 : L.c32;
		#pragma unroll
		for (int t = 0; t < 8; ++t)
			k[t] = med3u(k[t], p[t], c);
	}
	__builtin_amdgcn_sched_barrier(0);
	if (M >= 6) ts_stage<16>(k, L);
	if (M >= 5) ts_stage<8>(k, L);
	if (M >= 4) ts_stage<4>(k, L);
	if (M >= 3) ts_stage<2>(k, L);
	if (M >= 2) ts_stage<1>(k, L);
	ts_ce(k[0], k[4]); ts_ce(k[1], k[5]); ts_ce(k[2], k[6]); ts_ce(k[3], k[7]);
	ts_ce(k[0], k[2]); ts_ce(k[1], k[3]); ts_ce(k[4], k[6]); ts_ce(k[5], k[7]);
	ts_ce(k[0], k[1]); ts_ce(k[2], k[3]); ts_ce(k[4], k[5]); ts_ce(k[6], k[7]);
}
// inversions between the two runs level M has just merged, as a per-lane share: the keys carry the element's index in
// their low 9 bits and the runs are index ranges, so bit M+2 of the index says "later run".  Summed over the wave, the
// later run's elements sit on positions whose total is S; the earlier run's inversions are (all positions) - S -
// (positions inside its own run), the constants are added once by the caller (TS_INV_CONST).
template <int M> __device__ __forceinline__ uint32_t ts_level_count(const uint32_t (&k)[8], const TsLane &L)
{
	uint32_t acc = 0;
	#pragma unroll
	for (int t = 0; t < 8; ++t)
		acc += ((k[t] >> (M + 2)) & 1u) * (256u + t);
	const uint32_t pbase = ((uint32_t)L.lane << 3) & ((16u << (M - 1)) - 1u);
	return (acc >> 8) * pbase + (acc & 255u);
}
constexpr int ts_inv_const()
{
	int c = 0;
	for (int m = 1; m <= 6; ++m) {
		const int k = 8 << m, l = k / 2, runs = 512 / k;
		c += runs * (k * (k - 1) / 2 - l * (l - 1) / 2);
	}
	return c;
}
constexpr int TS_INV_CONST = ts_inv_const();

// sorts k ascending (levels: 4 -> only the first 128 keys, lanes 0-15, are sorted as one run; 6 -> all 512).  counting
// (wave-uniform; keys below 2^31): returns this lane's share of the inversion count, TS_INV_CONST + wave sum = inversions
__device__ __forceinline__ int ts_sort(uint32_t (&k)[8], const TsLane &L, bool counting, int levels)
{
	int share = 0;
	if (counting) {                                               // inside the lane: all 28 pairs, sign of the difference
		uint32_t h = 0;
		#pragma unroll
		for (int t = 0; t < 8; ++t)
			#pragma unroll
			for (int u = t + 1; u < 8; ++u)
				h = __builtin_amdgcn_alignbit(h, k[u] - k[t], 31);
		share = __popc(h);
	}
	// Batcher's 19-comparator network for 8
	ts_ce(k[0], k[1]); ts_ce(k[2], k[3]); ts_ce(k[4], k[5]); ts_ce(k[6], k[7]);
	ts_ce(k[0], k[2]); ts_ce(k[1], k[3]); ts_ce(k[4], k[6]); ts_ce(k[5], k[7]);
	ts_ce(k[1], k[2]); ts_ce(k[5], k[6]);
	ts_ce(k[0], k[4]); ts_ce(k[1], k[5]); ts_ce(k[2], k[6]); ts_ce(k[3], k[7]);
	ts_ce(k[2], k[4]); ts_ce(k[3], k[5]);
	ts_ce(k[1], k[2]); ts_ce(k[3], k[4]); ts_ce(k[5], k[6]);
	ts_level<1>(k, L); if (counting) share -= (int)ts_level_count<1>(k, L);
	ts_level<2>(k, L); if (counting) share -= (int)ts_level_count<2>(k, L);
	ts_level<3>(k, L); if (counting) share -= (int)ts_level_count<3>(k, L);
	ts_level<4>(k, L); if (counting) share -= (int)ts_level_count<4>(k, L);
	if (levels > 4) {
		ts_level<5>(k, L); if (counting) share -= (int)ts_level_count<5>(k, L);
		ts_level<6>(k, L); if (counting) share -= (int)ts_level_count<6>(k, L);
	}
	return share;
}

// ---------------------------------------------------------------- keys of z = y - T x
struct TsQuant { double scale, off; };                         // q = floor(z * scale + off)
__device__ __forceinline__ TsQuant ts_quant(float T, float ymin, float ymax, int n)
{
	// range of z over the row from (ymin, ymax, |T| (n/2 + 1)): no pass over the data per threshold
	const double half = fabs((double)T) * (double)(n / 2 + 1);
	const double lo = (double)ymin - half, span = ((double)ymax + half) - lo;
	float sp = (float)span;
	sp = sp > 1e-30f ? sp : 1e-30f;
	// any scale <= TS_QREAL / span will do (the margins are in quanta): fp32 reciprocal, pulled in by 2^-20
	const double scale = (double)(__builtin_amdgcn_rcpf(sp) * ((float)(TS_QREAL - 2u) * (1.f - 9.5367431640625e-7f)));
	TsQuant q;
	q.scale = scale;
	q.off = -lo * scale;
	return q;
}
// keys of the eight points idx0 .. idx0 + 7 (values yv) at threshold T: q = floor(y scale + (off - T scale x))
__device__ __forceinline__ void ts_keys_run(uint32_t (&k)[8], const float (&yv)[8], int idx0, int n, float T, const TsQuant &q)
{
	const double tds = (double)T * q.scale;
	double c = fma(-tds, (double)(idx0 - n / 2), q.off);
	#pragma unroll
	for (int t = 0; t < 8; ++t) {
		const uint32_t qv = (uint32_t)fma((double)yv[t], q.scale, c);   // v_cvt_u32_f64: truncates, clamps below zero
		const int idx = idx0 + t;
		const uint32_t pad = TS_QPAD0 + 8u * (uint32_t)(idx - n + 1);
		k[t] = ((idx < n ? qv : pad) << 9) | (uint32_t)idx;
		c -= tds;
	}
}
__device__ __forceinline__ uint32_t ts_key(float y, int idx, int n, float T, const TsQuant &q)
{
	const double c = fma(-(double)T * q.scale, (double)(idx - n / 2), q.off);
	const uint32_t qv = (uint32_t)fma((double)y, q.scale, c);
	const uint32_t pad = TS_QPAD0 + 8u * (uint32_t)(idx - n + 1);
	return ((idx < n ? qv : pad) << 9) | (uint32_t)idx;
}

// exact slope of the pair of points a, b (any order)
__device__ __forceinline__ float ts_pair_slope(const TsLds &s, int a, int b)
{
	const int i = a < b ? a : b, j = a < b ? b : a;
	return (s.y[ts_yaddr(j)] - s.y[ts_yaddr(i)]) / (float)(j - i);
}

// Exact order statistic over ALL pairs by an 8-bit radix select with true divisions (one wave): the way out for rows the
// rank search cannot finish - hundreds of tied slopes (erased carriers), NaNs.  ~120 k instructions.
__device__ __noinline__ float ts_slow_select(TsLds &s, int n, int lane, int target)
{
	uint32_t prefix = 0, mask = 0;
	int rank = target;
	uint32_t *hist = (uint32_t *)s.sk;
	for (int shift = 24; shift >= 0; shift -= 8) {
		for (int b = lane; b < 256; b += 64)
			hist[b] = 0;
		TS_SYNC();
		for (int d = 1; d < n; ++d) {
			const float fd = (float)d;
			for (int i = lane; i < n - d; i += 64) {
				const unsigned key = fkey((s.y[ts_yaddr(i + d)] - s.y[ts_yaddr(i)]) / fd);
				if ((key & mask) == prefix)
					atomicAdd(&hist[(key >> shift) & 255u], 1u);
			}
		}
		TS_SYNC();
		// lane b handles bins 4b .. 4b+3
		const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
		const int mine = (int)(h0 + h1 + h2 + h3);
		int incl = mine;
		#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const int v = __shfl_up(incl, o);
			if (lane >= o)
				incl += v;
		}
		const int excl = incl - mine;
		const bool here = rank >= excl && rank < incl;
		int bin = 0, rr = 0;
		if (here) {
			rr = rank - excl;
			bin = 4 * lane;
			if (rr >= (int)h0) { rr -= h0; ++bin; if (rr >= (int)h1) { rr -= h1; ++bin; if (rr >= (int)h2) { rr -= h2; ++bin; } } }
		}
		const unsigned long long m = __builtin_amdgcn_ballot_w64(here);
		const int src = m ? __builtin_ctzll(m) : 0;
		bin = __builtin_amdgcn_readlane(bin, src);
		rank = __builtin_amdgcn_readlane(rr, src);
		prefix |= (uint32_t)bin << shift;
		mask |= 255u << shift;
		TS_SYNC();
	}
	return fkey_inv(prefix);
}

// one key per lane, ascending over the lanes: the same network on single registers (21 steps of one move + one median)
__device__ __forceinline__ uint32_t ts_sort_lanes(uint32_t v, const TsLane &L)
{
	v = med3u(v, dpp_u<DPP_XOR1>(v), L.c1);
	v = med3u(v, dpp_u<DPP_XOR3>(v), L.c2); v = med3u(v, dpp_u<DPP_XOR1>(v), L.c1);
	v = med3u(v, dpp_u<DPP_HALF_MIRROR>(v), L.c4); v = med3u(v, dpp_u<DPP_XOR2>(v), L.c2); v = med3u(v, dpp_u<DPP_XOR1>(v), L.c1);
	v = med3u(v, dpp_u<DPP_MIRROR>(v), L.c8); v = med3u(v, swz_u<SWZ_XOR4>(v), L.c4); v = med3u(v, dpp_u<DPP_XOR2>(v), L.c2);
	v = med3u(v, dpp_u<DPP_XOR1>(v), L.c1);
	v = med3u(v, swz_u<SWZ_XOR31>(v), L.c16); v = med3u(v, dpp_u<DPP_ROR8>(v), L.c8); v = med3u(v, swz_u<SWZ_XOR4>(v), L.c4);
	v = med3u(v, dpp_u<DPP_XOR2>(v), L.c2); v = med3u(v, dpp_u<DPP_XOR1>(v), L.c1);
	v = med3u(v, (uint32_t)__builtin_amdgcn_ds_bpermute(L.a63, (int)v), L.c32); v = med3u(v, swz_u<SWZ_XOR16>(v), L.c16);
	v = med3u(v, dpp_u<DPP_ROR8>(v), L.c8); v = med3u(v, swz_u<SWZ_XOR4>(v), L.c4); v = med3u(v, dpp_u<DPP_XOR2>(v), L.c2);
	v = med3u(v, dpp_u<DPP_XOR1>(v), L.c1);
	return v;
}
__device__ __forceinline__ uint32_t wave_max_u(uint32_t v)
{
	uint32_t o;
	o = dpp_u<DPP_XOR1>(v); v = v > o ? v : o;
	o = dpp_u<DPP_XOR2>(v); v = v > o ? v : o;
	o = dpp_u<DPP_HALF_MIRROR>(v); v = v > o ? v : o;
	o = dpp_u<DPP_MIRROR>(v); v = v > o ? v : o;
	const uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
	const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
	return ab > cd ? ab : cd;
}

// y[0..n) in s.y (register-major, every lane's writes visible); x[i] = i - n/2.  Returns (slope, yint) in all lanes.
__device__ __forceinline__ float2 theil_sen_wave(TsLds &s, int n, int lane)
{
	const TsLane L = ts_lane(lane);
	const int count = n * (n - 1) / 2, target = count / 2, xoff = n / 2;
	float slope = 0.f;
	uint32_t k[8];
	// the lane's eight points 8 lane .. 8 lane + 7: read again where they are needed rather than held across the search
	auto load_y = [&](float (&yv)[8]) {
		#pragma unroll
		for (int t = 0; t < 8; ++t)
			yv[t] = 8 * lane + t < n ? s.y[t * TS_YS + lane] : 0.f;
	};
	if (lane < TS_PS - 64) {                                      // positions past the end
		#pragma unroll
		for (int t = 0; t < 8; ++t)
			s.sk[t * TS_PS + 64 + lane] = make_uint2(TS_SENTINEL, TS_SENTINEL);
	}
	bool sorted_z = false;                                        // k holds the row's keys sorted at a threshold next to the slope
	if (count > 0) {
		// ---- row statistics: range, least-squares slope (a starting point only: fp32)
		float ymin = 3.0e38f, ymax = -3.0e38f, sy = 0.f, sxy = 0.f;
		{
		float yv[8];
		load_y(yv);
		#pragma unroll
		for (int t = 0; t < 8; ++t) {
			const int i = 8 * lane + t;
			if (i < n) {
				ymin = fminf(ymin, yv[t]); ymax = fmaxf(ymax, yv[t]);
				sy += yv[t]; sxy += yv[t] * (float)(i - xoff);
			}
		}
		}
		ymin = wave_reduce_f(ymin, [](float a, float b) { return fminf(a, b); });
		ymax = wave_reduce_f(ymax, [](float a, float b) { return fmaxf(a, b); });
		sy = wave_reduce_f(sy, [](float a, float b) { return a + b; });
		sxy = wave_reduce_f(sxy, [](float a, float b) { return a + b; });
		bool done = false, slow = false;
		if (ymin == ymax)                                         // every slope is +0
			done = true;
		else if (!(ymax - ymin < 3.0e38f)) {                      // NaN / inf in the row
			slow = true;
		}
		float Ta = 0.f, Tb = 0.f, T = 0.f;
		int ca = 0, cb = count;
		bool hasA = false, hasB = false;
		TsQuant qs = {0.0, 0.0};
		const unsigned long long real_lanes = n >= 512 ? ~0ull : (1ull << ((n + 7) >> 3)) - 1ull;
		float Tp = 0.f, irho = 0.f;                               // irho = 1 / (slopes per unit of T around the target)
		int cp = 0, it = 0, c_at_T = 0;
		bool hasP = false, open_end = false, open_failed = false;
		float To_open = 0.f;
		if (!done && !slow) {
			{
				const float fn = (float)n, sx = fn * (fn - 1.f) * 0.5f - fn * (float)xoff;
				const float sxx = (fn - 1.f) * fn * (2.f * fn - 1.f) * (1.f / 6.f) - 2.f * (float)xoff * (fn * (fn - 1.f) * 0.5f) + fn * (float)xoff * (float)xoff;
				const float rn = __builtin_amdgcn_rcpf(fn);
				T = (sxy - sx * sy * rn) * __builtin_amdgcn_rcpf(sxx - sx * sx * rn);
				if (!(fabsf(T) < 1.0e30f))
					T = 0.f;
				T = ts_uni(T);
			}
		}
		for (;;) {                                                // search, then the list; again only after an open bracket that failed
		if (!done && !slow) {
			for (;; ++it) {
				if (it >= TS_MAX_IT) { slow = true; break; }
				// ---- exact #{s < T}, #{s <= T}
				qs = ts_quant(T, ymin, ymax, n);
				{
					float yv[8];
					load_y(yv);
					ts_keys_run(k, yv, 8 * lane, n, T, qs);
				}
				const int share = ts_sort(k, L, true, 6);
				int c_lt = TS_INV_CONST + wave_sum_i(share), c_le;
				TS_SYNC();                                  // earlier readers of sk are done
				#pragma unroll
				for (int t = 0; t < 8; ++t)
					s.sk[t * TS_PS + lane].x = k[t];
				TS_SYNC();
				{   // uncertain pairs: neighbours in sorted order whose keys are within TS_MQ quanta
					int dmin = (int)(s.sk[lane + 1].x - k[7]);
					#pragma unroll
					for (int t = 0; t < 7; ++t) {
						const int d = (int)(k[t + 1] - k[t]);
						dmin = d < dmin ? d : dmin;
					}
					int dlt = 0, dle = 0;
					if (__builtin_amdgcn_ballot_w64(dmin < TS_MQ1S)) {
						for (int kk = 1;; ++kk) {
							if (kk > TS_UNC_STEPS) { slow = true; break; }
							uint32_t h = 0;                       // bit 7 - t: the key kk places after my t-th one is within the margin
							#pragma unroll
							for (int t = 0; t < 8; ++t)
								h = __builtin_amdgcn_alignbit(h, s.sk[((t + kk) & 7) * TS_PS + lane + ((t + kk) >> 3)].x - k[t] - (uint32_t)TS_MQ1S, 31);
							if (!__builtin_amdgcn_ballot_w64(h != 0))
								break;
							while (__builtin_amdgcn_ballot_w64(h != 0)) {   // one pair per lane and round: the exact division
								if (h != 0) {
									const int bit = 31 - __clz(h), p = 8 * lane + 7 - bit;
									h &= ~(1u << bit);
									const int a = (int)(s.sk[ts_paddr(p)].x & 511u), b = (int)(s.sk[ts_paddr(p + kk)].x & 511u);
									const int counted = a > b;    // the earlier key belongs to the later point: an inversion
									const float sl = ts_pair_slope(s, a, b);
									dlt += (sl < T) - counted;
									dle += (sl <= T) - counted;
								}
							}
						}
						if (slow)
							break;
						dlt = wave_sum_i(dlt);
						dle = wave_sum_i(dle);
					}
					c_le = c_lt + dle;
					c_lt += dlt;
				}
				sorted_z = true;
				c_at_T = c_lt;
				if (c_lt <= target && target < c_le) { slope = T; done = true; break; }
				if (c_lt <= target) { Ta = T; ca = c_lt; hasA = true; }
				else { Tb = T; cb = c_lt; hasB = true; }
				if (hasA && hasB) {
					if (cb - ca <= TS_CAP)
						break;
					if (fkey(Tb) - fkey(Ta) <= 1u) { slope = Ta; done = true; break; }   // every slope in [Ta, Tb) is Ta
				}
				// ---- next threshold: secant on the last two exact counts (first step: density from the interquartile range)
				if (it == 0) {
					const uint32_t q1 = s.sk[ts_paddr(n / 4)].x >> 9, q3 = s.sk[ts_paddr((3 * n) / 4)].x >> 9;
					const float iqr = (float)(q3 > q1 ? q3 - q1 : 1u) * __builtin_amdgcn_rcpf((float)qs.scale);
					// gaussian noise sigma = IQR / 1.349: density of slopes at their median = sum_d (n - d) d / (2 sigma sqrt(pi))
					const float fn = (float)n;
					irho = ts_uni(iqr * (2.f * 1.7724539f * 6.f / 1.349f) * __builtin_amdgcn_rcpf(fn * (fn * fn - 1.f)));
				}
				if (hasP && c_lt != cp && T != Tp)
					irho = ts_uni((T - Tp) * __builtin_amdgcn_rcpf((float)(c_lt - cp)));
				// ---- close enough to list the pairs between T and an end that is not counted: T + / - (the ranks still missing +
				// TS_OPEN_MARGIN) / density.  The list stage finds EVERY pair in between; with the exact count on the T side
				// the wanted rank is the (target - ca)-th of them from below (T below the median slope) or the (cb - target)-th
				// from above; if the list turns out too short or too long, the other end is counted after all.
				{
					const bool below = c_lt <= target;
					const int need = below ? target - c_lt + 1 : c_lt - target;
					if (!open_failed && irho > 0.f && need <= TS_OPEN_NEED) {
						const float w = (float)(need + TS_OPEN_MARGIN) * irho * 1.25f;
						To_open = ts_uni(below ? T + w : T - w);
						if (fabsf(To_open) < 1.0e30f && To_open != T && (below ? (!hasB || To_open < Tb) : (!hasA || To_open > Ta))) {
							open_end = true;
							break;
						}
					}
				}
				int goal;
				if (it == 0 && !open_failed) goal = target;      // the next count most likely allows an open bracket
				else if (!hasA) goal = target - TS_MARGIN;
				else if (!hasB) goal = target + TS_MARGIN;
				else goal = (target - ca > cb - target) ? target - TS_MARGIN : target + TS_MARGIN;
				float Tn = T;
				bool ok = false;
				if (irho > 0.f && it < 10) {
					Tn = T + ((float)(goal - c_lt) + 0.5f) * irho;
					ok = fabsf(Tn) < 1.0e30f && Tn != T && (!hasA || Tn > Ta) && (!hasB || Tn < Tb);
				}
				if (!ok) {                                        // bisection in key space: always terminates
					const uint32_t ka = hasA ? fkey(Ta) : fkey(-1.0e30f), kb = hasB ? fkey(Tb) : fkey(1.0e30f);
					uint32_t km = ka + (kb - ka) / 2u;
					if (km == ka)
						km = kb;                                  // (only when one side is still open)
					Tn = fkey_inv(km);
					if (Tn == T) { slow = true; break; }
				}
				Tp = T; cp = c_lt; hasP = true;
				T = ts_uni(Tn);
			}
		}
		if (!done && !slow) {
			// ---- the pairs inside [Ta, Tb): their order differs between the keys at Ta and at Tb.  The keys in k / s.sk.x are
			// sorted at one end (T); key the same elements at the other end and look for neighbours that are not clearly
			// ascending there (sorted at Ta: s < Tb <=> the later key is not larger at Tb; sorted at Tb: s >= Ta <=> the same
			// at Ta), plus the pairs that are uncertain at T itself.
			const bool t_is_a = open_end ? c_at_T <= target : T == Ta;
			const float To = open_end ? To_open : (t_is_a ? Tb : Ta);
			const float La = t_is_a ? T : To, Lb = t_is_a ? To : T;   // the list's bracket [La, Lb)
			const TsQuant qo = ts_quant(To, ymin, ymax, n);
			#pragma unroll
			for (int t = 0; t < 8; ++t) {
				const int i = (int)(k[t] & 511u);
				s.sk[t * TS_PS + lane].y = ts_key(i < n ? s.y[ts_yaddr(i)] : 0.f, i, n, To, qo);
			}
			// a pair inside the bracket is at most this far apart in the sorted keys (minus the margin: compared with o - km)
			const float wq = fabsf(Lb - La) * (float)qs.scale * (float)(n - 1) * 1.0001f + 4.f;
			const int ws = wq < 4.0e6f ? (int)(((uint32_t)wq << 9) | 511u) : 0x7fffffff;
			TS_SYNC();
			int nc = 0;
			for (int kk = 1;; ++kk) {
				if (kk > 8 * (TS_PS - 64)) {
					slow = true; break;
				}
				uint32_t h = 0;
				int gmin = 0x7fffffff;
				#pragma unroll
				for (int t = 0; t < 8; ++t) {
					// (the lane's own keys at the other end are read again per step instead of held in eight more registers)
					const uint2 o = s.sk[((t + kk) & 7) * TS_PS + lane + ((t + kk) >> 3)];
					const int x2 = (int)o.x - ((int)k[t] + TS_MQ1S);                              // < 0: uncertain at T
					const int x1 = (int)o.y - ((int)s.sk[t * TS_PS + lane].y + TS_MQ1S);          // < 0: not clearly ascending at the other end
					// (a slot beyond n never ends the scan late: those keys are only 8 quanta apart, and a bracket wider than
					// the 4096 quanta between them and the real keys would walk through all of them)
					const int x2r = o.x < (TS_QPAD0 << 9) ? x2 : 0x7fffffff;
					gmin = x2r < gmin ? x2r : gmin;
					h = __builtin_amdgcn_alignbit(h, (uint32_t)(x1 | x2), 31);
				}
				unsigned long long hm = __builtin_amdgcn_ballot_w64(h != 0);
				while (hm) {                                      // one candidate per lane and round
					const bool mine = h != 0;
					const int slot = nc + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0));
					if (mine) {
						const int bit = 31 - __clz(h), p = 8 * lane + 7 - bit;
						h &= ~(1u << bit);
						if (slot < TS_LIST)
							s.lst[slot] = (uint32_t)p | ((uint32_t)(p + kk) << 16);
					}
					nc += __popcll(hm);
					hm = __builtin_amdgcn_ballot_w64(h != 0);
				}
				if (!(__builtin_amdgcn_ballot_w64(gmin <= ws) & real_lanes))
					break;
			}
			TS_SYNC();
			bool retry = false;
			if (nc > TS_LIST) {
				if (open_end)
					retry = true;
				else {
					slow = true;
				}
			}
			if (!slow && !retry) {
				bool in = false;
				uint32_t key = 0xffffffffu;
				if (lane < nc) {
					const uint32_t e = s.lst[lane];
					const float sl = ts_pair_slope(s, (int)(s.sk[ts_paddr(e & 0xffffu)].x & 511u), (int)(s.sk[ts_paddr(e >> 16)].x & 511u));
					in = sl >= La && sl < Lb;
					if (in)
						key = fkey(sl);
				}
				const int found = __popcll(__builtin_amdgcn_ballot_w64(in));
				// rank of the wanted slope among the listed ones: counted from the side whose count is exact
				const int idx = t_is_a ? target - c_at_T : found - (c_at_T - target);
				if (open_end) {
					if (idx < 0 || idx >= found)
						retry = true;                                           // the open end was too close: count it
				} else if (found != cb - ca) {
					slow = true;                                  // (cannot happen: the counts are exact)
				}
				if (!slow && !retry) {
					key = ts_sort_lanes(key, L);
					slope = fkey_inv(__builtin_amdgcn_readlane(key, idx));
				}
			}
			if (retry) {                                          // the search goes on with a count at the open end
				open_failed = true;
				open_end = false;
				Tp = T; cp = c_at_T; hasP = true;
				T = To;
				++it;
				continue;
			}
		}
		break;
		}
		if (slow) {
			slope = ts_slow_select(s, n, lane, target);
			sorted_z = false;
		}
	}
	// ---- intercepts b = y - slope*x, median (sorted position n/2)
	float yint = 0.f;
	bool have = false;
	if (sorted_z) {
		// The keys in k are the row sorted by y - T x with T within a few ulps' worth of ranks of the slope: b in that order is
		// sorted up to swaps of close neighbours, so the element on position n/2 is the median or next to it.  Its exact rank
		// is counted; if it is off, the next value below / above is tried (a few rounds), else the full sort below decides.
		uint32_t bk[8];
		#pragma unroll
		for (int t = 0; t < 8; ++t) {
			const int i = (int)(k[t] & 511u);
			bk[t] = i < n ? fkey(sub_mul_nofma(s.y[ts_yaddr(i)], slope, (float)(i - xoff))) : 0xffffffffu;
		}
		TS_SYNC();
		if (lane == (n / 2) >> 3) {
			#pragma unroll
			for (int t = 0; t < 8; ++t)
				if (t == ((n / 2) & 7))
					s.lst[0] = bk[t];
		}
		TS_SYNC();
		uint32_t cand = s.lst[0];
		for (int round = 0; round < 6 && !have; ++round) {
			int both = 0;                                         // #{b < cand} in the low half, #{b <= cand} in the high half (n <= 512): one wave sum
			#pragma unroll
			for (int t = 0; t < 8; ++t)
				both += (int)(bk[t] < cand) + ((int)(bk[t] <= cand) << 16);
			both = wave_sum_i(both);
			const int lt = both & 0xffff, le = both >> 16;
			if (lt <= n / 2 && n / 2 < le) {
				have = true;
			} else if (lt > n / 2) {                              // too high: the largest value below it
				uint32_t m = 0;
				#pragma unroll
				for (int t = 0; t < 8; ++t)
					m = (bk[t] < cand && bk[t] > m) ? bk[t] : m;
				cand = wave_max_u(m);
			} else {                                              // too low: the smallest value above it
				uint32_t m = 0;
				#pragma unroll
				for (int t = 0; t < 8; ++t)
					m = (bk[t] > cand && ~bk[t] > m) ? ~bk[t] : m;
				cand = ~wave_max_u(m);
			}
		}
		yint = fkey_inv(cand);
	}
	if (!have) {                                                  // one 512-key sort
		float yv[8];
		load_y(yv);
		#pragma unroll
		for (int t = 0; t < 8; ++t) {
			const int i = 8 * lane + t;
			k[t] = i < n ? fkey(sub_mul_nofma(yv[t], slope, (float)(i - xoff))) : 0xffffffffu;
		}
		ts_sort(k, L, false, 6);
		TS_SYNC();
		#pragma unroll
		for (int t = 0; t < 8; ++t)
			s.sk[t * TS_PS + lane].x = k[t];
		TS_SYNC();
		yint = n > 0 ? fkey_inv(s.sk[ts_paddr(n / 2)].x) : 0.f;
	}
	// the key order treats -0 < +0; nth_element would return whichever sits there: same value
	return make_float2(slope, yint);
}

// arg(d) for d = c conj(map(hard(c))) (decode.cc:486): the point turned back onto the positive real axis, so d.re > 0 and
// |d.im| <= d.re tan(pi/8) (8PSK) or <= d.re (QPSK) - the arc tangent of a ratio of magnitude <= 0.42 / <= 1, where a minimax
// polynomial in t^2 (6 / 10 coefficients; fitted in double, <= 1.0 / 1.4 ulp over the range in fp32 Horner form) and a division
// refined once replace the library's atan2f (38 vector instructions, most of them quadrant and special-case handling): 11 / 15.
// Anything else - an erased carrier (d = 0), NaN, a point exactly on a decision boundary pushed past the range by rounding -
// takes the library routine.
#define TS_OWN_ATAN 1
__device__ __forceinline__ float ts_phase(cf d, int mod_bits)
{
	const float lim = mod_bits == 3 ? 0.42f : 1.005f;               // (the fits cover 0.4225 / 1.00995)
	if (!TS_OWN_ATAN || !(d.re > 0.f) || !(fabsf(d.im) <= lim * d.re))
		return atan2f(d.im, d.re);
	const float r = __builtin_amdgcn_rcpf(d.re);
	float t = d.im * r;
	t = fmaf(fmaf(-d.re, t, d.im), r, t);                         // the quotient, correctly rounded but for rare ties
	const float z = t * t;
	float p;
	if (mod_bits == 3) {
		p = -5.951132927e-02f;
		p = fmaf(p, z, 1.054106507e-01f);
		p = fmaf(p, z, -1.423576012e-01f);
		p = fmaf(p, z, 1.999798680e-01f);
		p = fmaf(p, z, -3.333330347e-01f);
		p = fmaf(p, z, 9.999999993e-01f);
	} else {
		p = -1.718391760e-03f;
		p = fmaf(p, z, 1.059115275e-02f);
		p = fmaf(p, z, -3.059610274e-02f);
		p = fmaf(p, z, 5.738879400e-02f);
		p = fmaf(p, z, -8.370542419e-02f);
		p = fmaf(p, z, 1.094069024e-01f);
		p = fmaf(p, z, -1.426187097e-01f);
		p = fmaf(p, z, 1.999827962e-01f);
		p = fmaf(p, z, -3.333328490e-01f);
		p = fmaf(p, z, 9.999999978e-01f);
	}
	return p * t;
}

// one row: decode.cc:482-492.  The rotation of decode.cc:493-494 is not done here (round 4): this kernel is bound by vector
// instruction issue, the rotated row was 176 KB per frame written and read back, and its only consumers (k_back; the CONS_ROT
// tap) are latency-bound kernels that rotate each point where they use it (dev_common.h: rotate_point).
__device__ __forceinline__ void ts_row(TsLds &s, int f, int j, int lane, const ModeDesc &md, cf *__restrict__ cons_all,
	const cf *__restrict__ carr_all, float *__restrict__ slope_all, float *__restrict__ yint_all)
{
	const int cols = md.cols;
	cf *row = cons_all + (size_t)f * CONS_MAX + (size_t)j * cols;
	const cf *cr = carr_all ? carr_all + (size_t)f * CARR_MAX + (size_t)j * cols : nullptr;
	// decode.cc:482-487.  The lane's points lane + 64 q are fetched together, then turned into phases: one point per round trip (the
	// round-5 form) left a wave parked at s_waitcnt for 80 % of this stage - a fifth of its life (profiles/r06_theil_sen_by_stage.txt)
	cf pt[8], pv[8];
	#pragma unroll
	for (int q = 0; q < 8; ++q) {
		const int i = lane + 64 * q;
		if (i < cols) {
			if (cr) {
				pt[q] = cr[cols + i];
				pv[q] = cr[i];
			} else
				pt[q] = row[i];
		}
	}
	#pragma unroll
	for (int q = 0; q < 8; ++q) {
		const int i = lane + 64 * q;
		if (i < cols) {
			cf c = pt[q];
			if (cr) {                                             // decode.cc:474-475
				c = demod_or_erase(pt[q], pv[q]);
				row[i] = c;
			}
			cf d = cmul(c, cconj(md.mod_bits == 3 ? psk8_hard_map(c) : psk4_hard_map(c)));
			s.y[ts_yaddr(i)] = ts_phase(d, md.mod_bits);
		}
	}
	TS_SYNC();
	const float2 sy = theil_sen_wave(s, cols, lane);
	if (lane == 0) {
		slope_all[(size_t)f * ROWS_MAX + j] = sy.x;
		yint_all[(size_t)f * ROWS_MAX + j] = sy.y;
	}
}

// decode.cc:479-504: one wave per (frame, row)
// carr_all != nullptr: the row is formed here from the carriers of two consecutive symbols (the rates whose demodulator does not).
// Rows 0 .. TS_ROWS_DIRECT-1 of every frame: one wave each (grid = frames x TS_ROWS_DIRECT; mode 6 has exactly 50 rows, mode 10 has
// 42: its last eight waves leave at once).  The modes with more rows (7, 8, 9, 11, 12, 13: up to 126) get theirs from
// k_theil_sen_more: TS_MORE_WAVES persistent waves that stride over the units (frame, row 50 .. R-1), R = the largest row count
// the direct kernel met in this chunk (chunk_flags[0], an atomicMax; cleared by k_init_sync) - it leaves at once when no frame
// has more than 50 rows.  A launch of frames x 126 one-row waves, 60 % of which leave at once for mode 6, cost 0.15 ms per
// 8192 frames.  The loop costs registers (what depends on the row length and the lane alone is hoisted out of it; bounding it
// to the direct kernel's 96 spills five of them), so it is its own kernel.
#define TS_MORE_WAVES_N 20480
#define TS_MORE_OCC 5
constexpr int TS_ROWS_DIRECT = 50, TS_MORE_WAVES = TS_MORE_WAVES_N;
__global__ __launch_bounds__(64 * TS_ROWS_PER_WG, 5) void k_theil_sen(const SyncState *__restrict__ st_all, cf *__restrict__ cons_all,
	const cf *__restrict__ carr_all, float *__restrict__ slope_all, float *__restrict__ yint_all, int n_frames, int *__restrict__ chunk_flags)
{
	const int unit = (int)blockIdx.x * TS_ROWS_PER_WG + ((int)threadIdx.x >> 6);
	// (uniform per wave: said explicitly, or the row's addresses are computed per lane)
	const int f = __builtin_amdgcn_readfirstlane(unit / TS_ROWS_DIRECT), j = __builtin_amdgcn_readfirstlane(unit % TS_ROWS_DIRECT), lane = threadIdx.x & 63;
	if (f >= n_frames || !st_all[f].okay)
		return;
	const ModeDesc md = mode_desc(st_all[f].oper_mode);
	if (j == 0 && lane == 0 && md.rows > TS_ROWS_DIRECT)
		atomicMax(chunk_flags, md.rows);
	if (j >= md.rows)
		return;
	__shared__ TsLds s_all[TS_ROWS_PER_WG];
	ts_row(s_all[threadIdx.x >> 6], f, j, lane, md, cons_all, carr_all, slope_all, yint_all);
}
__global__ __launch_bounds__(64, TS_MORE_OCC) void k_theil_sen_more(const SyncState *__restrict__ st_all, cf *__restrict__ cons_all,
	const cf *__restrict__ carr_all, float *__restrict__ slope_all, float *__restrict__ yint_all, int n_frames, const int *__restrict__ chunk_flags)
{
	const int more = chunk_flags[0] - TS_ROWS_DIRECT;
	if (more <= 0)
		return;
	const int lane = threadIdx.x, units = n_frames * more;
	__shared__ TsLds s;
	for (int u = (int)blockIdx.x; u < units; u += TS_MORE_WAVES) {
		const int f = u / more, j = TS_ROWS_DIRECT + u % more;
		if (!st_all[f].okay)
			continue;
		const ModeDesc md = mode_desc(st_all[f].oper_mode);
		if (j >= md.rows)
			continue;
		__syncthreads();
		ts_row(s, f, j, lane, md, cons_all, carr_all, slope_all, yint_all);
	}
}
__global__ __launch_bounds__(64 * TS_ROWS_PER_WG, 5) void k_theil_sen_raw(int cols, int rows, const float *__restrict__ y, float *__restrict__ slope_all,
	float *__restrict__ yint_all)
{
	const int r = __builtin_amdgcn_readfirstlane((int)blockIdx.x * TS_ROWS_PER_WG + ((int)threadIdx.x >> 6)), lane = threadIdx.x & 63;
	if (r >= rows)
		return;
	__shared__ TsLds s_all[TS_ROWS_PER_WG];
	TsLds &s = s_all[threadIdx.x >> 6];
	for (int i = lane; i < cols; i += 64)
		s.y[ts_yaddr(i)] = y[(size_t)r * cols + i];
	TS_SYNC();
	const float2 sy = theil_sen_wave(s, cols, lane);
	if (lane == 0) { slope_all[r] = sy.x; yint_all[r] = sy.y; }
}

void launch_theil_sen(hipStream_t s, int n, const SyncState *st, cf *cons, const cf *carr, float *slope, float *yint, int *chunk_flags)
{
	hipLaunchKernelGGL(k_theil_sen, dim3((n * TS_ROWS_DIRECT + TS_ROWS_PER_WG - 1) / TS_ROWS_PER_WG), dim3(64 * TS_ROWS_PER_WG), 0, s, st, cons, carr,
		slope, yint, n, chunk_flags);
	hipLaunchKernelGGL(k_theil_sen_more, dim3(TS_MORE_WAVES), dim3(64), 0, s, st, cons, carr, slope, yint, n,
		chunk_flags);
}
void launch_theil_sen_raw(hipStream_t s, int rows, int cols, const float *y, float *slope, float *yint)
{
	hipLaunchKernelGGL(k_theil_sen_raw, dim3((rows + TS_ROWS_PER_WG - 1) / TS_ROWS_PER_WG), dim3(64 * TS_ROWS_PER_WG), 0, s, cols, rows, y, slope, yint);
}

}  // namespace rx
