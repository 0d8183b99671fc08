// k_sc.hip -- the list decoder's sign-following path, decoded alone (list size 1), and the certificate that it IS the list
// decoder's answer ("SC dominance") for gfx950.  decode.cc:530-555 needs lane 0's message, its CRC and the flip
// count - not the eight-path search - whenever lane 0 provably is that path.
//
// P* = the path of CODE::PolarListDecoder (decode.cc:201) that takes the sign of its LLR at every information leaf.  k_sc decodes
// P* with one lane's arithmetic of the list decoder - f_minsum, g_add, frozen penalties max(0, -llr) leaf by leaf, an aligned
// all-frozen node of 2..128 leaves in the butterfly order at once (k_polar.hip, oracle/polar.c: scl_node) - and carries
//   M*        P*'s path metric (fp32, same additions in the same order as lane 0's),
//   min_fork  min over the information leaves i of fl(M*(i) + |llr_i|): the metric of the candidate that leaves P* at leaf i.
// Rule: min_fork > M*(final)  =>  P* is lane 0 of the list decoder, for any list size.  Every candidate that is not P* either
// descends from a first deviation at some leaf i - it then carries at least fl(M*(i) + |llr_i|) for ever, fp32 sums of non-negative
// penalties being monotone - or from one of the placeholder paths (metric 1000 at the start) making P*'s own decisions - never
// cheaper than P* (monotone again), and P* wins ties through its candidate index (lane 0).  So P* has the smallest metric, ties
// broken its way, at every fork: never pruned, rank 0 throughout, lane 0 at the end with metric M*.  The syndrome certificate of
// k_back is the case M* = 0.  k_sc_finish then does decode.cc:532-555 for lane 0: CRC-32 of P*'s systematic bits, payload, flip
// count; a frame whose rule or CRC fails goes on to the list decoder's queue unchanged (its LLRs are copied there).
// Checked against the oracle's list decoder by tests/test_oracle_kat.py (the rule, CPU) and tests/test_gpu_parity.py (this kernel).
//
// One wavefront per codeword, persistent.  Where the tree lives: the input arrays of the nodes of 2^15, 2^14, 2^13 leaves in a
// level store in HBM (224 KB per resident decoder; each is written once and read once), of the current 4096-leaf node in LDS (16 KB),
// of the current nodes of 2048..128 leaves in registers (position x * 64 + q of such an array = element x of the lane that holds
// q: every f / g step down to 128 leaves is lane-local), the 64-leaf block below in one register per lane with DPP / permlane
// butterflies.  Partial sums are bits: one 64-bit register per lane for the current 4096-leaf node (bit x = position x * 64 + q),
// published as plain bit-packed words (bit i of the codeword = bit i % 64 of word i / 64) that the upper g steps read back.
// Uniform nodes are decided in one step: all-frozen ones of up to 128 leaves (the penalty sum above), all-information ones of any
// size (the SC decisions of such a node are the signs of its input LLRs; the smallest leaf magnitude on P* is the smallest input
// magnitude, so min_fork takes fl(M* + min |input|)).
#include "dev_common.h"
#include "kernels.h"
#include "polar_common.h"

namespace rx {

#define SC_WAVE_ORDER() __builtin_amdgcn_wave_barrier()

// ---- positions and lanes.  Lane l holds position q = l ^ ((l & 4) ? 3 : 0) of a 64-leaf block: then "the lane whose position
// differs in bit 2" is row_half_mirror (l ^ 7), and all six butterfly exchanges are single DPP / permlane instructions.
__device__ __forceinline__ int sc_pos(int lane) { return lane ^ ((lane & 4) ? 3 : 0); }
template <int H> __device__ __forceinline__ int xpos_i(int v, int lane)
{
	if constexpr (H == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);           // quad_perm [1,0,3,2]
	else if constexpr (H == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
	else if constexpr (H == 4) return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);     // row_half_mirror
	else if constexpr (H == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);     // row_ror:8
	else if constexpr (H == 16) {
		auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
		return (int)((lane & 16) ? r[0] : r[1]);
	} else {
		auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
		return (int)((lane & 32) ? r[0] : r[1]);
	}
}
template <int H> __device__ __forceinline__ float xpos(float v, int lane) { return __int_as_float(xpos_i<H>(__float_as_int(v), lane)); }
template <int H> __device__ __forceinline__ uint32_t xpos(uint32_t v, int lane) { return (uint32_t)xpos_i<H>((int)v, lane); }

// the decoder's running figures: M* and min_fork (as its bit pattern: non-negative floats order like unsigned integers)
struct ScAcc {
	float M;
	uint32_t fork;
	__device__ __forceinline__ void info(uint32_t mu_bits)        // an information leaf / all-information node with smallest magnitude mu
	{
		const float cand = M + __uint_as_float(mu_bits);
		fork = min(fork, __float_as_uint(cand));
	}
};

// sum over the 2^LV values of a node held one per lane (duplicated over the other position bits) in the butterfly halving
// order p[i] += p[i + h], h = n/2 .. 1 (oracle/polar.c: scl_node's rate-0 step): both partners form the same sum
template <int LV> __device__ __forceinline__ float sc_pen_sum(float pen, int lane)
{
	if constexpr (LV >= 6) pen = pen + xpos<32>(pen, lane);
	if constexpr (LV >= 5) pen = pen + xpos<16>(pen, lane);
	if constexpr (LV >= 4) pen = pen + xpos<8>(pen, lane);
	if constexpr (LV >= 3) pen = pen + xpos<4>(pen, lane);
	if constexpr (LV >= 2) pen = pen + xpos<2>(pen, lane);
	if constexpr (LV >= 1) pen = pen + xpos<1>(pen, lane);
	return pen;
}
template <int LV> __device__ __forceinline__ uint32_t sc_min_mag(uint32_t mu, int lane)
{
	if constexpr (LV >= 6) mu = min(mu, xpos<32>(mu, lane));
	if constexpr (LV >= 5) mu = min(mu, xpos<16>(mu, lane));
	if constexpr (LV >= 4) mu = min(mu, xpos<8>(mu, lane));
	if constexpr (LV >= 3) mu = min(mu, xpos<4>(mu, lane));
	if constexpr (LV >= 2) mu = min(mu, xpos<2>(mu, lane));
	if constexpr (LV >= 1) mu = min(mu, xpos<1>(mu, lane));
	return mu;
}
__device__ __forceinline__ float sc_pen(float v) { return v < 0.f ? -v : 0.f; }
__device__ __forceinline__ uint32_t sc_mag(float v) { return __float_as_uint(v) & 0x7fffffffu; }

// g step inside a 64-leaf block: the level-LV node's own value and its partner's (position bit LV-1), the left child's partial sum
template <int LV> __device__ __forceinline__ float sc_g_cross(float own, int ub, int lane, int q)
{
	const float oth = xpos<(1 << (LV - 1))>(own, lane);
	const bool hi = (q >> (LV - 1)) & 1;
	return g_add(hi ? oth : own, hi ? own : oth, ub);
}

// ---- the 8-leaf sub-tree (levels 3..0): a top-down walk that decides every uniform node it meets.  r[L] = this lane's LLR at
// level L (position = the low L bits of q, duplicated over the others); H = partial sums, one bit per leaf: only the bit of the
// lane's own position (mod the node size) is ever consumed, and the combines keep exactly that bit right.
struct ScBlock8 {
	ScAcc &acc;
	int &H;
	float r[4];
	const uint32_t fz;
	const int lane, q;

	template <int LV, int P0> __device__ __forceinline__ void node()
	{
		if constexpr (LV == 0) {
			const float r0 = r[0];
			if ((fz >> P0) & 1) {
				if (r0 < 0.f)
					acc.M -= r0;
			} else {
				acc.info(sc_mag(r0));
				H |= (r0 < 0.f ? 1 : 0) << P0;
			}
		} else {
			constexpr int N = 1 << LV, HALF = N / 2;
			constexpr uint32_t MASK = ((1u << N) - 1u) << P0;
			const uint32_t pat = fz & MASK;
			if (pat == MASK) {
				acc.M += sc_pen_sum<LV>(sc_pen(r[LV]), lane);
				return;
			}
			if (pat == 0) {
				acc.info(sc_min_mag<LV>(sc_mag(r[LV]), lane));
				H |= r[LV] < 0.f ? (int)MASK : 0;
				return;
			}
			r[LV - 1] = f_minsum(r[LV], xpos<HALF>(r[LV], lane));
			node<LV - 1, P0>();
			r[LV - 1] = sc_g_cross<LV>(r[LV], (H >> (P0 + (q & (HALF - 1)))) & 1, lane, q);
			node<LV - 1, P0 + HALF>();
			constexpr int lmask = ((1 << HALF) - 1) << P0;
			H = (H & ~lmask) | ((H ^ (H >> HALF)) & lmask);
		}
	}
};

// ---- the 64-leaf block: levels 6..4 as a loop over its eight 8-leaf sub-trees (the step pattern of the loops above it), the
// rest in ScBlock8.  fz0 / fz1: frozen bits of leaves 0..31 / 32..63.  Returns the partial sum of the lane's own position.
__device__ __forceinline__ int sc_walk64(float r6, uint32_t fz0, uint32_t fz1, ScAcc &acc, int lane, int q)
{
	unsigned long long H = 0;
	float r5 = 0.f, r4 = 0.f;
	#pragma unroll 1
	for (int b8 = 0, adv8 = 1; b8 < 8; b8 += adv8) {
		adv8 = 1;
		const int z8 = b8 ? __builtin_ctz(b8) + 3 : 6;            // the level whose array a g step produces here (6: none, all f)
		const uint32_t fzw = b8 < 4 ? fz0 : fz1;
		const int sh = (b8 & 3) * 8;
		int L8 = 0;
		float rn = 0.f;                                           // the uniform node's own value
		if (z8 >= 5) {
			r5 = z8 == 5 ? sc_g_cross<6>(r6, (int)((uint32_t)H >> (q & 31)) & 1, lane, q) : f_minsum(r6, xpos<32>(r6, lane));
			if (fzw == 0xffffffffu) {
				acc.M += sc_pen_sum<5>(sc_pen(r5), lane);
				L8 = 5;
			} else if (fzw == 0u) {
				acc.info(sc_min_mag<5>(sc_mag(r5), lane));
				L8 = 5;
				rn = r5;
			}
		}
		if (z8 >= 4 && !L8) {
			r4 = z8 == 4 ? sc_g_cross<5>(r5, (int)(H >> ((b8 - 2) * 8 + (q & 15))) & 1, lane, q) : f_minsum(r5, xpos<16>(r5, lane));
			const uint32_t pat = (fzw >> sh) & 0xffffu;
			if (pat == 0xffffu) {
				acc.M += sc_pen_sum<4>(sc_pen(r4), lane);
				L8 = 4;
			} else if (pat == 0u) {
				acc.info(sc_min_mag<4>(sc_mag(r4), lane));
				L8 = 4;
				rn = r4;
			}
		}
		if (!L8) {
			const float r3 = z8 == 3 ? sc_g_cross<4>(r4, (int)(H >> ((b8 - 1) * 8 + (q & 7))) & 1, lane, q) : f_minsum(r4, xpos<8>(r4, lane));
			int H8 = 0;
			ScBlock8 blk{ acc, H8, { 0.f, 0.f, 0.f, r3 }, (fzw >> sh) & 0xffu, lane, q };
			blk.node<3, 0>();
			H |= (unsigned long long)(uint32_t)H8 << (b8 * 8);
		} else {
			adv8 = 1 << (L8 - 3);
			if (rn < 0.f)                                         // all-information: the node's partial sums are its sign bits (own position)
				H |= (L8 == 5 ? 0xffffffffull : 0xffffull) << (b8 * 8);
		}
		const int bn8 = b8 + adv8;
		for (int m = L8 ? L8 + 1 : 4; m <= 6 && (bn8 & ((1 << (m - 3)) - 1)) == 0; ++m) {
			const int half = 1 << (m - 1), st = bn8 * 8 - 2 * half;
			const unsigned long long lmask = ((1ull << half) - 1ull) << st;
			H = (H & ~lmask) | ((H ^ (H >> half)) & lmask);
		}
	}
	return (int)(H >> q) & 1;
}

// ---- the level store: input arrays of the current nodes of 2^15 | 2^14 | 2^13 leaves
constexpr int SC_STORE_FLOATS = 32768 + 16384 + 8192;
__host__ __device__ constexpr int sc_off(int m) { return m == 15 ? 0 : (m == 14 ? 32768 * 4 : (32768 + 16384) * 4); }   // bytes

// One pass over the top of the tree: the array of the 4096-leaf node that starts at sub-tree s, from the level D above it.
//   KIND 0: f chain from the channel LLRs (s = 0, D = 4)   KIND 1: g of the channel LLRs, then f (s = 8, D = 4)
//   KIND 2: g of level 12 + D of the level store, then f (s = 4, 12: D = 3; s = 2 mod 4: D = 2; odd s: D = 1)
// A lane owns the columns x * 64 + lane (x = 0..63) of EVERY level: the chain below the first step is lane-local, each level
// between is written once (levels >= 13 to the store, level 12 to LDS).  g takes the left child's partial sums from the
// published words: lane l fetches word l of each of its sub-trees once, the loop picks word x with v_readlane.
// KIND 0 also leaves the hard decisions of the channel LLRs (xw, bit-packed like the codeword) for the flip count, and
// checks that every LLR is finite and small enough that no sum of 65536 of them overflows.
template <int LEV, int N> __device__ __forceinline__ void sc_emit(rsrc_t soft, float *a12, int voff, int idx, float (&t)[N])
{
	if constexpr (LEV >= 13) {
		#pragma unroll
		for (int k = 0; k < N; ++k)
			bstore<2>(soft, voff, sc_off(LEV) + k * 16384, t[k]);
		float u[N / 2];
		#pragma unroll
		for (int k = 0; k < N / 2; ++k)
			u[k] = f_minsum(t[k], t[k + N / 2]);
		sc_emit<LEV - 1, N / 2>(soft, a12, voff, idx, u);
	} else
		a12[idx] = t[0];
}
template <int D, int KIND>
__device__ __forceinline__ void sc_top_pass(rsrc_t soft, rsrc_t llr, float *a12, const unsigned long long *cw, unsigned long long *xw, int s, int lane,
	bool &finite)
{
	constexpr int NS = 1 << D, NH = NS / 2, XB = NS >= 16 ? 1 : 16 / NS;
	const rsrc_t src = KIND == 2 ? soft : llr;
	constexpr int src_off = KIND == 2 ? sc_off(12 + D) : 0;
	uint32_t wl[NH], wh[NH];
	if (KIND) {
		#pragma unroll
		for (int k = 0; k < NH; ++k) {
			const unsigned long long w = cw[(s - NH + k) * 64 + lane];
			wl[k] = (uint32_t)w;
			wh[k] = (uint32_t)(w >> 32);
		}
	}
	unsigned long long X[KIND == 0 ? NS : 1];
	const int sh = lane & 31;
	const bool up = lane >= 32;
	int voff = lane * 4;
	#pragma unroll 1
	for (int x0 = 0; x0 < 64; x0 += XB, voff += XB * 256) {
		float v[XB][NS];
		#pragma unroll
		for (int xb = 0; xb < XB; ++xb)
			#pragma unroll
			for (int k = 0; k < NS; ++k)
				v[xb][k] = KIND == 0 ? bload<0>(src, voff + xb * 256, src_off + k * 16384) : bload<2>(src, voff + xb * 256, src_off + k * 16384);
		#pragma unroll
		for (int xb = 0; xb < XB; ++xb) {
			const int x = x0 + xb;
			float t[NH];
			#pragma unroll
			for (int k = 0; k < NH; ++k) {
				if (KIND == 0)
					t[k] = f_minsum(v[xb][k], v[xb][k + NH]);
				else {
					const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)wl[k], x), hi = (uint32_t)__builtin_amdgcn_readlane((int)wh[k], x);
					t[k] = g_add(v[xb][k], v[xb][k + NH], (int)(((up ? hi : lo) >> sh) & 1u));
				}
			}
			if (KIND == 0) {
				#pragma unroll
				for (int k = 0; k < NS; ++k) {
					const unsigned long long bal = __ballot(v[xb][k] < 0.f);
					if (lane == x)
						X[k] = bal;
					finite &= sc_mag(v[xb][k]) < 0x71000000u;         // |llr| < 6e29 (and not a NaN)
				}
			}
			sc_emit<11 + D, NH>(soft, a12, voff + xb * 256, x * 64 + lane, t);
		}
	}
	if (KIND == 0) {
		#pragma unroll
		for (int k = 0; k < NS; ++k)
			xw[k * 64 + lane] = X[k];
	}
}

// a uniform node of 64 * CNT leaves on its register array (element x = position x * 64 + q)
template <int CNT> __device__ __forceinline__ void sc_rate0(const float (&r)[CNT], ScAcc &acc, int lane)
{
	float pz[CNT];
	#pragma unroll
	for (int x = 0; x < CNT; ++x)
		pz[x] = sc_pen(r[x]);
	#pragma unroll
	for (int hx = CNT / 2; hx >= 1; hx >>= 1)
		#pragma unroll
		for (int x = 0; x < hx; ++x)
			pz[x] = pz[x] + pz[x + hx];
	acc.M += sc_pen_sum<6>(pz[0], lane);
}
template <int CNT> __device__ __forceinline__ uint32_t sc_rate1(const float (&r)[CNT], ScAcc &acc, int lane)
{
	uint32_t mu = 0x7f800000u, bits = 0;
	#pragma unroll
	for (int x = 0; x < CNT; ++x) {
		mu = min(mu, sc_mag(r[x]));
		bits |= (r[x] < 0.f ? 1u : 0u) << x;
	}
	acc.info(sc_min_mag<6>(mu, lane));
	return bits;
}
template <int CNT> __device__ __forceinline__ void sc_f_half(float (&dst)[CNT], const float (&src)[2 * CNT])
{
	#pragma unroll
	for (int x = 0; x < CNT; ++x)
		dst[x] = f_minsum(src[x], src[x + CNT]);
}
template <int CNT> __device__ __forceinline__ void sc_g_half(float (&dst)[CNT], const float (&src)[2 * CNT], uint32_t hb)
{
	#pragma unroll
	for (int x = 0; x < CNT; ++x)
		dst[x] = g_add(src[x], src[x + CNT], (int)((hb >> x) & 1u));
}

// Persistent grid: workgroup = one wave = one decoder with its own level store; decoders take codewords from the run's counter.
__global__ __launch_bounds__(64) void k_sc(ListQueue *__restrict__ q, const ListSlot *__restrict__ slots, const float *__restrict__ llr_q,
	float *__restrict__ soft_all, unsigned long long *__restrict__ cw_q, unsigned long long *__restrict__ xw_q, ScStat *__restrict__ stat_q,
	const uint32_t *__restrict__ frozen2, const uint8_t *__restrict__ node_lev64)
{
	const int lane = threadIdx.x, qp = sc_pos(lane);
	__shared__ float a12[4096];                                   // the array of the current 4096-leaf node, [x][position]
	const int par = 0;
	const int run_n = (int)q->run_n[par];
	const unsigned run_head = q->run_head[par], cap = q->cap;
	if (run_n == 0)
		return;
	const rsrc_t soft = make_rsrc(soft_all + (size_t)blockIdx.x * SC_STORE_FLOATS, SC_STORE_FLOATS * 4);
	for (;;) {
		int unit = 0;
		if (lane == 0)
			unit = atomicAdd(&q->next_unit[par], 1);
		unit = __builtin_amdgcn_readfirstlane(unit);
		if (unit >= run_n)
			break;
		const int slot = (int)((run_head + (unsigned)unit) % cap);
		const int tab = slots[slot].oper_mode >= 10;                   // decode.cc:312,344
		const uint32_t *frozen = frozen2 + (tab ? 2048 : 0);
		const uint8_t *nlev = node_lev64 + (tab ? 1024 : 0);
		const rsrc_t llr = make_rsrc(llr_q + (size_t)slot * CODE_LEN, CODE_LEN * 4);
		unsigned long long *cw = cw_q + (size_t)slot * (CODE_LEN / 64), *xw = xw_q + (size_t)slot * (CODE_LEN / 64);
		ScAcc acc{ 0.f, 0x7f800000u };
		bool finite = true;
		#pragma unroll 1
		for (int s = 0; s < 16; ++s) {
			// ---------------- the array of this 4096-leaf node into LDS, through the level store
			if (s == 0) sc_top_pass<4, 0>(soft, llr, a12, cw, xw, s, lane, finite);
			else if (s == 8) sc_top_pass<4, 1>(soft, llr, a12, cw, xw, s, lane, finite);
			else if ((s & 3) == 0) sc_top_pass<3, 2>(soft, llr, a12, cw, xw, s, lane, finite);
			else if ((s & 1) == 0) sc_top_pass<2, 2>(soft, llr, a12, cw, xw, s, lane, finite);
			else sc_top_pass<1, 2>(soft, llr, a12, cw, xw, s, lane, finite);
			SC_WAVE_ORDER();
			// this node's 64 table bytes and frozen words, one block per lane; the block loop reads them with v_readlane
			const int nlv = nlev[s * 64 + lane];
			const uint32_t fzl = frozen[(s * 64 + lane) * 2], fzh = frozen[(s * 64 + lane) * 2 + 1];
			unsigned long long HR = 0;                                // partial sums: bit x = position x * 64 + qp
			float R11[32], R10[16], R9[8], R8[4], R7[2], R6[1];
			#pragma unroll 1
			for (int b = 0, adv = 1; b < 64; b += adv) {
				adv = 1;
				const int zb = b ? __builtin_ctz(b) + 6 : 12;         // the level whose array the one g step of this block produces (12: none)
				const int nl = __builtin_amdgcn_readlane(nlv, b), nl0 = nl & 15, nl1 = nl >> 4;
				const int Lt = nl0 > nl1 ? nl0 : nl1;                 // the largest uniform node that starts here (0: none)
				const bool frz = nl0 > nl1;
				const uint32_t hb = zb < 12 ? (uint32_t)(HR >> (b - (1 << (zb - 6)))) : 0u;
				int L2 = 0;
				uint32_t bits = 0;
				if (zb >= 11) {
					if (zb == 11) {
						#pragma unroll
						for (int x = 0; x < 32; ++x)
							R11[x] = g_add(a12[x * 64 + qp], a12[(x + 32) * 64 + qp], (int)((hb >> x) & 1u));
					} else {
						#pragma unroll
						for (int x = 0; x < 32; ++x)
							R11[x] = f_minsum(a12[x * 64 + qp], a12[(x + 32) * 64 + qp]);
					}
					if (Lt == 11) { bits = sc_rate1(R11, acc, lane); L2 = 11; }
				}
				if (zb >= 10 && !L2) {
					if (zb == 10) sc_g_half(R10, R11, hb); else sc_f_half(R10, R11);
					if (Lt == 10) { bits = sc_rate1(R10, acc, lane); L2 = 10; }
				}
				if (zb >= 9 && !L2) {
					if (zb == 9) sc_g_half(R9, R10, hb); else sc_f_half(R9, R10);
					if (Lt == 9) { bits = sc_rate1(R9, acc, lane); L2 = 9; }
				}
				if (zb >= 8 && !L2) {
					if (zb == 8) sc_g_half(R8, R9, hb); else sc_f_half(R8, R9);
					if (Lt == 8) { bits = sc_rate1(R8, acc, lane); L2 = 8; }
				}
				if (zb >= 7 && !L2) {
					if (zb == 7) sc_g_half(R7, R8, hb); else sc_f_half(R7, R8);
					if (Lt == 7) {
						if (frz) sc_rate0(R7, acc, lane); else bits = sc_rate1(R7, acc, lane);
						L2 = 7;
					}
				}
				if (!L2) {
					if (zb == 6) sc_g_half(R6, R7, hb); else sc_f_half(R6, R7);
					if (Lt == 6) {
						if (frz) sc_rate0(R6, acc, lane); else bits = sc_rate1(R6, acc, lane);
						L2 = 6;
					}
				}
				if (L2) {
					adv = 1 << (L2 - 6);
					HR |= (unsigned long long)bits << b;
				} else {
					const uint32_t fz0 = (uint32_t)__builtin_amdgcn_readlane((int)fzl, b), fz1 = (uint32_t)__builtin_amdgcn_readlane((int)fzh, b);
					HR |= (unsigned long long)sc_walk64(R6[0], fz0, fz1, acc, lane, qp) << b;
				}
				// partial-sum combines of the nodes of 128..4096 leaves that end here: left half ^= right half
				const int bn = b + adv;
				for (int m = L2 ? L2 + 1 : 7; m <= 12 && (bn & ((1 << (m - 6)) - 1)) == 0; ++m) {
					const int half = 1 << (m - 7), b0 = bn - 2 * half;
					const unsigned long long lmask = ((1ull << half) - 1ull) << b0;
					HR = (HR & ~lmask) | ((HR ^ (HR >> half)) & lmask);
				}
			}
			// publish the node's 4096 partial sums as 64 plain words: position q sits on lane q ^ ((q & 4) ? 3 : 0); bring the
			// bits to their natural lanes (quad_perm [3,2,1,0] on the upper half of every 8), then one ballot per word
			{
				uint32_t h0 = (uint32_t)HR, h1 = (uint32_t)(HR >> 32);
				const uint32_t s0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h0, 0x1B, 0xf, 0xf, false);
				const uint32_t s1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h1, 0x1B, 0xf, 0xf, false);
				if (lane & 4) { h0 = s0; h1 = s1; }
				unsigned long long mine = 0;
				#pragma unroll
				for (int u = 0; u < 64; ++u) {
					const unsigned long long bal = __ballot(((u < 32 ? h0 >> u : h1 >> (u - 32)) & 1u) != 0u);
					if (lane == u)
						mine = bal;
				}
				cw[s * 64 + lane] = mine;
			}
			// combines of the nodes of 2^13 .. 2^16 leaves that end here, on the published words (every word stays with its lane)
			const int sn = s + 1;
			for (int m = 13; m <= 16 && (sn & ((1 << (m - 12)) - 1)) == 0; ++m) {
				const int halfw = 64 << (m - 13), w0 = sn * 64 - 2 * halfw;
				for (int w = lane; w < halfw; w += 64)
					cw[w0 + w] ^= cw[w0 + halfw + w];
			}
			SC_WAVE_ORDER();
		}
		const bool all_finite = __ballot(!finite) == 0;
		if (lane == 0) {
			ScStat st;
			st.metric = acc.M;
			st.min_fork = __uint_as_float(acc.fork);
			st.ok = (all_finite && __uint_as_float(acc.fork) > acc.M) ? 1 : 0;   // the rule (a NaN compares false)
			st.pad = 0;
			stat_q[slot] = st;
		}
		SC_WAVE_ORDER();
	}
}

// ---------------------------------------------------------------- k_sc_finish: decode.cc:532-555 for the frames k_sc decided
// One workgroup per entry of the run.  Rule holds: P* is lane 0 - its systematic bits (P*'s codeword at the unfrozen positions,
// decode.cc:254-261), their CRC-32 (decode.cc:533-541); CRC zero: the reference takes lane 0 - payload (descrambled,
// decode.cc:613-615), best_lane 0, the flip count against the channel's hard decisions (decode.cc:546-555): the frame is finished.
// Rule or CRC fails: the frame takes a slot of the list decoder's queue and its LLRs are copied there - the general path, unchanged.
// slot_of[frame]: -2 - (slot in the SC ring) for a frame finished here (its LLRs stay there until the next chunk), else its slot
// in the list decoder's queue.
__global__ __launch_bounds__(256) void k_sc_finish(ListQueue *__restrict__ qs, const ListSlot *__restrict__ slots_s, const float *__restrict__ llr_s,
	const unsigned long long *__restrict__ cw_q, const unsigned long long *__restrict__ xw_q, const ScStat *__restrict__ stat_q, Tables tb, int descramble,
	ListQueue *__restrict__ ql, ListSlot *__restrict__ slots_l, float *__restrict__ llr_l, int *__restrict__ slot_of)
{
	const int rel = blockIdx.x, tid = threadIdx.x;
	const unsigned run_n = qs->run_n[0], run_head = qs->run_head[0], cap = qs->cap;
	if ((unsigned)rel >= run_n)
		return;
	const int slot = (int)((run_head + (unsigned)rel) % cap);
	const ListSlot ls = slots_s[slot];
	const ScStat st = stat_q[slot];
	__shared__ uint32_t bits[CODE_LEN / 32];
	__shared__ uint8_t mesg[MESG_BYTES_MAX];
	__shared__ uint32_t ctab[256], csh[1024], cpart[32];
	__shared__ uint32_t crc_sh;
	__shared__ int flips_red[4], slot_sh;
	const ModeDesc md = mode_desc(ls.oper_mode);
	bool done = false;
	if (st.ok) {
		const uint32_t *cw = (const uint32_t *)(cw_q + (size_t)slot * (CODE_LEN / 64));
		for (int w = tid; w < CODE_LEN / 32; w += 256)
			bits[w] = cw[w];
		ctab[tid] = tb.crc32_tab[tid];
		#pragma unroll
		for (int w = 0; w < 4; ++w)
			csh[tid + 256 * w] = tb.crc32_shift168[tid + 256 * w];
		__syncthreads();
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
		__syncthreads();
		constexpr int SEG = 168, NSEG = 32, TAIL = CRC_BITS / 8 - SEG * NSEG;   // 5384 = 32 * 168 + 8 (k_finish's scheme)
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
			for (int e = 0; e < NSEG; ++e) {
				crc = csh[crc & 255] ^ csh[256 + ((crc >> 8) & 255)] ^ csh[512 + ((crc >> 16) & 255)] ^ csh[768 + (crc >> 24)];
				crc ^= cpart[e];
			}
			for (int i = SEG * NSEG; i < SEG * NSEG + TAIL; ++i)
				crc = (crc >> 8) ^ ctab[(crc ^ mesg[i]) & 255];
			crc_sh = crc;
		}
		__syncthreads();
		done = crc_sh == 0;
	}
	if (done) {
		// decode.cc:546-554: received hard decision against decoded bit over the data bits = the unfrozen positions below the one
		// of message bit DATA_BITS
		const uint32_t *xwp = (const uint32_t *)(xw_q + (size_t)slot * (CODE_LEN / 64));
		const uint32_t *frozen = tb.frozen + (md.table ? CODE_LEN / 32 : 0);
		const int p_end = (tb.info_pos + (md.table ? MESG_BITS_MAX : 0))[DATA_BITS];
		int flips = 0;
		for (int w = tid; w * 32 < p_end; w += 256) {
			uint32_t m = ~frozen[w];
			if (p_end - w * 32 < 32)
				m &= (1u << (p_end - w * 32)) - 1u;
			flips += __builtin_popcount((bits[w] ^ xwp[w]) & m);
		}
		#pragma unroll
		for (int m = 32; m; m >>= 1)
			flips += __shfl_xor(flips, m);
		if ((tid & 63) == 0)
			flips_red[tid >> 6] = flips;
		for (int i = tid; i < PAYLOAD_BYTES; i += 256)
			ls.payload_now[i] = mesg[i] ^ (descramble ? tb.scramble[i] : (uint8_t)0);
		__syncthreads();
		if (tid == 0) {
			ls.res_now->best_lane = 0;
			ls.res_now->bit_flips = flips_red[0] + flips_red[1] + flips_red[2] + flips_red[3];
			slot_of[ls.frame] = -2 - slot;
			atomicAdd(&qs->certified, 1u);
			atomicAdd(&qs->done_total, 1u);
		}
		return;
	}
	if (tid == 0) {
		const unsigned e = atomicAdd(&ql->tail, 1u);
		const int lslot = (int)(e % ql->cap);
		slots_l[lslot] = ls;
		slot_of[ls.frame] = lslot;
		slot_sh = lslot;
	}
	__syncthreads();
	const float4 *src = (const float4 *)(llr_s + (size_t)slot * CODE_LEN);
	float4 *dst = (float4 *)(llr_l + (size_t)slot_sh * CODE_LEN);
	for (int i = tid; i < CODE_LEN / 4; i += 256)
		dst[i] = src[i];
}

// behind k_back of a chunk: the run of k_sc = everything that chunk put into the SC ring; behind k_sc_finish: the adaptive switch.
// Where k_sc decides few frames (below about -18.5 dB every path metric outgrows min_fork) its pass is spent for nothing:
// k_back then sends a probe sample only (one frame in sixteen) and the rest straight to the list decoder, and all of them
// again when an eighth of the sample is decided.  Either way every decision is exact: the list decoder is the general path.
__global__ void k_sc_plan(ListQueue *__restrict__ qs)
{
	const unsigned head = qs->head, n = qs->tail - head;
	qs->run_head[0] = head;
	qs->run_n[0] = n;
	qs->head = head + n;
	qs->next_unit[0] = 0;
	qs->tried = n;
	qs->certified = 0;
}
__global__ void k_sc_adapt(ListQueue *__restrict__ qs)
{
	const unsigned tried = qs->tried, done = qs->certified;
	if (qs->cert_on) {
		if (tried >= 64 && done * 8 < tried)
			qs->cert_on = 0;
	} else if (tried >= 8 && done * 8 >= tried)
		qs->cert_on = 1;
}

void launch_sc(hipStream_t s, int grid, ListQueue *q, const ListSlot *slots, const float *llr_q, float *soft, unsigned long long *cw_q,
	unsigned long long *xw_q, ScStat *stat_q, Tables tb)
{
	hipLaunchKernelGGL(k_sc, dim3(grid), dim3(64), 0, s, q, slots, llr_q, soft, cw_q, xw_q, stat_q, tb.frozen, tb.node_lev64);
}
void launch_sc_finish(hipStream_t s, int max_entries, ListQueue *qs, const ListSlot *slots_s, const float *llr_s, const unsigned long long *cw_q,
	const unsigned long long *xw_q, const ScStat *stat_q, Tables tb, int descramble, ListQueue *ql, ListSlot *slots_l, float *llr_l, int *slot_of)
{
	hipLaunchKernelGGL(k_sc_finish, dim3(max_entries), dim3(256), 0, s, qs, slots_s, llr_s, cw_q, xw_q, stat_q, tb, descramble, ql, slots_l, llr_l, slot_of);
}
void launch_sc_plan(hipStream_t s, ListQueue *qs) { hipLaunchKernelGGL(k_sc_plan, dim3(1), dim3(1), 0, s, qs); }
void launch_sc_adapt(hipStream_t s, ListQueue *qs) { hipLaunchKernelGGL(k_sc_adapt, dim3(1), dim3(1), 0, s, qs); }
size_t sc_store_bytes() { return (size_t)SC_STORE_FLOATS * sizeof(float); }

}  // namespace rx
