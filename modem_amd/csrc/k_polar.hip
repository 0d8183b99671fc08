// k_polar.hip -- D9 (polar successive-cancellation list decoder, N = 65536, L = 8) for gfx950;
// D10 lives in k_finish.hip.
//
// CODE::PolarListDecoder<SIMD<float,8>,16> (decode.cc:201,530): min-sum SCL.
// One wavefront decodes one codeword.  Lane l = (j << 3) | k : k = list path (the
// reference's SIMD lane), j = one of 8 butterflies processed per wave instruction.
//   f(a,b)   = sign(a) sign(b) min(|a|,|b|)              left child LLRs
//   g(a,b,u) = u ? b - a : a + b                          right child LLRs
// Tree levels 9..15 live in HBM as soft[level m][i][k] (fp32, 2 MiB per resident decoder, the
// reference's own soft[N+i] layout); level 8 - the array of the current 256-leaf node - in LDS
// (8 KB per decoder); levels 4..7 in registers as the arrays of the current node of each level;
// level 16 is the shared channel LLR vector; levels 3..0 (8-leaf sub-trees) never leave
// registers either: their butterflies are DPP / permlane exchanges ("Where the tree lives" below).
// Lane permutations after a fork are applied lazily exactly like the reference's vshuf at
// the g step and at the partial-sum combine: per level the composition of all leaf maps
// since that level's node started is kept as 3-bit fields packed in two registers per
// lane (W0: levels 0..9, W1: levels 10..16), so one fork costs two cross-lane gathers.
// Partial sums are bits: hard[i] is one byte per code position, bit k = path k; bytes are
// assembled with __ballot.  The root's hard[] is the re-encoded codeword x = u F, and the
// systematic message of decode.cc:254-261 is x at the unfrozen positions - no separate
// re-encode, no message back-trace.
// Selection rule at an information leaf: the 2L candidates are ranked by (metric,
// candidate index 2k+u); survivors are stored in rank order (same rule as the CPU oracle).
#include "dev_common.h"
#include "kernels.h"
#include "polar_common.h"
#include <cstdio>

namespace rx {

// ---- cheap cross-lane exchanges (no LDS crossbar): lane ^ 8 (DPP row_ror:8), lane ^ 16 and
// lane ^ 32 (gfx950 v_permlane16_swap / v_permlane32_swap)
__device__ __forceinline__ int xor8_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); }
__device__ __forceinline__ int xor16_i(int v, int lane)
{
	auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
	return (int)((lane & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ int xor32_i(int v, int lane)
{
	auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
	return (int)((lane & 32) ? r[0] : r[1]);
}
// exchange with the lane whose butterfly index j differs in bit zz (zz = 0,1,2)
template <int ZZ> __device__ __forceinline__ float xj(float v, int lane)
{
	int i = __float_as_int(v);
	i = ZZ == 0 ? xor8_i(i) : (ZZ == 1 ? xor16_i(i, lane) : xor32_i(i, lane));
	return __int_as_float(i);
}
// min / max over the 8 paths (lanes k = 0..7 of a group); result in every lane.  Path metrics are
// non-negative floats, so their bit patterns order like unsigned integers: v_min_u32 / v_max_u32 fuse with the
// DPP operand (one instruction per step) where the float forms need a canonicalising v_max each.
// LN = 4: the list of the reference's non-AVX2 build (SIMD<float,4>, decode.cc:168) runs on the same 8-lane layout
// with paths 4..7 dead (metric +inf): the live paths are one quad, so the reduction stops after the quad steps.
template <int LN> __device__ __forceinline__ float group8_min(float vf)
{
	uint32_t v = __float_as_uint(vf);
	v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
	v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
	if (LN == 8)
		v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));   // row_half_mirror
	return __uint_as_float(v);
}
template <int LN> __device__ __forceinline__ float group8_max(float vf)
{
	uint32_t v = __float_as_uint(vf);
	v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));
	v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));
	if (LN == 8)
		v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));
	return __uint_as_float(v);
}

constexpr uint32_t ID0 = 0x09249249u;   // 10 fields of 3 bits, each = 1
constexpr uint32_t ID1 = 0x00049249u;   //  7 fields

struct Maps {
	uint32_t w0, w1;
	__device__ __forceinline__ int get(int m) const { return m < 10 ? (w0 >> (3 * m)) & 7 : (w1 >> (3 * (m - 10))) & 7; }
	// a node of every level <= c starts at this leaf: those fields become the identity
	__device__ __forceinline__ void reset_upto(int c, int k)
	{
		int n0 = (c < 9 ? c : 9) + 1;
		uint32_t m0 = (1u << (3 * n0)) - 1;
		w0 = (w0 & ~m0) | ((ID0 * (uint32_t)k) & m0);
		if (c >= 10) {
			uint32_t m1 = (1u << (3 * (c - 9))) - 1;
			w1 = (w1 & ~m1) | ((ID1 * (uint32_t)k) & m1);
		}
	}
};

// Where the tree lives (v11):
//   levels 9..15  HBM level store, soft[level m][i][k] (fp32, the reference's own soft[N+i] layout)
//   level  8      LDS, 8 KB per decoder: the array of the current 256-leaf node, [x][lane]
//   levels 4..7   registers: r7[16], r6[8], r5[4], r4[2] = the arrays of the CURRENT node of each level.  Position
//                 i = x*8 + j of a level-L array is element x of lane (j, k), and the partner of x at level L+1 is
//                 x + 2^(L-3): every f step below level 8 is lane-local with compile-time register indices, a g step
//                 adds one cross-lane gather per element only when the lane map is not the identity
//   levels 0..3   registers + DPP / permlane butterflies (Block8 below)
// Partial sums of the current 256-leaf node are one register per lane (HR: bit x = position x*8 + j, own path); their
// combines are shifts and masks; every finished 128-leaf block is published to the global byte array (bit k = path
// k) with 16 ballots.

// Fused pass: levels m, m-1, ..., m-D+1 (all >= 8) from level m+1 in one sweep.  A lane owns butterfly column
// (j, k) at EVERY level (position i = x*8 + j, local index x), and the partner of local index x at
// level L is x + 2^(L-4): the whole f-chain below the first step is lane-local, so the intermediate
// levels are produced in registers and each level is written exactly once, never re-read.
//   KIND 0: first step f from level m+1      KIND 1: first step g (partial sums hb, lane map gl)
//   KIND 2: first step f from the shared channel LLRs   KIND 3: first step g from the shared channel LLRs
// The NG highest produced levels are in global memory (gs = base of the codeword's soft array), a lower one can only
// be level 8, in LDS (ls8).  Level L starts at element 8 << L of the level store.
// Global accesses are raw buffer loads / stores: one per-lane byte offset in a VGPR (lane * 4, or the mapped
// lane for the g step), everything else (level base, column, partner distance) in the scalar offset - so 16
// loads in flight cost 16 data registers and no 64-bit address pairs.
#define POLAR_NT_LEVEL 10   // level-store accesses of levels >= this are non-temporal (written once, read once a long time later: keeping
                            // them out of the caches leaves room for level 9, which is re-read soon).  r02 sweep, k_polar alone at 16
                            // decoders per CU, ms per 65536 codewords: none 316, >= 13 312, >= 12 304, >= 11 300, >= 10 290, all 317
#ifndef POLAR_WAVES_PER_SIMD
#define POLAR_WAVES_PER_SIMD 5     // register budget: 5 waves per SIMD = 96 VGPRs (13 dwords of scratch) so that Theil-Sen workgroups of the next
                                   // chunk fit beside 12 resident decoders per CU; 1 = unconstrained (125 VGPRs): 2 % faster alone, 12 % slower overlapped
#endif
struct PolarBufs {
	rsrc_t soft, llr, hard;    // this decoder's 2 MiB level store, the channel LLRs, the partial-sum bytes of what it decodes
	// List 4 decodes TWO codewords per wave (paths 0..3 = the first, 4..7 = the second, see k_polar): everything that is
	// shared by the paths of ONE codeword - the channel LLRs and the compact arrays of the first descent - exists twice
	int half_sel;              // per lane: 1 for the lanes of the second codeword, else 0
	int llr_half;              // byte distance of the second codeword's channel LLRs
	int kmask;                 // (k & kmask) == 0: the lane that writes a compact array (7: one codeword, 3: two)
};

// Compact arrays.  Until the first fork all eight paths hold the same LLRs, so the first left descent (t = 0)
// computes eight identical copies of every level.  DST_C stores such a level ONCE (position-major, 2^L floats
// at the start of the level's region, written by the k = 0 lanes as full 32-byte sectors) and SRC_C reads it
// back with the channel-LLR indexing - at t = 0 for the next pass of the descent and at t = 2^z for the g step
// of the right sibling on the left spine, the only other reader.  Values are unchanged; the level store sees
// 1.75 MB fewer writes and ~2 MB fewer reads per codeword.
// Recomputed arrays.  The right sibling on the left spine (t = 2^z, level z) is g(C, partial sums) of a compact
// array C (level z+1; the channel LLRs for z = 15): 4 bytes of C per position serve all eight paths.  Its own
// level-z array is read exactly once more, by the g step of ITS right child at t = 3 * 2^(z-1).  SKIP0 does not
// store it; SRC_R rebuilds both inputs of that g step from C and the level-(z+1) node's left partial sums (bit =
// the path's ancestor at t = 2^z, the very lane map the stored array would have been read with) with the same
// g_add, so every value is bit-identical: 2 MB fewer writes and ~1.5 MB fewer reads per codeword.
//   SRC_R 1: C = compact level m+2      SRC_R 2: C = the channel LLRs (m = 14)
// TERM (D = 1 only): the produced level is the array of an all-information node of 512..2048 leaves that is DECIDED, not
// walked: nothing is stored; the pass returns the lane's smallest magnitude (mu, for the stable-list test) and writes
// the sign bits of all eight paths as the node's partial-sum bytes (bit k = path k) straight to the byte array.  If
// the test fails the caller runs the ordinary pass and walks the node; the bytes are overwritten then.
template <int D, int KIND, int NG, bool SRC_C = false, bool DST_C = false, bool SKIP0 = false, int SRC_R = 0, bool TERM = false>
__device__ __forceinline__ void fused_pass(const PolarBufs &pb, float *ls8, int hb_g_off, int m, int lane, int gl,
	uint32_t *mu_out = nullptr, uint8_t *hard_t = nullptr)
{
	static_assert(!TERM || (D == 1 && NG == 0), "terminal passes produce one level");
	uint32_t mu = 0x7f800000u;
	constexpr int NT = 1 << (D - 1);          // level-m values per column of the lowest produced level
#define POLAR_XB3 2
#define POLAR_XB2 4
#define POLAR_XB1 8
	constexpr int XB = SRC_R ? (D == 3 ? 1 : (D == 2 ? 2 : 4)) : (D == 3 ? POLAR_XB3 : (D == 2 ? POLAR_XB2 : POLAR_XB1));   // columns batched: XB * 2 * NT loads in flight
	const int S = 1 << (m - D + 1 - 3);       // local indices at the lowest produced level
	const int half = 1 << (m - 3);            // partner distance (local) at level m+1
	const int j = lane >> 3, k = lane & 7;
	const int src_off = (8 << (m + 1)) * 4;   // byte offset of level m+1 in the level store
	const int vo_lane = lane * 4, vo_src = (KIND == 1 ? gl : lane) * 4, vo_j = j * 4;
	// Addressing of the global accesses: the per-lane offset register advances with x0, the column xb inside a batch is
	// an immediate (xb * 256 < 4096) and everything that depends on s2 or the level is a loop-invariant scalar - no
	// scalar arithmetic per access inside the loop.
	constexpr int XS = (KIND >= 2 || SRC_C || SRC_R) ? 32 : 256;      // source bytes per local index
	const rsrc_t C = (KIND >= 2 || SRC_R == 2) ? pb.llr : pb.soft;
	const int c_off = (KIND >= 2 || SRC_R == 2) ? 0 : (SRC_R ? (8 << (m + 2)) * 4 : src_off);
	int so_a[NT], so_h[NT], so_d[D][NT];
	#pragma unroll
	for (int s2 = 0; s2 < NT; ++s2) {
		so_a[s2] = c_off + s2 * S * XS;
		so_h[s2] = hb_g_off + s2 * S * 8;
		#pragma unroll
		for (int d = 0; d < D; ++d)
			so_d[d][s2] = ((8 << (m - d)) + s2 * S * (DST_C ? 8 : 64)) * 4;
	}
	const int hx = half * XS;
	int v_src = (KIND >= 2 || SRC_C || SRC_R) ? vo_j : vo_src, v_dst = DST_C ? vo_j : vo_lane, v_h = j;
	if (KIND >= 2 || SRC_R == 2)
		v_src += pb.half_sel * pb.llr_half;                   // the second codeword's channel LLRs
	else if (SRC_C || SRC_R == 1)
		v_src += pb.half_sel << ((SRC_R ? m + 2 : m + 1) + 2);   // its compact array sits behind the first one's (2^level floats)
	const int anc = gl & 7;
	#pragma unroll 1
	for (int x0 = 0; x0 < S; x0 += XB, v_src += XB * XS, v_dst += XB * (DST_C ? 32 : 256), v_h += XB * 8) {
		float a[XB][NT], b[XB][NT];
		int h[XB][NT];
		#pragma unroll
		for (int xb = 0; xb < XB; ++xb)
			#pragma unroll
			for (int s2 = 0; s2 < NT; ++s2)
				if (x0 + xb < S) {
					if (SRC_R) {
						const float a1 = bload(C, v_src + xb * XS, so_a[s2]), a2 = bload(C, v_src + xb * XS, so_a[s2] + 2 * hx);
						const float b1 = bload(C, v_src + xb * XS, so_a[s2] + hx), b2 = bload(C, v_src + xb * XS, so_a[s2] + 3 * hx);
						const int ha = bload_u8(pb.hard, v_h + xb * 8, so_h[s2] - (2 << m));
						const int hb = bload_u8(pb.hard, v_h + xb * 8, so_h[s2] - (2 << m) + half * 8);
						a[xb][s2] = g_add(a1, a2, (ha >> anc) & 1);
						b[xb][s2] = g_add(b1, b2, (hb >> anc) & 1);
					} else {
						if (KIND < 2 && !SRC_C && m + 1 >= POLAR_NT_LEVEL) {      // loop-invariant: two copies of the load group
							a[xb][s2] = bload<2>(C, v_src + xb * XS, so_a[s2]);
							b[xb][s2] = bload<2>(C, v_src + xb * XS, so_a[s2] + hx);
						} else {
							a[xb][s2] = bload<0>(C, v_src + xb * XS, so_a[s2]);
							b[xb][s2] = bload<0>(C, v_src + xb * XS, so_a[s2] + hx);
						}
					}
					if (KIND & 1)
						h[xb][s2] = bload_u8(pb.hard, v_h + xb * 8, so_h[s2]);
				}
		unsigned long long mine = 0;
		#pragma unroll
		for (int xb = 0; xb < XB; ++xb)
			if (x0 + xb < S) {
				float v[NT];
				#pragma unroll
				for (int s2 = 0; s2 < NT; ++s2) {
					v[s2] = (KIND & 1) ? g_add(a[xb][s2], b[xb][s2], (h[xb][s2] >> k) & 1) : f_minsum(a[xb][s2], b[xb][s2]);
					if (TERM) {
						mu = min(mu, __float_as_uint(v[s2]) & 0x7fffffffu);
						const unsigned long long bal = __ballot(v[s2] < 0.f);
						if (lane == xb)
							mine = bal;
					} else if (SKIP0) {
					} else if (NG > 0) {
						if (!DST_C) { if (m >= POLAR_NT_LEVEL) bstore<2>(pb.soft, v_dst + xb * 256, so_d[0][s2], v[s2]); else bstore<0>(pb.soft, v_dst + xb * 256, so_d[0][s2], v[s2]); }
						else if ((k & pb.kmask) == 0) bstore(pb.soft, v_dst + xb * 32 + (pb.half_sel << (m + 2)), so_d[0][s2], v[s2]);
					} else ls8[(x0 + xb + s2 * S) * 64 + lane] = v[s2];
				}
				#pragma unroll
				for (int d = 1; d < D; ++d) {
					const int n = NT >> d;
					#pragma unroll
					for (int s2 = 0; s2 < n; ++s2) {
						v[s2] = f_minsum(v[s2], v[s2 + n]);
						if (NG > d) {
							if (!DST_C) { if (m - d >= POLAR_NT_LEVEL) bstore<2>(pb.soft, v_dst + xb * 256, so_d[d][s2], v[s2]); else bstore<0>(pb.soft, v_dst + xb * 256, so_d[d][s2], v[s2]); }
							else if ((k & pb.kmask) == 0) bstore(pb.soft, v_dst + xb * 32 + (pb.half_sel << (m - d + 2)), so_d[d][s2], v[s2]);
						} else ls8[(x0 + xb + s2 * S) * 64 + lane] = v[s2];
					}
				}
			}
		if (TERM && lane < XB && x0 + lane < S)               // XB positions x 8 bytes
			*(unsigned long long *)(hard_t + (size_t)(x0 + lane) * 8) = mine;
	}
	if (TERM)
		*mu_out = mu;
}


// ---- the 8-leaf sub-tree in registers ------------------------------------------------------------------------
// r[L] = this lane's LLR at level L of the node being decoded (position = the low L bits of j, duplicated over the
// other bits of j); H = partial sums of the 8 leaves, one bit per position, for THIS lane's path - only the bit of
// the lane's own position (mod the node size) is ever consumed, and the combines keep exactly that bit right.
//
// node<LV, P0>() decodes the 2^LV leaves from P0 top-down:
//   all frozen      -> rate-0: with min-sum the leaf penalties of a sub-tree add up to sum max(0, -llr) over the node's
//                      OWN inputs (max(0,-f(a,b)) + max(0,-(a+b)) = max(0,-a) + max(0,-b) by cases, then induction);
//                      summed in the butterfly halving order the oracle fixes (the reference's -Ofast build leaves the
//                      order open) and added to the metric once;
//   all information -> rate-1: if the list is sorted and max_k M_k < min_k (M_k + mu_k), mu_k = the smallest magnitude
//                      of path k's node LLRs, then EVERY leaf below takes the stable-list fast path (min-sum leaf
//                      magnitudes never drop under mu_k: f keeps the smaller input magnitude, g adds two same-signed
//                      terms once the left decisions are sign decisions), so no metric changes, no path is replaced and
//                      the node's partial sums are the sign bits of its LLRs.  Bit-identical to the leaf walk.
//                      If the condition fails the node is walked (its children get their own chance);
//   otherwise       -> left child (f), right child (g with the left partial sums and the lane map), combine.
template <int LN> struct Block8 {
	float &M;
	Maps &A;
	int &H;
	float r[4];
	const uint32_t fz;
	const int t, lane, j, k;
	const bool hi_alive;       // list 4: paths 4..7 carry a second codeword (else they are dead: metric +inf)

	template <int B> __device__ __forceinline__ float xjb(float v) const { return xj<B>(v, lane); }
	__device__ __forceinline__ void reset_at(int p0) { const int tt = t + p0; A.reset_upto(tt ? __builtin_ctz(tt) : 16, k); }
	__device__ __forceinline__ bool list_is_stable(float mu) const
	{
		const float P = M + mu;
		const float Mprev = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(M), __float_as_int(M), 0x111, 0xf, 0xf, false));   // row_shr:1
		const bool dead = LN == 4 && k >= 4 && !hi_alive, first = (k & (LN - 1)) == 0;   // list 4: each quad is its own list
		const bool ok = dead || ((first || Mprev <= M) && group8_max<LN>(M) < group8_min<LN>(P));
		return __ballot(!ok) == 0;
	}

	template <int P0> __device__ __forceinline__ void leaf()
	{
		reset_at(P0);
		const float r0 = r[0];
		int ubit = 0;
		if ((fz >> P0) & 1) {
			if (r0 < 0.f)
				M -= r0;
		} else if (list_is_stable(fabsf(r0))) {
			// stable-list fast path: paths already sorted by metric and every penalised continuation is worse
			// than every free one -> each path just takes its own sign bit
			ubit = r0 < 0.f;
		} else {
			float v0 = M, v1 = M;
			if (r0 < 0.f) v0 = M - r0; else v1 = M + r0;
			const int umine = j & 1;
			const float val = umine ? v1 : v0;
			const int cidx = 2 * k + umine;
			int rank = 0;
			#pragma unroll
			for (int kk = 0; kk < 8; ++kk) {
				const float o0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v0), kk));
				const float o1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v1), kk));
				const int mine = LN == 8 || (kk >> 2) == (k >> 2);   // list 4: the candidates of the lane's own codeword only
				rank += mine & ((o0 < val) | ((o0 == val) & (2 * kk < cidx)));
				rank += mine & ((o1 < val) | ((o1 == val) & (2 * kk + 1 < cidx)));
			}
			// every row of 16 lanes holds the 16 candidates at position (u << 3) | k: scatter inside the row.  List 4: the
			// four survivors of each codeword go to its own four positions, the four losers to the upper half of the row
			const int pos = LN == 8 ? rank : (rank < 4 ? ((k & 4) | rank) : (8 | (k & 4) | (rank - 4)));
			const int dst = (lane & 48) | pos;
			int rc = __builtin_amdgcn_ds_permute(dst << 2, cidx);
			int rv = __builtin_amdgcn_ds_permute(dst << 2, __float_as_int(val));
			const int rcx = xor8_i(rc), rvx = xor8_i(rv);   // odd j sit at row position 8+k: take position k
			if (j & 1) { rc = rcx; rv = rvx; }
			M = __int_as_float(rv);                               // (a dead quad ranks its +inf metrics among themselves)
			const int parent = rc >> 1;
			ubit = rc & 1;
			if (__ballot(parent != k)) {
				A.w0 = __shfl((int)A.w0, (j << 3) | parent);
				A.w1 = __shfl((int)A.w1, (j << 3) | parent);
			}
		}
		H = (H & ~(1 << P0)) | (ubit << P0);
	}

	// ---- the same walk for a frozen pattern known at compile time, under the assumption that no fork of the block
	// reorders the list (every information leaf / node passes the stable-list test): then the lane maps of levels 0..2 stay
	// the identity, every gather and every map reset inside the block is a no-op, and the block is straight-line code - f / g
	// butterflies, frozen penalties and sign bits, with the stable-list tests ORed into one flag.  Same arithmetic in the
	// same order as node<3, 0>(); the caller checks the flag once and, if a test failed anywhere, restores the metric and
	// walks the block with node<3, 0>() (every speculative result is lane-local and thrown away).
	__device__ __forceinline__ bool unstable_lane(float mu) const
	{
		const float P = M + mu;
		const float Mprev = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(M), __float_as_int(M), 0x111, 0xf, 0xf, false));   // row_shr:1
		const bool dead = LN == 4 && k >= 4 && !hi_alive, first = (k & (LN - 1)) == 0;
		return !(dead || ((first || Mprev <= M) && group8_max<LN>(M) < group8_min<LN>(P)));
	}
	template <uint32_t FZ, int LV, int P0> __device__ __forceinline__ void fast_node(bool &bad)
	{
		constexpr int N = 1 << LV, HALF = N / 2;
		constexpr uint32_t MASK = ((1u << N) - 1u) << P0, pat = FZ & MASK;
		if constexpr (pat == MASK) {
			if constexpr (LV == 0) {
				if (r[0] < 0.f)
					M -= r[0];
			} else {
				float pen = r[LV] < 0.f ? -r[LV] : 0.f;
				if constexpr (LV >= 3) pen = pen + xjb<2>(pen);
				if constexpr (LV >= 2) pen = pen + xjb<1>(pen);
				pen = pen + xjb<0>(pen);
				M += pen;
			}
		} else if constexpr (pat == 0) {
			uint32_t mu = __float_as_uint(r[LV]) & 0x7fffffffu;
			if constexpr (LV >= 3) mu = min(mu, (uint32_t)xor32_i((int)mu, lane));
			if constexpr (LV >= 2) mu = min(mu, (uint32_t)xor16_i((int)mu, lane));
			if constexpr (LV >= 1) mu = min(mu, (uint32_t)xor8_i((int)mu));
			bad |= unstable_lane(__uint_as_float(mu));
			H |= r[LV] < 0.f ? (int)MASK : 0;
		} else {
			r[LV - 1] = f_minsum(r[LV], xjb<LV - 1>(r[LV]));
			fast_node<FZ, LV - 1, P0>(bad);
			{
				const float own = r[LV], oth = xjb<LV - 1>(own);
				const bool hi = (j >> (LV - 1)) & 1;
				const float a = hi ? oth : own, b = hi ? own : oth;
				const int ub = (H >> (P0 + (j & (HALF - 1)))) & 1;
				r[LV - 1] = g_add(a, b, ub);
			}
			fast_node<FZ, LV - 1, P0 + HALF>(bad);
			constexpr int lmask = ((1 << HALF) - 1) << P0;
			H = (H & ~lmask) | ((H ^ (H >> HALF)) & lmask);
		}
	}
	// true: the block is decided (H, M final); false: nothing has changed, walk it
	template <uint32_t FZ> __device__ __forceinline__ bool fast_block()
	{
		const float M0 = M, r3 = r[3];
		reset_at(0);
		H = 0;
		bool bad = false;
		fast_node<FZ, 3, 0>(bad);
		if (__ballot(bad) == 0)
			return true;
		M = M0;
		H = 0;
		r[3] = r3;
		return false;
	}

	template <int LV, int P0> __device__ __forceinline__ void node()
	{
		if constexpr (LV == 0) {
			leaf<P0>();
		} else {
			constexpr int N = 1 << LV, HALF = N / 2;
			constexpr uint32_t MASK = ((1u << N) - 1u) << P0;
			const uint32_t pat = fz & MASK;
			if (pat == MASK) {                                    // rate-0
				reset_at(P0);
				float pen = r[LV] < 0.f ? -r[LV] : 0.f;
				if constexpr (LV >= 3) pen = pen + xjb<2>(pen);
				if constexpr (LV >= 2) pen = pen + xjb<1>(pen);
				pen = pen + xjb<0>(pen);
				M += pen;
				H &= ~(int)MASK;
				return;
			}
			if (pat == 0) {                                       // rate-1, if provably the same as the walk
				uint32_t mu = __float_as_uint(r[LV]) & 0x7fffffffu;
				if constexpr (LV >= 3) mu = min(mu, (uint32_t)xor32_i((int)mu, lane));
				if constexpr (LV >= 2) mu = min(mu, (uint32_t)xor16_i((int)mu, lane));
				mu = min(mu, (uint32_t)xor8_i((int)mu));
				if (list_is_stable(__uint_as_float(mu))) {
					reset_at(P0);
					H = (H & ~(int)MASK) | (r[LV] < 0.f ? (int)MASK : 0);   // own-position bit = sign bit
					return;
				}
			}
			// left child
			r[LV - 1] = f_minsum(r[LV], xjb<LV - 1>(r[LV]));
			node<LV - 1, P0>();
			// right child: g at level LV with the left child's partial sums; the lane map since this node started
			{
				const int lk = A.get(LV);
				float own = r[LV];
				if (__ballot(lk != k))                            // not the identity: fetch the parent path's value
					own = __shfl(r[LV], (j << 3) | lk);
				const float oth = xjb<LV - 1>(own);
				const bool hi = (j >> (LV - 1)) & 1;
				const float a = hi ? oth : own, b = hi ? own : oth;
				const int ub = (H >> (P0 + (j & (HALF - 1)))) & 1;
				r[LV - 1] = g_add(a, b, ub);
			}
			node<LV - 1, P0 + HALF>();
			// partial-sum combine of this node
			{
				constexpr int lmask = ((1 << HALF) - 1) << P0;
				const int rm = A.get(LV - 1);
				int L = H;
				if (__ballot(rm != k))
					L = __shfl(H, (j << 3) | rm);
				H = (H & ~lmask) | ((L ^ (H >> HALF)) & lmask);
			}
		}
	}
};

// Persistent grid: block b decodes codewords b, b + gridDim.x, ... and owns ONE 2 MiB level store
// (soft_all + b * 2 MiB).  A grid smaller than the machine (launch_polar's `grid`) leaves LDS and registers on
// every CU for the other stages' kernels of the next chunk, which run concurrently on a second stream: this
// kernel is bound by memory latency / HBM traffic, those by VALU and LDS.
// The decoder is ONE wavefront per workgroup: LDS and vector-memory operations of a wave execute in program
// order, so a value stored by one lane is visible to a later load by any lane of the same wave without a
// workgroup barrier.  WAVE_ORDER only stops the compiler from moving memory operations across the point; a real
// barrier would also drain every outstanding store (s_waitcnt vmcnt(0)) after each tree pass.
#define WAVE_ORDER() __builtin_amdgcn_wave_barrier()

// (The per-phase cycle counters of rounds 2 - 4, -DPOLAR_PROF, and the experiments' knobs are in the history:
// git show c197dc0:modem_amd/csrc/k_polar.hip.)
// Decoders per workgroup.  A decoder is one wave and never meets another one (no barrier, its own 8 KB of LDS).  The
// hardware places at most 16 WORKGROUPS on a CU: sixteen one-wave decoders fill every slot, and whatever kernel another
// stream launches beside them waits until decoders leave (round 3: the first kernel queued behind a polar launch took
// 27 ms instead of 1, whichever kernel it was).  With four decoders per workgroup the other kernels do run beside the
// decoders - and the list decoder slows down by exactly their time alone (36.2 -> 43.6 ms per 8192 codewords, 178 k
// frames/s against 194 k): the machine has no idle issue slots to give.  So the default stays one decoder per workgroup,
// and the pipeline treats a polar launch as owning the machine (api_pipeline.cpp: run_pipeline).
#define POLAR_WPB 1
template <int LN>
__global__ __launch_bounds__(64 * POLAR_WPB, POLAR_WAVES_PER_SIMD) void k_polar(ListQueue *__restrict__ q, int par, const ListSlot *__restrict__ slots,
	const float *__restrict__ llr_q, float *__restrict__ soft_all, uint8_t *__restrict__ hard_q, const uint32_t *__restrict__ frozen2,
	const uint8_t *__restrict__ node_lev2, float *__restrict__ metric_q)
{
	const int lane = threadIdx.x & 63, j = lane >> 3, k = lane & 7;
	const int wave_in_block = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
	const int decoder = (int)blockIdx.x * POLAR_WPB + wave_in_block;
	__shared__ float ls8_all[POLAR_WPB][32 * 64];              // level 8 of the current 256-leaf node: [x][lane], one per decoder
	float *ls8 = ls8_all[wave_in_block];
	// The work: the entries run_head .. run_head + run_n - 1 of the list decoder's queue (kernels.h: ListQueue; k_queue_plan
	// decided the run - nothing at all while too few frames wait), entry e in slot e % cap.  Persistent decoders take units
	// from a shared counter: a fixed stride would leave the fast decoders idle in the last round.
	// List 8: a unit of work is one codeword.  List 4 (the reference's 128-bit build): a unit is a PAIR of entries
	// (2u, 2u+1) decoded side by side - paths 0..3 of every 8-lane group belong to the first, paths 4..7 to the second:
	// the same instructions, loads and stores serve two codewords, and every decision the code takes wave-wide (stable
	// list, rate-1 node) is taken only when it holds for both (otherwise both take the general path, which is always
	// right).  A pair is formed when both entries share the frozen table and sit in adjacent slots; a codeword without such a
	// partner is decoded alone in the low half (the high half dead: metrics +inf) into its own partial-sum array.
	const int run_n = (int)q->run_n[par];
	const unsigned run_head = q->run_head[par], cap = q->cap;
	const int n_units = LN == 4 ? (run_n + 1) / 2 : run_n;
	if (run_n == 0)
		return;
	for (;;) {
	int unit = 0;
	if (lane == 0)
		unit = atomicAdd(&q->next_unit[par], 1);
	unit = __builtin_amdgcn_readfirstlane(unit);
	if (unit >= n_units)
		break;
	int e_a = unit, e_b = -1, n_pass = 1;
	if (LN == 4) {
		e_a = 2 * unit;
		e_b = e_a + 1 < run_n ? e_a + 1 : -1;
		if (e_b >= 0) {
			const int sa = (int)((run_head + (unsigned)e_a) % cap), sb = (int)((run_head + (unsigned)e_b) % cap);
			const bool paired = sb == sa + 1 && (slots[sa].oper_mode >= 10) == (slots[sb].oper_mode >= 10);
			if (!paired)
				n_pass = 2;                                       // two single passes, e_a then e_b
		}
	}
	for (int pass = 0; pass < n_pass; ++pass) {
	const int slot = (int)((run_head + (unsigned)((n_pass == 2 && pass == 1) ? e_b : e_a)) % cap);
	const bool hi_alive = LN == 4 && n_pass == 1 && e_b >= 0;    // a second codeword in paths 4..7 (the next slot)
	const int mode = slots[slot].oper_mode;
	const uint32_t *frozen = frozen2 + (mode >= 10 ? 2048 : 0);   // decode.cc:312,344
	const uint8_t *node_lev = node_lev2 + (mode >= 10 ? 8192 : 0);
	const float *llr = llr_q + (size_t)slot * CODE_LEN;
	float *soft = soft_all + (size_t)decoder * (8 * CODE_LEN);   // level m >= 9 at soft + 8*2^m
	uint8_t *hard = hard_q + (size_t)slot * CODE_LEN;            // (a pair shares the first one's array: bits 0..3 | 4..7)
	PolarBufs pb;
	pb.soft = make_rsrc(soft, 8 * CODE_LEN * 4);
	pb.llr = make_rsrc(llr, (hi_alive ? 2 : 1) * CODE_LEN * 4);
	pb.hard = make_rsrc(hard, CODE_LEN);
	pb.half_sel = LN == 4 ? (k >> 2) : 0;
	pb.llr_half = hi_alive ? CODE_LEN * 4 : 0;
	pb.kmask = LN - 1;
	// lane 0 of a codeword's paths carries its only real path; the others start 1000 behind; dead paths: +inf
	float M = (LN == 4 && k >= 4 && !hi_alive) ? __builtin_inff() : ((k & (LN - 1)) == 0 ? 0.f : 1000.f);
	Maps A;
	A.w0 = ID0 * (uint32_t)k;
	A.w1 = ID1 * (uint32_t)k;

	// rate-1 test of a uniform node (the predicate of Block8::list_is_stable): mu = this lane's smallest |LLR| of the node
	auto stable = [&](uint32_t mu) -> bool {
		mu = min(mu, (uint32_t)xor8_i((int)mu));
		mu = min(mu, (uint32_t)xor16_i((int)mu, lane));
		mu = min(mu, (uint32_t)xor32_i((int)mu, lane));
		const float P = M + __uint_as_float(mu);
		const float Mprev = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(M), __float_as_int(M), 0x111, 0xf, 0xf, false));   // row_shr:1
		const bool dead = LN == 4 && k >= 4 && !hi_alive, first = (k & (LN - 1)) == 0;
		const bool ok = dead || ((first || Mprev <= M) && group8_max<LN>(M) < group8_min<LN>(P));
		return __ballot(!ok) == 0;
	};
	// A uniform node of 16..128 leaves decided in one step on its register array r (CNT = 2^(level-3) elements per
	// lane, blocks b .. b+CNT-1 of the current 256-leaf node).  Returns false if it has to be walked after all.
	//   all frozen      -> the sum of max(0, -llr) in the butterfly halving order the oracle fixes (positions i, i + n/2:
	//                      lane-local while the distance is >= 8 positions, then j^4, j^2, j^1), added to the metric once
	//   all information -> if the list provably stays stable: partial sums = sign bits
	auto try_node = [&](const auto &r, bool frozen_node, uint32_t &HR, int b) -> bool {
		constexpr int CNT = (int)(sizeof(r) / sizeof(float));
		const uint32_t nmask = ((1u << CNT) - 1u) << b;
		if (frozen_node) {
			float pz[CNT];
			#pragma unroll
			for (int x = 0; x < CNT; ++x)
				pz[x] = r[x] < 0.f ? -r[x] : 0.f;
			#pragma unroll
			for (int hx = CNT / 2; hx >= 1; hx >>= 1)
				#pragma unroll
				for (int x = 0; x < hx; ++x)
					pz[x] = pz[x] + pz[x + hx];
			float pen = pz[0];
			pen = pen + xj<2>(pen, lane);
			pen = pen + xj<1>(pen, lane);
			pen = pen + xj<0>(pen, lane);
			M += pen;
			HR &= ~nmask;
			return true;
		}
		uint32_t mu = 0x7f800000u;
		#pragma unroll
		for (int x = 0; x < CNT; ++x)
			mu = min(mu, __float_as_uint(r[x]) & 0x7fffffffu);
		if (!stable(mu))
			return false;
		uint32_t bits = 0;
		#pragma unroll
		for (int x = 0; x < CNT; ++x)
			bits |= (r[x] < 0.f ? 1u : 0u) << x;
		HR = (HR & ~nmask) | (bits << b);
		return true;
	};
	// partial-sum combine of a node of >= 512 leaves on the global byte array: one dword (4 positions) per lane
	auto combine_wide = [&](uint8_t *hp, int hh, int rm) {
		int rmk[8];
		#pragma unroll
		for (int kk = 0; kk < 8; ++kk)
			rmk[kk] = __builtin_amdgcn_readlane(rm, kk);   // lanes 0..7 hold paths 0..7
		const bool ident = __ballot(rm != k) == 0;
		uint32_t *pl = (uint32_t *)hp, *pr = (uint32_t *)(hp + hh);
		for (int i0 = 0; i0 < hh / 4; i0 += 64 * 4) {
			uint32_t xl[4], xr[4];
			#pragma unroll
			for (int u = 0; u < 4; ++u)
				if (i0 + u * 64 + lane < hh / 4) {
					xl[u] = pl[i0 + u * 64 + lane];
					xr[u] = pr[i0 + u * 64 + lane];
				}
			#pragma unroll
			for (int u = 0; u < 4; ++u)
				if (i0 + u * 64 + lane < hh / 4) {
					uint32_t y = xl[u];
					if (!ident) {                         // out bit k of every byte = in bit rm[k]
						y = 0;
						#pragma unroll
						for (int kk = 0; kk < 8; ++kk)
							y |= ((xl[u] >> rmk[kk]) & 0x01010101u) << kk;
					}
					pl[i0 + u * 64 + lane] = y ^ xr[u];
				}
		}
	};

	for (int t256 = 0, adv256 = 1; t256 < CODE_LEN / 256; t256 += adv256) {
		const int t = t256 * 256;
		adv256 = 1;
		// all-information node of 256..2048 leaves that starts here (static, from the frozen pattern): decided on its own
		// array - level 8 in LDS, levels 9..11 in the level store - before anything below it is computed
		// this node's 32 table bytes and 8 frozen words, one per lane, fetched while the level passes run; the block loop
		// reads them with v_readlane (no memory round trip per block)
		const int nlv = node_lev[t256 * 32 + (lane & 31)];
		const uint32_t fzv = frozen[t256 * 8 + (lane & 7)];
		const int LtT = t ? __builtin_amdgcn_readfirstlane(nlv) >> 4 : 0;
		int Ln = 0;
		// ---------------- level 8 of this 256-leaf node into LDS, through the level store
		{
			int cur, kind;                                    // next level to produce and how its first step works
			int ho_g = 0, gl = lane;
			if (t == 0) {
				cur = 15; kind = 2;
			} else {
				const int z = __builtin_ctz(t);               // right child of the level-(z+1) node starts here
				gl = (j << 3) | A.get(z + 1);
				ho_g = t - (1 << z);                          // left child's partial sums (published bytes)
				cur = z; kind = z == 15 ? 3 : 1;
			}
			bool try_big = LtT >= 9;
			int stop = try_big ? LtT + 1 : 8;
			for (;;) {
				while (cur >= stop) {
					// produced levels stay >= 8: three per pass down to 10, then (9, 8) and (8); NG = how many are above level 8
					const int D = cur >= 10 ? 3 : (cur == 9 ? 2 : 1);
					const bool spine = kind == 1 && (t & (t - 1)) == 0 && cur <= 14;
					#define FP(...) fused_pass<__VA_ARGS__>(pb, ls8, ho_g, cur, lane, gl)
					if (t == 0) {                                     // first left descent: compact stores
						if (cur == 15) FP(3, 2, 3, false, true);      // levels 15, 14, 13
						else if (cur == 12) FP(3, 0, 3, true, true);  // 12, 11, 10
						else FP(2, 0, 1, true, true);                 // 9 and 8 (LDS)
					} else if (spine) {
						// right sibling on the left spine: its source was stored compact at t = 0; its own top level
						// (if >= 9) is not stored, the one later reader recomputes it (SRC_R below)
						if (cur >= 11) FP(3, 1, 3, true, false, true);
						else if (cur == 10) FP(3, 1, 2, true, false, true);
						else if (cur == 9) FP(2, 1, 1, true, false, true);
						else FP(1, 1, 0, true);
					} else if (kind == 1 && (t >> cur) == 3 && cur <= 14) {
						// t = 3 * 2^cur: right child of the right sibling on the left spine
						if (cur == 14) FP(3, 1, 3, false, false, false, 2);
						else if (cur >= 11) FP(3, 1, 3, false, false, false, 1);
						else if (cur == 10) FP(3, 1, 2, false, false, false, 1);
						else if (cur == 9) FP(2, 1, 1, false, false, false, 1);
						else FP(1, 1, 0, false, false, false, 1);
					} else if (cur == 15) {
						FP(3, 3, 3, false, false, true);              // t = 32768: g of the channel LLRs, level 15 itself is recomputed by its reader
					} else if (kind == 0) {
						if (cur >= 11) FP(3, 0, 3);
						else if (cur == 10) FP(3, 0, 2);
						else if (cur == 9) FP(2, 0, 1);
						else FP(1, 0, 0);
					} else {
						if (cur >= 11) FP(3, 1, 3);
						else if (cur == 10) FP(3, 1, 2);
						else if (cur == 9) FP(2, 1, 1);
						else FP(1, 1, 0);
					}
					#undef FP
					WAVE_ORDER();
					cur -= D;
					kind = 0;
				}
				if (!try_big)
					break;
				// All-information node of 512..2048 leaves that starts here: decided by a terminal pass over its SOURCE - the
				// node's own array is never stored, none of the levels below it is computed.  (Taken when the node's level is
				// the next one to produce - in both frozen tables every such node is a right child, so that is always; a
				// node reached by an overshooting three-level pass is walked.)  If the list is not provably stable the
				// ordinary pass runs and the node is walked.
				try_big = false;
				stop = 8;
				if (cur == LtT) {
					uint32_t mu = 0x7f800000u;
					#define FT(...) fused_pass<1, __VA_ARGS__, true>(pb, ls8, ho_g, cur, lane, gl, &mu, hard + t)
					if (kind == 1 && (t & (t - 1)) == 0) FT(1, 0, true, false, false, 0);          // spine: compact source
					else if (kind == 1 && (t >> cur) == 3) FT(1, 0, false, false, false, 1);       // its right child: recomputed source
					else if (kind == 1) FT(1, 0, false, false, false, 0);
					else FT(0, 0, false, false, false, 0);
					#undef FT
					WAVE_ORDER();
					if (stable(mu))
						Ln = LtT;
					if (Ln)
						break;
				}
			}
		}
		if (!Ln && LtT == 8) {
			// rate-1 node of 256 leaves on the LDS array (32 values per lane)
			uint32_t mu = 0x7f800000u;
			#pragma unroll
			for (int x = 0; x < 32; ++x)
				mu = min(mu, __float_as_uint(ls8[x * 64 + lane]) & 0x7fffffffu);
			if (stable(mu)) {
				#pragma unroll
				for (int x0 = 0; x0 < 32; x0 += 16) {
					unsigned long long mine = 0;
					#pragma unroll
					for (int u = 0; u < 16; ++u) {
						const unsigned long long bal = __ballot(ls8[(x0 + u) * 64 + lane] < 0.f);
						if (lane == u)
							mine = bal;
					}
					if (lane < 16)
						*(unsigned long long *)(hard + t + (x0 + lane) * 8) = mine;
				}
				Ln = 8;
			}
		}
		if (Ln) {
			A.reset_upto(t ? __builtin_ctz(t) : 16, k);
			adv256 = 1 << (Ln - 8);
		} else {
			// ---------------- the 32 8-leaf blocks of this node; levels 7..4 in registers
			float r7[16], r6[8], r5[4], r4[2];
			uint32_t HR = 0;                                  // partial sums, own path: bit x = position x*8 + j of the node
			for (int b = 0, adv = 1; b < 32; b += adv) {
				const int t8 = t256 * 32 + b, tt = t8 * 8;
				adv = 1;
				const int zb = b ? __builtin_ctz(b) + 3 : 8;      // level of the largest node inside this one that starts here
				// Uniform node that starts here: level 4..7 = 16..128 leaves all frozen (nl0) or all information (nl1),
				// decided in one step on its register array; the 8-leaf case (level 3) is Block8's.
				const int nl = __builtin_amdgcn_readlane(nlv, b), nl0 = nl & 15, nl1 = nl >> 4;
				int Lt = nl0 > nl1 ? nl0 : nl1;
				if (Lt > 7)
					Lt = 0;                                       // b == 0: tried above
				int L2 = 0;
				// the one g step of this block is at level zb (right child of the level-(zb+1) node): lane map since that node
				// started, partial sums of its left child = the 2^(zb-3) HR bits before b
				const int glb = zb < 8 ? ((j << 3) | A.get(zb + 1)) : lane;
				const bool mapped = __ballot(glb != lane) != 0;
				const uint32_t hb = zb < 8 ? HR >> (b - (1 << (zb - 3))) : 0u;
				if (zb >= 7) {
					if (zb == 7) {
						#pragma unroll
						for (int x = 0; x < 16; ++x)
							r7[x] = g_add(ls8[x * 64 + glb], ls8[(x + 16) * 64 + glb], (hb >> x) & 1);
					} else {
						#pragma unroll
						for (int x = 0; x < 16; ++x)
							r7[x] = f_minsum(ls8[x * 64 + lane], ls8[(x + 16) * 64 + lane]);
					}
					if (Lt == 7 && try_node(r7, nl0 == 7, HR, b))
						L2 = 7;
				}
				if (zb >= 6 && !L2) {
					if (zb == 6) {
						float p[16];
						#pragma unroll
						for (int x = 0; x < 16; ++x)
							p[x] = r7[x];
						if (mapped) {
							#pragma unroll
							for (int x = 0; x < 16; ++x)
								p[x] = __shfl(p[x], glb);
						}
						#pragma unroll
						for (int x = 0; x < 8; ++x)
							r6[x] = g_add(p[x], p[x + 8], (hb >> x) & 1);
					} else {
						#pragma unroll
						for (int x = 0; x < 8; ++x)
							r6[x] = f_minsum(r7[x], r7[x + 8]);
					}
					if (Lt == 6 && try_node(r6, nl0 == 6, HR, b))
						L2 = 6;
				}
				if (zb >= 5 && !L2) {
					if (zb == 5) {
						float p[8];
						#pragma unroll
						for (int x = 0; x < 8; ++x)
							p[x] = r6[x];
						if (mapped) {
							#pragma unroll
							for (int x = 0; x < 8; ++x)
								p[x] = __shfl(p[x], glb);
						}
						#pragma unroll
						for (int x = 0; x < 4; ++x)
							r5[x] = g_add(p[x], p[x + 4], (hb >> x) & 1);
					} else {
						#pragma unroll
						for (int x = 0; x < 4; ++x)
							r5[x] = f_minsum(r6[x], r6[x + 4]);
					}
					if (Lt == 5 && try_node(r5, nl0 == 5, HR, b))
						L2 = 5;
				}
				if (zb >= 4 && !L2) {
					if (zb == 4) {
						float p[4];
						#pragma unroll
						for (int x = 0; x < 4; ++x)
							p[x] = r5[x];
						if (mapped) {
							#pragma unroll
							for (int x = 0; x < 4; ++x)
								p[x] = __shfl(p[x], glb);
						}
						r4[0] = g_add(p[0], p[2], hb & 1);
						r4[1] = g_add(p[1], p[3], (hb >> 1) & 1);
					} else {
						r4[0] = f_minsum(r5[0], r5[2]);
						r4[1] = f_minsum(r5[1], r5[3]);
					}
					if (Lt == 4 && try_node(r4, nl0 == 4, HR, b))
						L2 = 4;
				}
				if (L2) {
					A.reset_upto(tt ? __builtin_ctz(tt) : 16, k);
					adv = 1 << (L2 - 3);
				} else {
					float r3;
					if (zb == 3) {
						float p0 = r4[0], p1 = r4[1];
						if (mapped) {
							p0 = __shfl(p0, glb);
							p1 = __shfl(p1, glb);
						}
						r3 = g_add(p0, p1, hb & 1);
					} else
						r3 = f_minsum(r4[0], r4[1]);
					const uint32_t fz = ((uint32_t)__builtin_amdgcn_readlane((int)fzv, b >> 2) >> ((b & 3) * 8)) & 0xffu;
					// the 8 leaves, all in registers: a top-down walk of the 8-leaf sub-tree (Block8 above) that decides every
					// uniform node it meets (8, 4 or 2 leaves all frozen / all information) in one step
					int H = 0;
					{
						Block8<LN> blk{ M, A, H, { 0.f, 0.f, 0.f, r3 }, fz, tt, lane, j, k, hi_alive };
						// the five patterns that make up 97 % of the 8-leaf blocks of both frozen tables (0x01: 673 of 2108, 0x17: 447,
						// 0x00: 429, 0x7f: 323, 0xff: 164) as straight-line code; the rest, and any block with a reordering fork, is walked
						bool decided = false;
						if (fz == 0x01u) decided = blk.template fast_block<0x01u>();
						else if (fz == 0x17u) decided = blk.template fast_block<0x17u>();
						else if (fz == 0x00u) decided = blk.template fast_block<0x00u>();
						else if (fz == 0x7fu) decided = blk.template fast_block<0x7fu>();
						else if (fz == 0xffu) decided = blk.template fast_block<0xffu>();
						if (!decided)
							blk.template node<3, 0>();
					}
					HR = (HR & ~(1u << b)) | ((uint32_t)((H >> j) & 1) << b);   // this lane's own position, own path
				}
				// partial-sum combines of the nodes of 16..256 leaves that end here: hard[i] = perm(hard[i]) ^ hard[i + half]
				const int bn = b + adv;
				for (int m = L2 ? L2 + 1 : 4; m <= 8 && (bn & ((1 << (m - 3)) - 1)) == 0; ++m) {
					const int half = 1 << (m - 4), b0 = bn - 2 * half;
					const int rm = A.get(m - 1);
					uint32_t Lp = HR;
					if (__ballot(rm != k))
						Lp = (uint32_t)__shfl((int)HR, (j << 3) | rm);
					const uint32_t lmask = ((1u << half) - 1u) << b0;
					HR = (HR & ~lmask) | ((Lp ^ (HR >> half)) & lmask);
				}
			}
			// publish the node's 256 partial-sum bytes (bit k = path k): 32 ballots, lanes 0..31 store 8 bytes each
			{
				unsigned long long mine = 0;
				#pragma unroll
				for (int u = 0; u < 32; ++u) {
					const unsigned long long bal = __ballot((HR >> u) & 1);
					if (lane == u)
						mine = bal;
				}
				if (lane < 32)
					*(unsigned long long *)(hard + t + lane * 8) = mine;
			}
		}
		const int tn = t + 256 * adv256;
		for (int m = Ln ? Ln + 1 : 9; m <= 16 && (tn & ((1 << m) - 1)) == 0; ++m) {
			WAVE_ORDER();
			combine_wide(hard + tn - (1 << m), 1 << (m - 1), A.get(m - 1));
		}
		WAVE_ORDER();
	}
	if (j == 0) {
		if (LN == 8 || k < 4)
			metric_q[(size_t)slot * LIST + k] = M;
		else if (hi_alive)
			metric_q[(size_t)(slot + 1) * LIST + (k - 4)] = M;
	}
	WAVE_ORDER();
	}   // second single pass of an unpaired unit
	}   // next unit of this decoder
}

void launch_polar(hipStream_t s, int list, int grid, ListQueue *q, int par, const ListSlot *slots, const float *llr_q, float *soft, uint8_t *hard_q,
	Tables tb, float *metric_q)
{
	grid = (grid + POLAR_WPB - 1) / POLAR_WPB;                // workgroups of POLAR_WPB decoders
	if (list == 4)
		hipLaunchKernelGGL(k_polar<4>, dim3(grid), dim3(64 * POLAR_WPB), 0, s, q, par, slots, llr_q, soft, hard_q, tb.frozen, tb.node_lev, metric_q);
	else
		hipLaunchKernelGGL(k_polar<8>, dim3(grid), dim3(64 * POLAR_WPB), 0, s, q, par, slots, llr_q, soft, hard_q, tb.frozen, tb.node_lev, metric_q);
}

}  // namespace rx
