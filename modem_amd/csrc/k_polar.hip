// k_polar.hip -- D9 (polar successive-cancellation list decoder, N = 65536, L = 8) for gfx950;
// D10 lives in k_finish.hip.
//
// CODE::PolarListDecoder<SIMD<float,8>,16> (decode.cc:201,530): min-sum SCL.
// One wavefront decodes one codeword.  Lane l = (j << 3) | k : k = list path (the
// reference's SIMD lane), j = one of 8 butterflies processed per wave instruction.
//   f(a,b)   = sign(a) sign(b) min(|a|,|b|)              left child LLRs
//   g(a,b,u) = u ? b - a : a + b                          right child LLRs
// Tree levels 8..15 live in HBM as soft[level m][i][k] (fp32, 2 MiB per decoder, the
// reference's own soft[N+i] layout), levels 4..7 in LDS with the same [i][k] layout; level
// 16 is the shared channel LLR vector; levels 3..0 (8-leaf sub-trees) never leave
// registers: butterflies are cross-lane shuffles.
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

namespace rx {

// three VALU: v_xor, v_med3_f32 with |.| modifiers (median of (|a|, |b|, 0) = the smaller magnitude; unlike
// fminf no canonicalising v_max is emitted), v_and_or.  A zero result may carry a minus sign; no consumer can tell.
__device__ __forceinline__ float f_minsum(float a, float b)
{
	const uint32_t sgn = (__float_as_uint(a) ^ __float_as_uint(b)) & 0x80000000u;
	const float m = __builtin_amdgcn_fmed3f(fabsf(a), fabsf(b), 0.f);
	return __uint_as_float(sgn | __float_as_uint(m));
}
__device__ __forceinline__ float g_add(float a, float b, int u) { return u ? b - a : a + b; }

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

constexpr int UB = 8;      // iterations batched per pass so that 2*UB loads are in flight per lane
#ifndef POLAR_LDS_TOP
#define POLAR_LDS_TOP 7
#endif
constexpr int LDS_TOP = POLAR_LDS_TOP;
#ifndef POLAR_COMPACT
#define POLAR_COMPACT (POLAR_LDS_TOP == 7)
#endif // tree levels 4..7 (sub-trees of <= 128 leaves) live in LDS

// Fused pass: levels m, m-1, ..., m-D+1 from level m+1 in one sweep.  A lane owns butterfly column
// (j, k) at EVERY level (position i = x*8 + j, local index x), and the partner of local index x at
// level L is x + 2^(L-4): the whole f-chain below the first step is lane-local, so the intermediate
// levels are produced in registers and each level is written exactly once, never re-read.
//   KIND 0: first step f from level m+1      KIND 1: first step g (partial sums hb, lane map gl)
//   KIND 2: first step f from the shared channel LLRs   KIND 3: first step g from the shared channel LLRs
// Address spaces are compile-time: the NG highest produced levels (and the source iff SRC_G) are
// in global memory (gs = base of the codeword's soft array), the rest in LDS (ls); flat
// addressing would serialise the two memory pipes.  Level L starts at element 8 << L in either.
// Global accesses are raw buffer loads / stores: one per-lane byte offset in a VGPR (lane * 4, or the mapped
// lane for the g step), everything else (level base, column, partner distance) in the scalar offset - so 24
// loads in flight cost 24 data registers and no 64-bit address pairs.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, int bytes)
{
	return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, int voff, int soff) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); }
__device__ __forceinline__ void bstore(rsrc_t r, int voff, int soff, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0); }
__device__ __forceinline__ int bload_u8(rsrc_t r, int voff, int soff) { return (int)__builtin_amdgcn_raw_buffer_load_b8(r, voff, soff, 0); }

struct PolarBufs {
	rsrc_t soft, llr, hard;    // this codeword's 2 MiB level store, its 65536 channel LLRs, its 65536 partial-sum bytes
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
template <int D, int KIND, int NG, bool SRC_G, bool SRC_C = false, bool DST_C = false, bool SKIP0 = false, int SRC_R = 0>
__device__ __forceinline__ void fused_pass(const PolarBufs &pb, float *ls, int hb_g_off, const uint8_t *hb_l, int m, int lane, int gl)
{
	constexpr int NT = 1 << (D - 1);          // level-m values per column of the lowest produced level
#ifndef POLAR_XB3
#define POLAR_XB3 2
#endif
#ifndef POLAR_XB2
#define POLAR_XB2 4
#endif
#ifndef POLAR_XB1
#define POLAR_XB1 8
#endif
	constexpr int XB = SRC_R ? 1 : (D == 3 ? POLAR_XB3 : (D == 2 ? POLAR_XB2 : POLAR_XB1));   // columns batched: XB * 2 * NT loads in flight
	const int S = 1 << (m - D + 1 - 3);       // local indices at the lowest produced level
	const int half = 1 << (m - 3);            // partner distance (local) at level m+1
	const int j = lane >> 3, k = lane & 7;
	const float *src_l = ls + (8 << (m + 1));
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
					const int x = x0 + xb + s2 * S;
					if (SRC_R) {
						const float a1 = bload(C, v_src + xb * XS, so_a[s2]), a2 = bload(C, v_src + xb * XS, so_a[s2] + 2 * hx);
						const float b1 = bload(C, v_src + xb * XS, so_a[s2] + hx), b2 = bload(C, v_src + xb * XS, so_a[s2] + 3 * hx);
						const int ha = bload_u8(pb.hard, v_h + xb * 8, so_h[s2] - (2 << m));
						const int hb = bload_u8(pb.hard, v_h + xb * 8, so_h[s2] - (2 << m) + half * 8);
						a[xb][s2] = g_add(a1, a2, (ha >> anc) & 1);
						b[xb][s2] = g_add(b1, b2, (hb >> anc) & 1);
					} else if (KIND >= 2 || SRC_C || SRC_G) {
						a[xb][s2] = bload(C, v_src + xb * XS, so_a[s2]);
						b[xb][s2] = bload(C, v_src + xb * XS, so_a[s2] + hx);
					} else {
						const int o = KIND == 1 ? gl : lane;
						a[xb][s2] = src_l[x * 64 + o];
						b[xb][s2] = src_l[(x + half) * 64 + o];
					}
					if (KIND & 1)
						h[xb][s2] = (SRC_G || KIND == 3 || SRC_R) ? bload_u8(pb.hard, v_h + xb * 8, so_h[s2]) : hb_l[x * 8 + j];
				}
		#pragma unroll
		for (int xb = 0; xb < XB; ++xb)
			if (x0 + xb < S) {
				float v[NT];
				#pragma unroll
				for (int s2 = 0; s2 < NT; ++s2) {
					v[s2] = (KIND & 1) ? g_add(a[xb][s2], b[xb][s2], (h[xb][s2] >> k) & 1) : f_minsum(a[xb][s2], b[xb][s2]);
					const int idx = (8 << m) + (x0 + xb + s2 * S) * 64;
					if (SKIP0) {
					} else if (NG > 0) {
						if (!DST_C) bstore(pb.soft, v_dst + xb * 256, so_d[0][s2], v[s2]);
						else if (k == 0) bstore(pb.soft, v_dst + xb * 32, so_d[0][s2], v[s2]);
					} else ls[idx + lane] = v[s2];
				}
				#pragma unroll
				for (int d = 1; d < D; ++d) {
					const int n = NT >> d;
					#pragma unroll
					for (int s2 = 0; s2 < n; ++s2) {
						v[s2] = f_minsum(v[s2], v[s2 + n]);
						const int idx = (8 << (m - d)) + (x0 + xb + s2 * S) * 64;
						if (NG > d) {
							if (!DST_C) bstore(pb.soft, v_dst + xb * 256, so_d[d][s2], v[s2]);
							else if (k == 0) bstore(pb.soft, v_dst + xb * 32, so_d[d][s2], v[s2]);
						} else ls[idx + lane] = v[s2];
					}
				}
			}
	}
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

	template <int B> __device__ __forceinline__ float xjb(float v) const { return xj<B>(v, lane); }
	__device__ __forceinline__ void reset_at(int p0) { const int tt = t + p0; A.reset_upto(tt ? __builtin_ctz(tt) : 16, k); }
	__device__ __forceinline__ bool list_is_stable(float mu) const
	{
		const float P = M + mu;
		const float Mprev = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(M), __float_as_int(M), 0x111, 0xf, 0xf, false));   // row_shr:1
		const bool ok = k >= LN || ((k == 0 || Mprev <= M) && group8_max<LN>(M) < group8_min<LN>(P));
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
				rank += (o0 < val) | ((o0 == val) & (2 * kk < cidx));
				rank += (o1 < val) | ((o1 == val) & (2 * kk + 1 < cidx));
			}
			// every row of 16 lanes holds the 16 candidates at position (u << 3) | k: scatter inside the row
			const int dst = (lane & 48) | rank;
			int rc = __builtin_amdgcn_ds_permute(dst << 2, cidx);
			int rv = __builtin_amdgcn_ds_permute(dst << 2, __float_as_int(val));
			const int rcx = xor8_i(rc), rvx = xor8_i(rv);   // odd j sit at row position 8+k: take position k
			if (j & 1) { rc = rcx; rv = rvx; }
			M = k < LN ? __int_as_float(rv) : __builtin_inff();   // list 4: ranks 4..7 do not survive
			const int parent = rc >> 1;
			ubit = rc & 1;
			if (__ballot(parent != k)) {
				A.w0 = __shfl((int)A.w0, (j << 3) | parent);
				A.w1 = __shfl((int)A.w1, (j << 3) | parent);
			}
		}
		H = (H & ~(1 << P0)) | (ubit << P0);
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

template <int LN>
__global__ __launch_bounds__(64) void k_polar(int n_cw, const SyncState *__restrict__ st_all, const float *__restrict__ llr_all,
	float *__restrict__ soft_all, uint8_t *__restrict__ hard_all, const uint32_t *__restrict__ frozen2, const uint8_t *__restrict__ node_lev2,
	float *__restrict__ metric_all)
{
	const int lane = threadIdx.x, j = lane >> 3, k = lane & 7;
	for (int cw = blockIdx.x; cw < n_cw; cw += gridDim.x) {
	if (!st_all[cw].okay)
		continue;                                             // no header -> nothing to decode (decode.cc:450-451)
	const uint32_t *frozen = frozen2 + (st_all[cw].oper_mode >= 10 ? 2048 : 0);   // decode.cc:312,344
	const uint8_t *node_lev = node_lev2 + (st_all[cw].oper_mode >= 10 ? 8192 : 0);
	const float *llr = llr_all + (size_t)cw * CODE_LEN;
	float *soft = soft_all + (size_t)blockIdx.x * (8 * CODE_LEN);   // level m >= 8 at soft + 8*2^m
	uint8_t *hard = hard_all + (size_t)cw * CODE_LEN;
	__shared__ float ls[8 << (LDS_TOP + 1)];                  // level m <= 7 at ls + 8*2^m
	__shared__ __attribute__((aligned(8))) uint8_t lh[1 << LDS_TOP];   // partial sums of the current 128-leaf sub-tree
	PolarBufs pb;
	pb.soft = make_rsrc(soft, 8 * CODE_LEN * 4);
	pb.llr = make_rsrc(llr, CODE_LEN * 4);
	pb.hard = make_rsrc(hard, CODE_LEN);
	float M = k == 0 ? 0.f : (k < LN ? 1000.f : __builtin_inff());   // lane 0 carries the only real path; k >= LN: dead
	Maps A;
	A.w0 = ID0 * (uint32_t)k;
	A.w1 = ID1 * (uint32_t)k;
	float r3 = 0.f;

	for (int t8 = 0, adv = 1; t8 < CODE_LEN / 8; t8 += adv) {
		const int t = t8 * 8;
		// Uniform node that starts here (static, from the frozen pattern): level 4..7 = 16..128 leaves all frozen
		// (nl0) or all information (nl1); such a node is decided in one step on its own LLR array (below), the
		// 8-leaf case (level 3) in registers further down.  Ln = level of the node decided this way, 0 = none.
		const int nl = node_lev[t8], nl0 = nl & 15, nl1 = nl >> 4, Lt = nl0 > nl1 ? nl0 : nl1;
		int Ln = 0;
		adv = 1;
		// ---------------- LLRs of this 8-leaf sub-tree into r3
		{
			int cur, kind;                                    // next level to produce and how its first step works
			int ho_g = 0;
			const uint8_t *ho_l = lh;
			int gl = lane;
			if (t == 0) {
				cur = 15; kind = 2;
			} else {
				const int z = __builtin_ctz(t);               // right child of the level-(z+1) node starts here
				gl = (j << 3) | A.get(z + 1);
				ho_g = t - (1 << z);                          // left child's partial sums: global for sub-trees >= 128 leaves,
				ho_l = lh + ((t - (1 << z)) & ((1 << LDS_TOP) - 1));   // else inside the current LDS block
				cur = z; kind = z == 15 ? 3 : 1;
			}
			if (cur == 3) {
				float a = ls[(8 << 4) + gl], b = ls[(8 << 4) + gl + 64];
				r3 = g_add(a, b, (ho_l[j] >> k) & 1);
			} else {
				bool try_node = Lt >= 4, redo;
				int stop = try_node ? Lt : 4;
				do {
				while (cur >= stop) {
					#define FP(DD, KK, NGG, SG) fused_pass<DD, KK, NGG, SG>(pb, ls, ho_g, ho_l, cur, lane, gl)
					#define FPC(DD, KK, NGG, SC, DC) fused_pass<DD, KK, NGG, true, SC, DC>(pb, ls, ho_g, ho_l, cur, lane, gl)
					#define FPK(DD, NGG, SG) do { if (kind == 0) FP(DD, 0, NGG, SG); else FP(DD, 1, NGG, SG); } while (0)
					int D = 3;
					// NG = produced levels that are above LDS_TOP (global); SRC_G = the source level cur+1 is global
					static_assert(LDS_TOP == 7 || !POLAR_COMPACT, "the compact first descent is written for LDS_TOP = 7");
					if (POLAR_COMPACT && t == 0 && cur >= LDS_TOP + 2) {          // first left descent: compact stores
						if (cur == 15) FPC(3, 2, 3, false, true);
						else if (cur >= LDS_TOP + 3) FPC(3, 0, 3, true, true);
						else FPC(3, 0, 2, true, true);
					} else if (POLAR_COMPACT && kind == 1 && (t & (t - 1)) == 0 && cur >= LDS_TOP && cur <= 14 && cur == Lt) {
						// ... unless this very level is a node decided on its own array (below): then it is stored
						if (cur >= LDS_TOP + 3) FPC(3, 1, 3, true, false);
						else if (cur == LDS_TOP + 2) FPC(3, 1, 2, true, false);
						else if (cur == LDS_TOP + 1) FPC(3, 1, 1, true, false);
						else FPC(3, 1, 0, true, false);
					} else if (POLAR_COMPACT && kind == 1 && (t & (t - 1)) == 0 && cur >= LDS_TOP && cur <= 14) {
						// right sibling on the left spine: its source was stored compact at t = 0; its own top level
						// (if >= 8) is not stored, the one later reader recomputes it (SRC_R below)
						if (cur >= LDS_TOP + 3) fused_pass<3, 1, 3, true, true, false, true>(pb, ls, ho_g, ho_l, cur, lane, gl);
						else if (cur == LDS_TOP + 2) fused_pass<3, 1, 2, true, true, false, true>(pb, ls, ho_g, ho_l, cur, lane, gl);
						else if (cur == LDS_TOP + 1) fused_pass<3, 1, 1, true, true, false, true>(pb, ls, ho_g, ho_l, cur, lane, gl);
						else FPC(3, 1, 0, true, false);
					} else if (POLAR_COMPACT && kind == 1 && (t >> cur) == 3 && cur >= LDS_TOP && cur <= 14) {
						// t = 3 * 2^cur: right child of the right sibling on the left spine
						#define FPR(NGG, RR) fused_pass<3, 1, NGG, true, false, false, false, RR>(pb, ls, ho_g, ho_l, cur, lane, gl)
						if (cur == 14) FPR(3, 2);
						else if (cur >= LDS_TOP + 3) FPR(3, 1);
						else if (cur == LDS_TOP + 2) FPR(2, 1);
						else if (cur == LDS_TOP + 1) FPR(1, 1);
						else FPR(0, 1);
						#undef FPR
					}
					else if (cur == 15) { if (kind == 2) FP(3, 2, 3, true); else if (POLAR_COMPACT) fused_pass<3, 3, 3, true, false, false, true>(pb, ls, ho_g, ho_l, cur, lane, gl); else FP(3, 3, 3, true); }
					else if (cur >= LDS_TOP + 3) FPK(3, 3, true);
					else if (cur == LDS_TOP + 2) FPK(3, 2, true);
					else if (cur == LDS_TOP + 1) FPK(3, 1, true);
					else if (cur == LDS_TOP) FPK(3, 0, true);
					else if (cur >= 6) FPK(3, 0, false);
					else if (cur == 5) { FPK(2, 0, false); D = 2; }
					else { FPK(1, 0, false); D = 1; }
					#undef FPC
					#undef FPK
					#undef FP
					WAVE_ORDER();
					cur -= D;
					kind = 0;
				}
				redo = false;
				if (try_node) {
					// level Lt (in LDS) now holds the node's LLRs: cnt = 2^(Lt-3) values per lane, position x*8 + j
					try_node = false;
					const int cnt = 1 << (Lt - 3);
					const float *lv = ls + (8 << Lt) + lane;
					uint8_t *lhn = lh + (t & ((1 << LDS_TOP) - 1));
					if (Lt > LDS_TOP) {
						// rate-1 node of 256..2048 leaves: the same decision on its array in the level store (cnt = 32..256
						// values per lane); the sign bytes go straight to the partial-sum array.  None of the levels
						// below it is ever computed.
						const int base = (8 << Lt) * 4;
						uint32_t mu = 0x7f800000u;
						for (int x0 = 0; x0 < cnt; x0 += 16) {
							float v[16];
							#pragma unroll
							for (int u = 0; u < 16; ++u)
								v[u] = bload(pb.soft, lane * 4, base + (x0 + u) * 256);
							#pragma unroll
							for (int u = 0; u < 16; ++u)
								mu = min(mu, __float_as_uint(v[u]) & 0x7fffffffu);
						}
						mu = min(mu, (uint32_t)xor8_i((int)mu));
						mu = min(mu, (uint32_t)xor16_i((int)mu, lane));
						mu = min(mu, (uint32_t)xor32_i((int)mu, lane));
						const float P = M + __uint_as_float(mu);
						const float Mprev = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(M), __float_as_int(M), 0x111, 0xf, 0xf, false));   // row_shr:1
						const bool ok = k >= LN || ((k == 0 || Mprev <= M) && group8_max<LN>(M) < group8_min<LN>(P));
						if (__ballot(!ok) == 0) {
							for (int x0 = 0; x0 < cnt; x0 += 16) {
								float v[16];
								#pragma unroll
								for (int u = 0; u < 16; ++u)
									v[u] = bload(pb.soft, lane * 4, base + (x0 + u) * 256);
								unsigned long long mine = 0;
								#pragma unroll
								for (int u = 0; u < 16; ++u) {
									const unsigned long long bal = __ballot(v[u] < 0.f);
									if (lane == u)
										mine = bal;
								}
								if (lane < 16)                        // 16 positions x 8 bytes: one 128-byte store
									*(unsigned long long *)(hard + t + (x0 + lane) * 8) = mine;
							}
							Ln = Lt;
						} else {
							stop = 4;
							redo = cur >= 4;
						}
					} else if (nl0) {
						// rate-0 node: sum of max(0, -llr) in the butterfly halving order the oracle fixes
						// (positions i, i + n/2: lane-local while the distance is >= 8, then j^4, j^2, j^1)
						float pz[16];
						#pragma unroll
						for (int x = 0; x < 16; ++x) {
							const float v = x < cnt ? lv[x * 64] : 0.f;
							pz[x] = v < 0.f ? -v : 0.f;
						}
						#pragma unroll
						for (int hx = 8; hx >= 1; hx >>= 1)
							#pragma unroll
							for (int x = 0; x < hx; ++x)
								pz[x] = pz[x] + pz[x + hx];           // + 0 beyond cnt: exact
						float pen = pz[0];
						pen = pen + xj<2>(pen, lane);
						pen = pen + xj<1>(pen, lane);
						pen = pen + xj<0>(pen, lane);
						M += pen;
						if (lane < cnt)
							*(unsigned long long *)(lhn + lane * 8) = 0ull;
						Ln = Lt;
					} else {
						// rate-1 node: same argument as for 8 leaves, mu_k = the smallest magnitude of the whole node
						uint32_t mu = 0x7f800000u;
						#pragma unroll
						for (int x = 0; x < 16; ++x)
							if (x < cnt)
								mu = min(mu, __float_as_uint(lv[x * 64]) & 0x7fffffffu);
						mu = min(mu, (uint32_t)xor8_i((int)mu));
						mu = min(mu, (uint32_t)xor16_i((int)mu, lane));
						mu = min(mu, (uint32_t)xor32_i((int)mu, lane));
						const float P = M + __uint_as_float(mu);
						const float Mprev = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(M), __float_as_int(M), 0x111, 0xf, 0xf, false));   // row_shr:1
						const bool ok = k >= LN || ((k == 0 || Mprev <= M) && group8_max<LN>(M) < group8_min<LN>(P));
						if (__ballot(!ok) == 0) {
							for (int x = 0; x < cnt; ++x) {           // partial sums = sign bits, one byte per position
								const unsigned long long bal = __ballot(lv[x * 64] < 0.f);
								if (lane == 0)
									*(unsigned long long *)(lhn + x * 8) = bal;
							}
							Ln = Lt;
						} else {
							stop = 4;                                 // walk it after all: the 8-leaf groups get their own chance
							redo = cur >= 4;
						}
					}
				}
				} while (redo);
				if (Ln) {
					A.reset_upto(t ? __builtin_ctz(t) : 16, k);
					adv = 1 << (Ln - 3);
				} else
					r3 = f_minsum(ls[(8 << 4) + lane], ls[(8 << 4) + lane + 64]);
			}
		}
		if (!Ln) {
		const uint32_t fz = (frozen[t >> 5] >> (t & 31)) & 0xffu;
		// ---------------- the 8 leaves, all in registers: a top-down walk of the 8-leaf sub-tree (Block8 above) that
		// decides every uniform node it meets (8, 4 or 2 leaves all frozen / all information) in one step
		int H = 0;
		{
			Block8<LN> blk{ M, A, H, { 0.f, 0.f, 0.f, r3 }, fz, t, lane, j, k };
			blk.template node<3, 0>();
		}
		// ---------------- partial sums of the sub-tree: bytes (bit k = path k) via ballot
		{
			const unsigned long long bal = __ballot((H >> j) & 1);
			if (lane == 0)
				*(unsigned long long *)(lh + (t & ((1 << LDS_TOP) - 1))) = bal;
		}
		}   // !Ln
		auto combine = [&](int off, int hh, int rm) {        // LDS block: lh[i] = perm(lh[i], rm) ^ lh[i + hh]
			uint8_t *hp = lh + off;
			for (int it0 = 0; it0 < hh / 8; it0 += UB) {
				int xl[UB], xr[UB];
				#pragma unroll
				for (int u = 0; u < UB; ++u)
					if (it0 + u < hh / 8) {
						xl[u] = hp[(it0 + u) * 8 + j];
						xr[u] = hp[hh + (it0 + u) * 8 + j];
					}
				#pragma unroll
				for (int u = 0; u < UB; ++u)
					if (it0 + u < hh / 8) {
						const int bit = ((xl[u] >> rm) ^ (xr[u] >> k)) & 1;
						const unsigned long long bal = __ballot(bit != 0);
						if (lane == 0)
							*(unsigned long long *)(hp + (it0 + u) * 8) = bal;
					}
			}
		};
		// same for sub-trees of >= 256 leaves (hh >= 128): one dword (4 positions) per lane, 256 B per sweep
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
		const int tn = t + 8 * adv;
		for (int m = Ln ? Ln + 1 : 4; m <= LDS_TOP && (tn & ((1 << m) - 1)) == 0; ++m) {
			WAVE_ORDER();
			combine((tn - (1 << m)) & ((1 << LDS_TOP) - 1), 1 << (m - 1), A.get(m - 1));
		}
		if ((tn & ((1 << LDS_TOP) - 1)) == 0) {               // a 128-leaf sub-tree is complete: publish its bytes
			WAVE_ORDER();
			if (Ln > LDS_TOP) {
				// a node of >= 256 leaves wrote its bytes itself
			} else if (LDS_TOP == 7)
				((unsigned short *)(hard + tn - (1 << LDS_TOP)))[lane] = ((const unsigned short *)lh)[lane];
			else
				for (int q = lane; q < (1 << LDS_TOP) / 4; q += 64)
					((uint32_t *)(hard + tn - (1 << LDS_TOP)))[q] = ((const uint32_t *)lh)[q];
			for (int m = Ln > LDS_TOP ? Ln + 1 : LDS_TOP + 1; m <= 16 && (tn & ((1 << m) - 1)) == 0; ++m) {
				WAVE_ORDER();
				combine_wide(hard + tn - (1 << m), 1 << (m - 1), A.get(m - 1));
			}
		}
		WAVE_ORDER();
	}
	if (j == 0)
		metric_all[(size_t)cw * LIST + k] = M;
	WAVE_ORDER();
	}   // next codeword of this block
}

void launch_polar(hipStream_t s, int list, int n, int grid, const SyncState *st, const float *llr, float *soft, uint8_t *hard, Tables tb, float *metric)
{
	if (grid <= 0 || grid > n)
		grid = n;
	if (list == 4)
		hipLaunchKernelGGL(k_polar<4>, dim3(grid), dim3(64), 0, s, n, st, llr, soft, hard, tb.frozen, tb.node_lev, metric);
	else
		hipLaunchKernelGGL(k_polar<8>, dim3(grid), dim3(64), 0, s, n, st, llr, soft, hard, tb.frozen, tb.node_lev, metric);
}

}  // namespace rx
